// gx_tile.hip -- launch of the tile kernel (gx_tile_body.hpp): fills the kernel's arguments and picks the table tier,
// whose kernels live in a translation unit each.
#include "gx_tile_body.hpp"
#ifdef GX_DEV
#include <cstdlib>
#endif

namespace gx {

hipError_t launch_tile_lds(int mode, bool off64, const GxLds& lds, const void* io, dim3 grid, dim3 block, hipStream_t stream);
hipError_t launch_tile_l2(int mode, bool off64, const GxLds& lds, const void* io, dim3 grid, dim3 block, hipStream_t stream);
hipError_t launch_tile_rec(int mode, bool off64, const GxLds& lds, const void* io, dim3 grid, dim3 block, hipStream_t stream);
hipError_t launch_tile_recg(int mode, bool off64, const GxLds& lds, const void* io, dim3 grid, dim3 block, hipStream_t stream);
hipError_t launch_tile_hop(int mode, bool off64, const GxLds& lds, const void* io, dim3 grid, dim3 block, hipStream_t stream);
hipError_t launch_tile_lds_w(int mode, bool off64, const GxLds& lds, const void* io, dim3 grid, dim3 block, hipStream_t stream);
hipError_t launch_tile_hop_w(int mode, bool off64, const GxLds& lds, const void* io, dim3 grid, dim3 block, hipStream_t stream);

hipError_t launch_extract_tile(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, int num_cus,
                               const GxBatch& b, hipStream_t stream, unsigned long long* dev_stamps) {
    if (b.n == 0) return hipSuccess;
    const uint64_t tiles = (b.n + 63) >> 6;
    uint32_t blocks_per_cu = 163840u / lds.total_bytes;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu * lds.nwaves > 12) blocks_per_cu = 1;  // the kernel is built for <= 3 waves per SIMD
    uint64_t blocks = static_cast<uint64_t>(num_cus) * blocks_per_cu;
    const uint64_t need = (tiles + lds.nwaves - 1) / lds.nwaves;
    if (blocks > need) blocks = need;
    if (blocks > GX_STEAL_MAX) blocks = GX_STEAL_MAX;
    if (!b.steal) return hipErrorInvalidValue;
    dim3 grid(static_cast<unsigned>(blocks)), block(lds.nwaves * 64);
    TileIO io{};
    io.image = lds_image;
    io.at_global = at_global;
    io.data = static_cast<const uint8_t*>(b.data);
    io.off = b.offsets;
    io.n = b.n;
    io.match_id = b.match_id;
    io.caps = b.caps;
    io.packed = b.packed;
    io.overflow = b.overflow;
    io.narrow = b.narrow;
    io.oversize_flag = b.oversize_flag;
    io.seq = b.seq;
    io.steal = b.steal;
    io.steal_parity = b.steal_parity & 1u;
    io.state_out = b.state_out;
    if (b.state_out && (b.match_only == 0 || lds.tier > 1u || b.wide)) return hipErrorInvalidValue;   // (states: match-only batches on dense rows)
    io.wide_flags = b.wide_flags;
    io.wide_any = b.wide_any;
    // How much of a workgroup's tiles other workgroups may take (64ths; gx_tile_body.hpp).  With DENSE results the XCDs of a chip finish
    // an equal share up to a tenth apart (the odd ones later, on half of the boxes of the pool): an eighth evens that out -- 43 -> 6 us
    // between the median and the last wave -- at the price of 1 % (a returning global atomic per tile and wave in that eighth).  With
    // compact or u8 rows, or no captures at all, they are 0-3 % apart: sharing then costs what it gives or more (same-box comparisons:
    // profiles/r04_ab_tile.txt), so those launches share nothing.
#ifndef GX_SHARE64
#define GX_SHARE64 ((b.packed || b.match_only != 0) ? 0u : 8u)
#endif
    io.share64 = GX_SHARE64;
    io.max_groups = dev.max_groups;
    io.strip_eol = b.strip_eol;
#ifdef GX_DEV
    io.stamps = dev_stamps;
    io.dev_flags = getenv("GX_DEV_FLAGS") ? static_cast<uint32_t>(atoi(getenv("GX_DEV_FLAGS"))) : 0u;
    if (getenv("GX_DEV_SHARE64")) io.share64 = static_cast<uint32_t>(atoi(getenv("GX_DEV_SHARE64")));   // (0: nothing shared, 64: everything)
#else
    (void)dev_stamps;
#endif
    const bool want_caps = b.match_only == 0 && dev.has_capture;
    const int mode = !want_caps ? 0 : (lds.u_start != 0xFFFFFFFFu && lds.simple_ops) ? 1 : 2;
    const bool off64 = b.offsets64 != 0;
    if (b.wide) {   // UTF-16 code units: dense rows in LDS, or the hop tier (the other tiers take the narrowed copy: gx_api.cpp)
        if (!b.wide_flags || !b.wide_any) return hipErrorInvalidValue;
        if (lds.tier == 4) return (at_global && mode != 2) ? launch_tile_hop_w(mode, off64, lds, &io, grid, block, stream) : hipErrorInvalidValue;
        if (at_global || lds.tier != 0) return hipErrorInvalidValue;
        return launch_tile_lds_w(mode, off64, lds, &io, grid, block, stream);
    }
    if (lds.tier == 4) return (at_global && mode != 2) ? launch_tile_hop(mode, off64, lds, &io, grid, block, stream) : hipErrorInvalidValue;
    if (at_global && lds.tier == 3) return launch_tile_recg(mode, off64, lds, &io, grid, block, stream);
    if (at_global) return launch_tile_l2(mode, off64, lds, &io, grid, block, stream);
    if (lds.tier == 2) return launch_tile_rec(mode, off64, lds, &io, grid, block, stream);
    return launch_tile_lds(mode, off64, lds, &io, grid, block, stream);
}

}  // namespace gx
