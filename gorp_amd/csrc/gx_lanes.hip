// gx_lanes.hip -- the batch kernel for definitions whose dense rows do not fit LDS (range records in LDS; dense rows or
// records in global memory) and for uneven lines on any tables: every lane keeps ITS OWN line in registers.
//
// Replaces, per line, the same reference code as the tile kernel (gx_tile_body.hpp):
//   PolyMatcher.match core/autom/PolyMatcher.java:123-133, Automata.step/accept core/autom/Automata.java:133-139,
//   JDKRegexpCookedExtraction.match core/jdkre/JDKRegexpCookedExtraction.java:36-59, Gorp.extract core/Gorp.java:159-186.
//
// With large tables every byte position costs the lock-step wave one dependent lookup (an LDS record read, or a gather
// from L2), and what bounds the throughput is how many lines a CU has in flight.  The tile kernel stages whole tiles in
// LDS -- 250 bytes of LDS per line in flight, 6-10 waves per CU beside the tables.  Here a lane loads its line 16 bytes
// at a time into KCH x 4 registers (plain loads at the lane's own address: a lane returns to the same cache line for its
// next 16 bytes and finds it in the vector L1), the wave walks the KCH windows in lock step with ONE copy of the window
// code (the window is always taken from register slot 0, the others move down), and loads the next KCH x 16 bytes.  LDS
// holds only the tables, the capture registers and the wave's result rows, so one workgroup of 16 waves shares one
// copy of the tables.
// SORTED: a lane does not need its tile's lines to be neighbours in memory, so for uneven lines a workgroup orders the
// lines of a chunk by length first and forms its tiles from lines of similar length (a tile takes as long as its longest
// line); chunks are handed out by a counter in global memory.
#include "gx_walk.hpp"

namespace gx {

namespace {

struct LanesIO {
    const uint8_t* image;        // the LDS table image, in global memory
    const uint8_t* at_global;    // automaton rows / records in global memory
    const uint8_t* data;
    const void* off;
    uint64_t n;
    int32_t* match_id;
    int32_t* caps;
    uint16_t* packed;
    unsigned long long* overflow;
    int32_t narrow;              // compact rows as u8 (gx_device.hpp: GxBatch::narrow)
    uint32_t* oversize_flag;
    uint32_t seq;
    int32_t max_groups;
    int32_t strip_eol;
    uint32_t chunk_lines;        // SORTED: lines per workgroup chunk (a multiple of 64, at most 65536)
    uint32_t sort_lds;           // SORTED: LDS address of u16 perm[chunk_lines], then u32 hist[64], u32 cursor[64]
    uint32_t* chunk_ctr;         // SORTED: the launch's chunk counter in global memory and its value when the launch began
    uint32_t chunk_base;
#ifdef GX_DEV
    unsigned long long* stamps;  // per workgroup: core-clock cycles and 100 MHz ticks spent in the kernel
#endif
};

// One workgroup of up to 16 waves per CU: the waves of a workgroup share one copy of the tables in LDS.
template <typename OFF, int KCH, int TIER, bool CAPTURE, bool SIMPLE, bool PACKED, bool SORTED>
__global__ void __launch_bounds__(1024)
k_extract_lanes(GxLds L, LanesIO io) {
#ifdef GX_DEV
    const uint64_t dev_t0 = __builtin_amdgcn_s_memtime(), dev_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    {
        extern __shared__ __attribute__((aligned(16))) uint8_t gx_smem[];
        const uint4* src = reinterpret_cast<const uint4*>(io.image);
        uint4* dst = reinterpret_cast<uint4*>(gx_smem);
        for (uint32_t c = threadIdx.x; c < L.table_bytes / 16; c += blockDim.x) dst[c] = src[c];
        if (threadIdx.x == 0) lds_st<uint32_t>(L.counter, L.nwaves);  // the workgroup's tile counter
    }
    __syncthreads();

    const uint8_t* __restrict__ data = io.data;
    const OFF* __restrict__ off = static_cast<const OFF*>(io.off);
    const uint64_t n = io.n;
    WalkTab W;  // the automaton this launch walks: fused (captures) or match
    constexpr bool GT = TIER == TIER_L2 || TIER == TIER_RECG;  // tables and final records in global memory
    const uint8_t* base = GT ? io.at_global + (TIER == TIER_RECG ? L.c_base : 0u) : nullptr;
    W.at = (CAPTURE && TIER == TIER_L2) ? io.at_global + L.c_base : base;
    W.row_bytes = L.row_bytes;
    W.ops_off = L.ops_off;
    W.ops = L.ops;
    W.acc_tab = L.acc_tab;
    W.ncls = L.ncls;
    W.indexed = L.rec_indexed;
    const uint32_t row0 = CAPTURE ? L.u_start : L.m_start, dead_row = CAPTURE ? L.u_dead : L.m_dead;

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // per wave: the register block (register r of this lane = u16 at regs + r * 128; the column before register 0 is a
    // write-only dummy: GxLds::stage_bytes bytes), then the area the tile's result rows are transposed through
    const uint32_t wave_area = L.regs + wave * L.regs_wave_bytes;
    const uint32_t regs = wave_area + 128u + lane * 2u;
    const uint32_t out_area = wave_area + L.stage_bytes;
    const int G = io.max_groups;
    const uint32_t slots = 2u * static_cast<uint32_t>(G);
    const uint64_t tiles = (n + 63) >> 6;
    const uint8_t* data_end = data + static_cast<uint64_t>(off[n]);
    const uint8_t* fin_g = GT ? io.at_global + L.fin_tags : nullptr;
    const uint64_t grid = gridDim.x;

    auto next_ticket = [&]() {  // the workgroup's tile counter
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add((GX_LDS uint32_t*)(uintptr_t)L.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(t));
    };
    // SORTED (uneven lines): a tile takes as long as its longest line, so the workgroup orders the lines of a chunk by
    // length first (a counting sort over 64 length classes of 32 bytes, longest first, in LDS) and forms its tiles from
    // neighbours in that order.  A lane loads its own line whatever its address, so only the result rows lose their
    // contiguity (every lane stores its own).  Chunks are handed out by a counter in global memory.
    const uint32_t CH = SORTED ? io.chunk_lines : 64u;
    const uint32_t perm = io.sort_lds, hist = perm + 2u * CH, cursor = hist + 256u;
    const uint64_t nchunks = (n + CH - 1) / CH;
    auto length_class = [&](uint64_t i) {
        const uint64_t l = static_cast<uint64_t>(off[i + 1]) - static_cast<uint64_t>(off[i]);
        return 63u - static_cast<uint32_t>(l >> 5 < 63u ? l >> 5 : 63u);
    };
    uint32_t j = wave;  // (!SORTED) the wave's ticket: tile blockIdx.x + j * gridDim.x
    for (;;) {
        uint64_t first = 0;
        uint32_t cnt = 0, chunk_tiles = 0;
        if (SORTED) {
            if (threadIdx.x == 0)
                lds_st<uint32_t>(L.counter + 4u, __hip_atomic_fetch_add(io.chunk_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - io.chunk_base);
            __syncthreads();
            const uint64_t chunk = lds_ld<uint32_t>(L.counter + 4u);
            if (chunk >= nchunks) break;
            first = chunk * CH;
            cnt = static_cast<uint32_t>(n - first < CH ? n - first : CH);
            chunk_tiles = (cnt + 63u) >> 6;
            if (threadIdx.x < 64u) lds_st<uint32_t>(hist + 4u * threadIdx.x, 0u);
            if (threadIdx.x == 0) lds_st<uint32_t>(L.counter, 0u);
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < cnt; t += blockDim.x)
                __hip_atomic_fetch_add((GX_LDS uint32_t*)(uintptr_t)(hist + 4u * length_class(first + t)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
            if (threadIdx.x < 64u) {  // exclusive scan of the 64 counts (wave 0)
                const uint32_t v = lds_ld<uint32_t>(hist + 4u * lane);
                uint32_t incl = v;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t up = static_cast<uint32_t>(__shfl_up(static_cast<int>(incl), d));
                    if (lane >= static_cast<uint32_t>(d)) incl += up;
                }
                lds_st<uint32_t>(cursor + 4u * lane, incl - v);
            }
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < cnt; t += blockDim.x) {
                const uint32_t at = __hip_atomic_fetch_add((GX_LDS uint32_t*)(uintptr_t)(cursor + 4u * length_class(first + t)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                lds_st<uint16_t>(perm + 2u * at, static_cast<uint16_t>(t));
            }
            __syncthreads();
        }
        for (;;) {
            // ---- the next tile: lane l takes line i (valid: it has one); contiguous: the 64 lines are i - lane .. i - lane + 63,
            // so their result rows are one block of the output ----
            uint64_t i;
            bool valid;
            const bool contiguous = !SORTED;
            if (SORTED) {
                const uint32_t t = next_ticket();
                if (t >= chunk_tiles) break;
                const uint32_t x = (t << 6) + lane;
                valid = x < cnt;
                i = first + (valid ? lds_ld<uint16_t>(perm + 2u * x) : 0u);
            } else {
                const uint64_t tile = static_cast<uint64_t>(blockIdx.x) + static_cast<uint64_t>(j) * grid;  // (wave-uniform)
                if (tile >= tiles) break;
                i = (tile << 6) + lane;
                valid = i < n;
            }
            const uint64_t o0 = off[valid ? i : n], o1 = off[valid ? i + 1 : n];
            uint64_t len64 = o1 - o0;
            if (io.strip_eol) {  // the terminator belongs to the line in the CSR buffer (gx_split_lines), not to the String
                if (len64 > 0 && data[o0 + len64 - 1] == 0x0Au) --len64;
                if (len64 > 0 && data[o0 + len64 - 1] == 0x0Du) --len64;
            }
            // positions are 16-bit in the register block: a longer line is left to the follow-up launch of the per-line kernel.
            // Compact rows keep 0xFFFF for "unset" and promise that an offset above 65 534 is stored as 65 534 and counted
            // (include/gorp_hip.h): a line of exactly 65 535 bytes can have such an offset, so it goes the same way (the
            // per-line kernel writes through LineOut, which clamps and counts).
            const bool oversize = valid && len64 > (PACKED ? 65534u : 65535u);
            if (oversize) __hip_atomic_store(io.oversize_flag, io.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint32_t len = oversize ? 0u : static_cast<uint32_t>(len64);
            const uint8_t* line = data + o0;

            uint32_t row = row0;
            uint32_t acc = state_acc<TIER>(W, row);
            uint32_t lo4 = splat_byte0(acc), k4 = splat_byte1(acc);
            bool more = valid && !oversize && len > 0u;
            // 16 bytes of this lane's line at line offset at_byte (zeros beyond the line; never a byte beyond the buffer)
            auto line_chunk = [&](uint32_t at_byte, bool wanted) -> u32x4 {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (wanted && at_byte < len) {
                    const uint8_t* src = line + at_byte;
                    if (src + 16 <= data_end) v = reinterpret_cast<const UnalignedWindow*>(src)->v;  // (a plain load: see KCH below)
                    else {
                        uint32_t w[4] = {0, 0, 0, 0};
                        for (int b = 0; b < 16; ++b)
                            if (src + b < data_end) w[b >> 2] |= static_cast<uint32_t>(src[b]) << ((b & 3) * 8);
                        v = u32x4{w[0], w[1], w[2], w[3]};
                    }
                }
                return v;
            };
            for (uint32_t seg = 0; __any(more); seg += KCH * 16u) {
                u32x4 pre[KCH];  // this lane's bytes [seg, seg + 16 KCH) of its line
    #pragma unroll
                for (int k = 0; k < KCH; ++k) pre[k] = line_chunk(seg + 16u * k, more);
                // One copy of the window code, run KCH times: the window is always taken from slot 0 and the others move down
                // (4 register moves per slot and window).  Unrolled with static slots instead, the loop body was 80 KB of code
                // at 7 windows, for no gain.
    #pragma unroll 1
                for (int k = 0; k < KCH; ++k) {
                    const uint32_t rel = seg + 16u * k;
                    const bool act = more && rel < len;
                    if (!__any(act)) break;
                    const uint4 w0 = make_uint4(pre[0].x, pre[0].y, pre[0].z, pre[0].w);
    #pragma unroll
                    for (int q = 0; q + 1 < KCH; ++q) pre[q] = pre[q + 1];
                    const bool full0 = rel + 16u <= len;
                    const uint32_t bx = outside_bits(w0.x, lo4, k4), by = outside_bits(w0.y, lo4, k4);
                    const uint32_t bz = outside_bits(w0.z, lo4, k4), bw = outside_bits(w0.w, lo4, k4);
                    const bool ok0 = full0 & (((or3(bx, by, bz) | bw) & HI_BITS) == 0u);
                    bool step = act && !ok0;
                    uint32_t mask = 0xFFFFu;
                    if (step && !full0) {
                        mask = window_mask(0u, len, rel);
                        step = !partial_window_ok(bx, by, bz, bw, mask);
                    }
                    const bool masked = __any(step && !full0);
                    if (step) {
                        if (!masked) row = steps16<TIER, CAPTURE, false, SIMPLE>(w0, mask, W, row, rel, regs);
                        else row = steps16<TIER, CAPTURE, true, SIMPLE>(w0, mask, W, row, rel, regs);
                        acc = state_acc<TIER>(W, row);
                        lo4 = splat_byte0(acc);
                        k4 = splat_byte1(acc);
                    }
                    more = more && row != dead_row && rel + 16u < len;
                }
            }

            // ---- results ----
            const int32_t info = state_info<TIER>(W, row);
            if (!CAPTURE) {
                if (valid && !oversize) io.match_id[i] = info;
            } else {
                const bool full_tile = contiguous && __all(valid) && !__any(oversize);
                const uint64_t i0 = i - lane;
                if (PACKED) {
                    // u16 rows, or -- io.narrow, wave-uniform -- u8 rows (an offset above 254 stored as 254 and counted)
                    const bool narrow = io.narrow != 0;
                    const uint32_t row_b = narrow ? 1u + slots : 2u + 2u * slots;
                    const bool rows_aligned = (reinterpret_cast<uintptr_t>(io.packed) & 15u) == 0u;  // (16-byte stores below)
                    uint8_t* rows8 = reinterpret_cast<uint8_t*>(io.packed);
                    uint32_t clamped = 0;
                    if (full_tile && rows_aligned) {
                        // the tile's 64 rows are one contiguous block of the output: through the wave's row area, then 1 KiB of
                        // consecutive bytes per store instruction
                        const uint32_t my_out = out_area + lane * row_b;
                        if (narrow) {
                            const int32_t result = line_result<TIER>(info, L.fin_tags, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                                clamped += (pb > 254 ? 1u : 0u) + (pe > 254 ? 1u : 0u);
                                lds_st<uint8_t>(my_out + 1u + 2u * g, static_cast<uint8_t>(pb > 254 ? 254 : pb));
                                lds_st<uint8_t>(my_out + 2u + 2u * g, static_cast<uint8_t>(pe > 254 ? 254 : pe));
                            });
                            lds_st<uint8_t>(my_out, static_cast<uint8_t>(result));
                        } else {
                            const int32_t result = line_result<TIER>(info, L.fin_tags, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                                lds_st<uint16_t>(my_out + 2u + 4u * g, static_cast<uint16_t>(pb));
                                lds_st<uint16_t>(my_out + 4u + 4u * g, static_cast<uint16_t>(pe));
                            });
                            lds_st<uint16_t>(my_out, static_cast<uint16_t>(result));
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        uint8_t* out = rows8 + i0 * static_cast<uint64_t>(row_b);
                        for (uint32_t c = lane; c < 4u * row_b; c += 64u)
                            *reinterpret_cast<u32x4*>(out + (c << 4)) = lds_ld<u32x4>(out_area + (c << 4));
                    } else if (valid && !oversize) {
                        if (narrow) {
                            uint8_t* rp = rows8 + i * static_cast<uint64_t>(row_b);
                            const int32_t result = line_result<TIER>(info, L.fin_tags, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                                clamped += (pb > 254 ? 1u : 0u) + (pe > 254 ? 1u : 0u);
                                rp[1 + 2 * g] = static_cast<uint8_t>(pb > 254 ? 254 : pb);
                                rp[2 + 2 * g] = static_cast<uint8_t>(pe > 254 ? 254 : pe);
                            });
                            rp[0] = static_cast<uint8_t>(result);
                        } else {
                            uint16_t* rp = io.packed + i * static_cast<uint64_t>(1u + slots);
                            const int32_t result = line_result<TIER>(info, L.fin_tags, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                                rp[1 + 2 * g] = static_cast<uint16_t>(pb);
                                rp[2 + 2 * g] = static_cast<uint16_t>(pe);
                            });
                            rp[0] = static_cast<uint16_t>(result);
                        }
                    }
                    if (narrow && clamped && io.overflow) atomicAdd(io.overflow, static_cast<unsigned long long>(clamped));
                } else {
                    // dense int32 rows: every lane stores its own (8 bytes per group).  Taking them through the wave's row area 32
                    // lines at a time, as contiguous 16-byte stores, was measured: no faster (1.558 against 1.551 ms on config 3)
                    if (valid && !oversize) {
                        int32_t* cp = io.caps + i * static_cast<uint64_t>(slots);
                        io.match_id[i] = line_result<TIER>(info, L.fin_tags, fin_g, regs, len, G, [&](int g, int32_t pb, int32_t pe) {
                            *reinterpret_cast<u32x2*>(cp + 2 * g) = u32x2{static_cast<uint32_t>(pb), static_cast<uint32_t>(pe)};
                        });
                    }
                }
            }
            // the wave's LDS area is reused by the next tile
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!SORTED) j = next_ticket();
        }
        if (!SORTED) break;
        __syncthreads();  // (the next chunk rewrites perm and the counter)
    }
#ifdef GX_DEV
    if (io.stamps && threadIdx.x == 0) {
        io.stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - dev_t0;
        io.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - dev_r0;
    }
#endif
}

template <typename OFF, int TIER, bool CAPTURE, bool SIMPLE, bool PACKED, bool SORTED>
hipError_t launch_lanes_s(const GxLds& lds, const LanesIO& io, dim3 grid, hipStream_t stream) {
    // 208 bytes of the line in 52 registers.  Measured on config 3 (captures, dense rows), variants side by side on one device:
    //   windows per segment: 5 2.09 ms, 7 1.89 ms, 13 1.80 ms;
    //   loads with the non-temporal hint (as the tile kernel's, which read every byte once) 1.80 ms, plain 1.56 ms: a lane comes
    //   back to the same cache line for its next 16 bytes, and plain loads find it in the vector L1;
    //   the four lanes of a quad fetching 64 contiguous bytes of one line each and exchanging inside the quad: 1.66 ms;
    //   the next tile's offsets and first windows loaded while this tile's results are put together: 1.94-2.02 ms (the kernel
    //   then needs more than its 128 registers); no interval test for a window in which some lane's state has no interval:
    //   1.94 against 1.89 ms.
    //   Timing-only builds (wrong results): no line loads at all 1.28 ms, the same bytes loaded in lane order 1.39 ms, no
    //   record reads 1.84 ms (no change: the dependent LDS read per byte is not what bounds the walk).
    constexpr int KCH = 13;
    hipError_t e = allow_full_lds(&k_extract_lanes<OFF, KCH, TIER, CAPTURE, SIMPLE, PACKED, SORTED>);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_extract_lanes<OFF, KCH, TIER, CAPTURE, SIMPLE, PACKED, SORTED>), grid, dim3(lds.nwaves * 64), lds.total_bytes, stream, lds, io);
    return hipGetLastError();
}
template <typename OFF, int TIER, bool CAPTURE, bool SIMPLE, bool PACKED>
hipError_t launch_lanes_t(const GxLds& lds, const LanesIO& io, dim3 grid, hipStream_t stream) {
    if (io.chunk_lines) return launch_lanes_s<OFF, TIER, CAPTURE, SIMPLE, PACKED, true>(lds, io, grid, stream);
    return launch_lanes_s<OFF, TIER, CAPTURE, SIMPLE, PACKED, false>(lds, io, grid, stream);
}
template <typename OFF, int TIER>
hipError_t launch_lanes_m(bool capture, bool simple, bool packed, const GxLds& lds, const LanesIO& io, dim3 grid, hipStream_t stream) {
    if (!capture) return launch_lanes_t<OFF, TIER, false, false, false>(lds, io, grid, stream);
    if (simple) return packed ? launch_lanes_t<OFF, TIER, true, true, true>(lds, io, grid, stream) : launch_lanes_t<OFF, TIER, true, true, false>(lds, io, grid, stream);
    return packed ? launch_lanes_t<OFF, TIER, true, false, true>(lds, io, grid, stream) : launch_lanes_t<OFF, TIER, true, false, false>(lds, io, grid, stream);
}

}  // namespace

uint32_t lanes_sorted_tickets(uint64_t n, uint32_t chunk_lines, int num_cus) {
    const uint64_t nchunks = (n + chunk_lines - 1) / chunk_lines;
    return static_cast<uint32_t>(nchunks + (static_cast<uint64_t>(num_cus) < nchunks ? static_cast<uint64_t>(num_cus) : nchunks));
}

// lds: a layout from plan_lanes_launch (gx_api.cpp): tables, then per wave the register block with the result rows
// behind it.  Needs tables in global memory (tier 1 or 3) and, for captures, the fused automaton.
hipError_t launch_extract_lanes(const GxDev& dev, const GxLds& lds, const uint8_t* lds_image, const uint8_t* at_global, int num_cus,
                                const GxBatch& b, hipStream_t stream, unsigned long long* dev_stamps) {
    if (b.n == 0) return hipSuccess;
    const uint64_t tiles = (b.n + 63) >> 6;
    uint64_t blocks = static_cast<uint64_t>(num_cus);
    const uint64_t need = (tiles + lds.nwaves - 1) / lds.nwaves;
    if (blocks > need) blocks = need;
    LanesIO io{};
    if (lds.sort_chunk) {
        const uint64_t nchunks = (b.n + lds.sort_chunk - 1) / lds.sort_chunk;
        blocks = static_cast<uint64_t>(num_cus) < nchunks ? static_cast<uint64_t>(num_cus) : nchunks;
        io.chunk_lines = lds.sort_chunk;
        io.sort_lds = lds.sort_lds;
        io.chunk_ctr = b.chunk_ctr;
        io.chunk_base = b.chunk_base;
    }
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
    io.max_groups = dev.max_groups;
    io.strip_eol = b.strip_eol;
#ifdef GX_DEV
    io.stamps = dev_stamps;
#else
    (void)dev_stamps;
#endif
    const bool capture = b.match_only == 0 && dev.has_capture;
    const dim3 grid(static_cast<unsigned>(blocks));
    const bool packed = b.packed != nullptr;
    if (lds.tier == 3) {
        if (b.offsets64) return launch_lanes_m<uint64_t, TIER_RECG>(capture, lds.simple_ops != 0, packed, lds, io, grid, stream);
        return launch_lanes_m<uint32_t, TIER_RECG>(capture, lds.simple_ops != 0, packed, lds, io, grid, stream);
    }
    if (lds.tier == 2) {
        if (b.offsets64) return launch_lanes_m<uint64_t, TIER_REC>(capture, lds.simple_ops != 0, packed, lds, io, grid, stream);
        return launch_lanes_m<uint32_t, TIER_REC>(capture, lds.simple_ops != 0, packed, lds, io, grid, stream);
    }
    if (lds.tier == 0) {  // dense rows in LDS (uneven lines: the tile kernel wants a tile's lines to be neighbours in memory)
        if (b.offsets64) return launch_lanes_m<uint64_t, TIER_LDS>(capture, lds.simple_ops != 0, packed, lds, io, grid, stream);
        return launch_lanes_m<uint32_t, TIER_LDS>(capture, lds.simple_ops != 0, packed, lds, io, grid, stream);
    }
    if (b.offsets64) return launch_lanes_m<uint64_t, TIER_L2>(capture, lds.simple_ops != 0, packed, lds, io, grid, stream);
    return launch_lanes_m<uint32_t, TIER_L2>(capture, lds.simple_ops != 0, packed, lds, io, grid, stream);
}

}  // namespace gx
