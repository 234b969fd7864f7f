// gx_tile_hop_w.hip -- the tile kernel's UTF-16 instantiations for TIER_HOP (gx_tile_body.hpp: WIDE).
#include "gx_tile_body.hpp"

namespace gx {
hipError_t launch_tile_hop_w(int mode, bool off64, const GxLds& lds, const void* io, dim3 grid, dim3 block, hipStream_t stream) {
    const TileIO& t = *static_cast<const TileIO*>(io);
    if (mode == 2) return hipErrorInvalidValue;
    if (off64) return launch_tile_wide<uint64_t, TIER_HOP>(mode, lds, t, grid, block, stream);
    return launch_tile_wide<uint32_t, TIER_HOP>(mode, lds, t, grid, block, stream);
}
}  // namespace gx
