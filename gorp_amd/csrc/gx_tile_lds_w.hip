// gx_tile_lds_w.hip -- the tile kernel's UTF-16 instantiations for TIER_LDS (gx_tile_body.hpp: WIDE): `data` holds 16-bit code
// units, the units' low bytes are staged.  A translation unit of its own so that it compiles beside the others.
#include "gx_tile_body.hpp"

namespace gx {
hipError_t launch_tile_lds_w(int mode, bool off64, const GxLds& lds, const void* io, dim3 grid, dim3 block, hipStream_t stream) {
    const TileIO& t = *static_cast<const TileIO*>(io);
    if (off64) return launch_tile_wide<uint64_t, TIER_LDS>(mode, lds, t, grid, block, stream);
    return launch_tile_wide<uint32_t, TIER_LDS>(mode, lds, t, grid, block, stream);
}
}  // namespace gx
