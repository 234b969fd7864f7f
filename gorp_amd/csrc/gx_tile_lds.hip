// gx_tile_lds.hip -- the tile kernel's instantiations for TIER_LDS (see gx_tile_body.hpp).
#include "gx_tile_body.hpp"

namespace gx {
GX_TILE_TIER_ENTRY(launch_tile_lds, TIER_LDS)
}  // namespace gx
