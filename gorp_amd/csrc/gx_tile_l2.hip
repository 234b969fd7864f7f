// gx_tile_l2.hip -- the tile kernel's instantiations for TIER_L2 (see gx_tile_body.hpp).
#include "gx_tile_body.hpp"

namespace gx {
GX_TILE_TIER_ENTRY(launch_tile_l2, TIER_L2)
}  // namespace gx
