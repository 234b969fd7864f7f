// gx_tile_recg.hip -- the tile kernel's instantiations for TIER_RECG (see gx_tile_body.hpp).
#include "gx_tile_body.hpp"

namespace gx {
GX_TILE_TIER_ENTRY(launch_tile_recg, TIER_RECG)
}  // namespace gx
