// gx_tile_rec.hip -- the tile kernel's instantiations for TIER_REC (see gx_tile_body.hpp).
#include "gx_tile_body.hpp"

namespace gx {
GX_TILE_TIER_ENTRY(launch_tile_rec, TIER_REC)
}  // namespace gx
