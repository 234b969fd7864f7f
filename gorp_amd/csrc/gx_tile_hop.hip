// hop records over dense rows in global memory, class ids in the staging area (gx_hop.hpp): the fused automaton with
// "register := position" programs (mode 1), or the match automaton alone (mode 0: match-only batches)
#include "gx_tile_body.hpp"
namespace gx {
hipError_t launch_tile_hop(int mode, bool off64, const GxLds& lds, const void* io, dim3 grid, dim3 block, hipStream_t stream) {
    const TileIO& t = *static_cast<const TileIO*>(io);
    if (mode == 0) {
        if (off64) return launch_tile_k<uint64_t, TIER_HOP, 0>(lds, t, grid, block, stream);
        return launch_tile_k<uint32_t, TIER_HOP, 0>(lds, t, grid, block, stream);
    }
    if (off64) return launch_tile_k<uint64_t, TIER_HOP, 1>(lds, t, grid, block, stream);
    return launch_tile_k<uint32_t, TIER_HOP, 1>(lds, t, grid, block, stream);
}
}  // namespace gx
