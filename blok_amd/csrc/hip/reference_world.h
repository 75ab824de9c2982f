// Reader of the reference's world arrays (WorldSvoGpu, reference blok/include/resources.hpp:195-203).
#ifndef BLOK_REFERENCE_WORLD_H
#define BLOK_REFERENCE_WORLD_H
#include "blok_hip.h"
#include "tree.h"

namespace blok {
// Expands every sub-chunk octree into its filled unit voxels (world integer coordinates).
// Returns false with a reason for worlds outside the supported lattice.
bool extract_voxels(const blok_svo_node* nodes, size_t n_nodes, const blok_sub_chunk* subs, size_t n_subs,
                    std::vector<VoxelRec>& out, const char** why);
}
#endif
