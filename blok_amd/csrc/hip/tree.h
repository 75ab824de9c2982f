// Derived traversal structure: a tree of 4x4x4 nodes ("64-tree") over the world's filled voxels.
//
// The reference walks a binary octree of 16-B SvoNode records inside each 16^3 sub-chunk
// (reference assets/shaders/intersect.rint:70-206) and leaves the choice of sub-chunks to Vulkan RT
// hardware (reference blok/src/renderer_raytracing.cpp:15-254).  Both are replaced by one structure:
//
//   level L (root)  : one node covering 4^L voxels per axis
//   level l (L..2)  : 64-bit child mask + index of the first child node of level l-1
//   level 1 (brick) : 64-bit voxel mask  + index of the first material id (one u32 per set bit)
//
// Child c of a node has bit  (x & 3) | (y & 3) << 2 | (z & 3) << 4  of the node-local cell
// coordinate; children / material ids are stored in ascending bit order, so the rank of a bit
// (popcount of the lower set bits) is the offset from `base`.  Every level is stored in Morton
// order of its cells.  One node is 16 bytes = one dwordx4 load.
#ifndef BLOK_TREE_H
#define BLOK_TREE_H

#include <cstdint>
#include <vector>

namespace blok {

struct TreeNode {
    uint32_t mask_lo;
    uint32_t mask_hi;
    uint32_t base;      // level >= 2: index of child 0 in nodes[];  level 1: index into materials[]
    uint32_t reserved;
};
static_assert(sizeof(TreeNode) == 16, "one node is one 16-byte load");

constexpr uint32_t kMaxLevels = 7;   // 4^7 = 16384 voxels per axis (hit records carry int16 coordinates)

struct HostTree {
    std::vector<TreeNode> nodes;     // nodes[0] is the root
    std::vector<uint32_t> materials; // material id per filled voxel
    uint32_t levels = 0;
    int32_t  origin[3] = {0, 0, 0};  // world coordinate of the tree's min corner
    uint64_t n_voxels = 0;
};

struct VoxelRec { int32_t x, y, z; uint32_t material; };

// Builds the tree from a voxel list (duplicates: the last record wins).  Returns false and sets
// `why` if the extent does not fit kMaxLevels / int16.
bool build_tree(std::vector<VoxelRec>& voxels, HostTree& out, const char** why);

// Levels L..2 (root first) from the sorted, strictly increasing keys of the level-1 bricks, for a tree whose
// bricks are stored right after these nodes (level-2 nodes' `base` already includes upper.size()).
// Returns false if the keys are not strictly increasing (duplicate bricks).
bool build_upper_levels(const std::vector<uint64_t>& brick_keys, uint32_t levels, std::vector<TreeNode>& upper);

}  // namespace blok
#endif
