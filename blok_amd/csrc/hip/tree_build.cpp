// Host builder of the 64-tree (tree.h) and the reader of the reference's world arrays.
#include "tree.h"
#include "reference_world.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

namespace blok {

namespace {

inline uint64_t tree_key(uint32_t x, uint32_t y, uint32_t z, uint32_t levels) {
    uint64_t key = 0;
    for (uint32_t l = 0; l < levels; ++l) {
        const uint64_t digit = ((x >> (2 * l)) & 3u) | (((y >> (2 * l)) & 3u) << 2) | (((z >> (2 * l)) & 3u) << 4);
        key |= digit << (6 * l);
    }
    return key;
}

inline int32_t floor_to(int32_t v, int32_t m) {
    const int32_t r = v % m;
    return r < 0 ? v - r - m : v - r;
}

}  // namespace

bool build_tree(std::vector<VoxelRec>& voxels, HostTree& out, const char** why) {
    out = HostTree{};
    if (voxels.empty()) {
        out.levels = 1;
        out.nodes.push_back(TreeNode{0, 0, 0, 0});
        return true;
    }
    int32_t lo[3] = {std::numeric_limits<int32_t>::max(), std::numeric_limits<int32_t>::max(), std::numeric_limits<int32_t>::max()};
    int32_t hi[3] = {std::numeric_limits<int32_t>::min(), std::numeric_limits<int32_t>::min(), std::numeric_limits<int32_t>::min()};
    for (const VoxelRec& v : voxels) {
        lo[0] = std::min(lo[0], v.x); hi[0] = std::max(hi[0], v.x);
        lo[1] = std::min(lo[1], v.y); hi[1] = std::max(hi[1], v.y);
        lo[2] = std::min(lo[2], v.z); hi[2] = std::max(hi[2], v.z);
    }
    for (int a = 0; a < 3; ++a)
        if (lo[a] < -32768 || hi[a] > 32767) { *why = "world voxel coordinates exceed int16 (hit records carry int16)"; return false; }
    int64_t extent = 1;
    for (int a = 0; a < 3; ++a) {
        out.origin[a] = floor_to(lo[a], 16);
        extent = std::max<int64_t>(extent, int64_t(hi[a]) - out.origin[a] + 1);
    }
    uint32_t levels = 1;
    while ((int64_t(1) << (2 * levels)) < extent) ++levels;
    if (levels > kMaxLevels) { *why = "world extent exceeds 4^7 voxels per axis"; return false; }
    out.levels = levels;

    // sort by Morton key of 2-bit digits; stable so that the last duplicate can be kept
    const size_t n = voxels.size();
    std::vector<uint64_t> keys(n);
    for (size_t i = 0; i < n; ++i)
        keys[i] = tree_key(uint32_t(voxels[i].x - out.origin[0]), uint32_t(voxels[i].y - out.origin[1]),
                           uint32_t(voxels[i].z - out.origin[2]), levels);
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });

    std::vector<uint64_t> cur_keys;   // keys of the entities of the level below
    cur_keys.reserve(n);
    out.materials.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t v = order[i];
        if (i + 1 < n && keys[order[i + 1]] == keys[v]) continue;   // keep the last write
        cur_keys.push_back(keys[v]);
        out.materials.push_back(voxels[v].material);
    }
    out.n_voxels = out.materials.size();

    // bottom-up: level l groups the entities of level l-1 by key >> 6
    std::vector<std::vector<TreeNode>> level_nodes(levels + 1);
    for (uint32_t l = 1; l <= levels; ++l) {
        std::vector<uint64_t> next_keys;
        std::vector<TreeNode>& nodes = level_nodes[l];
        for (size_t i = 0; i < cur_keys.size(); ++i) {
            const uint64_t parent = cur_keys[i] >> 6;
            const uint32_t bit = uint32_t(cur_keys[i] & 63u);
            if (next_keys.empty() || next_keys.back() != parent) {
                next_keys.push_back(parent);
                nodes.push_back(TreeNode{0, 0, uint32_t(i), 0});
            }
            if (bit < 32) nodes.back().mask_lo |= 1u << bit; else nodes.back().mask_hi |= 1u << (bit - 32);
        }
        cur_keys.swap(next_keys);
    }
    // concatenate root-first; child indices of level l (>= 2) point into level l-1's block
    std::vector<uint32_t> start(levels + 2, 0);
    uint64_t total = 0;
    for (uint32_t l = levels; l >= 1; --l) { start[l] = uint32_t(total); total += level_nodes[l].size(); }
    if (total > 0xFFFFFFFFull) { *why = "more than 2^32 tree nodes"; return false; }
    out.nodes.resize(total);
    for (uint32_t l = levels; l >= 1; --l) {
        TreeNode* dst = out.nodes.data() + start[l];
        for (size_t i = 0; i < level_nodes[l].size(); ++i) {
            dst[i] = level_nodes[l][i];
            if (l >= 2) dst[i].base += start[l - 1];
        }
    }
    return true;
}

bool build_upper_levels(const std::vector<uint64_t>& brick_keys, uint32_t levels, std::vector<TreeNode>& upper) {
    upper.clear();
    for (size_t i = 1; i < brick_keys.size(); ++i)
        if (brick_keys[i] <= brick_keys[i - 1]) return false;
    if (levels <= 1) return brick_keys.size() == 1;              // the single brick is the root
    std::vector<std::vector<TreeNode>> level_nodes(levels + 1);
    std::vector<uint64_t> cur = brick_keys, next;
    for (uint32_t l = 2; l <= levels; ++l) {
        next.clear();
        std::vector<TreeNode>& nodes = level_nodes[l];
        for (size_t i = 0; i < cur.size(); ++i) {
            const uint64_t parent = cur[i] >> 6;
            const uint32_t bit = uint32_t(cur[i] & 63u);
            if (next.empty() || next.back() != parent) { next.push_back(parent); nodes.push_back(TreeNode{0, 0, uint32_t(i), 0}); }
            if (bit < 32) nodes.back().mask_lo |= 1u << bit; else nodes.back().mask_hi |= 1u << (bit - 32);
        }
        cur.swap(next);
    }
    std::vector<uint32_t> start(levels + 2, 0);
    uint64_t total = 0;
    for (uint32_t l = levels; l >= 2; --l) { start[l] = uint32_t(total); total += level_nodes[l].size(); }
    if (total + brick_keys.size() > 0xFFFFFFFFull) return false;
    start[1] = uint32_t(total);                                   // bricks follow the upper levels
    upper.resize(total);
    for (uint32_t l = levels; l >= 2; --l)
        for (size_t i = 0; i < level_nodes[l].size(); ++i) {
            TreeNode n = level_nodes[l][i];
            n.base += start[l - 1];
            upper[start[l] + i] = n;
        }
    return true;
}

// ------------------------------------------------------------------------------------------------
// Reading the reference's world: every sub-chunk's octree is expanded to its filled unit leaves.
// Visibility rules follow the shader that consumes these arrays: a child is only ever visited through
// a set childMask bit (intersect.rint:169), out-of-range node indices are skipped (:132), a node
// with childMask == 0 is a leaf and counts iff occupancy > 0 (:136-137).
bool extract_voxels(const blok_svo_node* nodes, size_t n_nodes, const blok_sub_chunk* subs, size_t n_subs,
                    std::vector<VoxelRec>& out, const char** why) {
    out.clear();
    struct Item { uint32_t node; int32_t x, y, z; uint32_t size; };
    std::vector<Item> stack;
    for (size_t s = 0; s < n_subs; ++s) {
        const blok_sub_chunk& sc = subs[s];
        const float size_f = sc.sub_chunk_size;
        const uint32_t size = uint32_t(size_f);
        if (!(size_f >= 1.0f) || float(size) != size_f || (size & (size - 1)) != 0 || size > 1024) {
            *why = "sub-chunk size is not a power-of-two number of unit voxels (voxelSize must be 1)";
            return false;
        }
        int32_t origin[3];
        for (int a = 0; a < 3; ++a) {
            const float m = sc.world_min[a];
            if (std::floor(m) != m || std::fabs(m) > 32768.0f || sc.world_max[a] != m + size_f) {
                *why = "sub-chunk bounds are not on the integer voxel lattice";
                return false;
            }
            origin[a] = int32_t(m);
        }
        const uint64_t limit = std::min<uint64_t>(uint64_t(sc.node_offset) + sc.node_count, n_nodes);
        stack.clear();
        stack.push_back(Item{sc.node_offset + sc.root_node_index, origin[0], origin[1], origin[2], size});
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            if (it.node >= limit) continue;
            const blok_svo_node& nd = nodes[it.node];
            if (nd.child_mask == 0u) {
                if (nd.occupancy > 0.0f) {
                    if (it.size != 1) { *why = "filled leaf above voxel level (not produced by SvoTree::insertVoxel)"; return false; }
                    out.push_back(VoxelRec{it.x, it.y, it.z, nd.material_id});
                }
                continue;
            }
            if (it.size == 1) { *why = "interior node below voxel level"; return false; }
            const uint32_t half = it.size / 2;
            for (uint32_t c = 0; c < 8; ++c) {
                if (!(nd.child_mask & (1u << c))) continue;
                stack.push_back(Item{sc.node_offset + nd.first_child + c,
                                     it.x + int32_t((c & 1u) ? half : 0), it.y + int32_t((c & 2u) ? half : 0),
                                     it.z + int32_t((c & 4u) ? half : 0), half});
            }
        }
    }
    return true;
}

}  // namespace blok
