// Host-side voxel data model behind include/blok_world.h: Morton codec, per-chunk SVO
// build, sub-chunk packing.  Produces the reference's record layouts byte for byte
// (reference blok/include/svo.hpp:23-28, blok/include/resources.hpp:170-184) so the arrays can
// be handed to blok_hip_upload_world exactly like WorldSvoGpu's.
//
// Storage differs from the reference on purpose: the reference keeps two dense C^3 arrays
// per chunk (16 MiB at C=128, reference blok/include/chunk.hpp:35-36), which is 8 GiB for a
// 1024^3 world.  Here a chunk keeps a write log that is sorted and de-duplicated (last write
// wins) at rebuild time; visiting that log in ascending local index x + y*C + z*C*C is the
// same sequence of insertVoxel calls as the reference's z,y,x scan
// (reference blok/src/chunk_manager.cpp:106-119), hence the same node array.
#include "blok_world.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

namespace {

constexpr uint32_t kNoChild = 0xFFFFFFFFu;          // reference blok/include/svo.hpp:20
constexpr uint32_t kSubDivisions = 8;               // reference blok/src/chunk_manager.cpp:17
constexpr int32_t  kMortonBias = 1 << 20;           // reference blok/include/morton.hpp:25

// --- Morton: byte-table dilation (same function as reference morton.hpp:12-21,35-44) -------
struct DilateTable {
    uint32_t spread[256];   // bit i of the byte -> bit 3*i
    uint8_t  squeeze[512];  // 9-bit group with stride-3 payload bits 0,3,6 -> 3 bits
    constexpr DilateTable() : spread(), squeeze() {
        for (uint32_t b = 0; b < 256; ++b) {
            uint32_t s = 0;
            for (uint32_t i = 0; i < 8; ++i) s |= ((b >> i) & 1u) << (3 * i);
            spread[b] = s;
        }
        for (uint32_t g = 0; g < 512; ++g)
            squeeze[g] = static_cast<uint8_t>((g & 1u) | ((g >> 2) & 2u) | ((g >> 4) & 4u));
    }
};
constexpr DilateTable kDilate{};

inline uint64_t dilate21(uint32_t v) {
    v &= 0x1FFFFFu;
    return  static_cast<uint64_t>(kDilate.spread[v & 0xFFu])
         | (static_cast<uint64_t>(kDilate.spread[(v >> 8) & 0xFFu]) << 24)
         | (static_cast<uint64_t>(kDilate.spread[(v >> 16) & 0x1Fu]) << 48);
}
inline uint32_t contract21(uint64_t code) {
    uint32_t v = 0;
    for (uint32_t k = 0; k < 7; ++k)  // 7 groups of 3 payload bits = 21 bits
        v |= static_cast<uint32_t>(kDilate.squeeze[(code >> (9 * k)) & 0x1FFu]) << (3 * k);
    return v;
}
inline uint64_t morton_encode(int32_t x, int32_t y, int32_t z) {
    return dilate21(static_cast<uint32_t>(x + kMortonBias))
         | (dilate21(static_cast<uint32_t>(y + kMortonBias)) << 1)
         | (dilate21(static_cast<uint32_t>(z + kMortonBias)) << 2);
}
inline uint32_t morton_octant(uint64_t code, uint32_t max_depth, uint32_t level) {
    return static_cast<uint32_t>(code >> (3u * (max_depth - 1u - level))) & 7u;
}

// --- per-chunk octree -----------------------------------------------------------------------
struct Octree {
    std::vector<blok_svo_node> nodes;
    uint32_t depth = 0;

    static blok_svo_node blank() { return blok_svo_node{0u, kNoChild, 0u, 0.0f}; }
    void reset() { nodes.clear(); nodes.push_back(blank()); }

    // reference blok/src/svo.cpp:59-101 (descent + sibling-block allocation :36-57 + mask
    // propagation :94-100).  The mask bits are OR-ed on the way down; the reference ORs them
    // on the way back up, which yields the same array.
    void insert(uint32_t x, uint32_t y, uint32_t z, uint32_t material, float density) {
        if (!(density > 0.0f)) return;
        const uint32_t dim = 1u << depth;
        if (x >= dim || y >= dim || z >= dim) return;
        const uint64_t code = morton_encode(static_cast<int32_t>(x), static_cast<int32_t>(y),
                                            static_cast<int32_t>(z));
        uint32_t at = 0;
        for (uint32_t level = 0; level < depth; ++level) {
            const uint32_t oct = morton_octant(code, depth, level);
            uint32_t first = nodes[at].first_child;
            if (first == kNoChild) {
                first = static_cast<uint32_t>(nodes.size());
                nodes.resize(nodes.size() + 8, blank());
                nodes[at].first_child = first;
            }
            nodes[at].child_mask |= 1u << oct;
            at = first + oct;
        }
        nodes[at].material_id = material;
        nodes[at].occupancy = density;
    }

    // reference blok/src/svo.cpp:103-130
    int64_t find_leaf(uint32_t x, uint32_t y, uint32_t z) const {
        const uint32_t dim = 1u << depth;
        if (x >= dim || y >= dim || z >= dim) return -1;
        const uint64_t code = morton_encode(static_cast<int32_t>(x), static_cast<int32_t>(y),
                                            static_cast<int32_t>(z));
        uint32_t at = 0;
        for (uint32_t level = 0; level < depth; ++level) {
            const uint32_t oct = morton_octant(code, depth, level);
            const blok_svo_node& n = nodes[at];
            if (!(n.child_mask & (1u << oct)) || n.first_child == kNoChild) return -1;
            at = n.first_child + oct;
        }
        return nodes[at].occupancy > 0.0f ? static_cast<int64_t>(at) : -1;
    }
};

struct VoxelWrite {
    uint32_t local;     // x + y*C + z*C*C
    uint32_t seq;       // write order, for last-write-wins
    uint32_t material;
    float    density;
};

struct ChunkRec {
    int32_t cx = 0, cy = 0, cz = 0;
    std::vector<VoxelWrite> log;
    bool log_sorted = true;
    bool dirty = true;
    Octree tree;
};

using ChunkKey = std::tuple<int32_t, int32_t, int32_t>;  // (cz, cy, cx): sorted packing order

}  // namespace

struct blok_world {
    uint32_t C = 0;
    uint32_t depth = 0;
    float voxel_size = 1.0f;
    std::map<ChunkKey, std::unique_ptr<ChunkRec>> chunks;
    std::vector<ChunkRec*> order;  // sorted view, rebuilt on demand
    std::vector<blok_svo_node> packed_nodes;
    std::vector<blok_sub_chunk> packed_subs;
    blok_material_library* material_lib = nullptr;   // non-owning, reference chunk_manager.hpp:24,30
    std::string error;

    // reference blok/src/chunk_manager.cpp:41-47 (floor division for negatives)
    int32_t chunk_of(int32_t g) const {
        const int32_t c = static_cast<int32_t>(C);
        return g >= 0 ? g / c : (g - c + 1) / c;
    }
    ChunkRec* chunk_at(int32_t cx, int32_t cy, int32_t cz, bool create) {
        auto it = chunks.find(ChunkKey{cz, cy, cx});
        if (it != chunks.end()) return it->second.get();
        if (!create) return nullptr;
        auto rec = std::make_unique<ChunkRec>();
        rec->cx = cx; rec->cy = cy; rec->cz = cz;
        rec->tree.depth = depth;
        rec->tree.reset();
        ChunkRec* raw = rec.get();
        chunks.emplace(ChunkKey{cz, cy, cx}, std::move(rec));
        order.clear();
        return raw;
    }
    void write(int32_t gx, int32_t gy, int32_t gz, uint32_t material, float density) {
        const int32_t cx = chunk_of(gx), cy = chunk_of(gy), cz = chunk_of(gz);
        ChunkRec* ch = chunk_at(cx, cy, cz, true);
        const uint32_t lx = static_cast<uint32_t>(gx - cx * static_cast<int32_t>(C));
        const uint32_t ly = static_cast<uint32_t>(gy - cy * static_cast<int32_t>(C));
        const uint32_t lz = static_cast<uint32_t>(gz - cz * static_cast<int32_t>(C));
        const uint32_t local = lx + ly * C + lz * C * C;   // reference chunk_manager.cpp:57-59
        if (!ch->log.empty() && local <= ch->log.back().local) ch->log_sorted = false;
        ch->log.push_back(VoxelWrite{local, static_cast<uint32_t>(ch->log.size()), material, density});
        ch->dirty = true;
    }
    static void settle(ChunkRec* ch) {  // sort by cell, keep the last write per cell
        if (ch->log_sorted) return;
        std::sort(ch->log.begin(), ch->log.end(), [](const VoxelWrite& a, const VoxelWrite& b) {
            return a.local != b.local ? a.local < b.local : a.seq < b.seq;
        });
        size_t out = 0;
        for (size_t i = 0; i < ch->log.size(); ++i) {
            if (i + 1 < ch->log.size() && ch->log[i + 1].local == ch->log[i].local) continue;
            ch->log[out] = ch->log[i];
            ch->log[out].seq = static_cast<uint32_t>(out);
            ++out;
        }
        ch->log.resize(out);
        ch->log_sorted = true;
    }
    const std::vector<ChunkRec*>& sorted() {
        if (order.size() != chunks.size()) {
            order.clear();
            for (auto& kv : chunks) order.push_back(kv.second.get());
        }
        return order;
    }
};

namespace {

// Descend `levels` octants towards sub-chunk (sx,sy,sz): reference
// blok/src/chunk_manager.cpp:144-193 (occupancy test) and :196-232 (root lookup) share this walk.
struct SubWalk { bool has_geometry; uint32_t root; };
SubWalk walk_to_sub_chunk(const std::vector<blok_svo_node>& nodes, uint32_t sx, uint32_t sy,
                          uint32_t sz, uint32_t levels) {
    SubWalk r{true, 0};
    uint32_t at = 0;           // for the occupancy test
    uint32_t root = 0;         // for the root lookup (ignores masks, like the reference)
    bool root_done = false;
    for (uint32_t level = 0; level < levels; ++level) {
        const uint32_t cell = kSubDivisions >> (level + 1);
        const uint32_t oct = ((sx / cell) & 1u) | (((sy / cell) & 1u) << 1) | (((sz / cell) & 1u) << 2);
        if (r.has_geometry) {
            const blok_svo_node& n = nodes[at];
            if (!(n.child_mask & (1u << oct)) || n.first_child == kNoChild) r.has_geometry = false;
            else {
                at = n.first_child + oct;
                if (at >= nodes.size()) r.has_geometry = false;
            }
        }
        if (!root_done) {
            const blok_svo_node& n = nodes[root];
            if (n.first_child == kNoChild) root_done = true;        // :220-222
            else {
                root = n.first_child + oct;
                if (root >= nodes.size()) { root = 0; root_done = true; }  // :226-228
            }
        }
    }
    if (r.has_geometry) {
        const blok_svo_node& n = nodes[at];
        r.has_geometry = n.child_mask != 0u || n.occupancy > 0.0f;   // :191-192
    }
    r.root = root;
    return r;
}

int fail(blok_world* w, const char* msg) {
    if (w) w->error = msg;
    return BLOK_ERR_INVALID_ARG;
}

}  // namespace

extern "C" {

uint64_t blok_morton_encode(int32_t x, int32_t y, int32_t z) { return morton_encode(x, y, z); }
void blok_morton_decode(uint64_t code, int32_t* x, int32_t* y, int32_t* z) {
    if (x) *x = static_cast<int32_t>(contract21(code)) - kMortonBias;
    if (y) *y = static_cast<int32_t>(contract21(code >> 1)) - kMortonBias;
    if (z) *z = static_cast<int32_t>(contract21(code >> 2)) - kMortonBias;
}
uint32_t blok_morton_octant(uint64_t code, uint32_t max_depth, uint32_t level) {
    return morton_octant(code, max_depth, level);
}

int blok_world_create(blok_world** out, uint32_t chunk_size, float voxel_size) {
    if (!out) return BLOK_ERR_INVALID_ARG;
    *out = nullptr;
    if (chunk_size < kSubDivisions || (chunk_size & (chunk_size - 1)) != 0 || chunk_size > 1024)
        return BLOK_ERR_INVALID_ARG;
    auto* w = new (std::nothrow) blok_world();
    if (!w) return BLOK_ERR_OOM;
    w->C = chunk_size;
    w->voxel_size = voxel_size;
    while ((1u << w->depth) < chunk_size) ++w->depth;   // reference chunk_manager.cpp:21-24
    *out = w;
    return BLOK_OK;
}
void blok_world_destroy(blok_world* w) { delete w; }
const char* blok_world_last_error(const blok_world* w) { return w ? w->error.c_str() : "null world"; }

int blok_world_set_voxel(blok_world* w, const float p[3], uint32_t material_id, float density) {
    if (!w || !p) return fail(w, "set_voxel: null argument");
    // reference chunk_manager.cpp:31-39: floor, 1:1 world->voxel mapping
    w->write(static_cast<int32_t>(std::floor(p[0])), static_cast<int32_t>(std::floor(p[1])),
             static_cast<int32_t>(std::floor(p[2])), material_id, density);
    return BLOK_OK;
}
void blok_world_set_material_library(blok_world* w, blok_material_library* lib) { if (w) w->material_lib = lib; }
blok_material_library* blok_world_get_material_library(const blok_world* w) { return w ? w->material_lib : nullptr; }

// reference blok/src/chunk_manager.cpp:91-102
int blok_world_set_voxel_rgb(blok_world* w, const float p[3], uint8_t r, uint8_t g, uint8_t b, float density) {
    if (!w || !p) return fail(w, "set_voxel_rgb: null argument");
    const uint32_t id = w->material_lib
        ? blok_material_library_from_color(w->material_lib, r, g, b)
        : (static_cast<uint32_t>(r) << 16) | (static_cast<uint32_t>(g) << 8) | static_cast<uint32_t>(b);
    return blok_world_set_voxel(w, p, id, density);
}

int blok_world_set_voxels(blok_world* w, const int32_t* xyz, const uint32_t* mats, size_t n) {
    if (!w || (n && (!xyz || !mats))) return fail(w, "set_voxels: null argument");
    for (size_t i = 0; i < n; ++i) w->write(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], mats[i], 1.0f);
    return BLOK_OK;
}
uint32_t blok_world_get_voxel_material(const blok_world* cw, const float p[3]) {
    if (!cw || !p) return 0;
    auto* w = const_cast<blok_world*>(cw);
    const int32_t gx = static_cast<int32_t>(std::floor(p[0]));
    const int32_t gy = static_cast<int32_t>(std::floor(p[1]));
    const int32_t gz = static_cast<int32_t>(std::floor(p[2]));
    const int32_t cx = w->chunk_of(gx), cy = w->chunk_of(gy), cz = w->chunk_of(gz);
    ChunkRec* ch = w->chunk_at(cx, cy, cz, false);
    if (!ch) return 0;
    blok_world::settle(ch);
    const int32_t c = static_cast<int32_t>(w->C);
    const uint32_t local = static_cast<uint32_t>(gx - cx * c) + static_cast<uint32_t>(gy - cy * c) * w->C +
                           static_cast<uint32_t>(gz - cz * c) * w->C * w->C;
    auto it = std::lower_bound(ch->log.begin(), ch->log.end(), local,
                               [](const VoxelWrite& a, uint32_t key) { return a.local < key; });
    if (it == ch->log.end() || it->local != local || !(it->density > 0.0f)) return 0;
    return it->material;
}

// reference blok/src/brush.cpp:13-63
int blok_world_apply_brush(blok_world* w, const float center[3], float radius, float value, int mode) {
    if (!w || !center || (mode != 0 && mode != 1)) return fail(w, "apply_brush: bad argument");
    int32_t lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = static_cast<int32_t>(std::floor(center[a] - radius));
        hi[a] = static_cast<int32_t>(std::floor(center[a] + radius)) + 1;        // brush.cpp:22
    }
    const int32_t C = static_cast<int32_t>(w->C);
    for (int32_t gz = lo[2]; gz < hi[2]; ++gz)
        for (int32_t gy = lo[1]; gy < hi[1]; ++gy)
            for (int32_t gx = lo[0]; gx < hi[0]; ++gx) {
                const int32_t cx = w->chunk_of(gx), cy = w->chunk_of(gy), cz = w->chunk_of(gz);
                ChunkRec* ch = w->chunk_at(cx, cy, cz, true);                      // created even if nothing is written
                const int32_t lx = gx - cx * C, ly = gy - cy * C, lz = gz - cz * C;
                const float ox = static_cast<float>(cx * C) * w->voxel_size, oy = static_cast<float>(cy * C) * w->voxel_size,
                            oz = static_cast<float>(cz * C) * w->voxel_size;
                const float dx = (ox + (static_cast<float>(lx) + 0.5f * w->voxel_size)) - center[0];
                const float dy = (oy + (static_cast<float>(ly) + 0.5f * w->voxel_size)) - center[1];
                const float dz = (oz + (static_cast<float>(lz) + 0.5f * w->voxel_size)) - center[2];
                if (std::sqrt(dx * dx + dy * dy + dz * dz) > radius) continue;
                blok_world::settle(ch);
                const uint32_t local = static_cast<uint32_t>(lx) + static_cast<uint32_t>(ly) * w->C + static_cast<uint32_t>(lz) * w->C * w->C;
                auto it = std::lower_bound(ch->log.begin(), ch->log.end(), local,
                                           [](const VoxelWrite& a, uint32_t key) { return a.local < key; });
                const bool have = it != ch->log.end() && it->local == local;
                const float d = have ? it->density : 0.0f;
                const float nd = mode == 0 ? std::max(d, value) : std::min(d, value);
                if (have) it->density = nd;                                        // in place: the log stays sorted
                else ch->log.insert(it, VoxelWrite{local, 0u, 0u, nd});            // material id 0, like a fresh dense cell
                ch->dirty = true;
            }
    return BLOK_OK;
}

int blok_world_rebuild_dirty(blok_world* w, int max_per_frame) {
    if (!w) return BLOK_ERR_INVALID_ARG;
    int rebuilt = 0;
    for (ChunkRec* ch : w->sorted()) {
        if (!ch->dirty) continue;
        if (rebuilt >= max_per_frame) break;           // reference chunk_manager.cpp:125-126
        blok_world::settle(ch);
        ch->tree.reset();
        const uint32_t C = w->C;
        for (const VoxelWrite& v : ch->log)
            ch->tree.insert(v.local % C, (v.local / C) % C, v.local / (C * C), v.material, v.density);
        ch->dirty = false;
        ++rebuilt;
    }
    return rebuilt;
}

int blok_world_pack(blok_world* w) {
    if (!w) return BLOK_ERR_INVALID_ARG;
    w->packed_nodes.clear();
    w->packed_subs.clear();
    uint32_t levels = 0;
    while ((1u << levels) < kSubDivisions) ++levels;    // reference chunk_manager.cpp:246-247
    uint64_t offset = 0;
    for (ChunkRec* ch : w->sorted()) {
        const auto& nodes = ch->tree.nodes;
        if (nodes.empty()) continue;
        if (offset + nodes.size() > 0xFFFFFFFFull) return fail(w, "pack: more than 2^32 nodes");
        // reference chunk_manager.cpp:255-263
        float origin[3] = {static_cast<float>(ch->cx * static_cast<int32_t>(w->C)),
                           static_cast<float>(ch->cy * static_cast<int32_t>(w->C)),
                           static_cast<float>(ch->cz * static_cast<int32_t>(w->C))};
        for (float& o : origin) o *= w->voxel_size;
        const float chunk_world = static_cast<float>(w->C) * w->voxel_size;
        const float sub_world = chunk_world / static_cast<float>(kSubDivisions);
        for (uint32_t sz = 0; sz < kSubDivisions; ++sz)
            for (uint32_t sy = 0; sy < kSubDivisions; ++sy)
                for (uint32_t sx = 0; sx < kSubDivisions; ++sx) {
                    const SubWalk sw = walk_to_sub_chunk(nodes, sx, sy, sz, levels);
                    if (!sw.has_geometry) continue;
                    blok_sub_chunk s{};
                    s.node_offset = static_cast<uint32_t>(offset);
                    s.root_node_index = sw.root;
                    s.node_count = static_cast<uint32_t>(nodes.size());
                    s.start_depth = levels;
                    s.world_min[0] = origin[0] + static_cast<float>(sx) * sub_world;
                    s.world_min[1] = origin[1] + static_cast<float>(sy) * sub_world;
                    s.world_min[2] = origin[2] + static_cast<float>(sz) * sub_world;
                    s.sub_chunk_size = sub_world;
                    for (int a = 0; a < 3; ++a) s.world_max[a] = s.world_min[a] + sub_world;
                    w->packed_subs.push_back(s);
                }
        w->packed_nodes.insert(w->packed_nodes.end(), nodes.begin(), nodes.end());
        offset += nodes.size();
    }
    return BLOK_OK;
}

size_t blok_world_node_count(const blok_world* w) { return w ? w->packed_nodes.size() : 0; }
size_t blok_world_sub_chunk_count(const blok_world* w) { return w ? w->packed_subs.size() : 0; }
const blok_svo_node* blok_world_nodes(const blok_world* w) { return w ? w->packed_nodes.data() : nullptr; }
const blok_sub_chunk* blok_world_sub_chunks(const blok_world* w) { return w ? w->packed_subs.data() : nullptr; }

size_t blok_world_chunk_count(const blok_world* w) { return w ? w->chunks.size() : 0; }
int blok_world_chunk_info(const blok_world* cw, size_t i, int32_t coord[3], uint64_t* n_nodes) {
    auto* w = const_cast<blok_world*>(cw);
    if (!w || i >= w->chunks.size()) return BLOK_ERR_INVALID_ARG;
    ChunkRec* ch = w->sorted()[i];
    if (coord) { coord[0] = ch->cx; coord[1] = ch->cy; coord[2] = ch->cz; }
    if (n_nodes) *n_nodes = ch->tree.nodes.size();
    return BLOK_OK;
}
const blok_svo_node* blok_world_chunk_nodes(const blok_world* cw, size_t i) {
    auto* w = const_cast<blok_world*>(cw);
    if (!w || i >= w->chunks.size()) return nullptr;
    return w->sorted()[i]->tree.nodes.data();
}
int64_t blok_world_find_leaf(const blok_world* cw, size_t i, uint32_t x, uint32_t y, uint32_t z) {
    auto* w = const_cast<blok_world*>(cw);
    if (!w || i >= w->chunks.size()) return -1;
    return w->sorted()[i]->tree.find_leaf(x, y, z);
}

}  // extern "C"
