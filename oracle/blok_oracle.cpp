// blok_oracle.cpp — CPU restatement of the reference's voxel trace path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing shipped may import, link or call this file: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load liboracle.so.
//
// What is restated, and from where (paths relative to the reference tree):
//   morton3d            blok/include/morton.hpp:12-58
//   SvoTree             blok/src/svo.cpp:36-130, blok/include/svo.hpp:20-46
//   ChunkManager        blok/src/chunk_manager.cpp:19-140,316-348 (dense per-chunk store, literal)
//   packChunksToGpuSvo  blok/src/chunk_manager.cpp:144-314 (chunks visited in sorted (cz,cy,cx)
//                       order; the reference's unordered_map order is unspecified)
//   intersect.rint      assets/shaders/intersect.rint:47-68,70-206 (line for line)
//   traceRayEXT         procedural-hit acceptance: a reported t is accepted iff
//                       tmin <= t <= current tmax, and then becomes the current tmax;
//                       closest hit = last accepted (Vulkan ray-tracing pipeline semantics,
//                       call site assets/shaders/raygen.rgen:217-229)
//   hit.rchit           assets/shaders/hit.rchit:46-76 (face LUT, material fetch, unpack)
//   primary ray         blok/src/cuda_tracer.cu:276-282 with zero jitter (the compute backend's
//                       matrix-free form; algebraically assets/shaders/raygen.rgen:201-205)
//
// PARITY PIN STATUS.  The reference holds no tests, golden vectors or fixtures for this path
// (SURVEY.md §4, §8c), and nothing on it except morton.hpp compiles here: svo.hpp includes
// <glm.hpp> and chunk_manager.hpp pulls vulkan.hpp/vk_mem_alloc.h, all absent from the tree
// and the image, and the GLSL shaders have no compiler or device.  Therefore:
//   - morton: PINNED against the reference header compiled in place (oracle/_ref, see Makefile);
//   - everything else: "parity unpinned" — a line-by-line restatement, cross-validated only
//     against independent brute-force formulations in this file (trace_voxels_bruteforce,
//     get_voxel_material point queries).
//
// Build: g++ -O2 -ffp-contract=off (no fast-math): every float op below is one IEEE-754
// binary32 operation, in the order written.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <tuple>
#include <vector>

namespace {

// ---------------------------------------------------------------- records (reference layouts)
struct SvoNode {            // blok/include/svo.hpp:23-28
    uint32_t childMask;
    uint32_t firstChild;
    uint32_t materialId;
    float    occupancy;
};
static_assert(sizeof(SvoNode) == 16, "SvoNode is 16 bytes");

struct SubChunkGpu {        // blok/include/resources.hpp:170-184
    uint32_t nodeOffset, rootNodeIndex, nodeCount, startDepth;
    float worldMin[3];
    float subChunkSize;
    float worldMax[3];
    float pad0;
};
static_assert(sizeof(SubChunkGpu) == 48, "SubChunkGpu is 48 bytes");

struct MaterialGpu {        // blok/include/material.hpp:88-114
    float albedo[3];
    uint32_t flags;
    float emission[3];
    float ior;
};
static_assert(sizeof(MaterialGpu) == 32, "MaterialGpu is 32 bytes");

struct Camera {             // blok/src/cuda_tracer.cu:51-58
    float pos[3], fwd[3], right[3], up[3];
    float fovScale, aspect;
};

struct Ray { float org[3]; float tmin; float dir[3]; float tmax; };

struct Hit {                // include/blok_hip.h: blok_hit
    float t;
    uint32_t materialId;
    int16_t voxel[3];
    uint8_t face;
    uint8_t hit;
};
static_assert(sizeof(Hit) == 16, "hit record is 16 bytes");

// Per-visit accounting (defines the algorithmic byte count).  ORC_NO_COUNTERS: the timing build of bench.py's
// cpu_baseline (same traversal, -O3 -march=native) drops the per-node / per-box increments; rays and hits stay.
#ifdef ORC_NO_COUNTERS
#define ORC_COUNT(stmt) do { } while (0)
#else
#define ORC_COUNT(stmt) do { stmt; } while (0)
#endif

struct Counters {
    uint64_t rays, hits;
    uint64_t subChunksEntered;   // S: intersection-shader invocations whose root slab test passed
    uint64_t nodesFetched;       // P: fetches at intersect.rint:133
    uint64_t iterLimitHits;      // times the MAX_ITER guard ended a walk with work left
    uint64_t stackLimitHits;     // pushes dropped by the MAX_STACK guard
    uint64_t maxStack, maxIter;
    uint64_t ties;               // brute-force formulations: rays whose minimum t is not unique
};

constexpr uint32_t INVALID_NODE_INDEX = 0xFFFFFFFFu;    // svo.hpp:20

// ---------------------------------------------------------------- morton (morton.hpp:12-58)
uint64_t spreadBits(uint32_t v) {
    uint64_t x = v & 0x1fffff;
    x = (x | (x << 32)) & 0x1f00000000ffffULL;
    x = (x | (x << 16)) & 0x1f0000ff0000ffULL;
    x = (x | (x << 8)) & 0x100f00f00f00f00fULL;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ULL;
    x = (x | (x << 2)) & 0x1249249249249249ULL;
    return x;
}
uint64_t mortonEncode(int32_t x, int32_t y, int32_t z) {
    const int32_t BIAS = 1 << 20;
    uint32_t xs = x + BIAS, ys = y + BIAS, zs = z + BIAS;
    return spreadBits(xs) | (spreadBits(ys) << 1) | (spreadBits(zs) << 2);
}
uint32_t compactBits(uint64_t v) {
    v &= 0x1249249249249249ULL;
    v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ULL;
    v = (v ^ (v >> 4)) & 0x100f00f00f00f00fULL;
    v = (v ^ (v >> 8)) & 0x1f0000ff0000ffULL;
    v = (v ^ (v >> 16)) & 0x1f00000000ffffULL;
    v = (v ^ (v >> 32)) & 0x1fffffULL;
    return static_cast<uint32_t>(v);
}
uint32_t octantFromCode(uint64_t code, uint32_t maxDepth, uint32_t level) {
    return static_cast<uint32_t>((code >> (3u * (maxDepth - 1u - level))) & 0x7ull);
}

// ---------------------------------------------------------------- SvoTree (svo.cpp)
struct SvoTree {
    std::vector<SvoNode> nodes;
    uint32_t maxDepth = 0;

    static SvoNode emptyNode() { return SvoNode{0u, INVALID_NODE_INDEX, 0u, 0.0f}; }   // :11-18
    void clear() { nodes.clear(); nodes.push_back(emptyNode()); }                       // :27-31

    uint32_t ensureChildren(uint32_t nodeIndex) {                                       // :36-57
        if (nodes[nodeIndex].firstChild != INVALID_NODE_INDEX) return nodes[nodeIndex].firstChild;
        const uint32_t firstChild = static_cast<uint32_t>(nodes.size());
        nodes.resize(nodes.size() + 8);
        for (uint32_t i = 0; i < 8; ++i) nodes[firstChild + i] = emptyNode();
        nodes[nodeIndex].firstChild = firstChild;
        return firstChild;
    }
    void insertVoxel(uint32_t x, uint32_t y, uint32_t z, uint32_t materialId, float density) {  // :59-101
        if (density <= 0.0f) return;
        const uint32_t dim = 1u << maxDepth;
        if (x >= dim || y >= dim || z >= dim) return;
        const uint64_t code = mortonEncode(x, y, z);
        uint32_t nodeIndex = 0;
        uint32_t pathNode[32], pathOct[32];
        for (uint32_t level = 0; level < maxDepth; ++level) {
            pathNode[level] = nodeIndex;
            const uint32_t oct = octantFromCode(code, maxDepth, level);
            pathOct[level] = oct;
            nodeIndex = ensureChildren(nodeIndex) + oct;
        }
        nodes[nodeIndex].materialId = materialId;
        nodes[nodeIndex].occupancy = density;
        for (int level = static_cast<int>(maxDepth) - 1; level >= 0; --level)
            nodes[pathNode[level]].childMask |= (1u << pathOct[level]);
    }
    int64_t findLeaf(uint32_t x, uint32_t y, uint32_t z) const {                         // :103-130
        const uint32_t dim = 1u << maxDepth;
        if (x >= dim || y >= dim || z >= dim) return -1;
        const uint64_t code = mortonEncode(x, y, z);
        uint32_t nodeIndex = 0;
        for (uint32_t level = 0; level < maxDepth; ++level) {
            const uint32_t oct = octantFromCode(code, maxDepth, level);
            const SvoNode& node = nodes[nodeIndex];
            if ((node.childMask & (1u << oct)) == 0u) return -1;
            if (node.firstChild == INVALID_NODE_INDEX) return -1;
            nodeIndex = node.firstChild + oct;
        }
        if (nodes[nodeIndex].occupancy <= 0.0f) return -1;
        return nodeIndex;
    }
};

// ---------------------------------------------------------------- ChunkManager (chunk_manager.cpp)
constexpr uint32_t SUB_CHUNK_DIVISIONS = 8;     // :17

struct Chunk {                                   // chunk.hpp:33-42
    int32_t cx, cy, cz;
    std::vector<float> density;
    std::vector<uint32_t> materialIds;
    bool dirty = true;
    SvoTree svo;
};

struct World {
    uint32_t C; float voxelSize; uint32_t maxDepth;
    std::map<std::tuple<int32_t, int32_t, int32_t>, Chunk*> chunks;   // key (cz,cy,cx)
    std::vector<SvoNode> globalNodes;
    std::vector<SubChunkGpu> globalSubChunks;

    World(uint32_t C_, float vs) : C(C_), voxelSize(vs), maxDepth(0) {
        while ((1u << maxDepth) < C) maxDepth++;                        // :21-24
    }
    ~World() { for (auto& kv : chunks) delete kv.second; }

    int32_t toChunk(int32_t g) const {                                  // :41-47
        const int32_t c = static_cast<int32_t>(C);
        return g >= 0 ? g / c : (g - c + 1) / c;
    }
    Chunk* getOrCreate(int32_t cx, int32_t cy, int32_t cz) {            // :61-75
        auto key = std::make_tuple(cz, cy, cx);
        auto it = chunks.find(key);
        if (it != chunks.end()) return it->second;
        Chunk* ch = new Chunk();
        ch->cx = cx; ch->cy = cy; ch->cz = cz;
        ch->density.assign(static_cast<size_t>(C) * C * C, 0.0f);
        ch->materialIds.assign(static_cast<size_t>(C) * C * C, 0u);
        ch->svo.maxDepth = maxDepth;
        ch->svo.clear();
        chunks[key] = ch;
        return ch;
    }
    void setVoxelMaterial(float px, float py, float pz, uint32_t materialId, float density) {   // :316-328
        const int32_t gx = int(std::floor(px)), gy = int(std::floor(py)), gz = int(std::floor(pz));
        const int32_t cx = toChunk(gx), cy = toChunk(gy), cz = toChunk(gz);
        const int32_t lx = gx - cx * int32_t(C), ly = gy - cy * int32_t(C), lz = gz - cz * int32_t(C);
        Chunk* ch = getOrCreate(cx, cy, cz);
        const size_t idx = size_t(lx) + size_t(ly) * C + size_t(lz) * C * C;   // :57-59
        ch->density[idx] = density;
        ch->materialIds[idx] = materialId;
        ch->dirty = true;
    }
    uint32_t getVoxelMaterial(float px, float py, float pz) const {    // :330-348
        const int32_t gx = int(std::floor(px)), gy = int(std::floor(py)), gz = int(std::floor(pz));
        const int32_t cx = toChunk(gx), cy = toChunk(gy), cz = toChunk(gz);
        auto it = chunks.find(std::make_tuple(cz, cy, cx));
        if (it == chunks.end()) return 0;
        const int32_t lx = gx - cx * int32_t(C), ly = gy - cy * int32_t(C), lz = gz - cz * int32_t(C);
        const size_t idx = size_t(lx) + size_t(ly) * C + size_t(lz) * C * C;
        const Chunk* ch = it->second;
        if (ch->density[idx] <= 0.0f) return 0;
        return ch->materialIds[idx];
    }
    int rebuildDirtyChunks(int maxPerFrame) {                           // :106-140
        int count = 0;
        for (auto& kv : chunks) {
            Chunk* ch = kv.second;
            if (!ch->dirty) continue;
            if (count >= maxPerFrame) break;
            ch->svo.clear();
            for (uint32_t z = 0; z < C; ++z)
                for (uint32_t y = 0; y < C; ++y)
                    for (uint32_t x = 0; x < C; ++x) {
                        const size_t idx = x + size_t(y) * C + size_t(z) * C * C;
                        const float d = ch->density[idx];
                        if (d > 0.0f) ch->svo.insertVoxel(x, y, z, ch->materialIds[idx], d);
                    }
            ch->dirty = false;
            count++;
        }
        return count;
    }

    // applyBrush, brush.cpp:13-63 (mode 0 ADD, 1 SUBTRACT; brush.hpp:14-21)
    void applyBrush(float cxw, float cyw, float czw, float radiusWS, float value, int mode) {
        const int32_t gvMin[3] = {int(std::floor(cxw - radiusWS)), int(std::floor(cyw - radiusWS)), int(std::floor(czw - radiusWS))};
        const int32_t gvMax[3] = {int(std::floor(cxw + radiusWS)) + 1, int(std::floor(cyw + radiusWS)) + 1, int(std::floor(czw + radiusWS)) + 1};
        for (int gz = gvMin[2]; gz < gvMax[2]; gz++)
        for (int gy = gvMin[1]; gy < gvMax[1]; gy++)
        for (int gx = gvMin[0]; gx < gvMax[0]; gx++) {
            const int32_t cx = toChunk(gx), cy = toChunk(gy), cz = toChunk(gz);
            Chunk* ch = getOrCreate(cx, cy, cz);
            const int32_t lx = gx - cx * int32_t(C), ly = gy - cy * int32_t(C), lz = gz - cz * int32_t(C);
            if (lx < 0 || lx >= int(C) || ly < 0 || ly >= int(C) || lz < 0 || lz >= int(C)) continue;
            // ch->svo.origin = chunk coordinate * C * voxelSize (chunk_manager.cpp:66-70)
            const float ox = static_cast<float>(cx * int32_t(C)) * voxelSize, oy = static_cast<float>(cy * int32_t(C)) * voxelSize,
                        oz = static_cast<float>(cz * int32_t(C)) * voxelSize;
            const float vx = ox + (static_cast<float>(lx) + 0.5f * voxelSize), vy = oy + (static_cast<float>(ly) + 0.5f * voxelSize),
                        vz = oz + (static_cast<float>(lz) + 0.5f * voxelSize);
            const float dx = vx - cxw, dy = vy - cyw, dz = vz - czw;
            const float dist = std::sqrt(dx * dx + dy * dy + dz * dz);           // glm::distance
            if (dist > radiusWS) continue;
            float& d = ch->density[size_t(lx) + size_t(ly) * C + size_t(lz) * C * C];
            if (mode == 0) d = std::max(d, value); else d = std::min(d, value);
            ch->dirty = true;
        }
    }

    static bool subChunkHasGeometry(const std::vector<SvoNode>& nodes, uint32_t subX, uint32_t subY,
                                    uint32_t subZ, uint32_t subDivisions) {     // :144-193
        if (nodes.empty()) return false;
        uint32_t subChunkDepth = 0;
        while ((1u << subChunkDepth) < subDivisions) subChunkDepth++;
        uint32_t nodeIndex = 0;
        for (uint32_t level = 0; level < subChunkDepth; level++) {
            const SvoNode& node = nodes[nodeIndex];
            const uint32_t levelDivisions = 1u << (level + 1);
            const uint32_t cellSize = subDivisions / levelDivisions;
            const uint32_t octant = ((subX / cellSize) & 1) | (((subY / cellSize) & 1) << 1) |
                                    (((subZ / cellSize) & 1) << 2);
            if ((node.childMask & (1u << octant)) == 0) return false;
            if (node.firstChild == 0xFFFFFFFFu) return false;
            nodeIndex = node.firstChild + octant;
            if (nodeIndex >= nodes.size()) return false;
        }
        const SvoNode& subRoot = nodes[nodeIndex];
        return subRoot.childMask != 0 || subRoot.occupancy > 0.0f;
    }
    static uint32_t findSubChunkRootNode(const std::vector<SvoNode>& nodes, uint32_t subX, uint32_t subY,
                                         uint32_t subZ, uint32_t subDivisions) { // :196-232
        if (nodes.empty()) return 0;
        uint32_t subChunkDepth = 0;
        while ((1u << subChunkDepth) < subDivisions) subChunkDepth++;
        uint32_t nodeIndex = 0;
        for (uint32_t level = 0; level < subChunkDepth; level++) {
            const SvoNode& node = nodes[nodeIndex];
            const uint32_t levelDivisions = 1u << (level + 1);
            const uint32_t cellSize = subDivisions / levelDivisions;
            const uint32_t octant = ((subX / cellSize) & 1) | (((subY / cellSize) & 1) << 1) |
                                    (((subZ / cellSize) & 1) << 2);
            if (node.firstChild == 0xFFFFFFFFu) return nodeIndex;
            nodeIndex = node.firstChild + octant;
            if (nodeIndex >= nodes.size()) return 0;
        }
        return nodeIndex;
    }
    void pack() {                                                       // :234-314
        globalNodes.clear();
        globalSubChunks.clear();
        uint32_t nodeOffset = 0;
        uint32_t subChunkDepth = 0;
        while ((1u << subChunkDepth) < SUB_CHUNK_DIVISIONS) subChunkDepth++;
        for (auto& kv : chunks) {
            const Chunk* ch = kv.second;
            const auto& nodes = ch->svo.nodes;
            if (nodes.empty()) continue;
            float chunkOrigin[3] = {static_cast<float>(ch->cx * static_cast<int32_t>(C)),
                                    static_cast<float>(ch->cy * static_cast<int32_t>(C)),
                                    static_cast<float>(ch->cz * static_cast<int32_t>(C))};
            for (float& o : chunkOrigin) o *= voxelSize;
            const float chunkWorldSize = static_cast<float>(C) * voxelSize;
            const float subChunkWorldSize = chunkWorldSize / static_cast<float>(SUB_CHUNK_DIVISIONS);
            for (uint32_t sz = 0; sz < SUB_CHUNK_DIVISIONS; sz++)
                for (uint32_t sy = 0; sy < SUB_CHUNK_DIVISIONS; sy++)
                    for (uint32_t sx = 0; sx < SUB_CHUNK_DIVISIONS; sx++) {
                        if (!subChunkHasGeometry(nodes, sx, sy, sz, SUB_CHUNK_DIVISIONS)) continue;
                        const uint32_t subRootNode = findSubChunkRootNode(nodes, sx, sy, sz, SUB_CHUNK_DIVISIONS);
                        SubChunkGpu sub{};
                        sub.nodeOffset = nodeOffset;
                        sub.rootNodeIndex = subRootNode;
                        sub.nodeCount = static_cast<uint32_t>(nodes.size());
                        sub.startDepth = subChunkDepth;
                        sub.worldMin[0] = chunkOrigin[0] + static_cast<float>(sx) * subChunkWorldSize;
                        sub.worldMin[1] = chunkOrigin[1] + static_cast<float>(sy) * subChunkWorldSize;
                        sub.worldMin[2] = chunkOrigin[2] + static_cast<float>(sz) * subChunkWorldSize;
                        sub.subChunkSize = subChunkWorldSize;
                        for (int a = 0; a < 3; ++a) sub.worldMax[a] = sub.worldMin[a] + subChunkWorldSize;
                        globalSubChunks.push_back(sub);
                    }
            globalNodes.insert(globalNodes.end(), nodes.begin(), nodes.end());
            nodeOffset += static_cast<uint32_t>(nodes.size());
        }
    }
};

// ---------------------------------------------------------------- intersect.rint
constexpr uint32_t MAX_STACK = 24u;     // :42
constexpr uint32_t MAX_ITER = 256u;     // :43

struct Vec3 { float x, y, z; };
inline Vec3 sub3(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 add3(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 mul3(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 scale3(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 min3v(Vec3 a, Vec3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline Vec3 max3v(Vec3 a, Vec3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }

// :47-55
inline void intersectAABB_Root(Vec3 origin, Vec3 invDir, Vec3 boxMin, Vec3 boxMax, float& tNear, float& tFar) {
    const Vec3 t0 = mul3(sub3(boxMin, origin), invDir);
    const Vec3 t1 = mul3(sub3(boxMax, origin), invDir);
    const Vec3 tmin = min3v(t0, t1), tmax = max3v(t0, t1);
    tNear = std::fmax(std::fmax(tmin.x, tmin.y), tmin.z);
    tFar = std::fmin(std::fmin(tmax.x, tmax.y), tmax.z);
}
// :58-68
inline uint32_t getHitFace(Vec3 hitPos, Vec3 center) {
    const Vec3 diff = sub3(hitPos, center);
    const Vec3 a = {std::fabs(diff.x), std::fabs(diff.y), std::fabs(diff.z)};
    if (a.x >= a.y && a.x >= a.z) return diff.x > 0.0f ? 0u : 1u;
    if (a.y >= a.z) return diff.y > 0.0f ? 2u : 3u;
    return diff.z > 0.0f ? 4u : 5u;
}
// :79   1.0 / mix(rayDir, vec3(1e-6), lessThan(abs(rayDir), vec3(1e-6)))
inline Vec3 safeInvDir(Vec3 d) {
    auto one = [](float c) { return 1.0f / (std::fabs(c) < 1e-6f ? 1e-6f : c); };
    return {one(d.x), one(d.y), one(d.z)};
}

// State of one traceRayEXT: the live interval and the committed procedural hit.
struct RayQuery {
    Vec3 org, dir;
    float tmin, tmax;        // gl_RayTminEXT, gl_RayTmaxEXT (tmax shrinks on every accepted hit)
    bool committed = false;
    float hitT = -1.0f;
    uint32_t hitKind = 0xFF, materialId = 0;
    Vec3 leafCenter{0, 0, 0};
    float leafSize = 1.0f;       // edge of the reported leaf = ChunkManager's voxelSize
    // reportIntersectionEXT
    bool report(float t, uint32_t kind, uint32_t mat, Vec3 center, float size = 1.0f) {
        if (!(t >= tmin && t <= tmax)) return false;
        tmax = t; committed = true; hitT = t; hitKind = kind; materialId = mat; leafCenter = center; leafSize = size;
        return true;
    }
};

struct StackItem { uint32_t nodeIndex; Vec3 center; float halfSize, tEntry, tExit; };  // :100-106

// main(), :70-206.  Returns true if the root slab test passed (the invocation "entered" the AABB).
bool intersectSubChunk(RayQuery& q, const SubChunkGpu& sub, const SvoNode* nodes, Counters& c) {
    const Vec3 rayOrg = q.org, rayDir = q.dir;
    const Vec3 invDir = safeInvDir(rayDir);
    const Vec3 wmin = {sub.worldMin[0], sub.worldMin[1], sub.worldMin[2]};
    const Vec3 wmax = {sub.worldMax[0], sub.worldMax[1], sub.worldMax[2]};
    float rootNear, rootFar;
    intersectAABB_Root(rayOrg, invDir, wmin, wmax, rootNear, rootFar);          // :82
    const float tMin = std::fmax(rootNear, q.tmin);                               // :85
    const float tMax = std::fmin(rootFar, q.tmax);                                // :86
    if (tMin > tMax) return false;                                                // :88

    uint32_t octantMask = 0u;                                                     // :94-97
    if (rayDir.x < 0.0f) octantMask |= 1u;
    if (rayDir.y < 0.0f) octantMask |= 2u;
    if (rayDir.z < 0.0f) octantMask |= 4u;

    StackItem stack[MAX_STACK];
    uint32_t stackPtr = 0u;
    const Vec3 rootSize = sub3(wmax, wmin);                                       // :112
    stack[stackPtr++] = StackItem{sub.nodeOffset + sub.rootNodeIndex,
                                  add3(wmin, scale3(rootSize, 0.5f)), rootSize.x * 0.5f, tMin, tMax};
    uint32_t iter = 0u;
    while (stackPtr > 0u && iter++ < MAX_ITER) {                                  // :123
        const StackItem item = stack[--stackPtr];
        if (item.tEntry >= q.tmax) continue;                                      // :128
        if (item.nodeIndex >= sub.nodeOffset + sub.nodeCount) continue;           // :132
        const SvoNode node = nodes[item.nodeIndex];                               // :133
        ORC_COUNT(c.nodesFetched++);
        if (node.childMask == 0u) {                                               // :136
            if (node.occupancy > 0.0f) {
                const Vec3 hitPos = add3(rayOrg, scale3(rayDir, item.tEntry));    // :138
                const uint32_t faceID = getHitFace(hitPos, item.center);
                q.report(item.tEntry, faceID, node.materialId, item.center, item.halfSize * 2.0f);      // :139-141
            }
            continue;
        }
        // :148-160 compute tPlane/xIn/yIn/zIn/spans, none of which is read afterwards.
        const float nextHalf = item.halfSize * 0.5f;                              // :162
        for (int i = 7; i >= 0; i--) {                                            // :164
            const uint32_t childIdx = uint32_t(i) ^ octantMask;
            if ((node.childMask & (1u << childIdx)) == 0u) continue;              // :169
            Vec3 childOff;
            childOff.x = (childIdx & 1u) != 0u ? item.halfSize : -item.halfSize;  // :173-175
            childOff.y = (childIdx & 2u) != 0u ? item.halfSize : -item.halfSize;
            childOff.z = (childIdx & 4u) != 0u ? item.halfSize : -item.halfSize;
            const Vec3 childCenter = add3(item.center, scale3(childOff, 0.5f));   // :177
            const Vec3 nh = {nextHalf, nextHalf, nextHalf};
            const Vec3 tC0 = mul3(sub3(sub3(childCenter, nh), rayOrg), invDir);   // :179
            const Vec3 tC1 = mul3(sub3(add3(childCenter, nh), rayOrg), invDir);   // :180
            const Vec3 tMinV = min3v(tC0, tC1), tMaxV = max3v(tC0, tC1);
            float cTmin = std::fmax(std::fmax(tMinV.x, tMinV.y), tMinV.z);        // :185
            float cTmax = std::fmin(std::fmin(tMaxV.x, tMaxV.y), tMaxV.z);        // :186
            cTmin = std::fmax(cTmin, item.tEntry);                                // :189
            cTmax = std::fmin(cTmax, item.tExit);                                 // :190
            if (cTmin < cTmax) {                                                  // :193
                if (stackPtr < MAX_STACK) {
                    stack[stackPtr++] = StackItem{sub.nodeOffset + node.firstChild + childIdx,
                                                  childCenter, nextHalf, cTmin, cTmax};
                    ORC_COUNT(if (stackPtr > c.maxStack) c.maxStack = stackPtr);
                } else {
                    ORC_COUNT(c.stackLimitHits++);
                }
            }
        }
    }
    ORC_COUNT(if (iter > c.maxIter) c.maxIter = std::min(iter, MAX_ITER));
    ORC_COUNT(if (stackPtr > 0u) c.iterLimitHits++);
    return true;
}

inline void commitHit(const RayQuery& q, Hit& out) {
    if (!q.committed) {                                   // miss.rmiss:25-27
        out.t = -1.0f; out.materialId = 0; out.voxel[0] = out.voxel[1] = out.voxel[2] = 0;
        out.face = 0xFF; out.hit = 0;
        return;
    }
    out.t = q.hitT;                                       // hit.rchit:74
    out.materialId = q.materialId;                        // hit.rchit:62 input
    // the record's voxel is the leaf's index on the voxel lattice: floor(centre / voxelSize) (= floor(centre) for the app's size 1)
    out.voxel[0] = static_cast<int16_t>(std::floor(q.leafCenter.x / q.leafSize));
    out.voxel[1] = static_cast<int16_t>(std::floor(q.leafCenter.y / q.leafSize));
    out.voxel[2] = static_cast<int16_t>(std::floor(q.leafCenter.z / q.leafSize));
    out.face = static_cast<uint8_t>(q.hitKind);           // hit.rchit:58
    out.hit = 1;
}

// -------- candidate generation ---------------------------------------------------------------
// The reference leaves "which sub-chunk AABBs does the ray touch" to RT hardware.  Two stand-ins:
//  (1) brute force: every sub-chunk, in array order (a BVH-less device) — the literal baseline;
//  (2) a lattice of sub-chunk slots walked front to back — scalable, and it defines S_r
//      (descriptors entered up to the hit) for the algorithmic byte count.
struct Lattice {
    float cell = 16.0f;
    int32_t origin[3] = {0, 0, 0};     // world coordinate of slot (0,0,0), in voxels
    int32_t dims[3] = {0, 0, 0};
    std::vector<int32_t> slot;         // sub-chunk index or -1
    int32_t at(int32_t x, int32_t y, int32_t z) const {
        if (x < 0 || y < 0 || z < 0 || x >= dims[0] || y >= dims[1] || z >= dims[2]) return -1;
        return slot[size_t(x) + size_t(dims[0]) * (size_t(y) + size_t(dims[1]) * size_t(z))];
    }
};

// t of the axis-aligned plane at world coordinate p: same two operations, same order, as every
// slab test in intersect.rint ((plane - rayOrg) * invDir, :48-49,:179-180).
inline float planeT(float p, float o, float inv) { return (p - o) * inv; }

void traceLattice(const Lattice& L, const SvoNode* nodes, const SubChunkGpu* subs, RayQuery& q, Counters& c) {
    const Vec3 inv = safeInvDir(q.dir);
    const float o[3] = {q.org.x, q.org.y, q.org.z};
    const float iv[3] = {inv.x, inv.y, inv.z};
    // entry/exit of the lattice box
    float tEnter = -INFINITY, tExit = INFINITY;
    for (int a = 0; a < 3; ++a) {
        const float lo = planeT(float(L.origin[a]), o[a], iv[a]);
        const float hi = planeT(float(L.origin[a]) + L.cell * float(L.dims[a]), o[a], iv[a]);
        tEnter = std::fmax(tEnter, std::fmin(lo, hi));
        tExit = std::fmin(tExit, std::fmax(lo, hi));
    }
    const float tStart = std::fmax(tEnter, q.tmin);
    if (!(tStart <= std::fmin(tExit, q.tmax))) return;
    // starting slot: per axis, count lattice planes already crossed at tStart (in ray order)
    int32_t idx[3], step[3];
    float tNext[3];
    for (int a = 0; a < 3; ++a) {
        step[a] = iv[a] > 0.0f ? 1 : -1;
        // largest j in [0, dims-1] whose j-th interior plane (ray order) has t <= tStart; the
        // plane t's are non-decreasing in ray order, so bisect
        int32_t lo = 0, hi = L.dims[a] - 1;
        while (lo < hi) {
            const int32_t j = (lo + hi + 1) / 2;
            const int32_t planeIdx = step[a] > 0 ? j : L.dims[a] - j;
            if (planeT(float(L.origin[a]) + L.cell * float(planeIdx), o[a], iv[a]) <= tStart) lo = j; else hi = j - 1;
        }
        const int32_t crossed = lo;
        idx[a] = step[a] > 0 ? crossed : L.dims[a] - 1 - crossed;
        const int32_t farPlane = step[a] > 0 ? idx[a] + 1 : idx[a];
        tNext[a] = planeT(float(L.origin[a]) + L.cell * float(farPlane), o[a], iv[a]);
    }
    float tCur = tStart;
    for (;;) {
        if (tCur >= q.tmax) break;     // later slots start at or beyond the committed hit (:128 drops them)
        const int32_t s = L.at(idx[0], idx[1], idx[2]);
        if (s >= 0 && intersectSubChunk(q, subs[s], nodes, c)) ORC_COUNT(c.subChunksEntered++);
        int a = 0;
        if (tNext[1] < tNext[a]) a = 1;
        if (tNext[2] < tNext[a]) a = 2;
        tCur = tNext[a];
        idx[a] += step[a];
        if (idx[a] < 0 || idx[a] >= L.dims[a]) break;
        const int32_t farPlane = step[a] > 0 ? idx[a] + 1 : idx[a];
        tNext[a] = planeT(float(L.origin[a]) + L.cell * float(farPlane), o[a], iv[a]);
    }
}

// The frame's TAA jitter in clip space, (2 jx / width, 2 jy / height): what getJitteredProjection adds to proj[2][0..1]
// (blok/src/renderer_postprocess.cpp:234-241,254-268).  With proj' the NDC of a view-space point is the un-jittered NDC minus
// the jitter, so the ray raygen.rgen:201-205 forms for NDC d under inverse(proj') is the un-jittered ray of d + jitter (y:
// before the Vulkan flip).  Set per call sequence by the tests (orc_set_jitter_clip); 0 = TAA off (:255-257).
float g_jitterClip[2] = {0.0f, 0.0f};

// Primary ray, blok/src/cuda_tracer.cu:276-282 (its jx = jy = 0) in the NDC form of raygen.rgen:201-205.
inline Ray primaryRay(const Camera& cam, uint32_t width, uint32_t height, uint32_t x, uint32_t y) {
    const float u = ((2.0f * ((float(x) + 0.5f) / float(width)) - 1.0f) + g_jitterClip[0]) * cam.fovScale * cam.aspect;
    const float v = ((1.0f - 2.0f * ((float(y) + 0.5f) / float(height))) - g_jitterClip[1]) * cam.fovScale;
    const Vec3 f = {cam.fwd[0], cam.fwd[1], cam.fwd[2]}, r = {cam.right[0], cam.right[1], cam.right[2]},
               up = {cam.up[0], cam.up[1], cam.up[2]};
    const Vec3 d = add3(add3(f, scale3(r, u)), scale3(up, v));
    const float len = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);            // normalize3, :61-64
    Ray ray;
    ray.org[0] = cam.pos[0]; ray.org[1] = cam.pos[1]; ray.org[2] = cam.pos[2];
    ray.dir[0] = d.x / len; ray.dir[1] = d.y / len; ray.dir[2] = d.z / len;
    ray.tmin = 0.001f;                                                           // raygen.rgen:225
    ray.tmax = 10000.0f;                                                         // raygen.rgen:227
    return ray;
}

inline RayQuery makeQuery(const Ray& r) {
    RayQuery q;
    q.org = {r.org[0], r.org[1], r.org[2]};
    q.dir = {r.dir[0], r.dir[1], r.dir[2]};
    q.tmin = r.tmin; q.tmax = r.tmax;
    return q;
}

void addCounters(Counters& dst, const Counters& src) {
    dst.rays += src.rays; dst.hits += src.hits;
    dst.subChunksEntered += src.subChunksEntered; dst.nodesFetched += src.nodesFetched;
    dst.iterLimitHits += src.iterLimitHits; dst.stackLimitHits += src.stackLimitHits;
    dst.maxStack = std::max(dst.maxStack, src.maxStack); dst.maxIter = std::max(dst.maxIter, src.maxIter);
    dst.ties += src.ties;
}

template <class Fn>
void parallelFor(size_t n, int threads, Fn&& fn) {
    threads = std::max(1, threads);
    if (threads == 1) { fn(0, size_t(0), n); return; }
    std::vector<std::thread> pool;
    std::atomic<size_t> next{0};
    const size_t grain = std::max<size_t>(256, n / (size_t(threads) * 16));
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            for (;;) {
                const size_t b = next.fetch_add(grain);
                if (b >= n) break;
                fn(t, b, std::min(n, b + grain));
            }
        });
    for (auto& th : pool) th.join();
}

}  // namespace

// =============================================================================== C interface
extern "C" {

void orc_set_jitter_clip(float jx_clip, float jy_clip) { g_jitterClip[0] = jx_clip; g_jitterClip[1] = jy_clip; }

uint64_t orc_morton_encode(int32_t x, int32_t y, int32_t z) { return mortonEncode(x, y, z); }
void orc_morton_decode(uint64_t code, int32_t* x, int32_t* y, int32_t* z) {   // morton.hpp:46-53
    const int32_t BIAS = 1 << 20;
    *x = static_cast<int32_t>(compactBits(code)) - BIAS;
    *y = static_cast<int32_t>(compactBits(code >> 1)) - BIAS;
    *z = static_cast<int32_t>(compactBits(code >> 2)) - BIAS;
}
uint32_t orc_morton_octant(uint64_t code, uint32_t maxDepth, uint32_t level) {
    return octantFromCode(code, maxDepth, level);
}

// ---- world (ChunkManager) ----
void* orc_world_new(uint32_t C, float voxelSize) { return new World(C, voxelSize); }
void orc_world_free(void* w) { delete static_cast<World*>(w); }
void orc_world_set_voxel(void* w, float x, float y, float z, uint32_t mat, float density) {
    static_cast<World*>(w)->setVoxelMaterial(x, y, z, mat, density);
}
void orc_world_set_voxels(void* w, const int32_t* xyz, const uint32_t* mats, size_t n) {
    World* W = static_cast<World*>(w);
    for (size_t i = 0; i < n; ++i)
        W->setVoxelMaterial(float(xyz[3 * i]), float(xyz[3 * i + 1]), float(xyz[3 * i + 2]), mats[i], 1.0f);
}
uint32_t orc_world_get_voxel_material(const void* w, float x, float y, float z) {
    return static_cast<const World*>(w)->getVoxelMaterial(x, y, z);
}
void orc_world_apply_brush(void* w, float x, float y, float z, float radius, float value, int mode) {
    static_cast<World*>(w)->applyBrush(x, y, z, radius, value, mode);
}
int orc_world_rebuild(void* w, int maxPerFrame) { return static_cast<World*>(w)->rebuildDirtyChunks(maxPerFrame); }
void orc_world_pack(void* w) { static_cast<World*>(w)->pack(); }
size_t orc_world_n_nodes(const void* w) { return static_cast<const World*>(w)->globalNodes.size(); }
size_t orc_world_n_subs(const void* w) { return static_cast<const World*>(w)->globalSubChunks.size(); }
const void* orc_world_nodes(const void* w) { return static_cast<const World*>(w)->globalNodes.data(); }
const void* orc_world_subs(const void* w) { return static_cast<const World*>(w)->globalSubChunks.data(); }
size_t orc_world_n_chunks(const void* w) { return static_cast<const World*>(w)->chunks.size(); }
int orc_world_chunk_info(const void* w, size_t i, int32_t coord[3], uint64_t* nNodes) {
    const World* W = static_cast<const World*>(w);
    if (i >= W->chunks.size()) return -1;
    auto it = W->chunks.begin();
    std::advance(it, i);
    coord[0] = it->second->cx; coord[1] = it->second->cy; coord[2] = it->second->cz;
    *nNodes = it->second->svo.nodes.size();
    return 0;
}
// test accessor: the dense arrays of chunk i (chunk.hpp:35-36), [x + y*C + z*C*C]
int orc_world_chunk_dense(const void* w, size_t i, const float** density, const uint32_t** ids) {
    const World* W = static_cast<const World*>(w);
    if (i >= W->chunks.size()) return -1;
    auto it = W->chunks.begin();
    std::advance(it, i);
    *density = it->second->density.data(); *ids = it->second->materialIds.data();
    return 0;
}
const void* orc_world_chunk_nodes(const void* w, size_t i) {
    const World* W = static_cast<const World*>(w);
    if (i >= W->chunks.size()) return nullptr;
    auto it = W->chunks.begin();
    std::advance(it, i);
    return it->second->svo.nodes.data();
}
int64_t orc_world_find_leaf(const void* w, size_t i, uint32_t x, uint32_t y, uint32_t z) {
    const World* W = static_cast<const World*>(w);
    if (i >= W->chunks.size()) return -1;
    auto it = W->chunks.begin();
    std::advance(it, i);
    return it->second->svo.findLeaf(x, y, z);
}

// ---- rays ----
void orc_primary_rays(const void* cam, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0,
                      uint32_t w, uint32_t h, uint32_t stride, void* raysOut) {
    const Camera& c = *static_cast<const Camera*>(cam);
    Ray* out = static_cast<Ray*>(raysOut);
    size_t k = 0;
    for (uint32_t y = y0; y < y0 + h; y += stride)
        for (uint32_t x = x0; x < x0 + w; x += stride) out[k++] = primaryRay(c, width, height, x, y);
}

// ---- trace: literal, every sub-chunk in array order ----
void orc_trace_bruteforce(const void* nodes, const void* subs, size_t nSubs, const void* rays, size_t nRays,
                          void* hitsOut, void* countersOut) {
    const SvoNode* N = static_cast<const SvoNode*>(nodes);
    const SubChunkGpu* S = static_cast<const SubChunkGpu*>(subs);
    const Ray* R = static_cast<const Ray*>(rays);
    Hit* H = static_cast<Hit*>(hitsOut);
    Counters c{};
    for (size_t i = 0; i < nRays; ++i) {
        RayQuery q = makeQuery(R[i]);
        for (size_t s = 0; s < nSubs; ++s)
            if (intersectSubChunk(q, S[s], N, c)) c.subChunksEntered++;
        commitHit(q, H[i]);
        c.rays++; c.hits += H[i].hit;
    }
    if (countersOut) *static_cast<Counters*>(countersOut) = c;
}

// ---- trace: literal shader behind a front-to-back lattice of sub-chunk slots ----
void* orc_lattice_build(const void* subs, size_t nSubs) {
    const SubChunkGpu* S = static_cast<const SubChunkGpu*>(subs);
    Lattice* L = new Lattice();
    if (nSubs == 0) { L->dims[0] = L->dims[1] = L->dims[2] = 1; L->slot.assign(1, -1); return L; }
    L->cell = S[0].subChunkSize;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t s = 0; s < nSubs; ++s)
        for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], S[s].worldMin[a]); hi[a] = std::fmax(hi[a], S[s].worldMax[a]); }
    for (int a = 0; a < 3; ++a) {
        L->origin[a] = int32_t(lo[a]);
        L->dims[a] = int32_t((hi[a] - lo[a]) / L->cell + 0.5f);
    }
    L->slot.assign(size_t(L->dims[0]) * L->dims[1] * L->dims[2], -1);
    for (size_t s = 0; s < nSubs; ++s) {
        int32_t i[3];
        for (int a = 0; a < 3; ++a) i[a] = int32_t((S[s].worldMin[a] - lo[a]) / L->cell + 0.5f);
        L->slot[size_t(i[0]) + size_t(L->dims[0]) * (size_t(i[1]) + size_t(L->dims[1]) * size_t(i[2]))] = int32_t(s);
    }
    return L;
}
void orc_lattice_free(void* L) { delete static_cast<Lattice*>(L); }

void orc_trace_lattice(const void* lattice, const void* nodes, const void* subs, const void* rays, size_t nRays,
                       void* hitsOut, void* countersOut, int threads) {
    const Lattice& L = *static_cast<const Lattice*>(lattice);
    const SvoNode* N = static_cast<const SvoNode*>(nodes);
    const SubChunkGpu* S = static_cast<const SubChunkGpu*>(subs);
    const Ray* R = static_cast<const Ray*>(rays);
    Hit* H = static_cast<Hit*>(hitsOut);
    std::vector<Counters> per(std::max(1, threads), Counters{});
    parallelFor(nRays, threads, [&](int t, size_t b, size_t e) {
        Counters& c = per[t];
        for (size_t i = b; i < e; ++i) {
            RayQuery q = makeQuery(R[i]);
            traceLattice(L, N, S, q, c);
            commitHit(q, H[i]);
            c.rays++; c.hits += H[i].hit;
        }
    });
    if (countersOut) {
        Counters total{};
        for (auto& c : per) addCounters(total, c);
        *static_cast<Counters*>(countersOut) = total;
    }
}

// Primary rays generated on the fly (no ray buffer): the CPU-baseline entry.  Traces the pixels
// (x0 + i*stride, y0 + j*stride) of the rectangle, row-major.
void orc_trace_primary(const void* lattice, const void* nodes, const void* subs, const void* cam,
                       uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                       uint32_t stride, void* hitsOut, void* countersOut, int threads) {
    const Lattice& L = *static_cast<const Lattice*>(lattice);
    const SvoNode* N = static_cast<const SvoNode*>(nodes);
    const SubChunkGpu* S = static_cast<const SubChunkGpu*>(subs);
    const Camera& C = *static_cast<const Camera*>(cam);
    Hit* H = static_cast<Hit*>(hitsOut);
    const uint32_t cols = (w + stride - 1) / stride, rows = (h + stride - 1) / stride;
    std::vector<Counters> per(std::max(1, threads), Counters{});
    parallelFor(size_t(cols) * rows, threads, [&](int t, size_t b, size_t e) {
        Counters& c = per[t];
        for (size_t i = b; i < e; ++i) {
            const uint32_t px = x0 + uint32_t(i % cols) * stride, py = y0 + uint32_t(i / cols) * stride;
            RayQuery q = makeQuery(primaryRay(C, width, height, px, py));
            traceLattice(L, N, S, q, c);
            Hit hit;
            commitHit(q, hit);
            if (H) H[i] = hit;
            c.rays++; c.hits += hit.hit;
        }
    });
    if (countersOut) {
        Counters total{};
        for (auto& c : per) addCounters(total, c);
        *static_cast<Counters*>(countersOut) = total;
    }
}

// ---- independent formulation: minimum over ALL filled voxels of the leaf-level slab test ----
// A filled unit voxel [v, v+1]^3 is reported by intersect.rint iff its own clipped interval is
// non-empty (the intervals of its ancestors contain it), with t = max(slab entry, tmin); the
// closest reported one wins.  Quadratic; tiny scenes only.  Counts rays whose minimum is not unique.
void orc_trace_voxels_bruteforce(const int32_t* xyz, const uint32_t* mats, size_t nVox, const void* rays,
                                 size_t nRays, void* hitsOut, void* countersOut) {
    const Ray* R = static_cast<const Ray*>(rays);
    Hit* H = static_cast<Hit*>(hitsOut);
    Counters c{};
    for (size_t i = 0; i < nRays; ++i) {
        const Vec3 org = {R[i].org[0], R[i].org[1], R[i].org[2]}, dir = {R[i].dir[0], R[i].dir[1], R[i].dir[2]};
        const Vec3 inv = safeInvDir(dir);
        float best = INFINITY; size_t bestV = 0; uint32_t nBest = 0;
        for (size_t v = 0; v < nVox; ++v) {
            const Vec3 lo = {float(xyz[3 * v]), float(xyz[3 * v + 1]), float(xyz[3 * v + 2])};
            const Vec3 hi = {lo.x + 1.0f, lo.y + 1.0f, lo.z + 1.0f};
            float tn, tf;
            intersectAABB_Root(org, inv, lo, hi, tn, tf);
            const float a = std::fmax(tn, R[i].tmin), b = std::fmin(tf, R[i].tmax);
            if (!(a < b)) continue;
            if (a < best) { best = a; bestV = v; nBest = 1; }
            else if (a == best) nBest++;
        }
        Hit& out = H[i];
        if (nBest == 0) { out.t = -1.0f; out.materialId = 0; out.voxel[0] = out.voxel[1] = out.voxel[2] = 0; out.face = 0xFF; out.hit = 0; }
        else {
            const Vec3 center = {float(xyz[3 * bestV]) + 0.5f, float(xyz[3 * bestV + 1]) + 0.5f, float(xyz[3 * bestV + 2]) + 0.5f};
            out.t = best; out.materialId = mats[bestV];
            out.voxel[0] = int16_t(xyz[3 * bestV]); out.voxel[1] = int16_t(xyz[3 * bestV + 1]); out.voxel[2] = int16_t(xyz[3 * bestV + 2]);
            out.face = uint8_t(getHitFace(add3(org, scale3(dir, best)), center));
            out.hit = 1;
            if (nBest > 1) c.ties++;
        }
        c.rays++; c.hits += out.hit;
    }
    if (countersOut) *static_cast<Counters*>(countersOut) = c;
}

// ---- hit.rchit: surface record for a hit (normal, albedo, roughness, metallic, emission) ----
void orc_shade_surface(const void* hit, const void* materials, float out[12]) {
    static const float FACE_NORMALS[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};  // :46-53
    const Hit& h = *static_cast<const Hit*>(hit);
    const MaterialGpu& mat = static_cast<const MaterialGpu*>(materials)[std::min(h.materialId, 65535u)];   // :62
    const float metallic = float((mat.flags >> 24) & 0xFFu) / 255.0f;             // :65
    const float roughness = float((mat.flags >> 16) & 0xFFu) / 255.0f;            // :66
    for (int a = 0; a < 3; ++a) out[a] = FACE_NORMALS[h.face][a];
    for (int a = 0; a < 3; ++a) out[3 + a] = mat.albedo[a];
    out[6] = std::fmax(roughness, 0.04f);                                          // :72
    out[7] = metallic;
    out[8] = h.t;
    for (int a = 0; a < 3; ++a) out[9 + a] = mat.emission[a];
}

}  // extern "C"

// =============================================================================================
// raygen.rgen:77-165,167-414 — RNG, sampling, sky, the per-pixel sample/bounce loop and the G-buffer.
// Vector helpers spell out GLSL's definitions: dot = x*x'+y*y'+z*z' left to right, normalize = v / sqrt(dot),
// mix(a,b,t) = a*(1-t) + b*t, reflect(I,N) = I - 2*dot(N,I)*N.  The primary ray uses the camera-basis form
// (see primaryRay) with the reference's pixel-centre jitter (raygen.rgen:193-202).
// Reference formats not reproduced: outNormalRoughness is RGBA16F, outAlbedoMetallic RGBA8, motion vectors RG16F
// (raygen.rgen:57-59); here every plane is float4 and motion vectors are not produced (they need prevViewProj,
// i.e. glm matrices that are absent from the reference tree).
namespace {

inline float dot3(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 normalize3(Vec3 v) { const float len = std::sqrt(dot3(v, v)); return {v.x / len, v.y / len, v.z / len}; }
inline Vec3 cross3(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline Vec3 mix3(Vec3 a, Vec3 b, float t) { return add3(scale3(a, 1.0f - t), scale3(b, t)); }
inline Vec3 neg3(Vec3 a) { return {-a.x, -a.y, -a.z}; }

constexpr float PI = 3.14159265359f;         // :74
constexpr float INV_PI = 0.31830988618f;     // :75

inline uint32_t pcg(uint32_t& state) {        // :77-82
    const uint32_t oldState = state;
    state = oldState * 747796405u + 2891336453u;
    const uint32_t word = ((oldState >> ((oldState >> 28u) + 4u)) ^ oldState) * 277803737u;
    return (word >> 22u) ^ word;
}
inline float randomFloat(uint32_t& state) { return float(pcg(state)) / 4294967295.0f; }      // :84-86
inline uint32_t initRNG(uint32_t px, uint32_t py, uint32_t screenWidth, uint32_t frameIndex, uint32_t sampleIndex) {  // :92-99
    uint32_t seed = px + py * screenWidth;
    seed ^= frameIndex * 747796405u;
    seed ^= sampleIndex * 1664525u;
    pcg(seed);
    pcg(seed);
    return seed;
}
inline Vec3 sampleCosineHemisphere(float ux, float uy, Vec3 N) {      // :101-113
    const float r = std::sqrt(ux);
    const float phi = 2.0f * PI * uy;
    const float x = r * std::cos(phi);
    const float y = r * std::sin(phi);
    const float z = std::sqrt(std::fmax(0.0f, 1.0f - ux));
    const Vec3 up = std::fabs(N.z) < 0.999f ? Vec3{0.0f, 0.0f, 1.0f} : Vec3{1.0f, 0.0f, 0.0f};
    const Vec3 tangent = normalize3(cross3(up, N));
    const Vec3 bitangent = cross3(N, tangent);
    return normalize3(add3(add3(scale3(tangent, x), scale3(bitangent, y)), scale3(N, z)));
}
inline Vec3 sampleGGX(float ux, float uy, Vec3 N, float roughness) {   // :115-130
    const float a = roughness * roughness;
    const float a2 = a * a;
    const float phi = 2.0f * PI * ux;
    const float cosTheta = std::sqrt((1.0f - uy) / (1.0f + (a2 - 1.0f) * uy));
    const float sinTheta = std::sqrt(std::fmax(0.0f, 1.0f - cosTheta * cosTheta));
    const Vec3 H = {sinTheta * std::cos(phi), sinTheta * std::sin(phi), cosTheta};
    const Vec3 up = std::fabs(N.z) < 0.999f ? Vec3{0.0f, 0.0f, 1.0f} : Vec3{1.0f, 0.0f, 0.0f};
    const Vec3 tangent = normalize3(cross3(up, N));
    const Vec3 bitangent = cross3(N, tangent);
    return normalize3(add3(add3(scale3(tangent, H.x), scale3(bitangent, H.y)), scale3(N, H.z)));
}
inline Vec3 fresnelSchlick(float cosTheta, Vec3 F0) {                   // :132-134
    const float w = std::pow(std::fmax(1.0f - cosTheta, 0.0f), 5.0f);
    return {F0.x + (1.0f - F0.x) * w, F0.y + (1.0f - F0.y) * w, F0.z + (1.0f - F0.z) * w};
}
inline Vec3 sunDirection() { return normalize3({0.5f, 0.8f, 0.3f}); }  // :142,185
inline Vec3 getSkyColor(Vec3 dir) {                                     // :136-148
    const float t = 0.5f * (dir.y + 1.0f);
    const Vec3 skyColor = mix3({0.8f, 0.85f, 0.95f}, {0.4f, 0.6f, 0.9f}, t);
    const float sunDot = std::fmax(dot3(dir, sunDirection()), 0.0f);
    const Vec3 sunColor = scale3(scale3({1.0f, 0.95f, 0.8f}, std::pow(sunDot, 128.0f)), 5.0f);
    const Vec3 sunGlow = scale3(scale3({1.0f, 0.9f, 0.7f}, std::pow(sunDot, 8.0f)), 0.3f);
    return add3(add3(skyColor, sunColor), sunGlow);
}
inline bool isEmissive(Vec3 e) { return dot3(e, {1.0f, 1.0f, 1.0f}) > 0.01f; }                 // :158-160
inline float luminance(Vec3 c) { return dot3(c, {0.2126f, 0.7152f, 0.0722f}); }                // :163-165

struct Payload { Vec3 radiance, normal, albedo; float roughness, metallic, hitT; };           // :62-69

// traceRayEXT + hit.rchit / miss.rmiss for the radiance payload (:217-229, hit.rchit:55-76, miss.rmiss:25-27)
inline void traceRadiance(const Lattice& L, const SvoNode* nodes, const SubChunkGpu* subs, const MaterialGpu* materials,
                          Vec3 org, Vec3 dir, Payload& payload, Counters& c) {
    RayQuery q;
    q.org = org; q.dir = dir; q.tmin = 0.001f; q.tmax = 10000.0f;
    traceLattice(L, nodes, subs, q, c);
    c.rays++;
    if (!q.committed) { payload.hitT = -1.0f; return; }
    c.hits++;
    static const Vec3 FACE_NORMALS[6] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    const MaterialGpu& mat = materials[std::min(q.materialId, 65535u)];
    payload.normal = FACE_NORMALS[q.hitKind];
    payload.albedo = {mat.albedo[0], mat.albedo[1], mat.albedo[2]};
    payload.roughness = std::fmax(float((mat.flags >> 16) & 0xFFu) / 255.0f, 0.04f);
    payload.metallic = float((mat.flags >> 24) & 0xFFu) / 255.0f;
    payload.hitT = q.hitT;
    payload.radiance = {mat.emission[0], mat.emission[1], mat.emission[2]};
}
// shadow ray: TerminateOnFirstHit | SkipClosestHit, miss shader clears the flag (:283-298, shadow.rmiss:13-15)
inline bool traceShadow(const Lattice& L, const SvoNode* nodes, const SubChunkGpu* subs, Vec3 org, Vec3 dir, Counters& c) {
    RayQuery q;
    q.org = org; q.dir = dir; q.tmin = 0.001f; q.tmax = 1000.0f;
    traceLattice(L, nodes, subs, q, c);
    c.rays++;
    return q.committed;
}

struct GBufferPixel { float color[4], worldPos[4], normalRoughness[4], albedoMetallic[4]; };

GBufferPixel shadePixel(const Lattice& L, const SvoNode* nodes, const SubChunkGpu* subs, const MaterialGpu* materials,
                        const Camera& cam, uint32_t width, uint32_t height, uint32_t px, uint32_t py,
                        uint32_t sampleCount, uint32_t maxBounces, uint32_t frameCount, Counters& c) {
    const Vec3 camPos = {cam.pos[0], cam.pos[1], cam.pos[2]};
    const Vec3 camF = {cam.fwd[0], cam.fwd[1], cam.fwd[2]}, camR = {cam.right[0], cam.right[1], cam.right[2]},
               camU = {cam.up[0], cam.up[1], cam.up[2]};
    Vec3 firstHitPos{0, 0, 0}, firstHitNormal{0, 0, 0}, firstHitAlbedo{0, 0, 0}, firstHitEmission{0, 0, 0};   // :173-181
    float firstHitRoughness = 0.0f, firstHitMetallic = 0.0f, firstHitDepth = 0.0f;
    bool hadFirstHit = false, firstHitWasEmissive = false;
    Vec3 accumulatedColor{0, 0, 0};
    const Vec3 sunDir = sunDirection();
    const Vec3 sunRadiance = {3.0f, 2.9f, 2.7f};

    for (uint32_t sampleIdx = 0u; sampleIdx < sampleCount; sampleIdx++) {                                     // :188
        uint32_t rng = initRNG(px, py, width, frameCount, sampleIdx);
        float pcx, pcy;                                                                                        // :193-199
        if (sampleIdx == 0u) { pcx = float(px) + 0.5f; pcy = float(py) + 0.5f; }
        else {
            const float jx = randomFloat(rng) - 0.5f;
            const float jy = randomFloat(rng) - 0.5f;
            pcx = float(px) + 0.5f + jx * 0.5f;
            pcy = float(py) + 0.5f + jy * 0.5f;
        }
        // :201-206 in camera-basis form (cuda_tracer.cu:276-282)
        const float u = ((2.0f * (pcx / float(width)) - 1.0f) + g_jitterClip[0]) * cam.fovScale * cam.aspect;
        const float v = ((1.0f - 2.0f * (pcy / float(height))) - g_jitterClip[1]) * cam.fovScale;
        Vec3 rayDir = normalize3(add3(add3(camF, scale3(camR, u)), scale3(camU, v)));
        Vec3 rayOrigin = camPos;
        Vec3 radiance{0, 0, 0}, throughput{1, 1, 1};

        for (uint32_t bounce = 0u; bounce < maxBounces; bounce++) {                                            // :212
            Payload payload{};
            payload.radiance = {0, 0, 0};
            payload.hitT = -1.0f;
            traceRadiance(L, nodes, subs, materials, rayOrigin, rayDir, payload, c);
            if (payload.hitT < 0.0f) {                                                                         // :232-235
                radiance = add3(radiance, mul3(throughput, getSkyColor(rayDir)));
                break;
            }
            const Vec3 hitPos = add3(rayOrigin, scale3(rayDir, payload.hitT));                                 // :238
            Vec3 N = payload.normal;
            const Vec3 albedo = payload.albedo, emission = payload.radiance;
            const float roughness = payload.roughness, metallic = payload.metallic;
            if (dot3(N, rayDir) > 0.0f) N = neg3(N);                                                           // :248-250
            if (bounce == 0u && sampleIdx == 0u && !hadFirstHit) {                                             // :253-263
                hadFirstHit = true;
                firstHitPos = hitPos; firstHitNormal = N; firstHitAlbedo = albedo; firstHitEmission = emission;
                firstHitRoughness = roughness; firstHitMetallic = metallic; firstHitDepth = payload.hitT;
                firstHitWasEmissive = isEmissive(emission);
            }
            if (isEmissive(emission)) {                                                                        // :265-277
                radiance = add3(radiance, mul3(throughput, emission));
                if (luminance(emission) > 5.0f || bounce > 0u) break;
            }
            const float NdotL = std::fmax(dot3(N, sunDir), 0.0f);                                              // :280
            if (NdotL > 0.0f && bounce == 0u) {
                const Vec3 shadowOrigin = add3(hitPos, scale3(N, 0.001f));
                const bool isShadowed = traceShadow(L, nodes, subs, shadowOrigin, sunDir, c);
                if (!isShadowed) {                                                                             // :300-325
                    const Vec3 diffuseColor = scale3(albedo, 1.0f - metallic);
                    radiance = add3(radiance, scale3(scale3(mul3(mul3(throughput, diffuseColor), sunRadiance), NdotL), INV_PI));
                    if (roughness < 0.9f) {
                        const Vec3 H = normalize3(sub3(sunDir, rayDir));
                        const float NdotH = std::fmax(dot3(N, H), 0.0f);
                        const float VdotH = std::fmax(dot3(neg3(rayDir), H), 0.0f);
                        const float a = roughness * roughness;
                        const float a2 = a * a;
                        const float denom = NdotH * NdotH * (a2 - 1.0f) + 1.0f;
                        const float D = a2 / (PI * denom * denom);
                        const Vec3 F0 = mix3({0.04f, 0.04f, 0.04f}, albedo, metallic);
                        const Vec3 F = fresnelSchlick(VdotH, F0);
                        const float Vis = 0.25f;
                        radiance = add3(radiance, scale3(mul3(scale3(scale3(mul3(throughput, F), D), Vis), sunRadiance), NdotL));
                    }
                }
            }
            if (bounce > 0u) {                                                                                 // :329-335
                const float p = std::fmin(std::fmax(std::fmax(throughput.x, throughput.y), throughput.z), 0.95f);
                if (randomFloat(rng) > p) break;
                throughput = {throughput.x / p, throughput.y / p, throughput.z / p};
            }
            const float ux = randomFloat(rng);                                                                 // :338
            const float uy = randomFloat(rng);
            const Vec3 F0 = mix3({0.04f, 0.04f, 0.04f}, albedo, metallic);                                     // :341-347
            const Vec3 V = neg3(rayDir);
            const float NdotV = std::fmax(dot3(N, V), 0.001f);
            const Vec3 F = fresnelSchlick(NdotV, F0);
            float specularWeight = (F.x + F.y + F.z) / 3.0f;
            specularWeight = specularWeight * (1.0f - metallic) + 1.0f * metallic;
            if (randomFloat(rng) < specularWeight) {                                                           // :349-360
                const Vec3 H = sampleGGX(ux, uy, N, std::fmax(roughness, 0.04f));
                const Vec3 newDir = sub3(rayDir, scale3(H, 2.0f * dot3(H, rayDir)));
                if (dot3(newDir, N) <= 0.0f) break;
                const float HdotV = std::fmax(dot3(H, V), 0.0f);
                const Vec3 Fh = fresnelSchlick(HdotV, F0);
                const float w = std::fmax(specularWeight, 0.001f);
                throughput = mul3(throughput, {Fh.x / w, Fh.y / w, Fh.z / w});
                rayDir = newDir;
            } else {                                                                                           // :361-367
                const Vec3 newDir = sampleCosineHemisphere(ux, uy, N);
                const Vec3 diffuseColor = scale3(albedo, 1.0f - metallic);
                const float w = std::fmax(1.0f - specularWeight, 0.001f);
                throughput = mul3(throughput, {diffuseColor.x / w, diffuseColor.y / w, diffuseColor.z / w});
                rayDir = newDir;
            }
            const float maxThroughput = std::fmax(std::fmax(throughput.x, throughput.y), throughput.z);       // :370-373
            if (maxThroughput > 10.0f) throughput = scale3(throughput, 10.0f / maxThroughput);
            rayOrigin = add3(hitPos, scale3(N, 0.002f));                                                       // :376
        }
        accumulatedColor = add3(accumulatedColor, radiance);                                                   // :379
    }
    const float n = float(sampleCount);
    Vec3 color = {accumulatedColor.x / n, accumulatedColor.y / n, accumulatedColor.z / n};                    // :383
    const float maxVal = std::fmax(std::fmax(color.x, color.y), color.z);                                     // :386-389
    if (maxVal > 100.0f) color = scale3(color, 100.0f / maxVal);
    if (!hadFirstHit) {                                                                                        // :395-400
        firstHitDepth = 10000.0f;
        firstHitPos = add3(camPos, scale3(normalize3(camF), 10000.0f));
        firstHitNormal = {0.0f, 1.0f, 0.0f};
        firstHitAlbedo = getSkyColor(normalize3(sub3(firstHitPos, camPos)));
    }
    const Vec3 finalAlbedo = firstHitWasEmissive ? firstHitEmission : firstHitAlbedo;                         // :403
    GBufferPixel g;
    g.color[0] = color.x; g.color[1] = color.y; g.color[2] = color.z; g.color[3] = 1.0f;                      // :392
    g.worldPos[0] = firstHitPos.x; g.worldPos[1] = firstHitPos.y; g.worldPos[2] = firstHitPos.z; g.worldPos[3] = firstHitDepth;
    g.normalRoughness[0] = firstHitNormal.x; g.normalRoughness[1] = firstHitNormal.y; g.normalRoughness[2] = firstHitNormal.z;
    g.normalRoughness[3] = firstHitRoughness;
    g.albedoMetallic[0] = finalAlbedo.x; g.albedoMetallic[1] = finalAlbedo.y; g.albedoMetallic[2] = finalAlbedo.z;
    g.albedoMetallic[3] = firstHitMetallic;
    return g;
}

}  // namespace

extern "C" {

// Renders the pixels (x0 + i*stride, y0 + j*stride) of the rectangle; four float4 planes, row-major.
void orc_render_paths(const void* lattice, const void* nodes, const void* subs, const void* materials, const void* cam,
                      uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t stride,
                      uint32_t sampleCount, uint32_t maxBounces, uint32_t frameCount,
                      float* color, float* worldPos, float* normalRoughness, float* albedoMetallic,
                      void* countersOut, int threads) {
    const Lattice& L = *static_cast<const Lattice*>(lattice);
    const SvoNode* N = static_cast<const SvoNode*>(nodes);
    const SubChunkGpu* S = static_cast<const SubChunkGpu*>(subs);
    const MaterialGpu* M = static_cast<const MaterialGpu*>(materials);
    const Camera& C = *static_cast<const Camera*>(cam);
    const uint32_t cols = (w + stride - 1) / stride, rows = (h + stride - 1) / stride;
    std::vector<Counters> per(std::max(1, threads), Counters{});
    parallelFor(size_t(cols) * rows, threads, [&](int t, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const uint32_t px = x0 + uint32_t(i % cols) * stride, py = y0 + uint32_t(i / cols) * stride;
            const GBufferPixel g = shadePixel(L, N, S, M, C, width, height, px, py, sampleCount, maxBounces, frameCount, per[t]);
            std::memcpy(color + 4 * i, g.color, 16);
            std::memcpy(worldPos + 4 * i, g.worldPos, 16);
            std::memcpy(normalRoughness + 4 * i, g.normalRoughness, 16);
            std::memcpy(albedoMetallic + 4 * i, g.albedoMetallic, 16);
        }
    });
    if (countersOut) {
        Counters total{};
        for (auto& c : per) addCounters(total, c);
        *static_cast<Counters*>(countersOut) = total;
    }
}

}  // extern "C"

extern "C" {

// cuda_tracer.cu:372-386 (progressive accumulation), :209-216 (ACES fit), :95-99 (gamma 2.2 to 8 bits)
extern "C" void orc_accumulate(float* accum, const float* color, uint32_t n, uint32_t* pixels) {
    auto toSRGB8 = [](float x) {
        x = std::fmin(std::fmax(x, 0.0f), 1.0f);
        const float g = std::pow(x, 1.0f / 2.2f);
        return uint32_t((unsigned char)(g * 255.0f + 0.5f));
    };
    auto aces = [](float x) {
        const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
        return std::fmin(std::fmax((x * (a * x + b)) / (x * (c * x + d) + e), 0.f), 1.f);
    };
    for (uint32_t idx = 0; idx < n; ++idx) {
        float sumx = accum[4 * idx], sumy = accum[4 * idx + 1], sumz = accum[4 * idx + 2], spp = accum[4 * idx + 3];
        sumx = sumx + color[4 * idx]; sumy = sumy + color[4 * idx + 1]; sumz = sumz + color[4 * idx + 2];
        spp += 1.f;
        accum[4 * idx] = sumx; accum[4 * idx + 1] = sumy; accum[4 * idx + 2] = sumz; accum[4 * idx + 3] = spp;
        const float inv = 1.0f / spp;
        if (pixels) pixels[idx] = toSRGB8(aces(sumx * inv)) | (toSRGB8(aces(sumy * inv)) << 8) | (toSRGB8(aces(sumz * inv)) << 16) | 0xFF000000u;
    }
}

// tonemap.comp:17-143 (literal; push constants exposure / saturationBoost / tonemapOperator)
namespace {
inline float tmLuminance(Vec3 c) { return dot3(c, {0.2126f, 0.7152f, 0.0722f}); }
inline float length3(Vec3 v) { return std::sqrt(dot3(v, v)); }
Vec3 postTonemapSaturationBoost(Vec3 tonemapped, Vec3 originalHdr, float boost) {          // :43-61
    if (boost <= 1.0f) return tonemapped;
    const float hdrLuma = tmLuminance(originalHdr);
    const float hdrSat = (hdrLuma > 0.0001f) ? length3(sub3(originalHdr, {hdrLuma, hdrLuma, hdrLuma})) / hdrLuma : 0.0f;
    const float ldrLuma = tmLuminance(tonemapped);
    const float ldrSat = (ldrLuma > 0.0001f) ? length3(sub3(tonemapped, {ldrLuma, ldrLuma, ldrLuma})) / ldrLuma : 0.0f;
    if (ldrSat > 0.0001f && ldrLuma > 0.01f) {
        const float satRatio = std::fmin(hdrSat / std::fmax(ldrSat, 0.001f), 2.0f);
        const float recovery = 1.0f * (1.0f - (boost - 1.0f)) + satRatio * (boost - 1.0f);   // mix(1.0, satRatio, boost - 1.0)
        return mix3({ldrLuma, ldrLuma, ldrLuma}, tonemapped, std::fmin(recovery, 1.5f));
    }
    return tonemapped;
}
Vec3 khronosPbrNeutral(Vec3 hdr) {                                                          // :65-82
    const float startCompression = 0.8f - 0.04f;
    const float desaturation = 0.15f;
    const float x = std::fmin(hdr.x, std::fmin(hdr.y, hdr.z));
    const float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
    hdr = {hdr.x - offset, hdr.y - offset, hdr.z - offset};
    const float peak = std::fmax(hdr.x, std::fmax(hdr.y, hdr.z));
    if (peak < startCompression) return hdr;
    const float d = 1.0f - startCompression;
    const float newPeak = 1.0f - d * d / (peak + d - startCompression);
    hdr = scale3(hdr, newPeak / peak);
    const float g = 1.0f - 1.0f / (desaturation * (peak - newPeak) + 1.0f);
    return mix3(hdr, {newPeak, newPeak, newPeak}, g);
}
Vec3 neutralTonemap(Vec3 hdr) {                                                             // :85-95
    const float peak = std::fmax(std::fmax(hdr.x, hdr.y), hdr.z);
    if (peak <= 1.0f) return hdr;
    const float compressed = 1.0f - std::exp(-(peak - 1.0f));
    const float scale = (1.0f + compressed) / peak;
    return scale3(hdr, scale);
}
}  // namespace

extern "C" void orc_tonemap(const float* hdrRgba, uint32_t n, float exposure, float saturationBoost, int op, uint32_t* out) {
    for (uint32_t i = 0; i < n; ++i) {                                                      // main(), :97-143
        Vec3 hdr = {hdrRgba[4 * i], hdrRgba[4 * i + 1], hdrRgba[4 * i + 2]};
        hdr = scale3(hdr, exposure);
        const Vec3 hdrOriginal = hdr;
        Vec3 ldr = op == 0 ? neutralTonemap(hdr) : khronosPbrNeutral(hdr);
        if (saturationBoost > 1.0f) ldr = postTonemapSaturationBoost(ldr, hdrOriginal, saturationBoost);
        else if (saturationBoost < 1.0f && saturationBoost > 0.0f) {
            const float luma = tmLuminance(ldr);
            ldr = mix3({luma, luma, luma}, ldr, saturationBoost);
        }
        auto unorm = [](float v) { return uint32_t(std::fmin(std::fmax(v, 0.0f), 1.0f) * 255.0f + 0.5f); };   // rgba8 store
        out[i] = unorm(ldr.x) | (unorm(ldr.y) << 8) | (unorm(ldr.z) << 16) | 0xFF000000u;
    }
}

// =============================================================================================
// material.hpp:17-114, material.cpp, vox_loader.cpp:60-462 — materials and the MagicaVoxel importer.
// The default palette (vox_loader.cpp:22-55, a 256-entry literal there) is produced from its construction rule;
// tests compare it with the reference's table read as text when the reference tree is present.
}  // extern "C"  (closed for the C++ section below)

#include <array>
#include <cstdio>
#include <istream>
#include <sstream>
#include <string>
#include <unordered_map>

namespace {

enum class MaterialType : uint8_t { Diffuse = 0, Metallic = 1, Glass = 2, Emissive = 3, Count = 4 };   // material.hpp:18-24

struct Material {                                   // material.hpp:27-44
    std::string name;
    float albedo[3] = {1.0f, 1.0f, 1.0f};
    float alpha = 1.0f, metallic = 0.0f, roughness = 0.5f, ior = 1.5f, specular = 0.5f;
    float emission[3] = {0.0f, 0.0f, 0.0f};
    float emissionPower = 0.0f;
    MaterialType type = MaterialType::Diffuse;
    int16_t voxPaletteIndex = -1;
};

inline float clampf(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }

MaterialGpu packMaterial(const Material& mat) {      // material.hpp:96-112
    MaterialGpu gpu;
    for (int a = 0; a < 3; ++a) gpu.albedo[a] = mat.albedo[a];
    const uint32_t metalBits = static_cast<uint32_t>(clampf(mat.metallic, 0.0f, 1.0f) * 255.0f);
    const uint32_t roughBits = static_cast<uint32_t>(clampf(mat.roughness, 0.0f, 1.0f) * 255.0f);
    const uint32_t typeBits = static_cast<uint32_t>(mat.type);
    const uint32_t alphaBits = static_cast<uint32_t>(clampf(mat.alpha, 0.0f, 1.0f) * 15.0f);
    const uint32_t specBits = static_cast<uint32_t>(clampf(mat.specular, 0.0f, 1.0f) * 255.0f);
    gpu.flags = (metalBits << 24) | (roughBits << 16) | (typeBits << 12) | (alphaBits << 8) | specBits;
    for (int a = 0; a < 3; ++a) gpu.emission[a] = mat.emission[a] * mat.emissionPower;
    gpu.ior = (mat.type == MaterialType::Glass) ? mat.ior : mat.emissionPower;
    return gpu;
}

struct MaterialLibrary {                             // material.cpp
    std::vector<Material> m_materials;
    std::unordered_map<std::string, uint32_t> m_nameToId;
    std::unordered_map<uint32_t, uint32_t> m_colorToId;
    std::array<uint32_t, 256> m_voxPaletteMap{};
    MaterialLibrary() { m_voxPaletteMap.fill(0); createDefaultMaterials(); }                 // :12-15
    void createDefaultMaterials() {                                                          // :17-28
        Material d;
        d.name = "default";
        d.albedo[0] = d.albedo[1] = d.albedo[2] = 0.8f;
        d.roughness = 0.5f; d.metallic = 0.0f; d.type = MaterialType::Diffuse;
        m_materials.push_back(d);
        m_nameToId["default"] = 0;
    }
    uint32_t addMaterial(const Material& mat) {                                              // :30-39
        const uint32_t id = static_cast<uint32_t>(m_materials.size());
        m_materials.push_back(mat);
        if (!mat.name.empty()) m_nameToId[mat.name] = id;
        return id;
    }
    uint32_t getOrCreateFromColor(uint32_t packedRGB) {                                      // :89-117
        auto it = m_colorToId.find(packedRGB);
        if (it != m_colorToId.end()) return it->second;
        Material mat;
        mat.albedo[0] = static_cast<float>((packedRGB >> 16) & 0xFF) / 255.0f;
        mat.albedo[1] = static_cast<float>((packedRGB >> 8) & 0xFF) / 255.0f;
        mat.albedo[2] = static_cast<float>(packedRGB & 0xFF) / 255.0f;
        mat.roughness = 0.5f; mat.metallic = 0.0f; mat.type = MaterialType::Diffuse;
        char nameBuf[32];
        snprintf(nameBuf, sizeof(nameBuf), "color_%06X", packedRGB);
        mat.name = nameBuf;
        const uint32_t id = addMaterial(mat);
        m_colorToId[packedRGB] = id;
        return id;
    }
    uint32_t getOrCreateFromColor(uint8_t r, uint8_t g, uint8_t b) {                          // :81-87
        return getOrCreateFromColor((uint32_t(r) << 16) | (uint32_t(g) << 8) | uint32_t(b));
    }
};

struct VoxVoxel { uint8_t x, y, z, colorIndex; };                                            // vox_loader.hpp:20-23
struct VoxMaterial {                                                                         // vox_loader.hpp:26-37
    MaterialType type{MaterialType::Diffuse};
    float roughness{0.5f}, metallic{0.0f}, ior{1.5f}, emission{0.0f}, flux{0.0f}, alpha{1.0f}, glow{0.0f}, specular{0.5f};
    bool hasProperties{false};
};
struct VoxModel { uint32_t sizeX = 0, sizeY = 0, sizeZ = 0; std::vector<VoxVoxel> voxels; };
struct VoxFile {
    std::vector<VoxModel> models;
    uint32_t palette[256];
    VoxMaterial materials[256];
    Material getMaterial(uint8_t paletteIndex) const {                                       // vox_loader.cpp:116-149
        Material mat;
        const uint32_t c = palette[paletteIndex];
        const uint8_t r = (c >> 0) & 0xFF, g = (c >> 8) & 0xFF, b = (c >> 16) & 0xFF, a = (c >> 24) & 0xFF;
        mat.albedo[0] = r / 255.0f; mat.albedo[1] = g / 255.0f; mat.albedo[2] = b / 255.0f;
        mat.alpha = a / 255.0f;
        const VoxMaterial& voxMat = materials[paletteIndex];
        if (voxMat.hasProperties) {
            mat.type = voxMat.type; mat.roughness = voxMat.roughness; mat.metallic = voxMat.metallic;
            mat.ior = voxMat.ior; mat.specular = voxMat.specular; mat.alpha = voxMat.alpha;
            if (voxMat.type == MaterialType::Emissive) {
                for (int k = 0; k < 3; ++k) mat.emission[k] = mat.albedo[k];
                mat.emissionPower = voxMat.emission > 0 ? voxMat.emission : voxMat.flux;
                if (mat.emissionPower <= 0) mat.emissionPower = 5.0f;
            }
        } else {
            mat.type = MaterialType::Diffuse; mat.roughness = 0.5f; mat.metallic = 0.0f;
        }
        mat.voxPaletteIndex = static_cast<int16_t>(paletteIndex);
        return mat;
    }
};

void defaultPalette(uint32_t pal[256]) {     // values of vox_loader.cpp:22-55, from MagicaVoxel's construction rule
    const uint32_t lv[6] = {0xff, 0xcc, 0x99, 0x66, 0x33, 0x00};
    const uint32_t rp[10] = {0xee, 0xdd, 0xbb, 0xaa, 0x88, 0x77, 0x55, 0x44, 0x22, 0x11};
    int k = 0;
    pal[k++] = 0x00000000;
    for (int r = 0; r < 6; ++r) for (int g = 0; g < 6; ++g) for (int b = 0; b < 6; ++b) {
        if (r == 5 && g == 5 && b == 5) continue;
        pal[k++] = 0xff000000u | (lv[b] << 16) | (lv[g] << 8) | lv[r];
    }
    for (int i = 0; i < 10; ++i) pal[k++] = 0xff000000u | rp[i];
    for (int i = 0; i < 10; ++i) pal[k++] = 0xff000000u | (rp[i] << 8);
    for (int i = 0; i < 10; ++i) pal[k++] = 0xff000000u | (rp[i] << 16);
    for (int i = 0; i < 10; ++i) pal[k++] = 0xff000000u | (rp[i] << 16) | (rp[i] << 8) | rp[i];
}

template <typename T> bool readValue(std::istream& file, T& value) {                         // vox_loader.cpp:58-62
    file.read(reinterpret_cast<char*>(&value), sizeof(T));
    return file.good();
}
bool readBytes(std::istream& file, void* buffer, size_t count) {                             // :63-66
    file.read(reinterpret_cast<char*>(buffer), static_cast<std::streamsize>(count));
    return file.good();
}
std::string readString(std::istream& file) {                                                 // :68-74
    int32_t len;
    if (!readValue(file, len) || len <= 0 || len > 1024) return "";
    std::string str(len, '\0');
    readBytes(file, str.data(), len);
    return str;
}
std::unordered_map<std::string, std::string> readDict(std::istream& file) {                  // :76-90
    std::unordered_map<std::string, std::string> dict;
    int32_t numPairs;
    if (!readValue(file, numPairs)) return dict;
    for (int32_t i = 0; i < numPairs; ++i) {
        std::string key = readString(file);
        std::string value = readString(file);
        if (!key.empty()) dict[key] = value;
    }
    return dict;
}
MaterialType parseVoxMaterialType(const std::string& t) {                                    // :98-105
    if (t == "_diffuse") return MaterialType::Diffuse;
    if (t == "_metal") return MaterialType::Metallic;
    if (t == "_glass") return MaterialType::Glass;
    if (t == "_emit") return MaterialType::Emissive;
    return MaterialType::Diffuse;
}
float parseFloat(const std::string& str, float defaultVal = 0.0f) {                          // :107-113
    try { return std::stof(str); } catch (...) { return defaultVal; }
}

bool loadVoxFile(std::istream& file, VoxFile& outVox, std::string& errorMsg) {               // :151-368
    char magic[4];
    if (!readBytes(file, magic, 4) || std::memcmp(magic, "VOX ", 4) != 0) { errorMsg = "Invalid VOX file: bad magic number"; return false; }
    int32_t version;
    if (!readValue(file, version)) { errorMsg = "Failed to read VOX version"; return false; }
    if (version < 150) { errorMsg = "Unsupported VOX version: " + std::to_string(version) + " (need >= 150)"; return false; }
    defaultPalette(outVox.palette);
    outVox.models.clear();
    VoxModel currentModel{};
    bool hasSize = false;
    char id[4];
    int32_t contentSize, childrenSize;
    if (!readBytes(file, id, 4) || std::memcmp(id, "MAIN", 4) != 0) { errorMsg = "Invalid VOX file: missing MAIN chunk"; return false; }
    if (!readValue(file, contentSize) || !readValue(file, childrenSize)) { errorMsg = "Failed to read MAIN chunk header"; return false; }
    if (contentSize > 0) file.seekg(contentSize, std::ios::cur);
    std::streampos endPos = file.tellg();
    endPos += childrenSize;
    while (file.tellg() < endPos && file.good()) {
        char cid[4];
        int32_t cContent, cChildren;
        if (!readBytes(file, cid, 4)) break;
        if (!readValue(file, cContent)) break;
        if (!readValue(file, cChildren)) break;
        std::streampos chunkEnd = file.tellg();
        chunkEnd += cContent;
        if (std::memcmp(cid, "SIZE", 4) == 0) {
            if (hasSize && !currentModel.voxels.empty()) { outVox.models.push_back(std::move(currentModel)); currentModel = VoxModel{}; }
            int32_t x = 0, y = 0, z = 0;
            readValue(file, x); readValue(file, y); readValue(file, z);
            currentModel.sizeX = static_cast<uint32_t>(x); currentModel.sizeY = static_cast<uint32_t>(y); currentModel.sizeZ = static_cast<uint32_t>(z);
            hasSize = true;
        } else if (std::memcmp(cid, "XYZI", 4) == 0) {
            int32_t numVoxels;
            if (!readValue(file, numVoxels)) { errorMsg = "Failed to read voxel count"; return false; }
            for (int32_t i = 0; i < numVoxels; ++i) {
                uint8_t x = 0, y = 0, z = 0, colorIndex = 0;
                readValue(file, x); readValue(file, y); readValue(file, z);
                if (!readValue(file, colorIndex)) break;   // the reference keeps pushing here (uninitialised bytes)
                currentModel.voxels.push_back(VoxVoxel{x, y, z, colorIndex});
            }
        } else if (std::memcmp(cid, "RGBA", 4) == 0) {
            for (int i = 0; i < 255; ++i) {
                uint32_t rgba;
                if (readValue(file, rgba)) outVox.palette[i + 1] = rgba;
            }
            uint32_t unused;
            readValue(file, unused);
        } else if (std::memcmp(cid, "MATL", 4) == 0) {
            int32_t materialId;
            if (!readValue(file, materialId)) { file.seekg(chunkEnd); continue; }
            auto props = readDict(file);
            if (materialId >= 0 && materialId < 256) {
                VoxMaterial& mat = outVox.materials[materialId];
                mat.hasProperties = true;
                auto it = props.find("_type");  if (it != props.end()) mat.type = parseVoxMaterialType(it->second);
                it = props.find("_rough");      if (it != props.end()) mat.roughness = parseFloat(it->second, 0.5f);
                it = props.find("_metal");      if (it != props.end()) mat.metallic = parseFloat(it->second, 0.0f);
                it = props.find("_ior");        if (it != props.end()) mat.ior = parseFloat(it->second, 1.5f);
                it = props.find("_emit");       if (it != props.end()) mat.emission = parseFloat(it->second, 0.0f);
                it = props.find("_flux");       if (it != props.end()) mat.flux = parseFloat(it->second, 0.0f);
                it = props.find("_alpha");      if (it != props.end()) mat.alpha = parseFloat(it->second, 1.0f);
                it = props.find("_sp");         if (it != props.end()) mat.specular = parseFloat(it->second, 0.5f);
                it = props.find("_g");          if (it != props.end()) mat.glow = parseFloat(it->second, 0.0f);
            }
        }
        file.seekg(chunkEnd);
        if (cChildren > 0) file.seekg(cChildren, std::ios::cur);
    }
    if (hasSize || !currentModel.voxels.empty()) outVox.models.push_back(std::move(currentModel));
    if (outVox.models.empty()) { errorMsg = "No models found in VOX file"; return false; }
    return true;
}

void importVoxMaterials(const VoxFile& vox, MaterialLibrary& matLib, uint32_t paletteToMaterial[256]) {   // :370-388
    for (int i = 1; i < 256; ++i) {
        Material mat = vox.getMaterial(static_cast<uint8_t>(i));
        char nameBuf[32];
        snprintf(nameBuf, sizeof(nameBuf), "vox_mat_%d", i);
        mat.name = nameBuf;
        const uint32_t matId = matLib.addMaterial(mat);
        paletteToMaterial[i] = matId;
        matLib.m_voxPaletteMap[static_cast<uint8_t>(i)] = matId;
    }
    paletteToMaterial[0] = 0;
}

uint32_t importVoxToChunks(const VoxFile& vox, World& chunkMgr, MaterialLibrary* matLib, float ox, float oy, float oz,
                           uint32_t modelIndex) {                                            // :390-430
    if (modelIndex >= vox.models.size()) return 0;
    const VoxModel& model = vox.models[modelIndex];
    uint32_t count = 0;
    for (const auto& v : model.voxels) {
        const float wx = ox + static_cast<float>(v.x), wy = oy + static_cast<float>(v.z), wz = oz + static_cast<float>(v.y);
        if (matLib) {
            chunkMgr.setVoxelMaterial(wx, wy, wz, matLib->m_voxPaletteMap[v.colorIndex], 1.0f);
        } else {
            const uint32_t c = vox.palette[v.colorIndex];
            const uint8_t r = (c >> 0) & 0xFF, g = (c >> 8) & 0xFF, b = (c >> 16) & 0xFF;
            // ChunkManager::setVoxel(worldPos, r, g, b) without a library: chunk_manager.cpp:91-102
            const uint32_t materialId = (uint32_t(r) << 16) | (uint32_t(g) << 8) | uint32_t(b);
            chunkMgr.setVoxelMaterial(wx, wy, wz, materialId, 1.0f);
        }
        count++;
    }
    return count;
}

}  // namespace

extern "C" {

void* orc_vox_load(const void* data, size_t n, char* err, size_t errLen) {
    std::istringstream in(std::string(static_cast<const char*>(data), n), std::ios::binary);
    auto* v = new VoxFile();
    std::string msg;
    if (!loadVoxFile(in, *v, msg)) { if (err && errLen) snprintf(err, errLen, "%s", msg.c_str()); delete v; return nullptr; }
    return v;
}
void orc_vox_free(void* v) { delete static_cast<VoxFile*>(v); }
uint32_t orc_vox_n_models(const void* v) { return uint32_t(static_cast<const VoxFile*>(v)->models.size()); }
void orc_vox_model_info(const void* v, uint32_t i, uint32_t size[3], uint32_t* nVoxels) {
    const VoxModel& m = static_cast<const VoxFile*>(v)->models[i];
    size[0] = m.sizeX; size[1] = m.sizeY; size[2] = m.sizeZ; *nVoxels = uint32_t(m.voxels.size());
}
const void* orc_vox_model_voxels(const void* v, uint32_t i) { return static_cast<const VoxFile*>(v)->models[i].voxels.data(); }
const uint32_t* orc_vox_palette(const void* v) { return static_cast<const VoxFile*>(v)->palette; }
void orc_default_palette(uint32_t out[256]) { defaultPalette(out); }
void* orc_matlib_new() { return new MaterialLibrary(); }
void orc_matlib_free(void* l) { delete static_cast<MaterialLibrary*>(l); }
uint32_t orc_matlib_size(const void* l) { return uint32_t(static_cast<const MaterialLibrary*>(l)->m_materials.size()); }
uint32_t orc_matlib_from_color(void* l, uint8_t r, uint8_t g, uint8_t b) { return static_cast<MaterialLibrary*>(l)->getOrCreateFromColor(r, g, b); }
void orc_matlib_pack(const void* l, void* out) {                                             // material.cpp:127-135
    const MaterialLibrary* lib = static_cast<const MaterialLibrary*>(l);
    MaterialGpu* o = static_cast<MaterialGpu*>(out);
    for (size_t i = 0; i < lib->m_materials.size(); ++i) o[i] = packMaterial(lib->m_materials[i]);
}
void orc_vox_import_materials(const void* v, void* l, uint32_t map[256]) {
    importVoxMaterials(*static_cast<const VoxFile*>(v), *static_cast<MaterialLibrary*>(l), map);
}
uint32_t orc_vox_import_to_world(const void* v, void* world, void* libOrNull, float ox, float oy, float oz, uint32_t model) {
    return importVoxToChunks(*static_cast<const VoxFile*>(v), *static_cast<World*>(world),
                             static_cast<MaterialLibrary*>(libOrNull), ox, oy, oz, model);
}

uint32_t orc_sizeof_counters(void) { return sizeof(Counters); }

}  // extern "C"
