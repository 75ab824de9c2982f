// Device-side build of the 64-tree from the reference's world arrays (the work of Renderer::addWorld's
// uploadSvoBuffers + buildChunkBlas/Tlas, reference blok/src/renderer_upload.cpp:237-312,
// blok/src/renderer_raytracing.cpp:15-254, on this backend).
//
//   1. copy SvoNode[] / SubChunkGpu[] to HBM (coalesced memcpy, 149 MB for the 1024^3 scene);
//   2. brick_mask_kernel: one lane per 4x4x4 brick of every sub-chunk walks the sub-chunk's binary octree
//      (same visibility rules as the shader: only through set childMask bits, out-of-range indices skipped,
//      childMask == 0 is a leaf that counts iff occupancy > 0; intersect.rint:132-137,169) and emits the
//      brick's 64-bit voxel mask;
//   3. compaction of non-empty bricks (hipcub scan), Morton-digit keys, radix sort -> level-1 order;
//   4. popcount scan -> material offsets; material_kernel: one lane per voxel re-walks to the leaf and stores
//      its materialId;
//   5. the few thousand nodes above brick level are grouped on the host from the sorted brick keys
//      (tree_build.cpp: build_upper_levels) and copied in front of the bricks.
// Integer / byte work, HBM- and latency-bound; no MFMA.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "gpu_build.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace blok {

namespace {

struct WalkCtx {
    const blok_svo_node* nodes;
    uint32_t n_nodes;
    const blok_sub_chunk* subs;
    uint32_t n_subs;
    uint32_t sub_size;        // voxels per sub-chunk edge (power of two >= 4)
    uint32_t bricks_side;     // sub_size / 4
    uint32_t bricks_per_sub;  // bricks_side^3
    uint32_t sub_shift;       // log2(sub_size)
};

__device__ __forceinline__ uint32_t node_limit(const WalkCtx& c, const blok_sub_chunk& s) {
    const uint64_t end = static_cast<uint64_t>(s.node_offset) + s.node_count;
    return static_cast<uint32_t>(end < c.n_nodes ? end : c.n_nodes);
}

// Error bits raised by the walks.
enum : uint32_t { kErrLeafAboveVoxel = 1u, kErrInteriorBelowVoxel = 2u };

__global__ __launch_bounds__(256) void brick_mask_kernel(const WalkCtx c, uint64_t* masks, uint32_t* non_empty, uint32_t* error) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (tid >= static_cast<uint64_t>(c.n_subs) * c.bricks_per_sub) return;
    const uint32_t s = static_cast<uint32_t>(tid / c.bricks_per_sub), b = static_cast<uint32_t>(tid % c.bricks_per_sub);
    const uint32_t vx = (b % c.bricks_side) * 4u, vy = ((b / c.bricks_side) % c.bricks_side) * 4u,
                   vz = (b / (c.bricks_side * c.bricks_side)) * 4u;
    const blok_sub_chunk sub = c.subs[s];
    const uint32_t limit = node_limit(c, sub);
    uint32_t at = sub.node_offset + sub.root_node_index;
    uint64_t mask = 0;
    bool alive = true;
    for (uint32_t shift = c.sub_shift; shift > 2u && alive; --shift) {       // down to the node covering the brick
        if (at >= limit) { alive = false; break; }
        const blok_svo_node nd = c.nodes[at];
        if (nd.child_mask == 0u) { if (nd.occupancy > 0.0f) atomicOr(error, kErrLeafAboveVoxel); alive = false; break; }
        const uint32_t h = shift - 1u;
        const uint32_t oct = ((vx >> h) & 1u) | (((vy >> h) & 1u) << 1) | (((vz >> h) & 1u) << 2);
        if (!(nd.child_mask & (1u << oct))) { alive = false; break; }
        at = sub.node_offset + nd.first_child + oct;
    }
    if (alive && at < limit) {
        const blok_svo_node n4 = c.nodes[at];
        if (n4.child_mask == 0u) { if (n4.occupancy > 0.0f) atomicOr(error, kErrLeafAboveVoxel); }
        else
            for (uint32_t ci = 0; ci < 8u; ++ci) {
                if (!(n4.child_mask & (1u << ci))) continue;
                const uint32_t a2 = sub.node_offset + n4.first_child + ci;
                if (a2 >= limit) continue;
                const blok_svo_node n2 = c.nodes[a2];
                if (n2.child_mask == 0u) { if (n2.occupancy > 0.0f) atomicOr(error, kErrLeafAboveVoxel); continue; }
                for (uint32_t gi = 0; gi < 8u; ++gi) {
                    if (!(n2.child_mask & (1u << gi))) continue;
                    const uint32_t a1 = sub.node_offset + n2.first_child + gi;
                    if (a1 >= limit) continue;
                    const blok_svo_node n1 = c.nodes[a1];
                    if (n1.child_mask != 0u) { atomicOr(error, kErrInteriorBelowVoxel); continue; }
                    if (!(n1.occupancy > 0.0f)) continue;
                    const uint32_t x = ((ci & 1u) << 1) | (gi & 1u), y = (ci & 2u) | ((gi >> 1) & 1u), z = ((ci >> 2) << 1) | (gi >> 2);
                    mask |= 1ull << (x | (y << 2) | (z << 4));
                }
            }
    }
    masks[tid] = mask;
    non_empty[tid] = mask != 0ull;
}

struct KeyCtx { int32_t origin[3]; uint32_t levels; };

// Key of a brick: the 2-bit digit triples of levels 1..L-1 of its corner, least significant level first.
__global__ __launch_bounds__(256) void brick_key_kernel(const WalkCtx c, const KeyCtx k, const uint64_t* masks,
                                                        const uint32_t* slot_of, uint64_t total, uint64_t* keys, uint32_t* src) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (tid >= total || masks[tid] == 0ull) return;
    const uint32_t s = static_cast<uint32_t>(tid / c.bricks_per_sub), b = static_cast<uint32_t>(tid % c.bricks_per_sub);
    const blok_sub_chunk& sub = c.subs[s];
    const uint32_t x = static_cast<uint32_t>(static_cast<int32_t>(sub.world_min[0]) - k.origin[0]) + (b % c.bricks_side) * 4u;
    const uint32_t y = static_cast<uint32_t>(static_cast<int32_t>(sub.world_min[1]) - k.origin[1]) + ((b / c.bricks_side) % c.bricks_side) * 4u;
    const uint32_t z = static_cast<uint32_t>(static_cast<int32_t>(sub.world_min[2]) - k.origin[2]) + (b / (c.bricks_side * c.bricks_side)) * 4u;
    uint64_t key = 0;
    for (uint32_t l = 1; l < k.levels; ++l) {
        const uint64_t digit = ((x >> (2 * l)) & 3u) | (((y >> (2 * l)) & 3u) << 2) | (((z >> (2 * l)) & 3u) << 4);
        key |= digit << (6 * (l - 1));
    }
    const uint32_t slot = slot_of[tid];
    keys[slot] = key;
    src[slot] = static_cast<uint32_t>(tid);
}

__global__ __launch_bounds__(256) void gather_mask_kernel(const uint64_t* masks, const uint32_t* src_sorted, uint32_t n,
                                                          uint64_t* masks_sorted, uint32_t* counts) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t m = masks[src_sorted[i]];
    masks_sorted[i] = m;
    counts[i] = __popcll(m);
}

// One lane per (brick, voxel bit): find the leaf again and store its material id at its rank.
__global__ __launch_bounds__(256) void material_kernel(const WalkCtx c, const uint64_t* masks_sorted, const uint32_t* src_sorted,
                                                       const uint32_t* mat_base, uint32_t n_bricks, uint32_t* materials) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    const uint32_t i = static_cast<uint32_t>(tid >> 6), bit = static_cast<uint32_t>(tid & 63u);
    if (i >= n_bricks) return;
    const uint64_t mask = masks_sorted[i];
    if (!((mask >> bit) & 1ull)) return;
    const uint32_t g = src_sorted[i];
    const uint32_t s = g / c.bricks_per_sub, b = g % c.bricks_per_sub;
    const uint32_t vx = (b % c.bricks_side) * 4u + (bit & 3u), vy = ((b / c.bricks_side) % c.bricks_side) * 4u + ((bit >> 2) & 3u),
                   vz = (b / (c.bricks_side * c.bricks_side)) * 4u + (bit >> 4);
    const blok_sub_chunk sub = c.subs[s];
    uint32_t at = sub.node_offset + sub.root_node_index;
    for (uint32_t shift = c.sub_shift; shift > 0u; --shift) {
        const blok_svo_node nd = c.nodes[at];
        const uint32_t h = shift - 1u;
        const uint32_t oct = ((vx >> h) & 1u) | (((vy >> h) & 1u) << 1) | (((vz >> h) & 1u) << 2);
        at = sub.node_offset + nd.first_child + oct;        // the mask bit proves this path exists and is in range
    }
    materials[mat_base[i] + __popcll(mask & ((1ull << bit) - 1ull))] = c.nodes[at].material_id;
}

__global__ __launch_bounds__(256) void brick_node_kernel(const uint64_t* masks_sorted, const uint32_t* mat_base, uint32_t n, uint4* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t m = masks_sorted[i];
    out[i] = make_uint4(static_cast<uint32_t>(m), static_cast<uint32_t>(m >> 32), mat_base[i], 0u);
}

struct DeviceBuffers {          // frees everything it still owns on scope exit
    std::vector<void*> ptrs;
    ~DeviceBuffers() { for (void* p : ptrs) if (p) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** p, size_t count) {
        *p = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
    void release(void* p) { for (void*& q : ptrs) if (q == p) q = nullptr; }
};

#define GB_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { *why = std::string(#call) + ": " + hipGetErrorString(e_); \
                          return e_ == hipErrorOutOfMemory ? GpuBuildStatus::OutOfMemory : GpuBuildStatus::HipError; } } while (0)

inline uint32_t blocks_for(uint64_t n) { return static_cast<uint32_t>((n + 255u) / 256u); }

// Common tail of the builders: compaction of the non-empty bricks, keys, radix sort, material offsets, material
// ids, upper levels, final node array.  `launch_keys` / `launch_materials` enqueue the source-specific kernels.
template <class KeyLauncher, class MaterialLauncher>
GpuBuildStatus finish_from_masks(DeviceBuffers& mem, uint64_t total, const uint64_t* d_masks, uint32_t* d_flag, uint32_t* d_slot,
                                 uint32_t levels, const int32_t lo[3], KeyLauncher launch_keys, MaterialLauncher launch_materials,
                                 GpuTree* out, std::string* why) {
    size_t temp_bytes = 0;
    GB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, d_flag, d_slot, static_cast<int>(total + 1)));
    size_t scan_bytes = temp_bytes;
    uint8_t* d_scratch;
    GB_TRY(mem.alloc(&d_scratch, scan_bytes));
    GB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scratch, temp_bytes, d_flag, d_slot, static_cast<int>(total + 1)));
    uint32_t n_bricks = 0;
    GB_TRY(hipMemcpy(&n_bricks, d_slot + total, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (n_bricks == 0) return GpuBuildStatus::UseHostBuilder;

    uint64_t *d_keys, *d_keys_sorted, *d_masks_sorted; uint32_t *d_src, *d_src_sorted, *d_counts, *d_mat_base;
    GB_TRY(mem.alloc(&d_keys, n_bricks)); GB_TRY(mem.alloc(&d_keys_sorted, n_bricks));
    GB_TRY(mem.alloc(&d_src, n_bricks)); GB_TRY(mem.alloc(&d_src_sorted, n_bricks));
    GB_TRY(mem.alloc(&d_masks_sorted, n_bricks));
    GB_TRY(mem.alloc(&d_counts, n_bricks + 1)); GB_TRY(mem.alloc(&d_mat_base, n_bricks + 1));
    launch_keys(d_slot, d_keys, d_src);
    GB_TRY(hipGetLastError());
    const int key_bits = std::max(1, static_cast<int>(6 * (levels - 1)));
    temp_bytes = 0;
    GB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys, d_keys_sorted, d_src, d_src_sorted, static_cast<int>(n_bricks), 0, key_bits));
    uint8_t* d_sort_scratch;
    GB_TRY(mem.alloc(&d_sort_scratch, temp_bytes));
    GB_TRY(hipcub::DeviceRadixSort::SortPairs(d_sort_scratch, temp_bytes, d_keys, d_keys_sorted, d_src, d_src_sorted, static_cast<int>(n_bricks), 0, key_bits));
    hipLaunchKernelGGL(gather_mask_kernel, dim3(blocks_for(n_bricks)), dim3(256), 0, nullptr, d_masks, d_src_sorted, n_bricks, d_masks_sorted, d_counts);
    GB_TRY(hipGetLastError());
    GB_TRY(hipMemset(d_counts + n_bricks, 0, sizeof(uint32_t)));
    temp_bytes = 0;
    GB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, d_counts, d_mat_base, static_cast<int>(n_bricks + 1)));
    if (temp_bytes > scan_bytes) { GB_TRY(mem.alloc(&d_scratch, temp_bytes)); scan_bytes = temp_bytes; }
    GB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scratch, temp_bytes, d_counts, d_mat_base, static_cast<int>(n_bricks + 1)));
    uint32_t n_voxels = 0;
    GB_TRY(hipMemcpy(&n_voxels, d_mat_base + n_bricks, sizeof(uint32_t), hipMemcpyDeviceToHost));

    uint32_t* d_materials;
    GB_TRY(mem.alloc(&d_materials, n_voxels));
    launch_materials(d_masks_sorted, d_src_sorted, d_mat_base, n_bricks, d_materials);
    GB_TRY(hipGetLastError());

    // the levels above the bricks, on the host, from the sorted keys
    std::vector<uint64_t> keys(n_bricks);
    GB_TRY(hipMemcpy(keys.data(), d_keys_sorted, n_bricks * sizeof(uint64_t), hipMemcpyDeviceToHost));
    std::vector<TreeNode> upper;
    if (!build_upper_levels(keys, levels, upper)) return GpuBuildStatus::UseHostBuilder;    // duplicate bricks: general path
    uint4* d_tree;
    const size_t n_tree = upper.size() + n_bricks;
    GB_TRY(mem.alloc(&d_tree, n_tree));
    GB_TRY(hipMemcpy(d_tree, upper.data(), upper.size() * sizeof(TreeNode), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(brick_node_kernel, dim3(blocks_for(n_bricks)), dim3(256), 0, nullptr, d_masks_sorted, d_mat_base, n_bricks,
                       d_tree + upper.size());
    GB_TRY(hipGetLastError());
    GB_TRY(hipDeviceSynchronize());

    mem.release(d_tree); mem.release(d_materials);
    out->d_nodes = d_tree; out->d_materials = d_materials;
    out->n_nodes = n_tree; out->n_voxels = n_voxels; out->levels = levels;
    for (int a = 0; a < 3; ++a) out->origin[a] = lo[a];
    return GpuBuildStatus::Ok;
}

}  // namespace

GpuBuildStatus gpu_build_tree(const blok_svo_node* nodes, size_t n_nodes, const blok_sub_chunk* subs, size_t n_subs,
                              GpuTree* out, std::string* why) {
    *out = GpuTree{};
    if (n_subs == 0 || n_nodes == 0) return GpuBuildStatus::UseHostBuilder;          // trivial / empty worlds
    if (n_nodes > 0xFFFFFFFFull || n_subs > 0x7FFFFFFFull) return GpuBuildStatus::UseHostBuilder;
    // ---- host: descriptor validation (same rules as extract_voxels) and the lattice of the tree
    const float size_f = subs[0].sub_chunk_size;
    const uint32_t S = static_cast<uint32_t>(size_f);
    if (!(size_f >= 1.0f) || static_cast<float>(S) != size_f || (S & (S - 1)) != 0 || S > 1024) {
        *why = "sub-chunk size is not a power-of-two number of unit voxels (voxelSize must be 1)";
        return GpuBuildStatus::Unsupported;
    }
    if (S < 4 || S > 64) return GpuBuildStatus::UseHostBuilder;
    int32_t lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (size_t s = 0; s < n_subs; ++s) {
        if (subs[s].sub_chunk_size != size_f) return GpuBuildStatus::UseHostBuilder;  // mixed sizes: general path
        for (int a = 0; a < 3; ++a) {
            const float m = subs[s].world_min[a];
            if (std::floor(m) != m || std::fabs(m) > 32768.0f || subs[s].world_max[a] != m + size_f) {
                *why = "sub-chunk bounds are not on the integer voxel lattice";
                return GpuBuildStatus::Unsupported;
            }
            lo[a] = std::min(lo[a], static_cast<int32_t>(m));
            hi[a] = std::max(hi[a], static_cast<int32_t>(m) + static_cast<int32_t>(S));
        }
    }
    for (int a = 0; a < 3; ++a)
        if (lo[a] < -32768 || hi[a] > 32768) { *why = "world voxel coordinates exceed int16 (hit records carry int16)"; return GpuBuildStatus::Unsupported; }
    for (size_t s = 0; s < n_subs; ++s)
        for (int a = 0; a < 3; ++a)
            if ((static_cast<int32_t>(subs[s].world_min[a]) - lo[a]) % 4 != 0) return GpuBuildStatus::UseHostBuilder;
    int64_t extent = 1;
    for (int a = 0; a < 3; ++a) extent = std::max<int64_t>(extent, int64_t(hi[a]) - lo[a]);
    uint32_t levels = 1;
    while ((int64_t(1) << (2 * levels)) < extent) ++levels;
    if (levels > kMaxLevels) { *why = "world extent exceeds 4^7 voxels per axis"; return GpuBuildStatus::Unsupported; }

    WalkCtx c{};
    c.n_nodes = static_cast<uint32_t>(n_nodes); c.n_subs = static_cast<uint32_t>(n_subs);
    c.sub_size = S; c.bricks_side = S / 4; c.bricks_per_sub = c.bricks_side * c.bricks_side * c.bricks_side;
    c.sub_shift = 0; while ((1u << c.sub_shift) < S) ++c.sub_shift;
    const uint64_t total = static_cast<uint64_t>(n_subs) * c.bricks_per_sub;
    if (total > 0x7FFFFFFFull) return GpuBuildStatus::UseHostBuilder;

    DeviceBuffers mem;
    blok_svo_node* d_nodes_ref; blok_sub_chunk* d_subs; uint64_t* d_masks; uint32_t *d_flag, *d_slot, *d_error;
    GB_TRY(mem.alloc(&d_nodes_ref, n_nodes));
    GB_TRY(mem.alloc(&d_subs, n_subs));
    GB_TRY(mem.alloc(&d_masks, total));
    GB_TRY(mem.alloc(&d_flag, total + 1));
    GB_TRY(mem.alloc(&d_slot, total + 1));
    GB_TRY(mem.alloc(&d_error, 1));
    GB_TRY(hipMemcpy(d_nodes_ref, nodes, n_nodes * sizeof(blok_svo_node), hipMemcpyHostToDevice));
    GB_TRY(hipMemcpy(d_subs, subs, n_subs * sizeof(blok_sub_chunk), hipMemcpyHostToDevice));
    GB_TRY(hipMemset(d_error, 0, sizeof(uint32_t)));
    GB_TRY(hipMemset(d_flag + total, 0, sizeof(uint32_t)));
    c.nodes = d_nodes_ref; c.subs = d_subs;

    hipLaunchKernelGGL(brick_mask_kernel, dim3(blocks_for(total)), dim3(256), 0, nullptr, c, d_masks, d_flag, d_error);
    GB_TRY(hipGetLastError());
    uint32_t error = 0;
    GB_TRY(hipMemcpy(&error, d_error, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (error & kErrLeafAboveVoxel) { *why = "filled leaf above voxel level (not produced by SvoTree::insertVoxel)"; return GpuBuildStatus::Unsupported; }
    if (error & kErrInteriorBelowVoxel) { *why = "interior node below voxel level"; return GpuBuildStatus::Unsupported; }
    KeyCtx k{}; for (int a = 0; a < 3; ++a) k.origin[a] = lo[a]; k.levels = levels;
    return finish_from_masks(mem, total, d_masks, d_flag, d_slot, levels, lo,
        [&](const uint32_t* slot, uint64_t* keys, uint32_t* src) {
            hipLaunchKernelGGL(brick_key_kernel, dim3(blocks_for(total)), dim3(256), 0, nullptr, c, k, d_masks, slot, total, keys, src);
        },
        [&](const uint64_t* masks_sorted, const uint32_t* src_sorted, const uint32_t* mat_base, uint32_t n_bricks, uint32_t* materials) {
            hipLaunchKernelGGL(material_kernel, dim3(blocks_for(static_cast<uint64_t>(n_bricks) * 64u)), dim3(256), 0, nullptr,
                               c, masks_sorted, src_sorted, mat_base, n_bricks, materials);
        }, out, why);
}

// ---- dense id grid (BASELINE.json configs[0..1]; reference dense store chunk.hpp:35-36) --------------------------
namespace {

struct DenseCtx {
    const uint32_t* ids;          // ids[x + y*nx + z*nx*ny]; filled iff != 0 when there is no density array
    const float* density;         // null, or: filled iff density > 0 (chunk_manager.cpp:121) and ids are material ids
    uint32_t nx, ny, nz;
    uint32_t bx, by, bz;          // bricks per axis
    uint32_t levels;
    uint32_t rx0, ry0, rz0, rbx, rby, rbz;   // brick sub-range handled by dense_brick_kernel (whole grid: 0,0,0,bx,by,bz)
};

__device__ __forceinline__ bool dense_filled(const DenseCtx& d, size_t i) {
    return d.density ? d.density[i] > 0.0f : d.ids[i] != 0u;
}

// One lane per brick: 16 rows of 4 consecutive ids; neighbouring lanes read neighbouring 16-byte groups of a row.
__global__ __launch_bounds__(256) void dense_brick_kernel(const DenseCtx d, uint64_t total, uint64_t* masks, uint32_t* non_empty) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (tid >= total) return;                                                  // total = rbx * rby * rbz
    const uint32_t b_x = d.rx0 + static_cast<uint32_t>(tid % d.rbx), b_y = d.ry0 + static_cast<uint32_t>((tid / d.rbx) % d.rby),
                   b_z = d.rz0 + static_cast<uint32_t>(tid / (static_cast<uint64_t>(d.rbx) * d.rby));
    uint64_t mask = 0;
    for (uint32_t z = 0; z < 4u; ++z)
        for (uint32_t y = 0; y < 4u; ++y) {
            const uint32_t vy = b_y * 4u + y, vz = b_z * 4u + z;
            if (vy >= d.ny || vz >= d.nz) continue;
            const size_t row = (static_cast<size_t>(vz) * d.ny + vy) * d.nx + b_x * 4u;
            for (uint32_t x = 0; x < 4u; ++x)
                if (b_x * 4u + x < d.nx && dense_filled(d, row + x)) mask |= 1ull << (x | (y << 2) | (z << 4));
        }
    const size_t g = b_x + (static_cast<size_t>(b_z) * d.by + b_y) * d.bx;
    masks[g] = mask;
    non_empty[g] = mask != 0ull;
}

__global__ __launch_bounds__(256) void dense_key_kernel(const DenseCtx d, const uint64_t* masks, const uint32_t* slot_of, uint64_t total,
                                                        uint64_t* keys, uint32_t* src) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (tid >= total || masks[tid] == 0ull) return;
    const uint32_t x = static_cast<uint32_t>(tid % d.bx) * 4u, y = static_cast<uint32_t>((tid / d.bx) % d.by) * 4u,
                   z = static_cast<uint32_t>(tid / (static_cast<uint64_t>(d.bx) * d.by)) * 4u;
    uint64_t key = 0;
    for (uint32_t l = 1; l < d.levels; ++l) {
        const uint64_t digit = ((x >> (2 * l)) & 3u) | (((y >> (2 * l)) & 3u) << 2) | (((z >> (2 * l)) & 3u) << 4);
        key |= digit << (6 * (l - 1));
    }
    const uint32_t slot = slot_of[tid];
    keys[slot] = key;
    src[slot] = static_cast<uint32_t>(tid);
}

__global__ __launch_bounds__(256) void dense_material_kernel(const DenseCtx d, const uint64_t* masks_sorted, const uint32_t* src_sorted,
                                                             const uint32_t* mat_base, uint32_t n_bricks, uint32_t* materials) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    const uint32_t i = static_cast<uint32_t>(tid >> 6), bit = static_cast<uint32_t>(tid & 63u);
    if (i >= n_bricks) return;
    const uint64_t mask = masks_sorted[i];
    if (!((mask >> bit) & 1ull)) return;
    const uint32_t g = src_sorted[i];
    const uint32_t vx = (g % d.bx) * 4u + (bit & 3u), vy = ((g / d.bx) % d.by) * 4u + ((bit >> 2) & 3u), vz = (g / (d.bx * d.by)) * 4u + (bit >> 4);
    materials[mat_base[i] + __popcll(mask & ((1ull << bit) - 1ull))] = d.ids[(static_cast<size_t>(vz) * d.ny + vy) * d.nx + vx];
}

}  // namespace

GpuBuildStatus gpu_build_tree_dense(const uint32_t* ids, uint32_t nx, uint32_t ny, uint32_t nz, const int32_t origin[3],
                                    GpuTree* out, std::string* why) {
    *out = GpuTree{};
    const uint64_t cells = static_cast<uint64_t>(nx) * ny * nz;
    if (cells == 0) return GpuBuildStatus::UseHostBuilder;
    for (int a = 0; a < 3; ++a) {
        const int64_t hi = int64_t(origin[a]) + (a == 0 ? nx : a == 1 ? ny : nz);
        if (origin[a] < -32768 || hi > 32768) { *why = "world voxel coordinates exceed int16 (hit records carry int16)"; return GpuBuildStatus::Unsupported; }
    }
    const uint32_t extent = std::max(nx, std::max(ny, nz));
    uint32_t levels = 1;
    while ((uint64_t(1) << (2 * levels)) < extent) ++levels;
    if (levels > kMaxLevels) { *why = "world extent exceeds 4^7 voxels per axis"; return GpuBuildStatus::Unsupported; }
    DenseCtx d{};
    d.nx = nx; d.ny = ny; d.nz = nz; d.levels = levels;
    d.bx = (nx + 3) / 4; d.by = (ny + 3) / 4; d.bz = (nz + 3) / 4;
    d.rbx = d.bx; d.rby = d.by; d.rbz = d.bz;
    const uint64_t total = static_cast<uint64_t>(d.bx) * d.by * d.bz;
    if (total > 0x7FFFFFFFull) return GpuBuildStatus::UseHostBuilder;
    DeviceBuffers mem;
    uint32_t* d_ids; uint64_t* d_masks; uint32_t *d_flag, *d_slot;
    GB_TRY(mem.alloc(&d_ids, cells));
    GB_TRY(mem.alloc(&d_masks, total));
    GB_TRY(mem.alloc(&d_flag, total + 1));
    GB_TRY(mem.alloc(&d_slot, total + 1));
    GB_TRY(hipMemcpy(d_ids, ids, cells * sizeof(uint32_t), hipMemcpyHostToDevice));
    GB_TRY(hipMemset(d_flag + total, 0, sizeof(uint32_t)));
    d.ids = d_ids;
    hipLaunchKernelGGL(dense_brick_kernel, dim3(blocks_for(total)), dim3(256), 0, nullptr, d, total, d_masks, d_flag);
    GB_TRY(hipGetLastError());
    const int32_t lo[3] = {origin[0], origin[1], origin[2]};
    return finish_from_masks(mem, total, d_masks, d_flag, d_slot, levels, lo,
        [&](const uint32_t* slot, uint64_t* keys, uint32_t* src) {
            hipLaunchKernelGGL(dense_key_kernel, dim3(blocks_for(total)), dim3(256), 0, nullptr, d, d_masks, slot, total, keys, src);
        },
        [&](const uint64_t* masks_sorted, const uint32_t* src_sorted, const uint32_t* mat_base, uint32_t n_bricks, uint32_t* materials) {
            hipLaunchKernelGGL(dense_material_kernel, dim3(blocks_for(static_cast<uint64_t>(n_bricks) * 64u)), dim3(256), 0, nullptr,
                               d, masks_sorted, src_sorted, mat_base, n_bricks, materials);
        }, out, why);
}

// ---- device-resident dense store (gpu_build.h: GpuVolume) ------------------------------------------------------------
namespace {

DenseCtx volume_ctx(const GpuVolume& v) {
    DenseCtx d{};
    d.ids = v.d_ids; d.density = v.d_density;
    d.nx = v.nx; d.ny = v.ny; d.nz = v.nz; d.bx = v.nbx; d.by = v.nby; d.bz = v.nbz; d.levels = v.levels;
    d.rbx = v.nbx; d.rby = v.nby; d.rbz = v.nbz;
    return d;
}

// Recomputes the masks of the bricks that contain voxels [lo, hi) (box-local voxel coordinates).
GpuBuildStatus keyed_refresh(GpuVolume* v, const uint32_t lo[3], const uint32_t hi[3], std::string* why);
GpuBuildStatus volume_refresh(GpuVolume* v, const uint32_t lo[3], const uint32_t hi[3], std::string* why) {
    if (v->keyed) return keyed_refresh(v, lo, hi, why);
    for (int a = 0; a < 3; ++a) { v->edit_lo[a] = std::min(v->edit_lo[a], lo[a]); v->edit_hi[a] = std::max(v->edit_hi[a], hi[a]); }
    DenseCtx d = volume_ctx(*v);
    d.rx0 = lo[0] / 4u; d.ry0 = lo[1] / 4u; d.rz0 = lo[2] / 4u;
    d.rbx = (hi[0] + 3u) / 4u - d.rx0; d.rby = (hi[1] + 3u) / 4u - d.ry0; d.rbz = (hi[2] + 3u) / 4u - d.rz0;
    const uint64_t total = static_cast<uint64_t>(d.rbx) * d.rby * d.rbz;
    if (!total) return GpuBuildStatus::Ok;
    hipLaunchKernelGGL(dense_brick_kernel, dim3(blocks_for(total)), dim3(256), 0, nullptr, d, total, v->d_masks, v->d_flag);
    GB_TRY(hipGetLastError());
    return GpuBuildStatus::Ok;
}

struct VoxelEdit { uint32_t index; uint32_t material; float density; };

__global__ __launch_bounds__(256) void volume_set_kernel(float* density, uint32_t* ids, const VoxelEdit* edits, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const VoxelEdit e = edits[i];                   // indices are unique (the host keeps the last write per voxel)
    density[e.index] = e.density;
    ids[e.index] = e.material;
}

struct BrushCtx {
    float* density;
    uint32_t nx, ny, nz;
    int32_t origin[3];
    int32_t lo[3];               // gvMin (world voxels), brush.cpp:20-21
    uint32_t ex, ey, ez;         // gvMax - gvMin
    int32_t chunk;
    float voxel_size, cx, cy, cz, radius, value;
    int mode;
};

__device__ __forceinline__ int32_t chunk_of(int32_t g, int32_t c) { return g >= 0 ? g / c : (g - c + 1) / c; }   // chunk_manager.cpp:41-47

// One lane per voxel of the brush's bounding box; float ops in the reference's order (brush.cpp:36-50), no contraction.
__global__ __launch_bounds__(256) void volume_brush_kernel(const BrushCtx b) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (tid >= static_cast<uint64_t>(b.ex) * b.ey * b.ez) return;
    const int32_t gx = b.lo[0] + static_cast<int32_t>(tid % b.ex), gy = b.lo[1] + static_cast<int32_t>((tid / b.ex) % b.ey),
                  gz = b.lo[2] + static_cast<int32_t>(tid / (static_cast<uint64_t>(b.ex) * b.ey));
    const int32_t ccx = chunk_of(gx, b.chunk), ccy = chunk_of(gy, b.chunk), ccz = chunk_of(gz, b.chunk);
    const int32_t lx = gx - ccx * b.chunk, ly = gy - ccy * b.chunk, lz = gz - ccz * b.chunk;
    const float ox = static_cast<float>(ccx * b.chunk) * b.voxel_size, oy = static_cast<float>(ccy * b.chunk) * b.voxel_size,
                oz = static_cast<float>(ccz * b.chunk) * b.voxel_size;
    const float half = 0.5f * b.voxel_size;
    const float vx = ox + (static_cast<float>(lx) + half), vy = oy + (static_cast<float>(ly) + half), vz = oz + (static_cast<float>(lz) + half);
    const float dx = vx - b.cx, dy = vy - b.cy, dz = vz - b.cz;
    const float dist = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
    if (dist > b.radius) return;
    const size_t i = static_cast<size_t>(gx - b.origin[0]) + (static_cast<size_t>(gz - b.origin[2]) * b.ny + static_cast<size_t>(gy - b.origin[1])) * b.nx;
    const float d = b.density[i];
    b.density[i] = b.mode == 0 ? (d < b.value ? b.value : d) : (b.value < d ? b.value : d);     // std::max / std::min, brush.cpp:52-57
}


// ---- keyed layout (gpu_build.h: GpuVolume::keyed) -------------------------------------------------------------------------------
// Key of a cell from its coordinates in units of its own size: `digits` 2-bit digit triples, least significant level first
// (x | y << 2 | z << 4 per digit, the tree's child bit order).  A brick's key has levels-1 digits, a level-l cell's levels-l.
__host__ __device__ inline uint64_t cell_key(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t digits) {
    uint64_t key = 0;
    for (uint32_t j = 0; j < digits; ++j)
        key |= static_cast<uint64_t>(((cx >> (2u * j)) & 3u) | (((cy >> (2u * j)) & 3u) << 2) | (((cz >> (2u * j)) & 3u) << 4)) << (6u * j);
    return key;
}
__device__ inline void key_cell(uint64_t key, uint32_t digits, uint32_t& cx, uint32_t& cy, uint32_t& cz) {
    cx = cy = cz = 0;
    for (uint32_t j = 0; j < digits; ++j) {
        const uint32_t d = static_cast<uint32_t>(key >> (6u * j)) & 63u;
        cx |= (d & 3u) << (2u * j); cy |= ((d >> 2) & 3u) << (2u * j); cz |= (d >> 4) << (2u * j);
    }
}

struct KeyedCtx {
    const float* density; const uint32_t* ids;
    uint32_t nx, ny, nz, levels;
    uint64_t* masks; uint8_t* dirty;
};
struct CellRange { uint32_t x0, y0, z0, nx, ny, nz; };      // cells [x0, x0 + nx) x ... in units of the level's cell size

// One lane per brick of the range: its 64-bit voxel mask from the dense store (filled iff density > 0, chunk_manager.cpp:121), to its key.
__global__ __launch_bounds__(256) void keyed_brick_kernel(const KeyedCtx v, const CellRange r) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (tid >= static_cast<uint64_t>(r.nx) * r.ny * r.nz) return;
    const uint32_t b_x = r.x0 + static_cast<uint32_t>(tid % r.nx), b_y = r.y0 + static_cast<uint32_t>((tid / r.nx) % r.ny),
                   b_z = r.z0 + static_cast<uint32_t>(tid / (static_cast<uint64_t>(r.nx) * r.ny));
    uint64_t mask = 0;
    for (uint32_t z = 0; z < 4u; ++z)
        for (uint32_t y = 0; y < 4u; ++y) {
            const uint32_t vy = b_y * 4u + y, vz = b_z * 4u + z;
            if (vy >= v.ny || vz >= v.nz) continue;
            const size_t row = (static_cast<size_t>(vz) * v.ny + vy) * v.nx + b_x * 4u;
            for (uint32_t x = 0; x < 4u; ++x)
                if (b_x * 4u + x < v.nx && v.density[row + x] > 0.0f) mask |= 1ull << (x | (y << 2) | (z << 4));
        }
    const uint64_t g = cell_key(b_x, b_y, b_z, v.levels - 1u);
    v.masks[g] = mask;
    v.dirty[g] = 1;                       // its material ids are read from the dense store at the next build (an id can change under an unchanged mask)
}

// The same for a SMALL range (an edit): one wave per brick, lane b = voxel b — the 64 density reads of a brick are 16 short rows instead of
// one lane's 64 strided loads (radius-8 brush, 216 bricks: 13.8 -> ~3 us).
__global__ __launch_bounds__(64) void keyed_brick_wave_kernel(const KeyedCtx v, const CellRange r) {
    const uint32_t t = blockIdx.x, b = threadIdx.x;
    const uint32_t b_x = r.x0 + t % r.nx, b_y = r.y0 + (t / r.nx) % r.ny, b_z = r.z0 + t / (r.nx * r.ny);
    const uint32_t vx = b_x * 4u + (b & 3u), vy = b_y * 4u + ((b >> 2) & 3u), vz = b_z * 4u + (b >> 4);
    const bool filled = vx < v.nx && vy < v.ny && vz < v.nz && v.density[(static_cast<size_t>(vz) * v.ny + vy) * v.nx + vx] > 0.0f;
    const uint64_t mask = __ballot(filled);
    if (b == 0) {
        const uint64_t g = cell_key(b_x, b_y, b_z, v.levels - 1u);
        v.masks[g] = mask;
        v.dirty[g] = 1;
    }
}

// The occupancy words above an edit, ALL levels in one workgroup (the ranges are a handful of cells per level; one launch instead of
// levels - 1): level l reads what level l - 1 has just written, a workgroup barrier and a device-scope fence in between.
struct OccLevels { const uint64_t* below[8]; uint64_t* occ[8]; uint32_t digits[8]; CellRange range[8]; uint32_t first, last; };
__global__ __launch_bounds__(256) void occupancy_levels_kernel(const OccLevels o) {
    for (uint32_t l = o.first; l <= o.last; ++l) {
        const CellRange r = o.range[l];
        const uint32_t total = r.nx * r.ny * r.nz;
        for (uint32_t tid = threadIdx.x; tid < total; tid += 256u) {
            const uint64_t c = cell_key(r.x0 + tid % r.nx, r.y0 + (tid / r.nx) % r.ny, r.z0 + tid / (r.nx * r.ny), o.digits[l]);
            const uint64_t* child = o.below[l] + c * 64u;
            uint64_t word = 0;
            for (uint32_t b = 0; b < 64u; ++b) word |= static_cast<uint64_t>(__hip_atomic_load(child + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) << b;
            o.occ[l][c] = word;
        }
        __threadfence();
        __syncthreads();
    }
}

// One lane per cell of level l in the range: its occupancy word from the 64 cells below it (the brick masks for l = 2).
__global__ __launch_bounds__(256) void occupancy_kernel(const uint64_t* below, uint64_t* occ, const uint32_t digits, const CellRange r) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (tid >= static_cast<uint64_t>(r.nx) * r.ny * r.nz) return;
    const uint64_t c = cell_key(r.x0 + static_cast<uint32_t>(tid % r.nx), r.y0 + static_cast<uint32_t>((tid / r.nx) % r.ny),
                                r.z0 + static_cast<uint32_t>(tid / (static_cast<uint64_t>(r.nx) * r.ny)), digits);
    const uint64_t* child = below + c * 64u;
    uint64_t word = 0;
    for (uint32_t b = 0; b < 64u; ++b) word |= static_cast<uint64_t>(child[b] != 0ull) << b;
    occ[c] = word;
}

// (word != 0) << 32 | popcount(word): one exclusive scan of these gives, per cell, the rank of its node within its level (high half) and the
// index of its first child within the level below (low half); packed[n] = 0 makes scanned[n] the level's totals.
__global__ __launch_bounds__(256) void pack_kernel(const uint64_t* occ, uint64_t n, uint64_t* packed) {
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (c > n) return;
    const uint64_t w = c < n ? occ[c] : 0ull;
    packed[c] = (static_cast<uint64_t>(w != 0ull) << 32) | static_cast<uint64_t>(__popcll(w));
}

// The small levels (<= 4096 cells each: everything above level 2 of a 1024^3 world) in ONE workgroup: pack and exclusive scan of each, the
// totals of every level — level 2's from its own (device-wide) scan — into info[level].
struct SmallLevels { const uint64_t* occ[8]; uint64_t* scanned[8]; uint64_t cells[8]; uint32_t first, last; const uint64_t* big_total[8]; };
__global__ __launch_bounds__(1024) void small_levels_kernel(const SmallLevels s, uint64_t* info) {
    using Scan = hipcub::BlockScan<uint64_t, 1024>;
    __shared__ typename Scan::TempStorage temp;
    for (uint32_t l = 2; l < 8; ++l) if (s.big_total[l] && threadIdx.x == 0) info[l] = *s.big_total[l];
    for (uint32_t l = s.first; l <= s.last; ++l) {
        const uint64_t n = s.cells[l];
        uint64_t v[4], sum = 0;
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint64_t c = static_cast<uint64_t>(threadIdx.x) * 4u + k;
            const uint64_t w = c < n ? s.occ[l][c] : 0ull;
            v[k] = (static_cast<uint64_t>(w != 0ull) << 32) | static_cast<uint64_t>(__popcll(w));
            sum += v[k];
        }
        uint64_t before, total;
        Scan(temp).ExclusiveSum(sum, before, total);
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint64_t c = static_cast<uint64_t>(threadIdx.x) * 4u + k;
            if (c < n) s.scanned[l][c] = before;
            before += v[k];
        }
        if (threadIdx.x == 0) { s.scanned[l][n] = total; info[l] = total; }
        __syncthreads();
    }
}

// The nodes of level l: a non-empty cell's word is its node's mask; base = where the level below starts + the cell's first child.
// Level 2 also lists its non-empty cells (cells2), for the brick gather.
__global__ __launch_bounds__(256) void level_nodes_kernel(const uint64_t* occ, const uint64_t* scanned, uint64_t n, uint32_t level_start, uint32_t below_start,
                                                          uint4* tree, uint32_t* cells2) {
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (c >= n) return;
    const uint64_t w = occ[c];
    if (w == 0ull) return;
    const uint64_t s = scanned[c];
    const uint32_t rank = static_cast<uint32_t>(s >> 32);
    tree[level_start + rank] = make_uint4(static_cast<uint32_t>(w), static_cast<uint32_t>(w >> 32), below_start + static_cast<uint32_t>(s), 0u);
    if (cells2) cells2[rank] = static_cast<uint32_t>(c);
}

// One wave per non-empty level-2 cell, lane b = its brick b: the brick's mask, key and voxel count at its place in the sorted list.
__global__ __launch_bounds__(64) void gather_bricks_kernel(const uint64_t* occ2, const uint64_t* scanned2, const uint32_t* cells2, const uint64_t* masks,
                                                           uint64_t* masks_sorted, uint32_t* src, uint32_t* counts, uint32_t n_bricks) {
    const uint32_t c = cells2[blockIdx.x], b = threadIdx.x;
    if (blockIdx.x == 0 && b == 0) counts[n_bricks] = 0u;            // the scan's extra element: its result there is the number of voxels
    const uint64_t w = occ2[c];
    if (!((w >> b) & 1ull)) return;
    const uint32_t i = static_cast<uint32_t>(scanned2[c]) + static_cast<uint32_t>(__popcll(w & ((1ull << b) - 1ull)));
    const uint32_t g = c * 64u + b;
    const uint64_t m = masks[g];
    masks_sorted[i] = m; src[i] = g; counts[i] = static_cast<uint32_t>(__popcll(m));
}

// Sixteen lanes per brick (a brick of the benchmark world holds 12 voxels on average; one lane per voxel BIT left four lanes in five
// idle: 55 us per rebuild of 230 K bricks): the voxels' material ids at their ranks — copied from the previous build's array when the brick has
// not been touched since (its mask, hence every rank, is the same: a short coalesced run), from the dense store otherwise.
__global__ __launch_bounds__(256) void keyed_material_kernel(const KeyedCtx v, const uint64_t* masks_sorted, const uint32_t* src, const uint32_t* mat_base,
                                                             uint32_t n_bricks, const uint32_t* old_base, const uint32_t* previous, uint32_t* materials) {
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    const uint32_t i = static_cast<uint32_t>(tid >> 4), q = static_cast<uint32_t>(tid & 15u);
    if (i >= n_bricks) return;
    const uint64_t mask = masks_sorted[i];
    const uint32_t count = static_cast<uint32_t>(__popcll(mask));
    const uint32_t g = src[i], out = mat_base[i];
    if (previous && !v.dirty[g] && old_base[g] != 0xFFFFFFFFu) {
        const uint32_t from = old_base[g];
        for (uint32_t r = q; r < count; r += 16u) materials[out + r] = previous[from + r];
        return;
    }
    uint32_t bx, by, bz;
    key_cell(g, v.levels - 1u, bx, by, bz);
    for (uint32_t k = 0; k < 4u; ++k) {                      // this lane's four voxel bits: q, q + 16, q + 32, q + 48 = row (y = q >> 2, x = q & 3) of the z-slices
        const uint32_t bit = q + 16u * k;
        if (!((mask >> bit) & 1ull)) continue;
        const uint32_t rank = static_cast<uint32_t>(__popcll(mask & ((1ull << bit) - 1ull)));
        materials[out + rank] = v.ids[(static_cast<size_t>(bz * 4u + (bit >> 4)) * v.ny + (by * 4u + ((bit >> 2) & 3u))) * v.nx + bx * 4u + (bit & 3u)];
    }
}

// Brick nodes, and the bookkeeping for the next build: where each brick's ids now lie, nothing dirty.
__global__ __launch_bounds__(256) void keyed_brick_nodes_kernel(const uint64_t* masks_sorted, const uint32_t* src, const uint32_t* mat_base, uint32_t n, uint4* out,
                                                                uint32_t* old_base, uint8_t* dirty) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t m = masks_sorted[i];
    out[i] = make_uint4(static_cast<uint32_t>(m), static_cast<uint32_t>(m >> 32), mat_base[i], 0u);
    old_base[src[i]] = mat_base[i];
    dirty[src[i]] = 0;
}

KeyedCtx keyed_ctx(const GpuVolume& v) {
    KeyedCtx k{};
    k.density = v.d_density; k.ids = v.d_ids; k.nx = v.nx; k.ny = v.ny; k.nz = v.nz; k.levels = v.levels; k.masks = v.d_masks; k.dirty = v.d_dirty;
    return k;
}

// Masks of the bricks that contain voxels [lo, hi) and the occupancy words above them, level by level.
GpuBuildStatus keyed_refresh(GpuVolume* v, const uint32_t lo[3], const uint32_t hi[3], std::string* why) {
    if (hi[0] <= lo[0] || hi[1] <= lo[1] || hi[2] <= lo[2]) return GpuBuildStatus::Ok;
    const KeyedCtx k = keyed_ctx(*v);
    CellRange ranges[9] = {};
    uint64_t totals[9] = {};
    for (uint32_t l = 1; l <= v->levels; ++l) {
        CellRange& r = ranges[l];
        r.x0 = lo[0] >> (2u * l); r.y0 = lo[1] >> (2u * l); r.z0 = lo[2] >> (2u * l);
        r.nx = ((hi[0] - 1u) >> (2u * l)) - r.x0 + 1u; r.ny = ((hi[1] - 1u) >> (2u * l)) - r.y0 + 1u; r.nz = ((hi[2] - 1u) >> (2u * l)) - r.z0 + 1u;
        totals[l] = static_cast<uint64_t>(r.nx) * r.ny * r.nz;
    }
    // an edit (few bricks): a wave per brick, then every level above in one workgroup; an upload (the whole box): a lane per brick, a launch per level
    const bool small = totals[1] <= 65536u && (v->levels < 2 || totals[2] <= 4096u);
    if (small) {
        hipLaunchKernelGGL(keyed_brick_wave_kernel, dim3(static_cast<uint32_t>(totals[1])), dim3(64), 0, nullptr, k, ranges[1]);
        GB_TRY(hipGetLastError());
        if (v->levels >= 2) {
            OccLevels o{};
            o.first = 2; o.last = v->levels;
            for (uint32_t l = 2; l <= v->levels; ++l) { o.below[l] = l == 2 ? v->d_masks : v->d_occ[l - 1]; o.occ[l] = v->d_occ[l]; o.digits[l] = v->levels - l; o.range[l] = ranges[l]; }
            hipLaunchKernelGGL(occupancy_levels_kernel, dim3(1), dim3(256), 0, nullptr, o);
            GB_TRY(hipGetLastError());
        }
    } else {
        for (uint32_t l = 1; l <= v->levels; ++l) {
            if (l == 1) hipLaunchKernelGGL(keyed_brick_kernel, dim3(blocks_for(totals[l])), dim3(256), 0, nullptr, k, ranges[l]);
            else hipLaunchKernelGGL(occupancy_kernel, dim3(blocks_for(totals[l])), dim3(256), 0, nullptr, l == 2 ? v->d_masks : v->d_occ[l - 1], v->d_occ[l], v->levels - l, ranges[l]);
            GB_TRY(hipGetLastError());
        }
    }
    for (int a = 0; a < 3; ++a) { v->edit_lo[a] = std::min(v->edit_lo[a], lo[a]); v->edit_hi[a] = std::max(v->edit_hi[a], hi[a]); }
    return GpuBuildStatus::Ok;
}

template <class T> hipError_t grow(T** p, uint64_t* capacity, uint64_t need, uint64_t headroom_num = 5, uint64_t headroom_den = 4) {
    if (*capacity >= need && *p) return hipSuccess;
    if (*p) { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) return e; (void)hipFree(*p); *p = nullptr; *capacity = 0; }
    const uint64_t want = std::max<uint64_t>(need * headroom_num / headroom_den, need) + 64u;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), want * sizeof(T));
    if (e == hipSuccess) *capacity = want;
    return e;
}

// The rebuild of a keyed volume: scans over the occupancy pyramid, the gather of the non-empty bricks, the node writes.
// BLOK_VOLUME_TIMING=1: host wall clock between the phases of a rebuild, to stderr (diagnostic; scripts/r04/edit_latency.py)
struct PhaseClock {
    bool on; std::chrono::steady_clock::time_point t; std::string line;
    PhaseClock() : on(std::getenv("BLOK_VOLUME_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
    void mark(const char* what) { if (!on) return; const auto n = std::chrono::steady_clock::now(); char b[64]; std::snprintf(b, sizeof b, " %s %.1f", what, std::chrono::duration<double, std::micro>(n - t).count()); line += b; t = n; }
    ~PhaseClock() { if (on) std::fprintf(stderr, "[keyed_build us]%s\n", line.c_str()); }
};

GpuBuildStatus keyed_build(GpuVolume* v, GpuTree* out, std::string* why) {
    PhaseClock clock;
    auto& S = v->scratch;
    const uint32_t L = v->levels;
    uint64_t cells[8] = {};                                                 // cells of level l = 64^(L - l)
    for (uint32_t l = 2; l <= L; ++l) cells[l] = 1ull << (6u * (L - l));
    // 1. per level: (word != 0) << 32 | popcount, exclusively scanned, totals in the extra element — levels of <= 4096 cells all in one
    //    workgroup, larger ones with a device-wide scan each
    if (!S.d_info) GB_TRY(hipMalloc(reinterpret_cast<void**>(&S.d_info), 32 * sizeof(uint64_t)));
    SmallLevels small{};
    small.first = 8; small.last = 0;
    for (uint32_t l = 2; l <= L; ++l) {
        if (!S.d_packed[l]) { GB_TRY(hipMalloc(reinterpret_cast<void**>(&S.d_packed[l]), (cells[l] + 1u) * sizeof(uint64_t))); GB_TRY(hipMalloc(reinterpret_cast<void**>(&S.d_scanned[l]), (cells[l] + 1u) * sizeof(uint64_t))); }
        if (cells[l] <= 4096u) { small.occ[l] = v->d_occ[l]; small.scanned[l] = S.d_scanned[l]; small.cells[l] = cells[l]; small.first = std::min(small.first, l); small.last = std::max(small.last, l); continue; }
        hipLaunchKernelGGL(pack_kernel, dim3(blocks_for(cells[l] + 1u)), dim3(256), 0, nullptr, v->d_occ[l], cells[l], S.d_packed[l]);
        GB_TRY(hipGetLastError());
        size_t need = 0;
        GB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, S.d_packed[l], S.d_scanned[l], static_cast<int>(cells[l] + 1u)));
        if (need > S.scan_temp_bytes) { if (S.d_scan_temp) { GB_TRY(hipDeviceSynchronize()); (void)hipFree(S.d_scan_temp); } S.d_scan_temp = nullptr; GB_TRY(hipMalloc(&S.d_scan_temp, need * 2 + 256)); S.scan_temp_bytes = need * 2 + 256; }
        size_t bytes = S.scan_temp_bytes;
        GB_TRY(hipcub::DeviceScan::ExclusiveSum(S.d_scan_temp, bytes, S.d_packed[l], S.d_scanned[l], static_cast<int>(cells[l] + 1u)));
        small.big_total[l] = S.d_scanned[l] + cells[l];
    }
    hipLaunchKernelGGL(small_levels_kernel, dim3(1), dim3(1024), 0, nullptr, small, S.d_info);      // (small.first > small.last: just the big levels' totals)
    GB_TRY(hipGetLastError());
    clock.mark("scans-enqueued");
    uint64_t totals[8] = {};
    GB_TRY(hipMemcpy(totals, S.d_info, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    clock.mark("totals-read");              // the one wait in the middle: the launch sizes below
    uint32_t n_nodes[9] = {};                                               // nodes per level; n_nodes[1] = bricks
    for (uint32_t l = 2; l <= L; ++l) { n_nodes[l] = static_cast<uint32_t>(totals[l] >> 32); if (l == 2) n_nodes[1] = static_cast<uint32_t>(totals[l]); }
    const uint32_t n_bricks = n_nodes[1];
    if (n_bricks == 0) return GpuBuildStatus::UseHostBuilder;
    uint32_t start[9] = {};                                                 // root first: start[L] = 0, ..., start[1] = where the bricks begin
    { uint64_t at = 0; for (uint32_t l = L; l >= 1; --l) { start[l] = static_cast<uint32_t>(at); at += n_nodes[l]; } if (at > 0xFFFFFFFFull) { *why = "volume: more than 2^32 tree nodes"; return GpuBuildStatus::Unsupported; } }
    const uint64_t n_tree = static_cast<uint64_t>(start[1]) + n_bricks;
    const int next = S.current == 0 ? 1 : 0;
    GB_TRY(grow(&S.d_tree[next], &S.tree_capacity[next], n_tree));
    // by its bound (no wait for the voxel count), with a sixteenth to spare: sized exactly, every edit that added a brick re-allocated 59 MB —
    // a device-wide wait, a free and a malloc, 210 us of the 0.44 ms a radius-8 brush took to become a tree (BLOK_VOLUME_TIMING, round 4)
    GB_TRY(grow(&S.d_materials[next], &S.material_capacity[next], static_cast<uint64_t>(n_bricks) * 64u, 17, 16));
    if (S.brick_capacity < n_bricks + 1ull) {
        GB_TRY(hipDeviceSynchronize());
        for (void* p : {static_cast<void*>(S.d_masks_sorted), static_cast<void*>(S.d_src), static_cast<void*>(S.d_counts), static_cast<void*>(S.d_mat_base)}) if (p) (void)hipFree(p);
        const uint64_t want = (n_bricks + 1ull) * 5u / 4u + 64u;
        GB_TRY(hipMalloc(reinterpret_cast<void**>(&S.d_masks_sorted), want * sizeof(uint64_t))); GB_TRY(hipMalloc(reinterpret_cast<void**>(&S.d_src), want * sizeof(uint32_t)));
        GB_TRY(hipMalloc(reinterpret_cast<void**>(&S.d_counts), want * sizeof(uint32_t))); GB_TRY(hipMalloc(reinterpret_cast<void**>(&S.d_mat_base), want * sizeof(uint32_t)));
        S.brick_capacity = want;
    }
    GB_TRY(grow(&S.d_cells2, &S.cells2_capacity, n_nodes[2]));
    clock.mark("grown");
    // 2. nodes of levels L .. 2, the list of non-empty level-2 cells
    for (uint32_t l = L; l >= 2; --l) {
        hipLaunchKernelGGL(level_nodes_kernel, dim3(blocks_for(cells[l])), dim3(256), 0, nullptr, v->d_occ[l], S.d_scanned[l], cells[l], start[l], start[l - 1],
                           S.d_tree[next], l == 2 ? S.d_cells2 : nullptr);
        GB_TRY(hipGetLastError());
    }
    clock.mark("level-nodes");
    // 3. the non-empty bricks in key order, their voxel counts, the material offsets
    hipLaunchKernelGGL(gather_bricks_kernel, dim3(n_nodes[2]), dim3(64), 0, nullptr, v->d_occ[2], S.d_scanned[2], S.d_cells2, v->d_masks, S.d_masks_sorted, S.d_src, S.d_counts, n_bricks);
    GB_TRY(hipGetLastError());
    {
        size_t need = 0;
        GB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, S.d_counts, S.d_mat_base, static_cast<int>(n_bricks + 1u)));
        if (need > S.scan_temp_bytes) { GB_TRY(hipDeviceSynchronize()); if (S.d_scan_temp) (void)hipFree(S.d_scan_temp); S.d_scan_temp = nullptr; GB_TRY(hipMalloc(&S.d_scan_temp, need * 2 + 256)); S.scan_temp_bytes = need * 2 + 256; }
        size_t bytes = S.scan_temp_bytes;
        GB_TRY(hipcub::DeviceScan::ExclusiveSum(S.d_scan_temp, bytes, S.d_counts, S.d_mat_base, static_cast<int>(n_bricks + 1u)));
    }
    clock.mark("gather+scan");
    // 4. material ids (untouched bricks from the previous build's array), brick nodes
    const KeyedCtx k = keyed_ctx(*v);
    const uint32_t* previous = S.have_previous_materials && S.current >= 0 ? S.d_materials[S.current] : nullptr;
    hipLaunchKernelGGL(keyed_material_kernel, dim3(blocks_for(static_cast<uint64_t>(n_bricks) * 16u)), dim3(256), 0, nullptr, k, S.d_masks_sorted, S.d_src, S.d_mat_base,
                       n_bricks, S.d_old_base, previous, S.d_materials[next]);
    GB_TRY(hipGetLastError());
    hipLaunchKernelGGL(keyed_brick_nodes_kernel, dim3(blocks_for(n_bricks)), dim3(256), 0, nullptr, S.d_masks_sorted, S.d_src, S.d_mat_base, n_bricks, S.d_tree[next] + start[1],
                       S.d_old_base, v->d_dirty);
    GB_TRY(hipGetLastError());
    clock.mark("nodes-enqueued");
    uint32_t n_voxels = 0;
    GB_TRY(hipMemcpy(&n_voxels, S.d_mat_base + n_bricks, sizeof(uint32_t), hipMemcpyDeviceToHost));
    clock.mark("count-read");      // also: everything above has completed
    S.current = next; S.have_previous_materials = true;
    out->d_nodes = S.d_tree[next]; out->d_materials = S.d_materials[next]; out->owned_by_volume = true;
    out->n_nodes = n_tree; out->n_voxels = n_voxels; out->levels = L;
    for (int a = 0; a < 3; ++a) out->origin[a] = v->origin[a];
    return GpuBuildStatus::Ok;
}

}  // namespace

GpuBuildStatus gpu_volume_create(const int32_t origin[3], uint32_t nx, uint32_t ny, uint32_t nz, uint32_t chunk, float voxel_size,
                                 GpuVolume* out, std::string* why, bool allow_keyed) {
    *out = GpuVolume{};
    if (!nx || !ny || !nz || !chunk || !(voxel_size == 1.0f)) { *why = "volume: empty box, zero chunk size or voxel size other than 1"; return GpuBuildStatus::Unsupported; }
    for (int a = 0; a < 3; ++a) {
        const int64_t hi = int64_t(origin[a]) + (a == 0 ? nx : a == 1 ? ny : nz);
        if (origin[a] < -32768 || hi > 32768) { *why = "world voxel coordinates exceed int16 (hit records carry int16)"; return GpuBuildStatus::Unsupported; }
    }
    const uint32_t extent = std::max(nx, std::max(ny, nz));
    uint32_t levels = 1;
    while ((uint64_t(1) << (2 * levels)) < extent) ++levels;
    if (levels > kMaxLevels) { *why = "world extent exceeds 4^7 voxels per axis"; return GpuBuildStatus::Unsupported; }
    GpuVolume v;
    v.nx = nx; v.ny = ny; v.nz = nz; v.nbx = (nx + 3) / 4; v.nby = (ny + 3) / 4; v.nbz = (nz + 3) / 4; v.levels = levels;
    for (int a = 0; a < 3; ++a) v.origin[a] = origin[a];
    v.chunk = chunk; v.voxel_size = voxel_size;
    if (v.bricks() > 0x7FFFFFFFull) { *why = "volume: more than 2^31 bricks"; return GpuBuildStatus::Unsupported; }
    // keyed layout (gpu_build.h) whenever its 64^(L-1) mask words are affordable: always up to 5 levels (16.7 M words), beyond that for
    // volumes that fill most of their cube
    v.n_keys = levels >= 2 ? 1ull << (6u * (levels - 1u)) : 0;
    v.keyed = allow_keyed && levels >= 2 && v.n_keys <= std::max<uint64_t>(1ull << 24, 8ull * v.bricks()) && v.n_keys <= 0x7FFFFFFFull;
    DeviceBuffers mem;
    GB_TRY(mem.alloc(&v.d_density, v.cells())); GB_TRY(mem.alloc(&v.d_ids, v.cells()));
    GB_TRY(hipMemset(v.d_density, 0, v.cells() * sizeof(float))); GB_TRY(hipMemset(v.d_ids, 0, v.cells() * sizeof(uint32_t)));
    std::vector<void*> keep = {v.d_density, v.d_ids};
    if (v.keyed) {
        GB_TRY(mem.alloc(&v.d_masks, v.n_keys)); GB_TRY(mem.alloc(&v.d_dirty, v.n_keys)); GB_TRY(mem.alloc(&v.scratch.d_old_base, v.n_keys));
        GB_TRY(hipMemset(v.d_masks, 0, v.n_keys * sizeof(uint64_t))); GB_TRY(hipMemset(v.d_dirty, 1, v.n_keys)); GB_TRY(hipMemset(v.scratch.d_old_base, 0xFF, v.n_keys * sizeof(uint32_t)));
        keep.insert(keep.end(), {static_cast<void*>(v.d_masks), static_cast<void*>(v.d_dirty), static_cast<void*>(v.scratch.d_old_base)});
        for (uint32_t l = 2; l <= levels; ++l) {
            const uint64_t n = 1ull << (6u * (levels - l));
            GB_TRY(mem.alloc(&v.d_occ[l], n)); GB_TRY(hipMemset(v.d_occ[l], 0, n * sizeof(uint64_t)));
            keep.push_back(v.d_occ[l]);
        }
    } else {
        GB_TRY(mem.alloc(&v.d_masks, v.bricks())); GB_TRY(mem.alloc(&v.d_flag, v.bricks() + 1)); GB_TRY(mem.alloc(&v.d_slot, v.bricks() + 1));
        GB_TRY(hipMemset(v.d_masks, 0, v.bricks() * sizeof(uint64_t))); GB_TRY(hipMemset(v.d_flag, 0, (v.bricks() + 1) * sizeof(uint32_t)));
        keep.insert(keep.end(), {static_cast<void*>(v.d_masks), static_cast<void*>(v.d_flag), static_cast<void*>(v.d_slot)});
    }
    GB_TRY(hipDeviceSynchronize());
    for (void* p : keep) mem.release(p);
    *out = v;
    return GpuBuildStatus::Ok;
}

void gpu_volume_destroy(GpuVolume* v) {
    if (!v) return;
    auto& S = v->scratch;
    for (void* p : {static_cast<void*>(v->d_density), static_cast<void*>(v->d_ids), static_cast<void*>(v->d_masks), static_cast<void*>(v->d_flag), static_cast<void*>(v->d_slot),
                    static_cast<void*>(v->d_dirty), S.d_scan_temp, static_cast<void*>(S.d_cells2), static_cast<void*>(S.d_masks_sorted), static_cast<void*>(S.d_src),
                    static_cast<void*>(S.d_counts), static_cast<void*>(S.d_mat_base), static_cast<void*>(S.d_old_base), static_cast<void*>(S.d_info),
                    static_cast<void*>(S.d_tree[0]), static_cast<void*>(S.d_tree[1]), static_cast<void*>(S.d_materials[0]), static_cast<void*>(S.d_materials[1])})
        if (p) (void)hipFree(p);
    for (int l = 0; l < 8; ++l) for (void* p : {static_cast<void*>(v->d_occ[l]), static_cast<void*>(S.d_packed[l]), static_cast<void*>(S.d_scanned[l])}) if (p) (void)hipFree(p);
    *v = GpuVolume{};
}

GpuBuildStatus gpu_volume_upload(GpuVolume* v, const float* density, const uint32_t* ids, std::string* why) {
    if (density) GB_TRY(hipMemcpy(v->d_density, density, v->cells() * sizeof(float), hipMemcpyHostToDevice));
    else GB_TRY(hipMemset(v->d_density, 0, v->cells() * sizeof(float)));
    if (ids) GB_TRY(hipMemcpy(v->d_ids, ids, v->cells() * sizeof(uint32_t), hipMemcpyHostToDevice));
    else GB_TRY(hipMemset(v->d_ids, 0, v->cells() * sizeof(uint32_t)));
    const uint32_t lo[3] = {0, 0, 0}, hi[3] = {v->nx, v->ny, v->nz};
    return volume_refresh(v, lo, hi, why);
}

GpuBuildStatus gpu_volume_download(const GpuVolume* v, float* density, uint32_t* ids, std::string* why) {
    if (density) GB_TRY(hipMemcpy(density, v->d_density, v->cells() * sizeof(float), hipMemcpyDeviceToHost));
    if (ids) GB_TRY(hipMemcpy(ids, v->d_ids, v->cells() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return GpuBuildStatus::Ok;
}

GpuBuildStatus gpu_volume_set_voxels(GpuVolume* v, const int32_t* xyz, const uint32_t* material, const float* density, size_t n,
                                     std::string* why) {
    if (!n) return GpuBuildStatus::Ok;
    std::vector<VoxelEdit> edits(n);
    uint32_t lo[3] = {v->nx, v->ny, v->nz}, hi[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        uint32_t l[3];
        for (int a = 0; a < 3; ++a) {
            const int64_t c = int64_t(xyz[3 * i + a]) - v->origin[a];
            if (c < 0 || c >= int64_t(a == 0 ? v->nx : a == 1 ? v->ny : v->nz)) { *why = "set_voxels: voxel outside the resident volume"; return GpuBuildStatus::Unsupported; }
            l[a] = static_cast<uint32_t>(c);
            lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], l[a] + 1u);
        }
        edits[i] = VoxelEdit{static_cast<uint32_t>(l[0] + (static_cast<size_t>(l[2]) * v->ny + l[1]) * v->nx), material ? material[i] : 0u,
                             density ? density[i] : 1.0f};
    }
    if (v->cells() > 0xFFFFFFFFull) { *why = "set_voxels: volume larger than 2^32 cells"; return GpuBuildStatus::Unsupported; }
    // sequential semantics: the last write of a voxel wins
    std::stable_sort(edits.begin(), edits.end(), [](const VoxelEdit& a, const VoxelEdit& b) { return a.index < b.index; });
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) { if (i + 1 < n && edits[i + 1].index == edits[i].index) continue; edits[m++] = edits[i]; }
    DeviceBuffers mem;
    VoxelEdit* d_edits;
    GB_TRY(mem.alloc(&d_edits, m));
    GB_TRY(hipMemcpy(d_edits, edits.data(), m * sizeof(VoxelEdit), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(volume_set_kernel, dim3(blocks_for(m)), dim3(256), 0, nullptr, v->d_density, v->d_ids, d_edits, static_cast<uint32_t>(m));
    GB_TRY(hipGetLastError());
    v->edit_may_add = true;                               // (a written density may be positive)
    const GpuBuildStatus st = volume_refresh(v, lo, hi, why);
    GB_TRY(hipDeviceSynchronize());
    return st;
}

GpuBuildStatus gpu_volume_brush(GpuVolume* v, const float center[3], float radius, float value, int mode, std::string* why) {
    BrushCtx b{};
    b.density = v->d_density; b.nx = v->nx; b.ny = v->ny; b.nz = v->nz;
    uint32_t lo[3], hi[3];
    // floor() of a NaN or of a value beyond int32 is undefined behaviour in the casts below: refuse such a brush up front
    if (!std::isfinite(radius) || !(std::fabs(radius) < 1.0e9f)) { *why = "brush: radius is not a finite number of voxels"; return GpuBuildStatus::Unsupported; }
    for (int a = 0; a < 3; ++a)
        if (!std::isfinite(center[a]) || !(std::fabs(center[a]) < 1.0e9f)) { *why = "brush: centre is not finite"; return GpuBuildStatus::Unsupported; }
    for (int a = 0; a < 3; ++a) {
        b.origin[a] = v->origin[a];
        const int32_t gmin = static_cast<int32_t>(std::floor(center[a] - radius));               // brush.cpp:20
        const int32_t gmax = static_cast<int32_t>(std::floor(center[a] + radius)) + 1;           // brush.cpp:21-22
        const int64_t dim = a == 0 ? v->nx : a == 1 ? v->ny : v->nz;
        if (gmin < v->origin[a] || gmax > v->origin[a] + dim) { *why = "brush: bounding box leaves the resident volume"; return GpuBuildStatus::Unsupported; }
        b.lo[a] = gmin;
        lo[a] = static_cast<uint32_t>(gmin - v->origin[a]); hi[a] = static_cast<uint32_t>(gmax - v->origin[a]);
    }
    b.ex = hi[0] - lo[0]; b.ey = hi[1] - lo[1]; b.ez = hi[2] - lo[2];
    b.chunk = static_cast<int32_t>(v->chunk); b.voxel_size = v->voxel_size;
    b.cx = center[0]; b.cy = center[1]; b.cz = center[2]; b.radius = radius; b.value = value; b.mode = mode;
    if (mode == 0 && value > 0.0f) v->edit_may_add = true;          // max(density, value) can fill; min(density, value) never does (brush.cpp:52-57)
    const uint64_t total = static_cast<uint64_t>(b.ex) * b.ey * b.ez;
    if (!total) return GpuBuildStatus::Ok;
    hipLaunchKernelGGL(volume_brush_kernel, dim3(blocks_for(total)), dim3(256), 0, nullptr, b);
    GB_TRY(hipGetLastError());
    return volume_refresh(v, lo, hi, why);
}

GpuBuildStatus gpu_volume_build(GpuVolume* v, GpuTree* out, std::string* why) {
    *out = GpuTree{};
    if (v->keyed) return keyed_build(v, out, why);
    const DenseCtx d = volume_ctx(*v);
    const uint64_t total = v->bricks();
    DeviceBuffers mem;
    return finish_from_masks(mem, total, v->d_masks, v->d_flag, v->d_slot, v->levels, v->origin,
        [&](const uint32_t* slot, uint64_t* keys, uint32_t* src) {
            hipLaunchKernelGGL(dense_key_kernel, dim3(blocks_for(total)), dim3(256), 0, nullptr, d, v->d_masks, slot, total, keys, src);
        },
        [&](const uint64_t* masks_sorted, const uint32_t* src_sorted, const uint32_t* mat_base, uint32_t n_bricks, uint32_t* materials) {
            hipLaunchKernelGGL(dense_material_kernel, dim3(blocks_for(static_cast<uint64_t>(n_bricks) * 64u)), dim3(256), 0, nullptr,
                               d, masks_sorted, src_sorted, mat_base, n_bricks, materials);
        }, out, why);
}

}  // namespace blok
