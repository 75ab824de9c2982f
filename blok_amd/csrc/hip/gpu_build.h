// Device-side build of the 64-tree (gpu_build.hip).
#ifndef BLOK_GPU_BUILD_H
#define BLOK_GPU_BUILD_H
#include <hip/hip_runtime.h>

#include <string>

#include "blok_hip.h"
#include "tree.h"

namespace blok {

struct GpuTree {
    uint4* d_nodes = nullptr;        // root first; ownership passes to the caller on Ok
    uint32_t* d_materials = nullptr;
    size_t n_nodes = 0;
    uint64_t n_voxels = 0;
    uint32_t levels = 0;
    int32_t origin[3] = {0, 0, 0};
    bool owned_by_volume = false;    // the arrays belong to a GpuVolume's scratch (never freed by the holder)
};

enum class GpuBuildStatus { Ok, UseHostBuilder, Unsupported, HipError, OutOfMemory };

// UseHostBuilder: the world is valid but outside what the kernels cover (empty, sub-chunks smaller than a brick or of
// mixed sizes, overlapping sub-chunks); the caller then takes the general host path (tree_build.cpp).
GpuBuildStatus gpu_build_tree(const blok_svo_node* nodes, size_t n_nodes, const blok_sub_chunk* subs, size_t n_subs,
                              GpuTree* out, std::string* why);

// Same, from a dense id grid ids[x + y*nx + z*nx*ny] (0 = empty) whose voxel (0,0,0) sits at world `origin`.
GpuBuildStatus gpu_build_tree_dense(const uint32_t* ids, uint32_t nx, uint32_t ny, uint32_t nz, const int32_t origin[3],
                                    GpuTree* out, std::string* why);

// ---- device-resident dense voxel store: the reference's Chunk::density / Chunk::materialIds (blok/src/chunk.hpp:33-42)
// for one box of the world, kept in HBM together with the 64-bit voxel mask of every 4^3 brick, so that edits
// (brush.cpp:13-63, chunk_manager.cpp:316-328) and the rebuild they trigger (chunk_manager.cpp:106-140) never leave
// the device.  A voxel is filled iff density > 0 (chunk_manager.cpp:121).
struct GpuVolume {
    float* d_density = nullptr;      // [x + y*nx + z*nx*ny]
    uint32_t* d_ids = nullptr;
    uint64_t* d_masks = nullptr;     // one per brick, brick (bx, by, bz) at bx + by*nbx + bz*nbx*nby
    uint32_t* d_flag = nullptr;      // mask != 0, total + 1 entries (the last one is a zero sentinel for the scan)
    uint32_t* d_slot = nullptr;      // scan scratch, total + 1
    uint32_t nx = 0, ny = 0, nz = 0, nbx = 0, nby = 0, nbz = 0, levels = 0;
    int32_t origin[3] = {0, 0, 0};
    uint32_t chunk = 128;            // ChunkManager's chunk edge (the brush computes voxel centres per chunk)
    float voxel_size = 1.0f;
    uint64_t cells() const { return static_cast<uint64_t>(nx) * ny * nz; }
    uint64_t bricks() const { return static_cast<uint64_t>(nbx) * nby * nbz; }

    // KEY layout (volumes whose tree has <= 5 levels, and larger ones that fill their cube; gpu_build.hip: rebuild in ~0.2 ms).  Bricks are
    // indexed by their key — the 2-bit digit triples of levels 1..L-1 of the brick's coordinates, least significant level first: the
    // order of the tree's level-1 nodes — so d_masks has 64^(L-1) entries and the non-empty bricks, taken in index order, ARE the sorted
    // brick list: no scan over all bricks, no sort.  Above the masks lies a pyramid of occupancy words: bit b of d_occ[l][c] = cell
    // 64 c + b of level l-1 holds a voxel (l = 2: brick 64 c + b), kept up to date by the edits for the cells they touch.  A non-zero
    // word IS the mask of the tree node of that cell, and the two exclusive scans of a level — of the words' popcounts and of
    // "word != 0" — are the nodes' child indices and the nodes' own ranks.  A rebuild is those scans (262 144 words at level 2 of a
    // 1024^3 world, a handful above), the gather of the non-empty bricks' masks and material ids, and the node writes.
    bool keyed = false;
    uint64_t n_keys = 0;                         // 64^(levels-1)
    uint64_t* d_occ[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // [2 .. levels]
    uint8_t* d_dirty = nullptr;                  // per brick key: its voxels may have changed since the last build (material ids are gathered again)
    // kept between rebuilds (grown on demand): scratch, and two sets of output arrays so that the tree the tracer holds stays valid while the next is built
    struct Scratch {
        void* d_scan_temp = nullptr; size_t scan_temp_bytes = 0;
        uint64_t* d_packed[8] = {};              // per level: (word != 0) << 32 | popcount(word), and ...
        uint64_t* d_scanned[8] = {};             // ... its exclusive scan: rank of the cell's node << 32 | index of its first child within the level below
        uint32_t* d_cells2 = nullptr; uint64_t cells2_capacity = 0;      // the non-empty level-2 cells, in order
        uint64_t* d_masks_sorted = nullptr; uint32_t *d_src = nullptr, *d_counts = nullptr, *d_mat_base = nullptr; uint64_t brick_capacity = 0;
        uint32_t* d_old_base = nullptr;          // per brick key: where the brick's material ids lie in the previous output (0xFFFFFFFF: nowhere)
        uint64_t* d_info = nullptr;              // device words the kernels pass totals through; [0..7] level totals, [8] voxels, [9..16] level offsets
        uint4* d_tree[2] = {nullptr, nullptr}; uint64_t tree_capacity[2] = {0, 0};
        uint32_t* d_materials[2] = {nullptr, nullptr}; uint64_t material_capacity[2] = {0, 0};
        int current = -1;                        // output set the tracer holds (-1: none)
        bool have_previous_materials = false;
    } scratch;
    // voxels edited since the last build (box-local, half-open); empty = lo > hi
    uint32_t edit_lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, edit_hi[3] = {0, 0, 0};
    bool edit_may_add = false;                   // ... and one of those edits may have FILLED a voxel (an ADD brush with a positive value, setVoxel): what the shadow rays' map has to know
};

GpuBuildStatus gpu_volume_create(const int32_t origin[3], uint32_t nx, uint32_t ny, uint32_t nz, uint32_t chunk, float voxel_size,
                                 GpuVolume* out, std::string* why, bool allow_keyed = true);
void gpu_volume_destroy(GpuVolume* v);
// Whole-box upload from host arrays (either may be null = zeros) and recomputation of every brick mask.
GpuBuildStatus gpu_volume_upload(GpuVolume* v, const float* density, const uint32_t* ids, std::string* why);
GpuBuildStatus gpu_volume_download(const GpuVolume* v, float* density, uint32_t* ids, std::string* why);
// = ChunkManager::setVoxelMaterial for n world voxels (later entries win); voxels outside the box -> Unsupported.
GpuBuildStatus gpu_volume_set_voxels(GpuVolume* v, const int32_t* xyz, const uint32_t* material, const float* density, size_t n,
                                     std::string* why);
// = applyBrush (brush.cpp:13-63): mode 0 ADD (max), 1 SUBTRACT (min); the brush's bounding box must lie in the box.
GpuBuildStatus gpu_volume_brush(GpuVolume* v, const float center[3], float radius, float value, int mode, std::string* why);
// 64-tree of the current contents (UseHostBuilder = the volume is empty).  keyed volumes: out->d_nodes / d_materials stay OWNED BY THE
// VOLUME (out->owned_by_volume) and remain valid until the build after the next.
GpuBuildStatus gpu_volume_build(GpuVolume* v, GpuTree* out, std::string* why);

}  // namespace blok
#endif
