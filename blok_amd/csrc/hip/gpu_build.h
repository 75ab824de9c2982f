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
};

GpuBuildStatus gpu_volume_create(const int32_t origin[3], uint32_t nx, uint32_t ny, uint32_t nz, uint32_t chunk, float voxel_size,
                                 GpuVolume* out, std::string* why);
void gpu_volume_destroy(GpuVolume* v);
// Whole-box upload from host arrays (either may be null = zeros) and recomputation of every brick mask.
GpuBuildStatus gpu_volume_upload(GpuVolume* v, const float* density, const uint32_t* ids, std::string* why);
GpuBuildStatus gpu_volume_download(const GpuVolume* v, float* density, uint32_t* ids, std::string* why);
// = ChunkManager::setVoxelMaterial for n world voxels (later entries win); voxels outside the box -> Unsupported.
GpuBuildStatus gpu_volume_set_voxels(GpuVolume* v, const int32_t* xyz, const uint32_t* material, const float* density, size_t n,
                                     std::string* why);
// = applyBrush (brush.cpp:13-63): mode 0 ADD (max), 1 SUBTRACT (min); the brush's bounding box must lie in the box.
GpuBuildStatus gpu_volume_brush(GpuVolume* v, const float center[3], float radius, float value, int mode, std::string* why);
// 64-tree of the current contents (UseHostBuilder = the volume is empty).
GpuBuildStatus gpu_volume_build(GpuVolume* v, GpuTree* out, std::string* why);

}  // namespace blok
#endif
