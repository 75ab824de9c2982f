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

}  // namespace blok
#endif
