// C ABI, device-resident dense voxel store (include/blok_hip.h: blok_hip_volume_*; kernels in gpu_build.hip).
#include "api_internal.h"

using namespace blok_api;

extern "C" {

namespace {
int volume_status(blok_hip_ctx* ctx, blok::GpuBuildStatus st, const std::string& why) {
    switch (st) {
        case blok::GpuBuildStatus::Ok: return BLOK_OK;
        case blok::GpuBuildStatus::Unsupported: return set_error(ctx, BLOK_ERR_UNSUPPORTED, why);
        case blok::GpuBuildStatus::OutOfMemory: return set_error(ctx, BLOK_ERR_OOM, why);
        case blok::GpuBuildStatus::HipError: return set_error(ctx, BLOK_ERR_HIP, why);
        default: return set_error(ctx, BLOK_ERR_INVALID_ARG, why.empty() ? "volume operation not applicable" : why);
    }
}
int need_volume(blok_hip_ctx* ctx) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!ctx->has_volume) return set_error(ctx, BLOK_ERR_NO_WORLD, "no resident volume (blok_hip_volume_create)");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return BLOK_OK;
}
}  // namespace

int blok_hip_volume_create(blok_hip_ctx* ctx, const int32_t origin[3], uint32_t nx, uint32_t ny, uint32_t nz,
                           uint32_t chunk_size, float voxel_size) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->has_volume) { blok::gpu_volume_destroy(&ctx->volume); ctx->has_volume = false; }
    const int32_t o[3] = {origin ? origin[0] : 0, origin ? origin[1] : 0, origin ? origin[2] : 0};
    std::string why;
    const blok::GpuBuildStatus st = blok::gpu_volume_create(o, nx, ny, nz, chunk_size, voxel_size, &ctx->volume, &why);
    ctx->has_volume = st == blok::GpuBuildStatus::Ok;
    return volume_status(ctx, st, why);
}

int blok_hip_volume_destroy(blok_hip_ctx* ctx) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (ctx->has_volume) { (void)hipSetDevice(ctx->device); blok::gpu_volume_destroy(&ctx->volume); ctx->has_volume = false; }
    return BLOK_OK;
}

int blok_hip_volume_upload(blok_hip_ctx* ctx, const float* density, const uint32_t* material_ids) {
    int rc = need_volume(ctx);
    if (rc != BLOK_OK) return rc;
    std::string why;
    return volume_status(ctx, blok::gpu_volume_upload(&ctx->volume, density, material_ids, &why), why);
}

int blok_hip_volume_download(blok_hip_ctx* ctx, float* density, uint32_t* material_ids) {
    int rc = need_volume(ctx);
    if (rc != BLOK_OK) return rc;
    std::string why;
    return volume_status(ctx, blok::gpu_volume_download(&ctx->volume, density, material_ids, &why), why);
}

int blok_hip_volume_set_voxels(blok_hip_ctx* ctx, const int32_t* xyz, const uint32_t* material_ids, const float* density, size_t n) {
    int rc = need_volume(ctx);
    if (rc != BLOK_OK) return rc;
    if (n && !xyz) return set_error(ctx, BLOK_ERR_INVALID_ARG, "null voxel list");
    std::string why;
    return volume_status(ctx, blok::gpu_volume_set_voxels(&ctx->volume, xyz, material_ids, density, n, &why), why);
}

int blok_hip_volume_apply_brush(blok_hip_ctx* ctx, const float center[3], float radius, float value, int mode) {
    int rc = need_volume(ctx);
    if (rc != BLOK_OK) return rc;
    if (!center || (mode != 0 && mode != 1) || !(radius >= 0.0f)) return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad brush arguments");
    std::string why;
    rc = volume_status(ctx, blok::gpu_volume_brush(&ctx->volume, center, radius, value, mode, &why), why);
    if (rc != BLOK_OK) return rc;
    BLOK_HIP_TRY(ctx, hipDeviceSynchronize());
    return BLOK_OK;
}

int blok_hip_volume_rebuild(blok_hip_ctx* ctx, const blok_material* materials, size_t n_materials) {
    int rc = need_volume(ctx);
    if (rc != BLOK_OK) return rc;
    if (n_materials && !materials) return set_error(ctx, BLOK_ERR_INVALID_ARG, "null material table");
    BLOK_HIP_TRY(ctx, hipDeviceSynchronize());            // frames still reading the previous tree
    blok::GpuTree gpu;
    std::string why;
    const blok::GpuBuildStatus st = blok::gpu_volume_build(&ctx->volume, &gpu, &why);
    if (st == blok::GpuBuildStatus::UseHostBuilder) {      // nothing filled: an empty world
        blok::HostTree tree;
        std::vector<blok::VoxelRec> none;
        const char* w = "";
        if (!blok::build_tree(none, tree, &w)) return set_error(ctx, BLOK_ERR_UNSUPPORTED, w);
        return install_tree(ctx, tree, materials, n_materials);
    }
    if (st != blok::GpuBuildStatus::Ok) return volume_status(ctx, st, why);
    free_world(ctx);
    ctx->d_nodes = gpu.d_nodes;
    ctx->d_tree_materials = gpu.d_materials;
    rc = install_materials(ctx, materials, n_materials);
    if (rc != BLOK_OK) { free_world(ctx); return rc; }
    ctx->stats.n_voxels = gpu.n_voxels;
    ctx->stats.n_tree_nodes = gpu.n_nodes;
    ctx->stats.tree_bytes = gpu.n_nodes * sizeof(blok::TreeNode) + gpu.n_voxels * sizeof(uint32_t);
    ctx->stats.levels = gpu.levels;
    for (int a = 0; a < 3; ++a) ctx->stats.origin[a] = gpu.origin[a];
    ctx->has_world = true;
    ctx->built_on_device = true;
    return rebuild_sun_map(ctx);
}

}  // extern "C"
