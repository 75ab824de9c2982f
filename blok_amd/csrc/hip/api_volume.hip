// C ABI, device-resident dense voxel store (include/blok_hip.h: blok_hip_volume_*; kernels in gpu_build.hip).
#include "api_internal.h"
#include <chrono>
#include <cstdlib>

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
    if (ctx->has_volume) { if (ctx->tree_owned_by_volume) { BLOK_HIP_TRY(ctx, hipDeviceSynchronize()); free_world(ctx); } blok::gpu_volume_destroy(&ctx->volume); ctx->has_volume = false; }
    const int32_t o[3] = {origin ? origin[0] : 0, origin ? origin[1] : 0, origin ? origin[2] : 0};
    std::string why;
    const blok::GpuBuildStatus st = blok::gpu_volume_create(o, nx, ny, nz, chunk_size, voxel_size, &ctx->volume, &why, ctx->volume_keyed_layout);
    ctx->has_volume = st == blok::GpuBuildStatus::Ok;
    return volume_status(ctx, st, why);
}

int blok_hip_set_volume_layout(blok_hip_ctx* ctx, int keyed) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->volume_keyed_layout = keyed != 0;
    return BLOK_OK;
}

int blok_hip_volume_destroy(blok_hip_ctx* ctx) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (ctx->has_volume) {
        (void)hipSetDevice(ctx->device);
        if (ctx->tree_owned_by_volume) { (void)hipDeviceSynchronize(); free_world(ctx); }      // the installed world lives in the volume's arrays: it goes with them
        blok::gpu_volume_destroy(&ctx->volume); ctx->has_volume = false;
    }
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
    // enqueued on the null stream, like the rebuild that follows it: nothing waits here (a failing kernel shows at the next call that does)
    return volume_status(ctx, blok::gpu_volume_brush(&ctx->volume, center, radius, value, mode, &why), why);
}

int blok_hip_volume_rebuild(blok_hip_ctx* ctx, const blok_material* materials, size_t n_materials) {
    int rc = need_volume(ctx);
    if (rc != BLOK_OK) return rc;
    if (n_materials && !materials) return set_error(ctx, BLOK_ERR_INVALID_ARG, "null material table");
    const bool timing = std::getenv("BLOK_VOLUME_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    BLOK_HIP_TRY(ctx, hipDeviceSynchronize());            // frames still reading a tree of an earlier build
    const auto t_synced = std::chrono::steady_clock::now();
    blok::GpuVolume& v = ctx->volume;
    // the voxels edited since the last build, in world coordinates (for the shadow rays' last-occluder map)
    int32_t lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = v.origin[a] + static_cast<int32_t>(std::min<uint32_t>(v.edit_lo[a], 0x7FFFFFFFu)); hi[a] = v.origin[a] + static_cast<int32_t>(v.edit_hi[a]); }
    const bool edited = v.edit_lo[0] < v.edit_hi[0] && v.edit_lo[1] < v.edit_hi[1] && v.edit_lo[2] < v.edit_hi[2];
    if (!edited) { lo[0] = lo[1] = lo[2] = 0; hi[0] = hi[1] = hi[2] = 0; }
    blok::GpuTree gpu;
    std::string why;
    const bool may_add = v.edit_may_add;
    const blok::GpuBuildStatus st = blok::gpu_volume_build(&v, &gpu, &why);
    const auto t_built = std::chrono::steady_clock::now();
    for (int a = 0; a < 3; ++a) { v.edit_lo[a] = 0xFFFFFFFFu; v.edit_hi[a] = 0u; }
    v.edit_may_add = false;
    if (st == blok::GpuBuildStatus::UseHostBuilder) {      // nothing filled: an empty world
        blok::HostTree tree;
        std::vector<blok::VoxelRec> none;
        const char* w = "";
        if (!blok::build_tree(none, tree, &w)) return set_error(ctx, BLOK_ERR_UNSUPPORTED, w);
        return install_tree(ctx, tree, materials, n_materials);
    }
    if (st != blok::GpuBuildStatus::Ok) return volume_status(ctx, st, why);
    // the world it replaces: same lattice = the maps laid out over it stay valid where nothing was edited
    const bool same_lattice = ctx->has_world && ctx->built_on_device && ctx->stats.levels == gpu.levels && ctx->stats.origin[0] == gpu.origin[0] &&
                              ctx->stats.origin[1] == gpu.origin[1] && ctx->stats.origin[2] == gpu.origin[2] && ctx->world_voxel_size == 1.0f;
    // the material table is uploaded again only when it differs from the installed one
    const bool same_materials = same_lattice && ctx->d_materials && ctx->n_materials == n_materials && ctx->volume_materials.size() == n_materials * sizeof(blok_material) &&
                                (n_materials == 0 || std::memcmp(ctx->volume_materials.data(), materials, n_materials * sizeof(blok_material)) == 0);
    float* keep_sun = nullptr; bool keep_has_sun = false;
    blok_material* keep_mat = nullptr; size_t keep_n_mat = 0;
    if (same_lattice) { keep_sun = ctx->d_sun_map; keep_has_sun = ctx->has_sun_map; ctx->d_sun_map = nullptr; }      // survive free_world
    if (same_materials) { keep_mat = ctx->d_materials; keep_n_mat = ctx->n_materials; ctx->d_materials = nullptr; }
    free_world(ctx);
    ctx->d_nodes = gpu.d_nodes;
    ctx->d_tree_materials = gpu.d_materials;
    ctx->tree_owned_by_volume = gpu.owned_by_volume;
    if (same_materials) { ctx->d_materials = keep_mat; ctx->n_materials = keep_n_mat; }
    else {
        rc = install_materials(ctx, materials, n_materials);
        if (rc != BLOK_OK) { if (keep_sun) (void)hipFree(keep_sun); free_world(ctx); return rc; }
        ctx->volume_materials.assign(reinterpret_cast<const unsigned char*>(materials), reinterpret_cast<const unsigned char*>(materials) + n_materials * sizeof(blok_material));
    }
    ctx->stats.n_voxels = gpu.n_voxels;
    ctx->stats.n_tree_nodes = gpu.n_nodes;
    ctx->stats.tree_bytes = gpu.n_nodes * sizeof(blok::TreeNode) + gpu.n_voxels * sizeof(uint32_t);
    ctx->stats.levels = gpu.levels;
    for (int a = 0; a < 3; ++a) ctx->stats.origin[a] = gpu.origin[a];
    ctx->has_world = true;
    // (an edited world keeps the view's order: a brush changes few tiles' costs, and the tiles that have become live are walked by their
    // search waves until the next sort — which comes at the base interval again; measured, a brush of radius 10 before every frame: 207 us
    // per frame with the order kept, 270 with it dropped)
    ctx->order.interval_now = ctx->order.interval;
    ctx->built_on_device = true;
    if (same_lattice) { ctx->d_sun_map = keep_sun; ctx->has_sun_map = keep_has_sun; }
    const auto t_installed = std::chrono::steady_clock::now();
    rc = update_sun_map(ctx, lo, hi, same_lattice, may_add);
    if (timing) {
        const auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::fprintf(stderr, "[volume_rebuild us] wait-for-device %.1f build %.1f install %.1f sun-map %.1f\n", us(t_begin, t_synced), us(t_synced, t_built), us(t_built, t_installed), us(t_installed, std::chrono::steady_clock::now()));
    }
    return rc;
}

}  // extern "C"
