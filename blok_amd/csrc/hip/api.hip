// C ABI of the gfx950 backend (include/blok_hip.h).  Owns device memory; every HIP call is checked.
#include <hip/hip_runtime.h>

#include <cmath>
#include <unordered_map>
#include <utility>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "blok_hip.h"
#include "gpu_build.h"
#include <hip/hip_fp16.h>
#include "post_kernels.h"
#include "post_core.h"
#include "reference_world.h"
#include "trace_kernels.h"
#include "path_args.h"
#include "tree.h"

struct blok_hip_ctx {
    int device = 0;
    uint32_t width = 0, height = 0;
    // derived structure in HBM
    uint4* d_nodes = nullptr;
    uint32_t* d_tree_materials = nullptr;
    blok_material* d_materials = nullptr;
    size_t n_materials = 0;
    bool has_world = false;
    bool built_on_device = false;     // structure built by gpu_build.hip (else tree_build.cpp on the host)
    bool force_host_build = false;
    blok_world_stats stats{};
    // scratch frame for the host-output entry points
    blok_hit* d_frame = nullptr;
    size_t frame_capacity = 0;
    // progressive accumulation (CudaTracer::m_dAccum / m_frameIndex / m_prevCam, reference cuda_tracer.hpp:51-55)
    float* d_accum = nullptr;
    float* d_color = nullptr;
    size_t accum_pixels = 0;
    uint32_t accum_frames = 0;
    blok_camera prev_cam{};
    bool has_prev_cam = false;
    // image-space chain (post_core.h): history ping-pong [2] and per-frame planes, all width x height
    struct Post {
        size_t pixels = 0;
        float *hist_color[2] = {nullptr, nullptr}, *moments[2] = {nullptr, nullptr}, *world_pos[2] = {nullptr, nullptr};
        uint16_t* hist_len[2] = {nullptr, nullptr};
        float* unit_normals[2] = {nullptr, nullptr};   // float4: normalize(binary16 normal)
        uint16_t* motion = nullptr;          // half2
        float *variance = nullptr, *ping = nullptr, *pong = nullptr, *taa_hist[2] = {nullptr, nullptr};
        float* widen = nullptr;              // scratch for state downloads
        int cur = 0, taa_cur = 0;
        bool has_motion = false, taa_has_history = false;
        // blok_hip_draw_frame_rt: the frame's own planes and its camera history
        float *rt_planes[4] = {nullptr, nullptr, nullptr, nullptr}, *rt_denoised = nullptr, *rt_resolved = nullptr;
        uint32_t *rt_ldr = nullptr, *rt_final = nullptr;
        uint32_t rt_frame = 0;
        blok_camera rt_prev_cam{};
    } post;
    // device-resident dense store (gpu_build.h: GpuVolume)
    blok::GpuVolume volume;
    bool has_volume = false;
    // "last occluder" map of the shadow rays (beam.h: prism_far), rebuilt with every world
    float* d_sun_map = nullptr;
    bool sun_map_enabled = true, has_sun_map = false;
    blok::SunMapArgs sun{};
    // beam pre-pass (beam.h): start parameters per beam tile, one buffer per stream (launches on one stream are
    // ordered, frames in flight on different streams must not share)
    uint32_t beam_tile = 32;
    std::unordered_map<hipStream_t, std::pair<float*, size_t>> beam_buffers;
    // timing
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool timing = false, timed = false;
    std::string error;
};

namespace {

thread_local std::string g_create_error;

int set_error(blok_hip_ctx* ctx, int status, const std::string& msg) {
    if (ctx) ctx->error = msg; else g_create_error = msg;
    return status;
}

#define BLOK_HIP_TRY(ctx, call)                                                                    \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return set_error(ctx, e_ == hipErrorOutOfMemory ? BLOK_ERR_OOM : BLOK_ERR_HIP,         \
                             std::string(#call) + ": " + hipGetErrorString(e_));                   \
    } while (0)

void free_post(blok_hip_ctx* ctx) {
    auto& P = ctx->post;
    for (int k = 0; k < 2; ++k)
        for (void* p : {static_cast<void*>(P.hist_color[k]), static_cast<void*>(P.moments[k]), static_cast<void*>(P.world_pos[k]),
                        static_cast<void*>(P.hist_len[k]), static_cast<void*>(P.unit_normals[k]), static_cast<void*>(P.taa_hist[k])})
            if (p) (void)hipFree(p);
    for (void* p : {static_cast<void*>(P.motion), static_cast<void*>(P.variance), static_cast<void*>(P.ping), static_cast<void*>(P.pong), static_cast<void*>(P.widen),
                    static_cast<void*>(P.rt_planes[0]), static_cast<void*>(P.rt_planes[1]), static_cast<void*>(P.rt_planes[2]), static_cast<void*>(P.rt_planes[3]),
                    static_cast<void*>(P.rt_denoised), static_cast<void*>(P.rt_resolved), static_cast<void*>(P.rt_ldr), static_cast<void*>(P.rt_final)})
        if (p) (void)hipFree(p);
    P = blok_hip_ctx::Post{};
}

void free_world(blok_hip_ctx* ctx) {
    if (ctx->d_nodes) (void)hipFree(ctx->d_nodes);
    if (ctx->d_tree_materials) (void)hipFree(ctx->d_tree_materials);
    if (ctx->d_materials) (void)hipFree(ctx->d_materials);
    if (ctx->d_sun_map) (void)hipFree(ctx->d_sun_map);
    ctx->d_sun_map = nullptr; ctx->has_sun_map = false;
    ctx->d_nodes = nullptr; ctx->d_tree_materials = nullptr; ctx->d_materials = nullptr;
    ctx->n_materials = 0; ctx->has_world = false; ctx->built_on_device = false; ctx->stats = blok_world_stats{};
}

int install_materials(blok_hip_ctx* ctx, const blok_material* materials, size_t n_materials) {
    if (!n_materials) return BLOK_OK;
    BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_materials), n_materials * sizeof(blok_material)));
    BLOK_HIP_TRY(ctx, hipMemcpy(ctx->d_materials, materials, n_materials * sizeof(blok_material), hipMemcpyHostToDevice));
    ctx->n_materials = n_materials;
    return BLOK_OK;
}

int rebuild_sun_map(blok_hip_ctx* ctx);

int install_tree(blok_hip_ctx* ctx, const blok::HostTree& tree, const blok_material* materials, size_t n_materials) {
    free_world(ctx);
    const size_t node_bytes = tree.nodes.size() * sizeof(blok::TreeNode);
    const size_t mat_bytes = std::max<size_t>(tree.materials.size(), 1) * sizeof(uint32_t);
    BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_nodes), node_bytes));
    BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_tree_materials), mat_bytes));
    BLOK_HIP_TRY(ctx, hipMemcpy(ctx->d_nodes, tree.nodes.data(), node_bytes, hipMemcpyHostToDevice));
    if (!tree.materials.empty())
        BLOK_HIP_TRY(ctx, hipMemcpy(ctx->d_tree_materials, tree.materials.data(),
                                    tree.materials.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    const int rc_mat = install_materials(ctx, materials, n_materials);
    if (rc_mat != BLOK_OK) return rc_mat;
    ctx->stats.n_voxels = tree.n_voxels;
    ctx->stats.n_tree_nodes = tree.nodes.size();
    ctx->stats.tree_bytes = node_bytes + tree.materials.size() * sizeof(uint32_t);
    ctx->stats.levels = tree.levels;
    for (int a = 0; a < 3; ++a) ctx->stats.origin[a] = tree.origin[a];
    ctx->has_world = true;
    return rebuild_sun_map(ctx);
}

// Rebuilds the shadow rays' last-occluder map for the installed world (one wave per texel of the plane perpendicular to the
// sun; texel = 4 voxels, coarser for worlds wider than 2048 voxels so that the map stays <= 512 x 512).
int rebuild_sun_map(blok_hip_ctx* ctx) {
    if (ctx->d_sun_map) { (void)hipDeviceSynchronize(); (void)hipFree(ctx->d_sun_map); ctx->d_sun_map = nullptr; }
    ctx->has_sun_map = false;
    if (!ctx->has_world || ctx->stats.levels == 0 || ctx->stats.n_voxels == 0) return BLOK_OK;
    blok::SunMapArgs& m = ctx->sun;
    m = blok::SunMapArgs{};
    m.trace.nodes = ctx->d_nodes; m.trace.materials = ctx->d_tree_materials;
    for (int a = 0; a < 3; ++a) m.trace.origin[a] = ctx->stats.origin[a];
    m.trace.levels = ctx->stats.levels;
    // the shader's sun: normalize(vec3(0.5, 0.8, 0.3)) (raygen.rgen:142,185) with path_core.h's operation order
    const float sx = 0.5f, sy = 0.8f, sz = 0.3f;
    const float len = std::sqrt(sx * sx + sy * sy + sz * sz);
    m.s[0] = sx / len; m.s[1] = sy / len; m.s[2] = sz / len;
    // u = normalize(s x Y), v = s x u: any orthonormal complement serves
    const double s[3] = {m.s[0], m.s[1], m.s[2]};
    double u[3] = {-s[2], 0.0, s[0]};
    const double ul = std::sqrt(u[0] * u[0] + u[2] * u[2]);
    for (double& c : u) c /= ul;
    const double v[3] = {s[1] * u[2] - s[2] * u[1], s[2] * u[0] - s[0] * u[2], s[0] * u[1] - s[1] * u[0]};
    for (int a = 0; a < 3; ++a) { m.u[a] = static_cast<float>(u[a]); m.v[a] = static_cast<float>(v[a]); }
    const double W = std::ldexp(1.0, 2 * static_cast<int>(ctx->stats.levels));          // the tree's cube edge
    double lo[2] = {1e30, 1e30}, hi[2] = {-1e30, -1e30};
    for (int c = 0; c < 8; ++c) {
        const double p[3] = {ctx->stats.origin[0] + ((c & 1) ? W : 0.0), ctx->stats.origin[1] + ((c & 2) ? W : 0.0), ctx->stats.origin[2] + ((c & 4) ? W : 0.0)};
        const double pu = m.u[0] * p[0] + m.u[1] * p[1] + m.u[2] * p[2], pv = m.v[0] * p[0] + m.v[1] * p[1] + m.v[2] * p[2];
        lo[0] = std::min(lo[0], pu); hi[0] = std::max(hi[0], pu); lo[1] = std::min(lo[1], pv); hi[1] = std::max(hi[1], pv);
    }
    const double extent = std::max(hi[0] - lo[0], hi[1] - lo[1]);
    m.texel = static_cast<float>(std::max(4.0, std::ceil(extent / 512.0)));
    m.u0 = static_cast<float>(std::floor(lo[0]) - m.texel); m.v0 = static_cast<float>(std::floor(lo[1]) - m.texel);
    m.nu = static_cast<uint32_t>(std::ceil((hi[0] - m.u0) / m.texel)) + 1u; m.nv = static_cast<uint32_t>(std::ceil((hi[1] - m.v0) / m.texel)) + 1u;
    BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_sun_map), static_cast<size_t>(m.nu) * m.nv * sizeof(float)));
    m.map = ctx->d_sun_map;
    blok::launch_sun_map(m, nullptr);
    BLOK_HIP_TRY(ctx, hipGetLastError());
    BLOK_HIP_TRY(ctx, hipDeviceSynchronize());
    ctx->has_sun_map = true;
    return BLOK_OK;
}

int ensure_frame(blok_hip_ctx* ctx, size_t records) {
    if (records <= ctx->frame_capacity) return BLOK_OK;
    if (ctx->d_frame) (void)hipFree(ctx->d_frame);
    ctx->d_frame = nullptr; ctx->frame_capacity = 0;
    BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_frame), records * sizeof(blok_hit)));
    ctx->frame_capacity = records;
    return BLOK_OK;
}

blok::TraceArgs base_args(const blok_hip_ctx* ctx, const blok_camera* cam) {
    blok::TraceArgs a{};
    a.nodes = ctx->d_nodes;
    a.materials = ctx->d_tree_materials;
    for (int i = 0; i < 3; ++i) a.origin[i] = ctx->stats.origin[i];
    a.levels = ctx->stats.levels;
    if (cam) a.cam = *cam;
    a.frame_w = ctx->width; a.frame_h = ctx->height;
    a.mat_table = ctx->d_materials;
    a.n_materials = static_cast<uint32_t>(ctx->n_materials);
    a.tmin = BLOK_RAY_TMIN; a.tmax = BLOK_RAY_TMAX;
    return a;
}

// The stream's beam buffer, grown to n floats.
int beam_buffer(blok_hip_ctx* ctx, hipStream_t stream, size_t n, float** out) {
    auto& slot = ctx->beam_buffers[stream];
    if (slot.second < n) {
        if (slot.first) { BLOK_HIP_TRY(ctx, hipStreamSynchronize(stream)); (void)hipFree(slot.first); }
        slot = {nullptr, 0};
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&slot.first), n * sizeof(float)));
        slot.second = n;
    }
    *out = slot.first;
    return BLOK_OK;
}

// Fills the beam fields of a Rect / Tiles launch and returns the number of beam tiles (0 = pre-pass off).
int prepare_beam(blok_hip_ctx* ctx, blok::RayMode mode, blok::TraceArgs& args, hipStream_t stream, uint32_t tiles_of_rank, uint32_t* n_beams) {
    *n_beams = 0;
    if (mode == blok::RayMode::Rays || !ctx->beam_tile) return BLOK_OK;
    args.beam_tile = ctx->beam_tile;
    if (mode == blok::RayMode::Tiles && args.tile % args.beam_tile) args.beam_tile = 16;   // tiles are multiples of 16
    if (args.beam_tile % blok::kWaveW || args.beam_tile % blok::kWaveH) { args.beam_tile = 0; return BLOK_OK; }   // a wave must not straddle beam tiles (non-default footprints)
    args.beam_bx = (args.w + args.beam_tile - 1u) / args.beam_tile;
    *n_beams = blok::beam_tiles(mode, args, tiles_of_rank);
    return beam_buffer(ctx, stream, *n_beams, &args.beam);
}

// Rect / Tiles launches run the beam pre-pass first, on the same stream (tiles_of_rank: Tiles only).
int launch_timed(blok_hip_ctx* ctx, blok::RayMode mode, blok::TraceArgs args, uint32_t blocks, hipStream_t stream,
                 uint32_t tiles_of_rank = 0) {
    uint32_t n_beams = 0;
    if (blocks) { const int rc = prepare_beam(ctx, mode, args, stream, tiles_of_rank, &n_beams); if (rc != BLOK_OK) return rc; }
    if (ctx->timing) BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, stream));
    if (n_beams) blok::launch_beam(mode, args, n_beams, stream);
    blok::launch_trace(mode, args, blocks, stream);
    BLOK_HIP_TRY(ctx, hipGetLastError());
    if (ctx->timing) { BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_end, stream)); ctx->timed = true; }
    return BLOK_OK;
}

int check_trace(blok_hip_ctx* ctx, const blok_camera* cam) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!cam) return set_error(ctx, BLOK_ERR_INVALID_ARG, "camera is null");
    {   // a camera that is not a finite, non-degenerate basis would make every ray NaN (and the beam pre-pass cull nothing)
        const float* f = reinterpret_cast<const float*>(cam);
        for (size_t i = 0; i < sizeof(blok_camera) / sizeof(float); ++i)
            if (!std::isfinite(f[i])) return set_error(ctx, BLOK_ERR_INVALID_ARG, "camera has a non-finite component");
        if (!(cam->tan_half_fov > 0.0f) || !(cam->aspect > 0.0f)) return set_error(ctx, BLOK_ERR_INVALID_ARG, "camera needs tan_half_fov > 0 and aspect > 0");
    }
    if (!ctx->has_world) return set_error(ctx, BLOK_ERR_NO_WORLD, "no world uploaded");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return BLOK_OK;
}

}  // namespace

extern "C" {

uint32_t blok_hip_abi_version(void) { return (1u << 16) | 0u; }

const char* blok_hip_last_error(const blok_hip_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int blok_hip_create(blok_hip_ctx** out_ctx, int device_ordinal, uint32_t width, uint32_t height) {
    if (!out_ctx) return BLOK_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    if (!width || !height) return set_error(nullptr, BLOK_ERR_INVALID_ARG, "zero-sized frame");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return set_error(nullptr, BLOK_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= count)
        return set_error(nullptr, BLOK_ERR_INVALID_ARG, "device ordinal out of range");
    hipDeviceProp_t prop{};
    e = hipGetDeviceProperties(&prop, device_ordinal);
    if (e != hipSuccess) return set_error(nullptr, BLOK_ERR_NO_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(nullptr, BLOK_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this backend is built for gfx950 only");
    e = hipSetDevice(device_ordinal);
    if (e != hipSuccess) return set_error(nullptr, BLOK_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    auto* ctx = new (std::nothrow) blok_hip_ctx();
    if (!ctx) return set_error(nullptr, BLOK_ERR_OOM, "host allocation failed");
    ctx->device = device_ordinal; ctx->width = width; ctx->height = height;
    if (hipEventCreate(&ctx->ev_begin) != hipSuccess || hipEventCreate(&ctx->ev_end) != hipSuccess) {
        delete ctx;
        return set_error(nullptr, BLOK_ERR_HIP, "hipEventCreate failed");
    }
    *out_ctx = ctx;
    return BLOK_OK;
}

int blok_hip_resize(blok_hip_ctx* ctx, uint32_t width, uint32_t height) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!width || !height) return set_error(ctx, BLOK_ERR_INVALID_ARG, "zero-sized frame");
    ctx->width = width; ctx->height = height;     // the accumulation buffer is re-created on the next progressive frame
    return BLOK_OK;
}

void blok_hip_destroy(blok_hip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    free_world(ctx);
    if (ctx->d_frame) (void)hipFree(ctx->d_frame);
    free_post(ctx);
    if (ctx->has_volume) blok::gpu_volume_destroy(&ctx->volume);
    for (auto& kv : ctx->beam_buffers) if (kv.second.first) (void)hipFree(kv.second.first);
    if (ctx->d_accum) (void)hipFree(ctx->d_accum);
    if (ctx->d_color) (void)hipFree(ctx->d_color);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    delete ctx;
}

int blok_hip_upload_world(blok_hip_ctx* ctx, const blok_svo_node* nodes, size_t n_nodes,
                          const blok_sub_chunk* sub_chunks, size_t n_sub_chunks,
                          const blok_material* materials, size_t n_materials) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if ((n_nodes && !nodes) || (n_sub_chunks && !sub_chunks) || (n_materials && !materials))
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "null array with non-zero count");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // device-side build (gpu_build.hip); worlds it does not cover take the general host path below
    if (!ctx->force_host_build) {
        blok::GpuTree gpu;
        std::string reason;
        const blok::GpuBuildStatus st = blok::gpu_build_tree(nodes, n_nodes, sub_chunks, n_sub_chunks, &gpu, &reason);
        if (st == blok::GpuBuildStatus::Unsupported) return set_error(ctx, BLOK_ERR_UNSUPPORTED, reason);
        if (st == blok::GpuBuildStatus::HipError) return set_error(ctx, BLOK_ERR_HIP, reason);
        if (st == blok::GpuBuildStatus::OutOfMemory) return set_error(ctx, BLOK_ERR_OOM, reason);
        if (st == blok::GpuBuildStatus::Ok) {
            free_world(ctx);
            ctx->d_nodes = gpu.d_nodes;
            ctx->d_tree_materials = gpu.d_materials;
            const int rc = install_materials(ctx, materials, n_materials);
            if (rc != BLOK_OK) { free_world(ctx); return rc; }
            ctx->stats.n_voxels = gpu.n_voxels;
            ctx->stats.n_tree_nodes = gpu.n_nodes;
            ctx->stats.tree_bytes = gpu.n_nodes * sizeof(blok::TreeNode) + gpu.n_voxels * sizeof(uint32_t);
            ctx->stats.levels = gpu.levels;
            for (int a = 0; a < 3; ++a) ctx->stats.origin[a] = gpu.origin[a];
            ctx->stats.n_ref_nodes = n_nodes;
            ctx->stats.n_sub_chunks = n_sub_chunks;
            ctx->has_world = true;
            ctx->built_on_device = true;
            return rebuild_sun_map(ctx);
        }
    }
    std::vector<blok::VoxelRec> voxels;
    const char* why = "";
    if (!blok::extract_voxels(nodes, n_nodes, sub_chunks, n_sub_chunks, voxels, &why))
        return set_error(ctx, BLOK_ERR_UNSUPPORTED, why);
    blok::HostTree tree;
    if (!blok::build_tree(voxels, tree, &why)) return set_error(ctx, BLOK_ERR_UNSUPPORTED, why);
    const int rc = install_tree(ctx, tree, materials, n_materials);
    if (rc != BLOK_OK) return rc;
    ctx->stats.n_ref_nodes = n_nodes;
    ctx->stats.n_sub_chunks = n_sub_chunks;
    return BLOK_OK;
}

int blok_hip_set_host_build(blok_hip_ctx* ctx, int enabled) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->force_host_build = enabled != 0;
    return BLOK_OK;
}

int blok_hip_world_built_on_device(const blok_hip_ctx* ctx) { return ctx && ctx->has_world && ctx->built_on_device ? 1 : 0; }

int blok_hip_download_tree(const blok_hip_ctx* ctx, void* nodes_out, size_t node_capacity, uint32_t* materials_out,
                           size_t material_capacity) {
    if (!ctx || !ctx->has_world) return BLOK_ERR_NO_WORLD;
    if ((nodes_out && node_capacity < ctx->stats.n_tree_nodes) || (materials_out && material_capacity < ctx->stats.n_voxels))
        return BLOK_ERR_INVALID_ARG;
    if (nodes_out && hipMemcpy(nodes_out, ctx->d_nodes, ctx->stats.n_tree_nodes * sizeof(blok::TreeNode), hipMemcpyDeviceToHost) != hipSuccess)
        return BLOK_ERR_HIP;
    if (materials_out && ctx->stats.n_voxels &&
        hipMemcpy(materials_out, ctx->d_tree_materials, ctx->stats.n_voxels * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        return BLOK_ERR_HIP;
    return BLOK_OK;
}

int blok_hip_upload_dense(blok_hip_ctx* ctx, const uint32_t* ids, uint32_t nx, uint32_t ny, uint32_t nz,
                          const int32_t origin[3], const blok_material* materials, size_t n_materials) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!ids || !nx || !ny || !nz || (n_materials && !materials))
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad dense grid arguments");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int32_t o[3] = {origin ? origin[0] : 0, origin ? origin[1] : 0, origin ? origin[2] : 0};
    if (!ctx->force_host_build) {                      // device-side build straight from the grid (gpu_build.hip)
        blok::GpuTree gpu;
        std::string reason;
        const blok::GpuBuildStatus st = blok::gpu_build_tree_dense(ids, nx, ny, nz, o, &gpu, &reason);
        if (st == blok::GpuBuildStatus::Unsupported) return set_error(ctx, BLOK_ERR_UNSUPPORTED, reason);
        if (st == blok::GpuBuildStatus::HipError) return set_error(ctx, BLOK_ERR_HIP, reason);
        if (st == blok::GpuBuildStatus::OutOfMemory) return set_error(ctx, BLOK_ERR_OOM, reason);
        if (st == blok::GpuBuildStatus::Ok) {
            free_world(ctx);
            ctx->d_nodes = gpu.d_nodes;
            ctx->d_tree_materials = gpu.d_materials;
            const int rc = install_materials(ctx, materials, n_materials);
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
    }
    std::vector<blok::VoxelRec> voxels;
    for (uint32_t z = 0; z < nz; ++z)
        for (uint32_t y = 0; y < ny; ++y) {
            const uint32_t* row = ids + (static_cast<size_t>(z) * ny + y) * nx;
            for (uint32_t x = 0; x < nx; ++x)
                if (row[x]) voxels.push_back(blok::VoxelRec{o[0] + int32_t(x), o[1] + int32_t(y), o[2] + int32_t(z), row[x]});
        }
    blok::HostTree tree;
    const char* why = "";
    if (!blok::build_tree(voxels, tree, &why)) return set_error(ctx, BLOK_ERR_UNSUPPORTED, why);
    return install_tree(ctx, tree, materials, n_materials);
}

int blok_hip_world_stats(const blok_hip_ctx* ctx, blok_world_stats* out) {
    if (!ctx || !out) return BLOK_ERR_INVALID_ARG;
    if (!ctx->has_world) return BLOK_ERR_NO_WORLD;
    *out = ctx->stats;
    return BLOK_OK;
}

int blok_hip_trace_primary_device(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0,
                                  uint32_t w, uint32_t h, void* out_hits_dev, void* out_rgba_dev, void* hip_stream) {
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    if ((!out_hits_dev && !out_rgba_dev) || !w || !h || x0 + w > ctx->width || y0 + h > ctx->height)
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "rectangle outside the frame or no output");
    blok::TraceArgs a = base_args(ctx, cam);
    a.x0 = x0; a.y0 = y0; a.w = w; a.h = h;
    a.out = static_cast<blok_hit*>(out_hits_dev);
    a.out_rgba = static_cast<uint32_t*>(out_rgba_dev);
    const uint32_t blocks = blok::rect_grid_blocks(w, h);
    return launch_timed(ctx, blok::RayMode::Rect, a, blocks, static_cast<hipStream_t>(hip_stream));
}

int blok_hip_trace_primary(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0,
                           uint32_t w, uint32_t h, blok_hit* out_hits_host) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!out_hits_host) return set_error(ctx, BLOK_ERR_INVALID_ARG, "null output");
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    const size_t n = static_cast<size_t>(w) * h;
    rc = ensure_frame(ctx, n);
    if (rc != BLOK_OK) return rc;
    rc = blok_hip_trace_primary_device(ctx, cam, x0, y0, w, h, ctx->d_frame, nullptr, nullptr);
    if (rc != BLOK_OK) return rc;
    BLOK_HIP_TRY(ctx, hipMemcpy(out_hits_host, ctx->d_frame, n * sizeof(blok_hit), hipMemcpyDeviceToHost));
    return BLOK_OK;
}

uint32_t blok_hip_tiles_for_rank(uint32_t width, uint32_t height, uint32_t tile, uint32_t rank, uint32_t n_ranks) {
    if (!tile || !n_ranks || rank >= n_ranks) return 0;
    const uint32_t total = ((width + tile - 1) / tile) * ((height + tile - 1) / tile);
    return total > rank ? (total - rank + n_ranks - 1) / n_ranks : 0;
}

int blok_hip_trace_tiles_device(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t tile, uint32_t rank,
                                uint32_t n_ranks, void* out_hits_dev, void* out_rgba_dev, void* hip_stream) {
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    if ((!out_hits_dev && !out_rgba_dev) || tile < 16 || (tile % blok::kTileW) || (tile % blok::kTileH) || (tile & 15u) || !n_ranks || rank >= n_ranks)
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "tile must be a multiple of 16 and rank < n_ranks");
    blok::TraceArgs a = base_args(ctx, cam);
    a.tile = tile; a.rank = rank; a.n_ranks = n_ranks;
    a.tiles_x = (ctx->width + tile - 1) / tile;
    a.tiles_total = a.tiles_x * ((ctx->height + tile - 1) / tile);
    a.out = static_cast<blok_hit*>(out_hits_dev);
    a.out_rgba = static_cast<uint32_t*>(out_rgba_dev);
    const uint32_t mine = blok_hip_tiles_for_rank(ctx->width, ctx->height, tile, rank, n_ranks);
    const uint32_t blocks = mine * (tile / blok::kTileW) * (tile / blok::kTileH);
    return launch_timed(ctx, blok::RayMode::Tiles, a, blocks, static_cast<hipStream_t>(hip_stream), mine);
}

int blok_hip_untile_device(blok_hip_ctx* ctx, const void* gathered_dev, uint32_t elem_bytes, uint32_t tile,
                           uint32_t n_ranks, uint32_t tiles_per_rank_max, void* out_frame_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!gathered_dev || !out_frame_dev || !tile || !n_ranks || !tiles_per_rank_max || (elem_bytes != 16 && elem_bytes != 4))
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad untile arguments (elem_bytes must be 16 or 4)");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    blok::UntileArgs u{};
    u.gathered = gathered_dev;
    u.frame = out_frame_dev;
    u.elem_bytes = elem_bytes;
    u.frame_w = ctx->width; u.frame_h = ctx->height; u.tile = tile; u.n_ranks = n_ranks;
    u.tiles_per_rank_max = tiles_per_rank_max;
    u.tiles_x = (ctx->width + tile - 1) / tile;
    blok::launch_untile(u, static_cast<hipStream_t>(hip_stream));
    BLOK_HIP_TRY(ctx, hipGetLastError());
    return BLOK_OK;
}

int blok_hip_trace_rays(blok_hip_ctx* ctx, const blok_ray* rays_host, size_t n, blok_hit* out_hits_host) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!ctx->has_world) return set_error(ctx, BLOK_ERR_NO_WORLD, "no world uploaded");
    if (n == 0) return BLOK_OK;
    if (!rays_host || !out_hits_host || n > 0x7FFFFFFFu) return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad ray arguments");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_frame(ctx, n);
    if (rc != BLOK_OK) return rc;
    blok_ray* d_rays = nullptr;
    BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_rays), n * sizeof(blok_ray)));
    hipError_t e = hipMemcpy(d_rays, rays_host, n * sizeof(blok_ray), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        blok::TraceArgs a = base_args(ctx, nullptr);
        a.rays = d_rays; a.n_rays = static_cast<uint32_t>(n); a.out = ctx->d_frame;
        rc = launch_timed(ctx, blok::RayMode::Rays, a, static_cast<uint32_t>((n + blok::kBlock - 1) / blok::kBlock), nullptr);
        if (rc == BLOK_OK) e = hipMemcpy(out_hits_host, ctx->d_frame, n * sizeof(blok_hit), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_rays);
    if (rc != BLOK_OK) return rc;
    if (e != hipSuccess) return set_error(ctx, BLOK_ERR_HIP, std::string("trace_rays copy: ") + hipGetErrorString(e));
    return BLOK_OK;
}

int blok_hip_shade_rgba8(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0, uint32_t w,
                         uint32_t h, uint32_t* out_rgba8_host) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!out_rgba8_host) return set_error(ctx, BLOK_ERR_INVALID_ARG, "null output");
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    const size_t n = static_cast<size_t>(w) * h;
    rc = ensure_frame(ctx, (n + 3) / 4);                       // n RGBA8 pixels fit in n/4 16-byte records
    if (rc != BLOK_OK) return rc;
    rc = blok_hip_trace_primary_device(ctx, cam, x0, y0, w, h, nullptr, ctx->d_frame, nullptr);
    if (rc != BLOK_OK) return rc;
    BLOK_HIP_TRY(ctx, hipMemcpy(out_rgba8_host, ctx->d_frame, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return BLOK_OK;
}

int blok_hip_trace_paths_device(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0, uint32_t w,
                                uint32_t h, uint32_t spp, uint32_t max_bounces, uint32_t frame_index,
                                const blok_gbuffer* planes, void* hip_stream) {
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    if (!planes || !w || !h || x0 + w > ctx->width || y0 + h > ctx->height || !spp || !max_bounces)
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad path-trace arguments");
    if (!ctx->n_materials) return set_error(ctx, BLOK_ERR_INVALID_ARG, "path tracing needs a material table");
    blok::PathArgs p{};
    p.trace = base_args(ctx, cam);
    p.trace.x0 = x0; p.trace.y0 = y0; p.trace.w = w; p.trace.h = h;
    p.spp = spp; p.max_bounces = max_bounces; p.frame_count = frame_index;
    p.color = planes->color; p.world_pos = planes->world_pos;
    p.normal_roughness = planes->normal_roughness; p.albedo_metallic = planes->albedo_metallic;
    if (ctx->sun_map_enabled && ctx->has_sun_map) {         // shadow rays stop at the last occluder of their column
        const blok::SunMapArgs& m = ctx->sun;
        p.sun_map = ctx->d_sun_map;
        for (int a = 0; a < 3; ++a) { p.sun_u[a] = m.u[a]; p.sun_v[a] = m.v[a]; }
        p.sun_u0 = m.u0; p.sun_v0 = m.v0; p.sun_inv_texel = 1.0f / m.texel; p.sun_nu = m.nu; p.sun_nv = m.nv;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    uint32_t n_beams = 0;                                   // the primary rays of every sample start behind the beam pre-pass
    rc = prepare_beam(ctx, blok::RayMode::Rect, p.trace, stream, 0, &n_beams);
    if (rc != BLOK_OK) return rc;
    if (ctx->timing) BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, stream));
    if (n_beams) blok::launch_beam(blok::RayMode::Rect, p.trace, n_beams, stream);
    blok::launch_paths(p, blok::rect_grid_blocks(w, h), stream);
    BLOK_HIP_TRY(ctx, hipGetLastError());
    if (ctx->timing) { BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_end, stream)); ctx->timed = true; }
    return BLOK_OK;
}

int blok_hip_trace_paths(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                         uint32_t spp, uint32_t max_bounces, uint32_t frame_index, const blok_gbuffer* planes_host) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!planes_host) return set_error(ctx, BLOK_ERR_INVALID_ARG, "null planes");
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    const size_t n = static_cast<size_t>(w) * h, bytes = n * 4 * sizeof(float);
    float* host[4] = {planes_host->color, planes_host->world_pos, planes_host->normal_roughness, planes_host->albedo_metallic};
    float* dev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
        if (host[i]) e = hipMalloc(reinterpret_cast<void**>(&dev[i]), bytes);
    if (e == hipSuccess) {
        const blok_gbuffer planes_dev{dev[0], dev[1], dev[2], dev[3]};
        rc = blok_hip_trace_paths_device(ctx, cam, x0, y0, w, h, spp, max_bounces, frame_index, &planes_dev, nullptr);
        for (int i = 0; i < 4 && rc == BLOK_OK && e == hipSuccess; ++i)
            if (host[i]) e = hipMemcpy(host[i], dev[i], bytes, hipMemcpyDeviceToHost);
    }
    for (float* d : dev) if (d) (void)hipFree(d);
    if (rc != BLOK_OK) return rc;
    if (e != hipSuccess) return set_error(ctx, e == hipErrorOutOfMemory ? BLOK_ERR_OOM : BLOK_ERR_HIP,
                                          std::string("trace_paths: ") + hipGetErrorString(e));
    return BLOK_OK;
}

int blok_hip_tonemap_device(blok_hip_ctx* ctx, const float* hdr_dev, uint32_t n_pixels, float exposure,
                            float saturation_boost, int tonemap_operator, void* out_rgba8_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!hdr_dev || !out_rgba8_dev) return set_error(ctx, BLOK_ERR_INVALID_ARG, "null tonemap buffer");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    blok::TonemapArgs t{hdr_dev, static_cast<uint32_t*>(out_rgba8_dev), n_pixels, exposure, saturation_boost, tonemap_operator};
    blok::launch_tonemap(t, static_cast<hipStream_t>(hip_stream));
    BLOK_HIP_TRY(ctx, hipGetLastError());
    return BLOK_OK;
}

int blok_hip_tonemap(blok_hip_ctx* ctx, const float* hdr_host, uint32_t n_pixels, float exposure, float saturation_boost,
                     int tonemap_operator, uint32_t* out_rgba8_host) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!hdr_host || !out_rgba8_host) return set_error(ctx, BLOK_ERR_INVALID_ARG, "null tonemap buffer");
    if (!n_pixels) return BLOK_OK;
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    float* d_in = nullptr; uint32_t* d_out = nullptr;
    BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_in), static_cast<size_t>(n_pixels) * 16));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_out), static_cast<size_t>(n_pixels) * 4);
    int rc = BLOK_OK;
    if (e == hipSuccess) e = hipMemcpy(d_in, hdr_host, static_cast<size_t>(n_pixels) * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) rc = blok_hip_tonemap_device(ctx, d_in, n_pixels, exposure, saturation_boost, tonemap_operator, d_out, nullptr);
    if (e == hipSuccess && rc == BLOK_OK) e = hipMemcpy(out_rgba8_host, d_out, static_cast<size_t>(n_pixels) * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_in); if (d_out) (void)hipFree(d_out);
    if (rc != BLOK_OK) return rc;
    if (e != hipSuccess) return set_error(ctx, BLOK_ERR_HIP, std::string("tonemap: ") + hipGetErrorString(e));
    return BLOK_OK;
}

int blok_hip_reset_accum(blok_hip_ctx* ctx) {                      // reference cuda_tracer.cu:450-454
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->accum_frames = 0;
    if (!ctx->d_accum) return BLOK_OK;
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    BLOK_HIP_TRY(ctx, hipMemset(ctx->d_accum, 0, ctx->accum_pixels * 4 * sizeof(float)));
    return BLOK_OK;
}

namespace {
// reference cuda_tracer.cu:456-472: any basis component, position or fov moving by more than 1e-5
bool camera_changed(const blok_hip_ctx* ctx, const blok_camera& c) {
    if (!ctx->has_prev_cam) return true;
    const float* a = reinterpret_cast<const float*>(&c);
    const float* b = reinterpret_cast<const float*>(&ctx->prev_cam);
    for (size_t i = 0; i < sizeof(blok_camera) / sizeof(float); ++i)
        if (std::fabs(a[i] - b[i]) > 1e-5f) return true;
    return false;
}
}  // namespace

int blok_hip_draw_frame_accumulate(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t spp_per_frame, uint32_t max_bounces,
                                   uint32_t* out_rgba8_host, uint32_t* out_frames_accumulated) {
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    if (!spp_per_frame || !max_bounces) return set_error(ctx, BLOK_ERR_INVALID_ARG, "spp and bounces must be positive");
    const size_t n = static_cast<size_t>(ctx->width) * ctx->height;
    if (ctx->accum_pixels != n) {                                      // first use or resize: (re)allocate and clear
        if (ctx->d_accum) (void)hipFree(ctx->d_accum);
        if (ctx->d_color) (void)hipFree(ctx->d_color);
        ctx->d_accum = ctx->d_color = nullptr; ctx->accum_pixels = 0;
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_accum), n * 4 * sizeof(float)));
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_color), n * 4 * sizeof(float)));
        ctx->accum_pixels = n;
        ctx->has_prev_cam = false;
    }
    if (camera_changed(ctx, *cam)) { rc = blok_hip_reset_accum(ctx); if (rc != BLOK_OK) return rc; }      // cuda_tracer.cu:485
    ctx->prev_cam = *cam; ctx->has_prev_cam = true;                                                         // :486
    const blok_gbuffer planes{ctx->d_color, nullptr, nullptr, nullptr};
    rc = blok_hip_trace_paths_device(ctx, cam, 0, 0, ctx->width, ctx->height, spp_per_frame, max_bounces, ctx->accum_frames,
                                     &planes, nullptr);
    if (rc != BLOK_OK) return rc;
    rc = ensure_frame(ctx, (n + 3) / 4);
    if (rc != BLOK_OK) return rc;
    blok::AccumArgs a{ctx->d_color, ctx->d_accum, out_rgba8_host ? reinterpret_cast<uint32_t*>(ctx->d_frame) : nullptr,
                      static_cast<uint32_t>(n)};
    blok::launch_accumulate(a, nullptr);
    BLOK_HIP_TRY(ctx, hipGetLastError());
    ctx->accum_frames += 1;                                                                                  // ++m_frameIndex
    if (out_frames_accumulated) *out_frames_accumulated = ctx->accum_frames;
    if (out_rgba8_host) BLOK_HIP_TRY(ctx, hipMemcpy(out_rgba8_host, ctx->d_frame, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    else BLOK_HIP_TRY(ctx, hipDeviceSynchronize());
    return BLOK_OK;
}

int blok_hip_accum_download(blok_hip_ctx* ctx, float* out_rgba32f_host) {
    if (!ctx || !out_rgba32f_host) return BLOK_ERR_INVALID_ARG;
    if (!ctx->d_accum) return set_error(ctx, BLOK_ERR_INVALID_ARG, "no accumulation buffer yet");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    BLOK_HIP_TRY(ctx, hipMemcpy(out_rgba32f_host, ctx->d_accum, ctx->accum_pixels * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return BLOK_OK;
}

// ---- image-space chain ------------------------------------------------------------------------------------------------
namespace {
int post_alloc_bytes(blok_hip_ctx* ctx, void** p, size_t bytes) {
    BLOK_HIP_TRY(ctx, hipMalloc(p, bytes));
    BLOK_HIP_TRY(ctx, hipMemset(*p, 0, bytes));
    return BLOK_OK;
}
#define post_alloc(ctx, pp, count) post_alloc_bytes((ctx), reinterpret_cast<void**>(pp), (count) * sizeof(**(pp)))
int ensure_post(blok_hip_ctx* ctx) {
    const size_t n = static_cast<size_t>(ctx->width) * ctx->height;
    auto& P = ctx->post;
    if (P.pixels == n) return BLOK_OK;
    free_post(ctx);
    int rc = BLOK_OK;
    for (int k = 0; k < 2 && rc == BLOK_OK; ++k) {
        rc = post_alloc(ctx, &P.hist_color[k], 4 * n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.moments[k], 2 * n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.world_pos[k], 4 * n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.hist_len[k], n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.unit_normals[k], 4 * n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.taa_hist[k], 4 * n);
    }
    if (rc == BLOK_OK) rc = post_alloc(ctx, &P.motion, 2 * n);
    if (rc == BLOK_OK) rc = post_alloc(ctx, &P.variance, n);
    if (rc == BLOK_OK) rc = post_alloc(ctx, &P.ping, 4 * n);
    if (rc == BLOK_OK) rc = post_alloc(ctx, &P.pong, 4 * n);
    if (rc == BLOK_OK) rc = post_alloc(ctx, &P.widen, 2 * n);
    if (rc != BLOK_OK) { free_post(ctx); return rc; }
    P.pixels = n;
    return BLOK_OK;
}
blok::DenoiseSettings to_settings(const blok_denoise_settings& s) {
    blok::DenoiseSettings d;
    d.temporal_alpha = s.temporal_alpha; d.moment_alpha = s.moment_alpha; d.variance_clip_gamma = s.variance_clip_gamma;
    d.depth_threshold = s.depth_threshold; d.normal_threshold = s.normal_threshold;
    d.phi_color = s.phi_color; d.phi_normal = s.phi_normal; d.phi_depth = s.phi_depth;
    d.atrous_iterations = s.atrous_iterations; d.variance_boost = s.variance_boost; d.min_history_length = s.min_history_length;
    return d;
}
}  // namespace

void blok_denoise_settings_default(blok_denoise_settings* s) {       // renderer_denoising.hpp:49-66
    if (!s) return;
    s->temporal_alpha = 0.05f; s->moment_alpha = 0.2f; s->variance_clip_gamma = 1.5f;
    s->depth_threshold = 0.1f; s->normal_threshold = 0.95f;
    s->phi_color = 0.5f; s->phi_normal = 128.0f; s->phi_depth = 0.1f;
    s->atrous_iterations = 4; s->variance_boost = 1.5f; s->min_history_length = 4;
}

int blok_hip_denoise_device(blok_hip_ctx* ctx, const blok_gbuffer* planes, const float* motion_dev, const float prev_view_proj[16],
                            uint32_t frame_count, const blok_denoise_settings* settings, float* out_color_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!planes || !planes->color || !planes->world_pos || !planes->normal_roughness || !prev_view_proj || !out_color_dev)
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "denoise: colour, world position and normal planes, prevViewProj and an output are required");
    blok_denoise_settings def;
    blok_denoise_settings_default(&def);
    const blok_denoise_settings& S = settings ? *settings : def;
    if (S.atrous_iterations < 0 || S.atrous_iterations > 5) return set_error(ctx, BLOK_ERR_INVALID_ARG, "denoise: 0..5 a-trous iterations");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_post(ctx);
    if (rc != BLOK_OK) return rc;
    auto& P = ctx->post;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const int cur = P.cur, prev = cur ^ 1;
    blok::PostFrame f{};
    f.w = ctx->width; f.h = ctx->height; f.frame_count = frame_count; f.s = to_settings(S);
    for (int k = 0; k < 16; ++k) f.prev_view_proj[k] = prev_view_proj[k];

    blok::TemporalArgs t{};
    t.f = f;
    t.color = planes->color; t.world_pos = planes->world_pos; t.normal_roughness = planes->normal_roughness; t.motion_in = motion_dev;
    t.prev_color = P.hist_color[prev]; t.prev_moments = P.moments[prev]; t.prev_world_pos = P.world_pos[prev];
    t.prev_hist_len = P.hist_len[prev]; t.prev_unit_normals = P.unit_normals[prev];
    t.out_color = P.hist_color[cur]; t.out_moments = P.moments[cur]; t.hist_world_pos = P.world_pos[cur];
    t.out_hist_len = P.hist_len[cur]; t.unit_normals = P.unit_normals[cur]; t.motion = P.motion;
    blok::launch_temporal(t, stream);

    blok::VarianceArgs v{};
    v.f = f;
    v.color = P.hist_color[cur]; v.moments = P.moments[cur]; v.world_pos = P.world_pos[cur];
    v.hist_len = P.hist_len[cur]; v.unit_normals = P.unit_normals[cur]; v.variance = P.variance;
    blok::launch_variance(v, stream);

    // iteration 0 reads the temporal output; then ping <-> pong (renderer_denoising.cpp:520-536); the last one writes the caller's plane
    const float* in = P.hist_color[cur];
    for (int it = 0; it < S.atrous_iterations; ++it) {
        blok::AtrousArgs a{};
        a.w = f.w; a.h = f.h; a.step = 1 << it; a.phi_color = S.phi_color; a.phi_depth = S.phi_depth;
        a.color = in; a.variance = P.variance; a.world_pos = P.world_pos[cur]; a.unit_normals = P.unit_normals[cur];
        a.out = it == S.atrous_iterations - 1 ? out_color_dev : ((it & 1) ? P.pong : P.ping);
        blok::launch_atrous(a, stream);
        in = a.out;
    }
    if (S.atrous_iterations == 0)
        BLOK_HIP_TRY(ctx, hipMemcpyAsync(out_color_dev, P.hist_color[cur], P.pixels * 4 * sizeof(float), hipMemcpyDeviceToDevice, stream));
    BLOK_HIP_TRY(ctx, hipGetLastError());
    P.cur = prev;                                          // swapHistoryBuffers
    P.has_motion = true;
    return BLOK_OK;
}

int blok_hip_denoise_state(blok_hip_ctx* ctx, float* history_color, float* moments, float* history_length, float* variance, float* motion) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    auto& P = ctx->post;
    if (!P.pixels || !P.has_motion) return set_error(ctx, BLOK_ERR_INVALID_ARG, "no denoised frame yet");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    BLOK_HIP_TRY(ctx, hipDeviceSynchronize());
    const int last = P.cur ^ 1;                            // the slot the last frame wrote
    const size_t n = P.pixels;
    if (history_color) BLOK_HIP_TRY(ctx, hipMemcpy(history_color, P.hist_color[last], 4 * n * sizeof(float), hipMemcpyDeviceToHost));
    if (moments) BLOK_HIP_TRY(ctx, hipMemcpy(moments, P.moments[last], 2 * n * sizeof(float), hipMemcpyDeviceToHost));
    if (variance) BLOK_HIP_TRY(ctx, hipMemcpy(variance, P.variance, n * sizeof(float), hipMemcpyDeviceToHost));
    if (history_length) {
        blok::launch_widen(P.hist_len[last], P.widen, n, nullptr);
        BLOK_HIP_TRY(ctx, hipMemcpy(history_length, P.widen, n * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (motion) {
        blok::launch_widen(P.motion, P.widen, 2 * n, nullptr);
        BLOK_HIP_TRY(ctx, hipMemcpy(motion, P.widen, 2 * n * sizeof(float), hipMemcpyDeviceToHost));
    }
    return BLOK_OK;
}

namespace {
__global__ __launch_bounds__(256) void narrow_motion_kernel(const float* src, uint16_t* dst, size_t n) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x;
    if (i < n) dst[i] = __half_as_ushort(__float2half_rn(src[i]));
}
}  // namespace

int blok_hip_taa_device(blok_hip_ctx* ctx, const float* color_dev, const float* motion_dev, float feedback_min, float feedback_max,
                        uint32_t frame_count, float* out_color_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!color_dev || !out_color_dev) return set_error(ctx, BLOK_ERR_INVALID_ARG, "taa: null plane");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_post(ctx);
    if (rc != BLOK_OK) return rc;
    auto& P = ctx->post;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (motion_dev) {
        const size_t n = 2 * P.pixels;
        hipLaunchKernelGGL(narrow_motion_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream, motion_dev, P.motion, n);
        P.has_motion = true;
    } else if (!P.has_motion) {
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "taa: no motion vectors (pass a plane or denoise a frame first)");
    }
    blok::TaaArgs a{};
    a.w = ctx->width; a.h = ctx->height; a.frame_count = frame_count; a.feedback_min = feedback_min; a.feedback_max = feedback_max;
    a.color = color_dev; a.history = P.taa_hist[P.taa_cur ^ 1]; a.motion = P.motion;
    a.out = out_color_dev; a.out_history = P.taa_hist[P.taa_cur];
    blok::launch_taa(a, stream);
    BLOK_HIP_TRY(ctx, hipGetLastError());
    P.taa_cur ^= 1;
    return BLOK_OK;
}

int blok_hip_sharpen_device(blok_hip_ctx* ctx, const uint32_t* rgba8_dev, float strength, uint32_t* out_rgba8_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!rgba8_dev || !out_rgba8_dev || rgba8_dev == out_rgba8_dev) return set_error(ctx, BLOK_ERR_INVALID_ARG, "sharpen: two distinct planes are required");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    blok::SharpenArgs a{};
    a.w = ctx->width; a.h = ctx->height; a.strength = strength; a.in = rgba8_dev; a.out = out_rgba8_dev;
    blok::launch_sharpen(a, static_cast<hipStream_t>(hip_stream));
    BLOK_HIP_TRY(ctx, hipGetLastError());
    return BLOK_OK;
}

namespace {
// Column-major prevViewProj of a camera basis: uv = ndc.xy * 0.5 + 0.5 reproduces the basis' own pixel mapping
// (x + 0.5 = u * width, y + 0.5 = v * height), the role FrameUBO::prevViewProj plays for the reference's shaders.
void view_proj_of(const blok_camera& c, float M[16]) {
    const double ta = double(c.tan_half_fov) * double(c.aspect), t = double(c.tan_half_fov);
    double rows[4][4] = {};
    for (int a = 0; a < 3; ++a) {
        rows[0][a] = double(c.right[a]) / ta; rows[0][3] -= double(c.right[a]) * double(c.pos[a]) / ta;
        rows[1][a] = -double(c.up[a]) / t;    rows[1][3] += double(c.up[a]) * double(c.pos[a]) / t;
        rows[2][a] = double(c.fwd[a]);        rows[2][3] -= double(c.fwd[a]) * double(c.pos[a]);
    }
    for (int a = 0; a < 4; ++a) rows[3][a] = rows[2][a];
    for (int col = 0; col < 4; ++col) for (int r = 0; r < 4; ++r) M[col * 4 + r] = static_cast<float>(rows[r][col]);
}
}  // namespace

void blok_camera_view_proj(const blok_camera* cam, float out_view_proj[16]) { if (cam && out_view_proj) view_proj_of(*cam, out_view_proj); }

int blok_hip_draw_frame_rt(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t spp, uint32_t max_bounces,
                           const blok_denoise_settings* settings, uint32_t* out_rgba8_host, uint32_t* out_frame_count) {
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    if (!spp || !max_bounces) return set_error(ctx, BLOK_ERR_INVALID_ARG, "spp and bounces must be positive");
    rc = ensure_post(ctx);
    if (rc != BLOK_OK) return rc;
    auto& P = ctx->post;
    const size_t n = P.pixels;
    if (!P.rt_final) {
        for (int k = 0; k < 4 && rc == BLOK_OK; ++k) rc = post_alloc(ctx, &P.rt_planes[k], 4 * n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.rt_denoised, 4 * n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.rt_resolved, 4 * n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.rt_ldr, n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.rt_final, n);
        if (rc != BLOK_OK) { free_post(ctx); return rc; }
        P.rt_frame = 0;
    }
    const uint32_t frame = P.rt_frame;
    float prev_vp[16];
    view_proj_of(frame ? P.rt_prev_cam : *cam, prev_vp);            // Denoiser::updatePreviousFrameData: last frame's matrices
    const blok_gbuffer planes{P.rt_planes[0], P.rt_planes[1], P.rt_planes[2], P.rt_planes[3]};
    rc = blok_hip_trace_paths_device(ctx, cam, 0, 0, ctx->width, ctx->height, spp, max_bounces, frame, &planes, nullptr);
    if (rc == BLOK_OK) rc = blok_hip_denoise_device(ctx, &planes, nullptr, prev_vp, frame, settings, P.rt_denoised, nullptr);
    if (rc == BLOK_OK) rc = blok_hip_taa_device(ctx, P.rt_denoised, nullptr, 0.93f, 0.98f, frame, P.rt_resolved, nullptr);       // renderer_postprocess.hpp:104-106
    if (rc == BLOK_OK) rc = blok_hip_tonemap_device(ctx, P.rt_resolved, static_cast<uint32_t>(n), 1.0f, 1.15f, 1, P.rt_ldr, nullptr);   // :110-113
    if (rc == BLOK_OK) rc = blok_hip_sharpen_device(ctx, P.rt_ldr, 0.5f, P.rt_final, nullptr);                                     // :117-118
    if (rc != BLOK_OK) return rc;
    P.rt_prev_cam = *cam;
    P.rt_frame = frame + 1;
    if (out_frame_count) *out_frame_count = P.rt_frame;
    if (out_rgba8_host) BLOK_HIP_TRY(ctx, hipMemcpy(out_rgba8_host, P.rt_final, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    else BLOK_HIP_TRY(ctx, hipDeviceSynchronize());
    return BLOK_OK;
}

int blok_hip_post_reset(blok_hip_ctx* ctx) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_post(ctx);
    return BLOK_OK;
}

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

int blok_hip_set_sun_map(blok_hip_ctx* ctx, int enabled) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->sun_map_enabled = enabled != 0;
    return BLOK_OK;
}

int blok_hip_set_beam(blok_hip_ctx* ctx, uint32_t beam_tile_pixels) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (beam_tile_pixels != 0 && beam_tile_pixels != 8 && beam_tile_pixels != 16 && beam_tile_pixels != 32 && beam_tile_pixels != 64)
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "beam tile must be 0 (off), 8, 16, 32 or 64 pixels");
    ctx->beam_tile = beam_tile_pixels;
    return BLOK_OK;
}

int blok_hip_set_timing(blok_hip_ctx* ctx, int enabled) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->timing = enabled != 0;
    ctx->timed = false;
    return BLOK_OK;
}

int blok_hip_last_kernel_ms(blok_hip_ctx* ctx, float* out_ms) {
    if (!ctx || !out_ms) return BLOK_ERR_INVALID_ARG;
    if (!ctx->timed) return set_error(ctx, BLOK_ERR_INVALID_ARG, "no timed launch (enable with blok_hip_set_timing)");
    BLOK_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_end));
    BLOK_HIP_TRY(ctx, hipEventElapsedTime(out_ms, ctx->ev_begin, ctx->ev_end));
    return BLOK_OK;
}

}  // extern "C"
