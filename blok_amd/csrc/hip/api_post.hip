// C ABI, image-space chain: denoiser, TAA, sharpen, the one-call ray-tracing frame (include/blok_hip.h; kernels in post_core.h).
#include "api_internal.h"
#include "../common/taa_jitter.h"

using namespace blok_api;

extern "C" {

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
    if (P.pixels == n && P.width == ctx->width && P.height == ctx->height) return BLOK_OK;     // same shape: history stays valid
    free_post(ctx);       // any resize — also one that keeps the pixel count (320x200 -> 200x320) — drops the history planes
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
    P.pixels = n; P.width = ctx->width; P.height = ctx->height;
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

// One frame through temporal accumulation, variance and the a-trous iterations; the frame's planes come in `in` (float4 planes,
// or the reference-format halves of normal + roughness / motion).
static int denoise_frame(blok_hip_ctx* ctx, const blok::TemporalArgs& in, const float prev_view_proj[16], uint32_t frame_count,
                         const blok_denoise_settings* settings, float* out_color_dev, void* hip_stream) {
    if (!in.color || !in.world_pos || (!in.normal_roughness && !in.normal_roughness_h) || !prev_view_proj || !out_color_dev)
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

    blok::TemporalArgs t = in;
    t.f = f;

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
    const float* src = P.hist_color[cur];
    for (int it = 0; it < S.atrous_iterations; ++it) {
        blok::AtrousArgs a{};
        a.w = f.w; a.h = f.h; a.step = 1 << it; a.phi_color = S.phi_color; a.phi_depth = S.phi_depth;
        a.color = src; a.variance = P.variance; a.world_pos = P.world_pos[cur]; a.unit_normals = P.unit_normals[cur];
        a.out = it == S.atrous_iterations - 1 ? out_color_dev : ((it & 1) ? P.pong : P.ping);
        blok::launch_atrous(a, stream);
        src = a.out;
    }
    if (S.atrous_iterations == 0)
        BLOK_HIP_TRY(ctx, hipMemcpyAsync(out_color_dev, P.hist_color[cur], P.pixels * 4 * sizeof(float), hipMemcpyDeviceToDevice, stream));
    BLOK_HIP_TRY(ctx, hipGetLastError());
    P.cur = prev;                                          // swapHistoryBuffers
    P.has_motion = true;
    return BLOK_OK;
}

int blok_hip_denoise_device(blok_hip_ctx* ctx, const blok_gbuffer* planes, const float* motion_dev, const float prev_view_proj[16],
                            uint32_t frame_count, const blok_denoise_settings* settings, float* out_color_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!planes) return set_error(ctx, BLOK_ERR_INVALID_ARG, "denoise: null planes");
    blok::TemporalArgs t{};
    t.color = planes->color; t.world_pos = planes->world_pos; t.normal_roughness = planes->normal_roughness; t.motion_in = motion_dev;
    return denoise_frame(ctx, t, prev_view_proj, frame_count, settings, out_color_dev, hip_stream);
}

int blok_hip_denoise_ref_device(blok_hip_ctx* ctx, const blok_gbuffer_ref* planes, const float prev_view_proj[16],
                                uint32_t frame_count, const blok_denoise_settings* settings, float* out_color_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!planes) return set_error(ctx, BLOK_ERR_INVALID_ARG, "denoise: null planes");
    blok::TemporalArgs t{};
    t.color = planes->color; t.world_pos = planes->world_pos; t.normal_roughness_h = planes->normal_roughness; t.motion_in_h = planes->motion;
    return denoise_frame(ctx, t, prev_view_proj, frame_count, settings, out_color_dev, hip_stream);
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
        // the frame's G-buffer in the reference's image formats (raygen.rgen:55-59): 2 x RGBA32F, RGBA16F, RGBA8, RG16F = 48 B/pixel
        for (int k = 0; k < 2 && rc == BLOK_OK; ++k) rc = post_alloc(ctx, &P.rt_planes[k], 4 * n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.rt_normal_roughness_h, 4 * n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.rt_albedo_metallic_u8, n);
        if (rc == BLOK_OK) rc = post_alloc(ctx, &P.rt_motion_h, 2 * n);
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
    const blok_gbuffer_ref planes{P.rt_planes[0], P.rt_planes[1], P.rt_normal_roughness_h, P.rt_albedo_metallic_u8, P.rt_motion_h};
    // the frame's projection carries jitterSequence[frame mod 16] (renderer_draw.cpp:64-81; the index advances once per frame,
    // renderer_postprocess.cpp:660-663); prevViewProj above stays un-jittered, as Denoiser::updatePreviousFrameData is fed
    // (renderer_draw.cpp:313-328 passes the base matrices)
    const float saved_jitter[2] = {ctx->jitter_px[0], ctx->jitter_px[1]};
    if (ctx->rt_taa_jitter) blok::taa_jitter_px(frame, ctx->jitter_px);
    rc = blok_hip_trace_paths_ref_device(ctx, cam, 0, 0, ctx->width, ctx->height, spp, max_bounces, frame, prev_vp, &planes, nullptr);
    ctx->jitter_px[0] = saved_jitter[0]; ctx->jitter_px[1] = saved_jitter[1];
    if (rc == BLOK_OK) rc = blok_hip_denoise_ref_device(ctx, &planes, prev_vp, frame, settings, P.rt_denoised, nullptr);
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

}  // extern "C"
