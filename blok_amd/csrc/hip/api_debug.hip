// The entries of include/blok_hip_debug.h: tuning switches that were measured against each other, read-backs and hooks for the tests and the
// experiment scripts.  Nothing a drop-in user of include/blok_hip.h needs; split from api.hip in round 4.
#include "api_internal.h"
#include <cstdlib>
#include <cstdio>

using namespace blok_api;

extern "C" {

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

// The pre-pass alone: start parameter (and node visits) per beam tile of the rectangle, to the host.
int blok_hip_beam_prepass(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                          float* out_t0_host, uint32_t* out_visits_host, size_t capacity) {
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    if (!out_t0_host || !rect_inside(ctx, x0, y0, w, h)) return set_error(ctx, BLOK_ERR_INVALID_ARG, "rectangle outside the frame or no output");
    if (!ctx->beam_tile) return set_error(ctx, BLOK_ERR_INVALID_ARG, "the beam pre-pass is off (blok_hip_set_beam)");
    blok::TraceArgs a = base_args(ctx, cam);
    a.x0 = x0; a.y0 = y0; a.w = w; a.h = h;
    uint32_t n_beams = 0;
    rc = prepare_beam(ctx, blok::RayMode::Rect, a, nullptr, 0, &n_beams);
    if (rc != BLOK_OK) return rc;
    if (!n_beams || capacity < n_beams) return set_error(ctx, BLOK_ERR_INVALID_ARG, "output too small for the rectangle's beam tiles");
    uint32_t* d_visits = nullptr;
    if (out_visits_host) BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_visits), n_beams * sizeof(uint32_t)));
    a.debug_visits = d_visits;
    a.miss_in_walk = 1u;                               // nothing is written but the start parameters
    blok::launch_beam(blok::RayMode::Rect, a, n_beams, nullptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out_t0_host, a.beam, n_beams * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && d_visits) e = hipMemcpy(out_visits_host, d_visits, n_beams * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (d_visits) (void)hipFree(d_visits);
    BLOK_HIP_TRY(ctx, e);
    return BLOK_OK;
}

// The counting sort behind a moving camera's frames (tile_order.h), on the caller's costs: order, its inverse, the live prefix and the
// depth sums, to the host.
int blok_hip_debug_class_order(blok_hip_ctx* ctx, const uint32_t* cost_host, uint32_t tiles_x, uint32_t tiles_y, uint32_t radius, const float* beam_host, uint32_t n_beams,
                               uint32_t* out_order_host, uint32_t* out_rank_of_host, uint32_t* out_live, float* out_depth_sums3) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!cost_host || !out_order_host || !out_rank_of_host || !out_live || !tiles_x || !tiles_y || static_cast<uint64_t>(tiles_x) * tiles_y > (1u << 24) || radius > 8u)
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "class order: costs, outputs, a grid of at most 2^24 tiles and a radius of at most 8");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n = tiles_x * tiles_y;
    uint32_t *d_cost = nullptr, *d_order = nullptr, *d_rank = nullptr, *d_live = nullptr;
    float *d_beam = nullptr, *d_depth = nullptr;
    void* d_scratch = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_cost), n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_order), n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_rank), n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_live), sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_depth), blok::kOrderDepthPartials * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_beam), (n_beams ? n_beams : 1u) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d_scratch, blok::tile_order_class_sort_bytes(tiles_x, tiles_y));
    if (e == hipSuccess) e = hipMemcpy(d_cost, cost_host, n * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_beams && beam_host) e = hipMemcpy(d_beam, beam_host, n_beams * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_order, 0xFF, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(d_rank, 0xFF, n * sizeof(uint32_t));
    if (e == hipSuccess) e = blok::launch_tile_order_class_sort(d_cost, tiles_x, tiles_y, radius, d_scratch, d_order, d_rank, d_live, d_beam, nullptr, 0u, beam_host ? n_beams : 0u, d_depth, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_order_host, d_order, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_rank_of_host, d_rank, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_live, d_live, sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_depth_sums3) {
        float part[blok::kOrderDepthPartials * 3];
        e = hipMemcpy(part, d_depth, sizeof(part), hipMemcpyDeviceToHost);
        out_depth_sums3[0] = out_depth_sums3[1] = out_depth_sums3[2] = 0.0f;
        for (uint32_t k = 0; k < blok::kOrderDepthPartials; ++k) for (int c = 0; c < 3; ++c) out_depth_sums3[c] += part[k * 3 + c];
    }
    for (void* p : {static_cast<void*>(d_cost), static_cast<void*>(d_order), static_cast<void*>(d_rank), static_cast<void*>(d_live), static_cast<void*>(d_depth), static_cast<void*>(d_beam), d_scratch})
        if (p) (void)hipFree(p);
    BLOK_HIP_TRY(ctx, e);
    return BLOK_OK;
}

// Walks exactly the listed 8x8-pixel wave tiles of the rectangle, in list order (walk workgroup j takes entry j).
int blok_hip_trace_wave_tiles_device(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                     const uint32_t* tiles_host, const float* t0_host, size_t n_tiles, void* out_hits_dev, void* out_rgba_dev, void* hip_stream) {
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    if ((!out_hits_dev && !out_rgba_dev) || !rect_inside(ctx, x0, y0, w, h)) return set_error(ctx, BLOK_ERR_INVALID_ARG, "rectangle outside the frame or no output");
    if (blok::kBlock != 64) return set_error(ctx, BLOK_ERR_UNSUPPORTED, "wave-tile lists need the one-wave-per-workgroup build");
    if (!n_tiles) return BLOK_OK;
    if (!tiles_host) return set_error(ctx, BLOK_ERR_INVALID_ARG, "null tile list");
    const uint32_t bx_count = (w + blok::kTileW - 1u) / blok::kTileW, by_count = (h + blok::kTileH - 1u) / blok::kTileH;
    if (static_cast<uint64_t>(bx_count) * by_count >= (1u << blok::kListTaskBits) || n_tiles > 0x7FFFFFFFu) return set_error(ctx, BLOK_ERR_UNSUPPORTED, "rectangle too large for a wave-tile list");
    for (size_t i = 0; i < n_tiles; ++i) {
        if (tiles_host[i] >= bx_count * by_count) return set_error(ctx, BLOK_ERR_INVALID_ARG, "wave tile index outside the rectangle");
        if (t0_host && !(t0_host[i] >= 0.0f && t0_host[i] < blok::kBeamNone)) return set_error(ctx, BLOK_ERR_INVALID_ARG, "start parameters must be finite and >= 0");
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    blok::TraceArgs a = base_args(ctx, cam);
    a.x0 = x0; a.y0 = y0; a.w = w; a.h = h;
    a.out = static_cast<blok_hit*>(out_hits_dev);
    a.out_rgba = static_cast<uint32_t*>(out_rgba_dev);
    a.beam_tile = blok::kWaveW;                          // one entry per "search": the list's capacity arithmetic
    rc = live_list(ctx, a, stream, static_cast<uint32_t>(n_tiles), 1u);
    if (rc != BLOK_OK) return rc;
    const size_t cap = a.list.seg_capacity;
    const uint32_t cls = blok::kListUnknownClass;          // one class: the caller's order is the order
    std::vector<unsigned long long> entries(cap * blok::kListSegments * blok::kListClasses, 0ull), ctl(blok::kListSegments * blok::kListCtlWords, 0ull);
    uint32_t count[blok::kListSegments] = {};
    for (size_t i = 0; i < n_tiles; ++i) {                 // entry i -> segment i mod 8, slot i / 8: walk workgroup i takes it
        const uint32_t seg = static_cast<uint32_t>(i % blok::kListSegments);
        const float t0 = t0_host ? t0_host[i] : 0.0f;
        uint32_t bits; std::memcpy(&bits, &t0, sizeof(bits));
        entries[(seg * blok::kListClasses + cls) * cap + i / blok::kListSegments] = (static_cast<unsigned long long>(a.list.serial) << 44) | (static_cast<unsigned long long>(tiles_host[i]) << 23) | (bits >> 8);
        count[seg] += 1u;
    }
    for (uint32_t seg = 0; seg < blok::kListSegments; ++seg)
        for (uint32_t c = 0; c < blok::kListClasses; ++c)
            ctl[seg * blok::kListCtlWords + blok::kListFinal + c] = (static_cast<unsigned long long>(a.list.serial) << 32) | (c == cls ? count[seg] : 0u);
    BLOK_HIP_TRY(ctx, hipStreamSynchronize(stream));     // the stream's list may still be in use by an earlier launch
    BLOK_HIP_TRY(ctx, hipMemcpy(a.list.entries, entries.data(), entries.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    BLOK_HIP_TRY(ctx, hipMemcpy(a.list.ctl, ctl.data(), ctl.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    ctx->beam_buffers[stream].list_hint_valid = false;   // this list says nothing about the next frame's
    a.list.hint = nullptr;
    const uint32_t walkers = static_cast<uint32_t>((n_tiles + blok::kListSegments - 1u) / blok::kListSegments) * blok::kListSegments;
    for (uint32_t c = 0; c < blok::kListClasses; ++c) a.list.walkers[c] = c == cls ? walkers / blok::kListSegments : 0u;
    if (ctx->timing) BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, stream));
    blok::launch_list_walk(blok::RayMode::Rect, a, nullptr, walkers, stream);
    BLOK_HIP_TRY(ctx, hipGetLastError());
    if (ctx->timing) { BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_end, stream)); ctx->timed = true; }
    return BLOK_OK;
}

int blok_hip_set_debug_wave_clocks(blok_hip_ctx* ctx, void* clocks_dev) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->debug_clocks = static_cast<uint32_t*>(clocks_dev);
    return BLOK_OK;
}

int blok_hip_set_path_start(blok_hip_ctx* ctx, int resume_from_anchor, int wave_tile_beam) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->path_resume = resume_from_anchor != 0; ctx->path_fine_beam = wave_tile_beam != 0;
    return BLOK_OK;
}

int blok_hip_set_ray_batching(blok_hip_ctx* ctx, int enabled) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (enabled < 0 || enabled > 3) return set_error(ctx, BLOK_ERR_INVALID_ARG, "ray batching: 0 off, 1 by kind, 2 by sample and kind, 3 and the bounce rounds' tail pool");
    ctx->ray_batching = static_cast<uint32_t>(enabled);
    if (const char* caps = std::getenv("BLOK_TAIL_CAPS")) {      // "24,32": experiments
        unsigned a = 0, b = 0;
        if (std::sscanf(caps, "%u,%u", &a, &b) == 2 && a && b) { ctx->tail_cap = a; ctx->tail_cap_parked = b; }
    }
    return BLOK_OK;
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

int blok_hip_set_fused(blok_hip_ctx* ctx, int enabled) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (enabled < 0 || enabled > 5) return set_error(ctx, BLOK_ERR_INVALID_ARG, "launch form: 0 (two launches), 1 (one persistent launch with queues), 2 (joint launch), 3 (automatic), 4 (list-fed joint launch) or 5 (list-fed walk behind the beam launch)");
    ctx->launch_form = enabled;
    return BLOK_OK;
}

int blok_hip_frame_queue_stalls(blok_hip_ctx* ctx, uint32_t* out_stalled_waves) {
    if (!ctx || !out_stalled_waves) return BLOK_ERR_INVALID_ARG;
    *out_stalled_waves = 0;
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    BLOK_HIP_TRY(ctx, hipDeviceSynchronize());
    for (auto& kv : ctx->beam_buffers) {
        if (kv.second.gave_up) {                                         // joint launch: walk waves that stopped waiting for their tile's search
            uint32_t n = 0;
            BLOK_HIP_TRY(ctx, hipMemcpy(&n, kv.second.gave_up, sizeof(n), hipMemcpyDeviceToHost));
            *out_stalled_waves += n;
        }
        if (!kv.second.ctl) continue;
        for (uint32_t part = 0; part < blok::kFrameParts; ++part) {      // all possible parts: the words of unused ones stay 0
            uint32_t n = 0;
            BLOK_HIP_TRY(ctx, hipMemcpy(&n, kv.second.ctl + part * blok::kFramePartWords + blok::kFrameStalledWord, sizeof(n), hipMemcpyDeviceToHost));
            *out_stalled_waves += n;
        }
    }
    return BLOK_OK;
}

int blok_hip_last_launch_kind(const blok_hip_ctx* ctx) { return ctx ? ctx->last_launch_kind : -1; }

int blok_hip_set_tile_ordering(blok_hip_ctx* ctx, int resort_every_n_frames) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (resort_every_n_frames < 0) return set_error(ctx, BLOK_ERR_INVALID_ARG, "tile ordering: interval must be >= 0");
    ctx->order.enabled = resort_every_n_frames != 0;
    if (resort_every_n_frames) ctx->order.interval = ctx->order.interval_now = static_cast<uint32_t>(resort_every_n_frames);
    return BLOK_OK;
}

int blok_hip_set_rank_tile_ordering(blok_hip_ctx* ctx, int enabled) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->order.rank_tiles = enabled != 0;
    return BLOK_OK;
}

int blok_hip_debug_force_order_shift(blok_hip_ctx* ctx, int enabled, uint32_t shift_x, uint32_t shift_y) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->order.debug_shift = enabled != 0; ctx->order.debug_sx = shift_x; ctx->order.debug_sy = shift_y;
    return BLOK_OK;
}

int64_t blok_hip_last_fallback_tiles(const blok_hip_ctx* ctx) {
    if (!ctx || !ctx->order.h_fallback) return -1;
    return static_cast<int64_t>(*static_cast<volatile uint32_t*>(ctx->order.h_fallback));
}

int blok_hip_set_moving_order(blok_hip_ctx* ctx, int enabled) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->order.moving = enabled != 0;
    return BLOK_OK;
}

int blok_hip_last_order_use(const blok_hip_ctx* ctx, int32_t* out_shift_x, int32_t* out_shift_y) {
    if (!ctx) return -1;
    if (out_shift_x) *out_shift_x = static_cast<int32_t>(ctx->order.last_sx);
    if (out_shift_y) *out_shift_y = static_cast<int32_t>(ctx->order.last_sy);
    return ctx->order.last_use;
}

int blok_hip_set_joint_prefix_limit(blok_hip_ctx* ctx, uint32_t max_walk_waves) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->order.prefix_limit = max_walk_waves;
    return BLOK_OK;
}

int blok_hip_set_list_classes(blok_hip_ctx* ctx, int enabled) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->list_classes = enabled != 0;
    return BLOK_OK;
}

int blok_hip_set_miss_writer(blok_hip_ctx* ctx, int in_walk) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->miss_in_walk = in_walk != 0;
    return BLOK_OK;
}

int blok_hip_set_beam_budget(blok_hip_ctx* ctx, uint32_t max_node_visits) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->beam_budget = max_node_visits;
    return BLOK_OK;
}

}  // extern "C"
