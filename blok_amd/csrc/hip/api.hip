// C ABI of the gfx950 backend (include/blok_hip.h).  Owns device memory; every HIP call is checked.
#include "api_internal.h"
#include <cstdlib>

namespace blok_api {


thread_local std::string g_create_error;

int set_error(blok_hip_ctx* ctx, int status, const std::string& msg) {
    if (ctx) ctx->error = msg; else g_create_error = msg;
    return status;
}


void free_post(blok_hip_ctx* ctx) {
    auto& P = ctx->post;
    for (int k = 0; k < 2; ++k)
        for (void* p : {static_cast<void*>(P.hist_color[k]), static_cast<void*>(P.moments[k]), static_cast<void*>(P.world_pos[k]),
                        static_cast<void*>(P.hist_len[k]), static_cast<void*>(P.unit_normals[k]), static_cast<void*>(P.taa_hist[k])})
            if (p) (void)hipFree(p);
    for (void* p : {static_cast<void*>(P.motion), static_cast<void*>(P.variance), static_cast<void*>(P.ping), static_cast<void*>(P.pong), static_cast<void*>(P.widen),
                    static_cast<void*>(P.rt_planes[0]), static_cast<void*>(P.rt_planes[1]), static_cast<void*>(P.rt_planes[2]), static_cast<void*>(P.rt_planes[3]),
                    static_cast<void*>(P.rt_denoised), static_cast<void*>(P.rt_resolved), static_cast<void*>(P.rt_ldr), static_cast<void*>(P.rt_final),
                    static_cast<void*>(P.rt_normal_roughness_h), static_cast<void*>(P.rt_motion_h), static_cast<void*>(P.rt_albedo_metallic_u8)})
        if (p) (void)hipFree(p);
    P = blok_hip_ctx::Post{};
}

void free_world(blok_hip_ctx* ctx) {
    if (ctx->d_nodes && !ctx->tree_owned_by_volume) (void)hipFree(ctx->d_nodes);
    if (ctx->d_tree_materials && !ctx->tree_owned_by_volume) (void)hipFree(ctx->d_tree_materials);
    ctx->tree_owned_by_volume = false;
    if (ctx->d_materials) (void)hipFree(ctx->d_materials);
    if (ctx->d_sun_map) (void)hipFree(ctx->d_sun_map);
    ctx->d_sun_map = nullptr; ctx->has_sun_map = false;
    if (ctx->d_dense_tiled) (void)hipFree(ctx->d_dense_tiled);
    if (ctx->d_dense_bits) (void)hipFree(ctx->d_dense_bits);
    ctx->d_dense_tiled = ctx->d_dense_bits = nullptr; ctx->has_dense = false;
    ctx->d_nodes = nullptr; ctx->d_tree_materials = nullptr; ctx->d_materials = nullptr;
    ctx->n_materials = 0; ctx->has_world = false; ctx->built_on_device = false; ctx->stats = blok_world_stats{};
    ctx->world_voxel_size = 1.0f;
}

int install_materials(blok_hip_ctx* ctx, const blok_material* materials, size_t n_materials) {
    if (!n_materials) return BLOK_OK;
    BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_materials), n_materials * sizeof(blok_material)));
    BLOK_HIP_TRY(ctx, hipMemcpy(ctx->d_materials, materials, n_materials * sizeof(blok_material), hipMemcpyHostToDevice));
    ctx->n_materials = n_materials;
    return BLOK_OK;
}


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
    ctx->has_world = true; ctx->world_version += 1u;
    ctx->world_voxel_size = ctx->pending_voxel_size;
    return rebuild_sun_map(ctx);
}

// Rebuilds the shadow rays' last-occluder map for the installed world (one wave per texel of the plane perpendicular to the
// sun; texel = 1 voxel, coarser for worlds wider than 1024 voxels so that the map stays <= 2048 x 2048).
int rebuild_sun_map(blok_hip_ctx* ctx) {
    if (ctx->d_sun_map) { (void)hipDeviceSynchronize(); (void)hipFree(ctx->d_sun_map); ctx->d_sun_map = nullptr; }
    ctx->has_sun_map = false;
    ctx->sun_loose_edits = 0; ctx->sun_tighten_pending = false;      // a whole new map owes nothing
    if (!ctx->has_world || ctx->stats.levels == 0 || ctx->stats.n_voxels == 0) return BLOK_OK;
    if (ctx->world_voxel_size != 1.0f) return BLOK_OK;      // the map is laid out in world units = voxel units; other voxel sizes go without it
    blok::SunMapArgs& m = ctx->sun;
    m = blok::SunMapArgs{};
    m.trace.nodes = ctx->d_nodes; m.trace.materials = ctx->d_tree_materials;
    for (int a = 0; a < 3; ++a) m.trace.origin[a] = ctx->stats.origin[a];
    m.trace.levels = ctx->stats.levels;
    m.trace.voxel_size = 1.0f; m.trace.inv_voxel_size = 1.0f;
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
    // (round 4: texels of one voxel instead of four — a texel's bound is the farthest occluder of its whole column, so finer texels cap more
    // shadow rays at once: configs[4] 39.3 -> 38.4 ms, B 84.2 -> 83.2, C 62.5 -> 61.1, profiles/r04_sun_texel_ab.txt; 13 MB at 1024^3, 6 ms to build)
    m.texel = static_cast<float>(std::max(1.0, std::ceil(extent / 2048.0)));
    m.u0 = static_cast<float>(std::floor(lo[0]) - m.texel); m.v0 = static_cast<float>(std::floor(lo[1]) - m.texel);
    m.nu = static_cast<uint32_t>(std::ceil((hi[0] - m.u0) / m.texel)) + 1u; m.nv = static_cast<uint32_t>(std::ceil((hi[1] - m.v0) / m.texel)) + 1u;
    BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_sun_map), static_cast<size_t>(m.nu) * m.nv * sizeof(float)));
    m.map = ctx->d_sun_map;
    m.iu0 = m.iv0 = 0; m.sub_nu = m.nu; m.sub_nv = m.nv;
    blok::launch_sun_map(m, nullptr);
    BLOK_HIP_TRY(ctx, hipGetLastError());
    BLOK_HIP_TRY(ctx, hipDeviceSynchronize());
    ctx->has_sun_map = true;
    return BLOK_OK;
}

// After an edit of the installed world that left its lattice alone (same origin, same levels), [lo, hi) = the edited box in world voxels.
// The map is an UPPER bound per texel (a shadow ray's tmax is capped there: a larger value caps later, never wrongly), so it is kept
// valid cheaply — an edit that can only have emptied voxels (may_add false) changes nothing; one that may have filled some raises the texels
// whose prism meets the box to the box's farthest corner along the sun (one tiny launch) — and made TIGHT again by the search (one wave
// per texel through the new tree, 30-60 us: round 3 ran it behind every edit, a seventh of the brush -> tree latency) only every
// kSunMapLooseEdits edits, over the union of their boxes, a band of rows per edit.  Anything else: the whole map.
constexpr uint32_t kSunMapLooseEdits = 32;
// (round 4, texels of one voxel: the union of 32 scattered boxes can be the whole map — 3 M searches, 5 ms behind one edit.  The texels to be
// made tight are kept as a rectangle and searched a band of rows at a time, at most kSunMapBandTexels per edit: about 50 us.)
constexpr uint32_t kSunMapBandTexels = 32768;
int update_sun_map(blok_hip_ctx* ctx, const int32_t lo[3], const int32_t hi[3], bool same_lattice, bool may_add) {
    if (!same_lattice || !ctx->has_sun_map || !ctx->d_sun_map || ctx->world_voxel_size != 1.0f || !ctx->has_world || ctx->stats.n_voxels == 0) {
        ctx->sun_loose_edits = 0; ctx->sun_tighten_pending = false;
        return rebuild_sun_map(ctx);
    }
    blok::SunMapArgs& m = ctx->sun;
    m.trace.nodes = ctx->d_nodes; m.trace.materials = ctx->d_tree_materials;
    if (lo[0] >= hi[0] || lo[1] >= hi[1] || lo[2] >= hi[2]) return BLOK_OK;            // nothing was edited
    if (ctx->sun_loose_edits == 0) for (int a = 0; a < 3; ++a) { ctx->sun_loose_lo[a] = lo[a]; ctx->sun_loose_hi[a] = hi[a]; }
    else for (int a = 0; a < 3; ++a) { ctx->sun_loose_lo[a] = std::min(ctx->sun_loose_lo[a], lo[a]); ctx->sun_loose_hi[a] = std::max(ctx->sun_loose_hi[a], hi[a]); }
    ctx->sun_loose_edits += 1;
    // the texels whose prism meets a box of world voxels (one more on every side), and the box's farthest corner along the sun
    const auto texels_of = [&](const int32_t* blo, const int32_t* bhi, uint32_t& u0, uint32_t& u1, uint32_t& v0, uint32_t& v1, double& far_depth) {
        double ulo = 1e30, uhi = -1e30, vlo = 1e30, vhi = -1e30; far_depth = -1e30;
        for (int c = 0; c < 8; ++c) {
            const double p[3] = {double((c & 1) ? bhi[0] : blo[0]), double((c & 2) ? bhi[1] : blo[1]), double((c & 4) ? bhi[2] : blo[2])};
            const double pu = m.u[0] * p[0] + m.u[1] * p[1] + m.u[2] * p[2], pv = m.v[0] * p[0] + m.v[1] * p[1] + m.v[2] * p[2];
            ulo = std::min(ulo, pu); uhi = std::max(uhi, pu); vlo = std::min(vlo, pv); vhi = std::max(vhi, pv);
            far_depth = std::max(far_depth, m.s[0] * p[0] + m.s[1] * p[1] + m.s[2] * p[2]);
        }
        const auto texel_of = [&](double x, double x0, uint32_t n) { const double t = std::floor((x - x0) / m.texel); return static_cast<int64_t>(std::min<double>(std::max<double>(t, -1.0), double(n))); };
        const int64_t iu0 = std::max<int64_t>(texel_of(ulo, m.u0, m.nu) - 1, 0), iu1 = std::min<int64_t>(texel_of(uhi, m.u0, m.nu) + 1, int64_t(m.nu) - 1);
        const int64_t iv0 = std::max<int64_t>(texel_of(vlo, m.v0, m.nv) - 1, 0), iv1 = std::min<int64_t>(texel_of(vhi, m.v0, m.nv) + 1, int64_t(m.nv) - 1);
        if (iu1 < iu0 || iv1 < iv0) return false;
        u0 = static_cast<uint32_t>(iu0); u1 = static_cast<uint32_t>(iu1); v0 = static_cast<uint32_t>(iv0); v1 = static_cast<uint32_t>(iv1);
        return true;
    };
    if (ctx->sun_loose_edits >= kSunMapLooseEdits) {
        // the union of the loose edits' boxes joins the texels still to be made tight
        uint32_t u0, u1, v0, v1; double unused;
        if (texels_of(ctx->sun_loose_lo, ctx->sun_loose_hi, u0, u1, v0, v1, unused)) {
            if (ctx->sun_tighten_pending) { u0 = std::min(u0, ctx->sun_tighten_u0); u1 = std::max(u1, ctx->sun_tighten_u1); v0 = std::min(v0, ctx->sun_tighten_v0); v1 = std::max(v1, ctx->sun_tighten_v1); }
            ctx->sun_tighten_u0 = u0; ctx->sun_tighten_u1 = u1; ctx->sun_tighten_v0 = v0; ctx->sun_tighten_v1 = v1; ctx->sun_tighten_pending = true;
        }
        ctx->sun_loose_edits = 0;
    }
    bool launched = false;
    if (ctx->sun_tighten_pending) {
        // a band of rows through the tree as it is NOW (this edit included): exact for it; later edits raise it again
        const uint32_t width = ctx->sun_tighten_u1 - ctx->sun_tighten_u0 + 1u, rows_left = ctx->sun_tighten_v1 - ctx->sun_tighten_v0 + 1u;
        const uint32_t rows = std::min(rows_left, std::max(1u, kSunMapBandTexels / width));
        m.iu0 = ctx->sun_tighten_u0; m.iv0 = ctx->sun_tighten_v0; m.sub_nu = width; m.sub_nv = rows;
        blok::launch_sun_map(m, nullptr);
        launched = true;
        if (rows == rows_left) ctx->sun_tighten_pending = false; else ctx->sun_tighten_v0 += rows;
    }
    if (may_add) {
        uint32_t u0, u1, v0, v1; double far_depth;
        if (texels_of(lo, hi, u0, u1, v0, v1, far_depth)) {
            m.iu0 = u0; m.iv0 = v0; m.sub_nu = u1 - u0 + 1u; m.sub_nv = v1 - v0 + 1u;
            blok::launch_sun_map_raise(m, static_cast<float>(far_depth + 1.0e-3 * std::fabs(far_depth) + 0.01), nullptr);      // (rounded up: the bound must not fall short in float)
            launched = true;
        }
    }
    if (!launched) return BLOK_OK;
    BLOK_HIP_TRY(ctx, hipGetLastError());
    // launches on other (non-blocking) streams do not wait for the null stream: the path entries wait for this marker instead
    if (!ctx->sun_event) BLOK_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->sun_event, hipEventDisableTiming));
    BLOK_HIP_TRY(ctx, hipEventRecord(ctx->sun_event, nullptr));
    ctx->sun_event_pending = true;
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
    a.voxel_size = ctx->world_voxel_size; a.inv_voxel_size = 1.0f / ctx->world_voxel_size;
    if (cam) a.cam = *cam;
    a.frame_w = ctx->width; a.frame_h = ctx->height;
    a.jitter_clip[0] = (2.0f * ctx->jitter_px[0]) / static_cast<float>(ctx->width);          // getJitterClipSpace, renderer_postprocess.cpp:234-241
    a.jitter_clip[1] = (2.0f * ctx->jitter_px[1]) / static_cast<float>(ctx->height);
    a.mat_table = ctx->d_materials;
    a.n_materials = static_cast<uint32_t>(ctx->n_materials);
    a.tmin = BLOK_RAY_TMIN; a.tmax = BLOK_RAY_TMAX;
    a.beam_budget = ctx->beam_budget;
    a.debug_clocks = ctx->debug_clocks;
    return a;
}

// The stream's beam buffer, grown to n floats.
int beam_buffer(blok_hip_ctx* ctx, hipStream_t stream, size_t n, float** out) {
    auto& slot = ctx->beam_buffers[stream];
    if (slot.n_beam < n) {
        if (slot.beam) { BLOK_HIP_TRY(ctx, hipStreamSynchronize(stream)); (void)hipFree(slot.beam); }
        slot.beam = nullptr; slot.n_beam = 0;
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&slot.beam), n * sizeof(float)));
        slot.n_beam = n;
    }
    *out = slot.beam;
    return BLOK_OK;
}

// The stream's work queue for a one-launch frame with n_beams beam tasks, and the size of its persistent grid.
int prepare_queue(blok_hip_ctx* ctx, blok::RayMode mode, const blok::TraceArgs& args, hipStream_t stream, uint32_t n_beams,
                  blok::FrameQueue* queue, uint32_t* n_blocks) {
    const size_t per_beam = static_cast<size_t>(args.beam_tile / blok::kWaveW) * (args.beam_tile / blok::kWaveH);
    const uint32_t parts = ctx->frame_parts;
    const size_t part_capacity = static_cast<size_t>((n_beams + parts - 1u) / parts) * per_beam;
    const size_t need = part_capacity * parts;
    auto& slot = ctx->beam_buffers[stream];
    if (!slot.ctl) {
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&slot.ctl), blok::kFrameCtlWords * sizeof(uint32_t)));
        BLOK_HIP_TRY(ctx, hipMemsetAsync(slot.ctl, 0, blok::kFrameCtlWords * sizeof(uint32_t), stream));      // in stream order, before the launch
    }
    if (slot.capacity < need) {
        if (slot.entries) { BLOK_HIP_TRY(ctx, hipStreamSynchronize(stream)); (void)hipFree(slot.entries); }
        slot.entries = nullptr; slot.capacity = 0;
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&slot.entries), need * sizeof(unsigned long long)));
        BLOK_HIP_TRY(ctx, hipMemsetAsync(slot.entries, 0xFF, need * sizeof(unsigned long long), stream));   // every slot empty; a launch leaves them so
        slot.capacity = need;
    }
    queue->ctl = slot.ctl; queue->entries = slot.entries; queue->n_beam = n_beams; queue->part_capacity = static_cast<uint32_t>(part_capacity);
    queue->n_parts = parts; queue->chunk = ctx->frame_chunk;
    const int m = mode == blok::RayMode::Rect ? 0 : 1;
    const uint32_t lv = args.levels < 16u ? args.levels : 15u;
    if (!ctx->frame_blocks_per_cu[m][lv]) {
        const int n = blok::frame_blocks_per_cu(mode, args);
        if (n <= 0) return set_error(ctx, BLOK_ERR_HIP, "frame kernel does not fit a compute unit");
        ctx->frame_blocks_per_cu[m][lv] = n;
    }
    // every resident wave slot of the chip, but never more waves than there are tasks; a multiple of the part count
    const size_t resident = static_cast<size_t>(ctx->cu_count) * ctx->frame_blocks_per_cu[m][lv];
    const size_t waves = std::min(resident, std::max<size_t>(static_cast<size_t>(n_beams) * per_beam, n_beams));
    *n_blocks = static_cast<uint32_t>((waves + parts - 1u) / parts * parts);
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

// A non-empty rectangle inside the frame; written so that x0 + w cannot wrap.
bool rect_inside(const blok_hip_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h) {
    return w && h && x0 < ctx->width && y0 < ctx->height && w <= ctx->width - x0 && h <= ctx->height - y0;
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

}  // namespace blok_api

using namespace blok_api;

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
    ctx->cu_count = prop.multiProcessorCount;
    if (const char* e = std::getenv("BLOK_FRAME_PARTS")) { const long v = std::atol(e); if (v >= 1 && v <= long(blok::kFrameParts)) ctx->frame_parts = uint32_t(v); }
    if (const char* e = std::getenv("BLOK_FRAME_CHUNK")) { const long v = std::atol(e); if (v >= 1 && v <= 64) ctx->frame_chunk = uint32_t(v); }
    if (hipEventCreate(&ctx->ev_begin) != hipSuccess || hipEventCreate(&ctx->ev_end) != hipSuccess) {
        delete ctx;
        return set_error(nullptr, BLOK_ERR_HIP, "hipEventCreate failed");
    }
    if (order_buffers(ctx, 0u, nullptr) != BLOK_OK || hipDeviceSynchronize() != hipSuccess) {      // the frame's scheduling buffers: here, so that no launch allocates
        const std::string why = ctx->error;
        blok_hip_destroy(ctx);
        return set_error(nullptr, BLOK_ERR_OOM, "scheduling buffers: " + why);
    }
    *out_ctx = ctx;
    return BLOK_OK;
}

int blok_hip_resize(blok_hip_ctx* ctx, uint32_t width, uint32_t height) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!width || !height) return set_error(ctx, BLOK_ERR_INVALID_ARG, "zero-sized frame");
    ctx->width = width; ctx->height = height;     // the accumulation buffer is re-created on the next progressive frame
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = order_buffers(ctx, 0u, nullptr);      // grown here (a blocking entry), not by the next launch
    if (rc != BLOK_OK) return rc;
    BLOK_HIP_TRY(ctx, hipDeviceSynchronize());
    return BLOK_OK;
}

static void free_stream_scratch(blok_hip_ctx::StreamScratch& sc) {
    for (void* p : {static_cast<void*>(sc.beam), static_cast<void*>(sc.ctl), static_cast<void*>(sc.entries), static_cast<void*>(sc.tile_map),
                    static_cast<void*>(sc.slots), static_cast<void*>(sc.gave_up), static_cast<void*>(sc.list_entries), static_cast<void*>(sc.list_ctl), sc.tail_pool})
        if (p) (void)hipFree(p);
    if (sc.list_hint) (void)hipHostFree(sc.list_hint);
    sc = blok_hip_ctx::StreamScratch{};
}

int blok_hip_release_stream(blok_hip_ctx* ctx, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the stream's launches may still be using its scratch, and another stream's pending sort may be waiting for its marker
    BLOK_HIP_TRY(ctx, hipDeviceSynchronize());
    auto it = ctx->beam_buffers.find(stream);
    if (it != ctx->beam_buffers.end()) { free_stream_scratch(it->second); ctx->beam_buffers.erase(it); }
    forget_device_activity(ctx, true, stream);
    return BLOK_OK;
}

void blok_hip_destroy(blok_hip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    free_world(ctx);
    if (ctx->d_frame) (void)hipFree(ctx->d_frame);
    free_post(ctx);
    if (ctx->has_volume) blok::gpu_volume_destroy(&ctx->volume);
    forget_device_activity(ctx);
    for (auto& kv : ctx->beam_buffers) free_stream_scratch(kv.second);
    if (ctx->d_list_cost) (void)hipFree(ctx->d_list_cost);
    free_order(ctx);
    if (ctx->order.h_live) (void)hipHostFree(ctx->order.h_live);
    if (ctx->order.h_depth) (void)hipHostFree(ctx->order.h_depth);
    if (ctx->order.h_fallback) (void)hipHostFree(ctx->order.h_fallback);
    if (ctx->order.d_fallback) (void)hipFree(ctx->order.d_fallback);
    if (ctx->order.done) (void)hipEventDestroy(ctx->order.done);
    if (ctx->d_accum) (void)hipFree(ctx->d_accum);
    if (ctx->d_color) (void)hipFree(ctx->d_color);
    if (ctx->sun_event) (void)hipEventDestroy(ctx->sun_event);
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
    // ChunkManager(chunkSize, voxelSize) with voxelSize != 1 (reference chunk_manager.cpp:19-25, 234-314): the descriptors' bounds
    // are voxel coordinates times voxelSize.  For a power of two the division is exact, the structure is built in voxel units
    // as for voxelSize 1, and the walk multiplies its integer planes by voxelSize (trace_core.h).
    const float vs = ctx->voxel_size;
    std::vector<blok_sub_chunk> scaled;
    if (vs != 1.0f) {
        scaled.assign(sub_chunks, sub_chunks + n_sub_chunks);
        const float inv = 1.0f / vs;
        for (auto& sc : scaled) {
            for (int a = 0; a < 3; ++a) { sc.world_min[a] *= inv; sc.world_max[a] *= inv; }
            sc.sub_chunk_size *= inv;
        }
        sub_chunks = scaled.data();
    }
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
            ctx->has_world = true; ctx->world_version += 1u;
            ctx->built_on_device = true;
            ctx->world_voxel_size = vs;
            return rebuild_sun_map(ctx);
        }
    }
    std::vector<blok::VoxelRec> voxels;
    const char* why = "";
    if (!blok::extract_voxels(nodes, n_nodes, sub_chunks, n_sub_chunks, voxels, &why))
        return set_error(ctx, BLOK_ERR_UNSUPPORTED, why);
    blok::HostTree tree;
    if (!blok::build_tree(voxels, tree, &why)) return set_error(ctx, BLOK_ERR_UNSUPPORTED, why);
    ctx->pending_voxel_size = vs;                      // install_tree frees the old world (voxel size back to 1), then takes this one
    const int rc = install_tree(ctx, tree, materials, n_materials);
    ctx->pending_voxel_size = 1.0f;
    if (rc != BLOK_OK) return rc;
    ctx->stats.n_ref_nodes = n_nodes;
    ctx->stats.n_sub_chunks = n_sub_chunks;
    return BLOK_OK;
}

int blok_hip_set_voxel_size(blok_hip_ctx* ctx, float voxel_size) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    int e = 0;
    const float m = std::frexp(voxel_size, &e);
    if (!(voxel_size > 0.0f) || m != 0.5f || e < -7 || e > 9)        // 2^-8 .. 2^8
        return set_error(ctx, BLOK_ERR_UNSUPPORTED, "voxel size must be a power of two in [1/256, 256]: any other size puts box planes off the exactly representable lattice the bit-exact walk relies on");
    ctx->voxel_size = voxel_size;
    return BLOK_OK;
}

// Keeps the id grid itself on the device for the dense-grid kernel: 8^3-cell tiles + one occupancy bit per tile.
static int keep_dense_grid(blok_hip_ctx* ctx, const uint32_t* ids, uint32_t nx, uint32_t ny, uint32_t nz, const int32_t o[3]) {
    const uint32_t tx = (nx + 7u) / 8u, ty = (ny + 7u) / 8u, tz = (nz + 7u) / 8u;
    const uint64_t tiles = static_cast<uint64_t>(tx) * ty * tz;
    if (tiles > 0x7FFFFFFFull / 512u * 8u) return set_error(ctx, BLOK_ERR_UNSUPPORTED, "dense grid too large for the dense-grid path");
    const size_t cells = static_cast<size_t>(nx) * ny * nz, words = (tiles + 31u) / 32u;
    uint32_t* d_ids = nullptr;
    BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_ids), cells * sizeof(uint32_t)));
    hipError_t e = hipMemcpy(d_ids, ids, cells * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_dense_tiled), tiles * 512u * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_dense_bits), words * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(ctx->d_dense_bits, 0, words * sizeof(uint32_t));
    if (e == hipSuccess) {
        blok::launch_dense_tile(d_ids, nx, ny, nz, tx, ty, tz, ctx->d_dense_tiled, ctx->d_dense_bits, nullptr);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    (void)hipFree(d_ids);
    if (e != hipSuccess) return set_error(ctx, e == hipErrorOutOfMemory ? BLOK_ERR_OOM : BLOK_ERR_HIP, std::string("dense grid: ") + hipGetErrorString(e));
    ctx->dense_tiles[0] = tx; ctx->dense_tiles[1] = ty; ctx->dense_tiles[2] = tz;
    for (int a = 0; a < 3; ++a) ctx->dense_origin[a] = o[a];
    ctx->has_dense = true;
    return BLOK_OK;
}

int blok_hip_set_dense_dda(blok_hip_ctx* ctx, int enabled) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->dense_dda = enabled != 0;
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
            ctx->has_world = true; ctx->world_version += 1u;
            ctx->built_on_device = true;
            const int rc_sun = rebuild_sun_map(ctx);
            if (rc_sun != BLOK_OK || !ctx->dense_dda) return rc_sun;
            return keep_dense_grid(ctx, ids, nx, ny, nz, o);
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
    const int rc_tree = install_tree(ctx, tree, materials, n_materials);
    if (rc_tree != BLOK_OK || !ctx->dense_dda) return rc_tree;
    return keep_dense_grid(ctx, ids, nx, ny, nz, o);
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
    if ((!out_hits_dev && !out_rgba_dev) || !rect_inside(ctx, x0, y0, w, h))
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "rectangle outside the frame or no output");
    blok::TraceArgs a = base_args(ctx, cam);
    a.x0 = x0; a.y0 = y0; a.w = w; a.h = h;
    a.out = static_cast<blok_hit*>(out_hits_dev);
    a.out_rgba = static_cast<uint32_t*>(out_rgba_dev);
    if (ctx->dense_dda && ctx->has_dense) {               // the dense-grid kernel over the uploaded grid itself (dense_kernels.hip)
        blok::DenseArgs d{};
        d.trace = a;
        for (int k = 0; k < 3; ++k) d.trace.origin[k] = ctx->dense_origin[k];
        d.tiled = ctx->d_dense_tiled; d.tile_bits = ctx->d_dense_bits;
        d.tx = ctx->dense_tiles[0]; d.ty = ctx->dense_tiles[1]; d.tz = ctx->dense_tiles[2];
        d.bit_words = static_cast<uint32_t>((static_cast<uint64_t>(d.tx) * d.ty * d.tz + 31u) / 32u);
        hipStream_t stream = static_cast<hipStream_t>(hip_stream);
        if (ctx->timing) BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, stream));
        blok::launch_dense(d, stream);
        BLOK_HIP_TRY(ctx, hipGetLastError());
        if (ctx->timing) { BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_end, stream)); ctx->timed = true; }
        return BLOK_OK;
    }
    const uint32_t blocks = blok::rect_grid_blocks(w, h);
    return launch_timed(ctx, blok::RayMode::Rect, a, blocks, static_cast<hipStream_t>(hip_stream));
}

int blok_hip_trace_primary(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0,
                           uint32_t w, uint32_t h, blok_hit* out_hits_host) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!out_hits_host) return set_error(ctx, BLOK_ERR_INVALID_ARG, "null output");
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    if (!rect_inside(ctx, x0, y0, w, h)) return set_error(ctx, BLOK_ERR_INVALID_ARG, "rectangle outside the frame");   // before anything is sized by it
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

// Shared by the one-frame and the several-frame entry: n_frames cameras, one beam + trace launch pair.
static int trace_tile_frames(blok_hip_ctx* ctx, const blok_camera* cams, uint32_t n_frames, uint32_t tile, uint32_t rank, uint32_t n_ranks,
                             uint32_t frame_stride_tiles, void* out_hits_dev, void* out_rgba_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!cams || !n_frames || n_frames > blok::kMaxTileFrames) return set_error(ctx, BLOK_ERR_INVALID_ARG, "1 to 8 cameras per launch");
    for (uint32_t f = 0; f < n_frames; ++f) { const int rc = check_trace(ctx, cams + f); if (rc != BLOK_OK) return rc; }
    if ((!out_hits_dev && !out_rgba_dev) || tile < 16 || (tile % blok::kTileW) || (tile % blok::kTileH) || (tile & 15u) || !n_ranks || rank >= n_ranks)
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "tile must be a multiple of 16 and rank < n_ranks");
    const uint32_t mine = blok_hip_tiles_for_rank(ctx->width, ctx->height, tile, rank, n_ranks);
    if (n_frames > 1 && frame_stride_tiles < mine) return set_error(ctx, BLOK_ERR_INVALID_ARG, "frame stride is smaller than the rank's tile count");
    blok::TraceArgs a = base_args(ctx, cams);
    a.tile = tile; a.rank = rank; a.n_ranks = n_ranks;
    a.tiles_x = (ctx->width + tile - 1) / tile;
    a.tiles_total = a.tiles_x * ((ctx->height + tile - 1) / tile);
    a.out = static_cast<blok_hit*>(out_hits_dev);
    a.out_rgba = static_cast<uint32_t*>(out_rgba_dev);
    const uint32_t blocks = mine * (tile / blok::kTileW) * (tile / blok::kTileH);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (n_frames == 1) return launch_timed(ctx, blok::RayMode::Tiles, a, blocks, stream, mine);
    if (!blocks) return BLOK_OK;
    blok::TileFrames frames{};
    for (uint32_t f = 0; f < n_frames; ++f) frames.cam[f] = cams[f];
    frames.n_frames = n_frames;
    frames.frame_stride = static_cast<size_t>(frame_stride_tiles) * tile * tile;
    return launch_timed(ctx, blok::RayMode::Tiles, a, blocks, stream, mine, &frames);      // blocks / beam tiles per frame are filled in there
}

int blok_hip_trace_tiles_device(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t tile, uint32_t rank,
                                uint32_t n_ranks, void* out_hits_dev, void* out_rgba_dev, void* hip_stream) {
    return trace_tile_frames(ctx, cam, 1, tile, rank, n_ranks, 0, out_hits_dev, out_rgba_dev, hip_stream);
}

int blok_hip_trace_tile_frames_device(blok_hip_ctx* ctx, const blok_camera* cams, uint32_t n_frames, uint32_t tile, uint32_t rank,
                                      uint32_t n_ranks, uint32_t frame_stride_tiles, void* out_hits_dev, void* out_rgba_dev, void* hip_stream) {
    return trace_tile_frames(ctx, cams, n_frames, tile, rank, n_ranks, frame_stride_tiles, out_hits_dev, out_rgba_dev, hip_stream);
}

int blok_hip_untile_frames_device(blok_hip_ctx* ctx, const void* gathered_dev, uint32_t elem_bytes, uint32_t tile, uint32_t n_ranks,
                                  uint32_t tiles_per_rank_max, uint32_t n_frames, uint32_t frame_stride_tiles, void* out_frames_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!gathered_dev || !out_frames_dev || !tile || !n_ranks || !tiles_per_rank_max || !n_frames || (elem_bytes != 16 && elem_bytes != 4))
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad untile arguments (elem_bytes must be 16 or 4)");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    blok::UntileArgs u{};
    u.gathered = gathered_dev;
    u.frame = out_frames_dev;
    u.elem_bytes = elem_bytes;
    u.frame_w = ctx->width; u.frame_h = ctx->height; u.tile = tile; u.n_ranks = n_ranks;
    u.tiles_per_rank_max = tiles_per_rank_max;
    u.tiles_x = (ctx->width + tile - 1) / tile;
    u.gathered_frame_stride = static_cast<size_t>(frame_stride_tiles) * tile * tile;
    blok::launch_untile(u, n_frames, static_cast<hipStream_t>(hip_stream));
    BLOK_HIP_TRY(ctx, hipGetLastError());
    return BLOK_OK;
}

int blok_hip_untile_device(blok_hip_ctx* ctx, const void* gathered_dev, uint32_t elem_bytes, uint32_t tile,
                           uint32_t n_ranks, uint32_t tiles_per_rank_max, void* out_frame_dev, void* hip_stream) {
    return blok_hip_untile_frames_device(ctx, gathered_dev, elem_bytes, tile, n_ranks, tiles_per_rank_max, 1, 0, out_frame_dev, hip_stream);
}

size_t blok_hip_compact_words(uint32_t tile, uint32_t n_tiles) { return 1u + static_cast<size_t>(n_tiles) * (1u + static_cast<size_t>(tile) * tile); }

int blok_hip_compact_tile_frames_device(blok_hip_ctx* ctx, const void* rgba_tiles_dev, uint32_t tile, uint32_t n_tiles, uint32_t n_frames,
                                        uint32_t frame_stride_tiles, void* out_words_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!rgba_tiles_dev || !out_words_dev || !tile || !n_frames || (n_frames > 1 && frame_stride_tiles < n_tiles))
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad compact arguments");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    BLOK_HIP_TRY(ctx, hipMemsetAsync(out_words_dev, 0, n_frames * sizeof(uint32_t), stream));          // the count words
    blok::CompactArgs a{static_cast<const uint32_t*>(rgba_tiles_dev), static_cast<uint32_t*>(out_words_dev), tile, n_tiles, n_frames,
                        static_cast<size_t>(frame_stride_tiles) * tile * tile};
    blok::launch_compact_tiles(a, stream);
    BLOK_HIP_TRY(ctx, hipGetLastError());
    return BLOK_OK;
}

int blok_hip_compact_tiles_device(blok_hip_ctx* ctx, const void* rgba_tiles_dev, uint32_t tile, uint32_t n_tiles,
                                  void* out_words_dev, void* hip_stream) {
    return blok_hip_compact_tile_frames_device(ctx, rgba_tiles_dev, tile, n_tiles, 1, 0, out_words_dev, hip_stream);
}

}   // extern "C"
namespace blok_api {
int scatter_frames(blok_hip_ctx* ctx, bool codes, const void* gathered_dev, const void* const* rank_ptrs_dev, uint32_t n_ranks, size_t rank_stride_words,
                   uint32_t tile, uint32_t max_records, uint32_t n_frames, void* out_frames_rgba_dev, void* tile_state_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if ((!gathered_dev && !rank_ptrs_dev) || !out_frames_rgba_dev || !tile || (tile & 1u) || !n_ranks || !n_frames) return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad scatter arguments");
    if (codes && !blok_hip_exchange_code_bits(ctx)) return set_error(ctx, BLOK_ERR_UNSUPPORTED, "the material table is too large for 16-bit pixel codes");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    blok::ScatterArgs a{};
    a.gathered = static_cast<const uint32_t*>(gathered_dev); a.frame = static_cast<uint32_t*>(out_frames_rgba_dev);
    a.rank_ptrs = reinterpret_cast<const uint32_t* const*>(rank_ptrs_dev);
    a.tile_state = static_cast<uint8_t*>(tile_state_dev);
    a.frame_w = ctx->width; a.frame_h = ctx->height; a.tile = tile; a.n_ranks = n_ranks;
    a.tiles_x = (ctx->width + tile - 1) / tile; a.tiles_total = a.tiles_x * ((ctx->height + tile - 1) / tile);
    a.max_records = max_records; a.n_frames = n_frames; a.rank_stride = rank_stride_words;
    a.record_words = 1u + (codes ? tile * tile / 2u : tile * tile);
    a.mat_table = ctx->d_materials; a.n_materials = static_cast<uint32_t>(ctx->n_materials);
    // the stream's tile map: zero when a launch begins and when it ends (the tile kernel puts back what the map kernel set)
    auto& slot = ctx->beam_buffers[stream];
    const size_t need = static_cast<size_t>(a.tiles_total) * n_frames;
    if (slot.n_tile_map < need) {
        if (slot.tile_map) { BLOK_HIP_TRY(ctx, hipStreamSynchronize(stream)); (void)hipFree(slot.tile_map); }
        slot.tile_map = nullptr; slot.n_tile_map = 0;
        BLOK_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&slot.tile_map), need * sizeof(uint32_t)));
        BLOK_HIP_TRY(ctx, hipMemsetAsync(slot.tile_map, 0, need * sizeof(uint32_t), stream));
        slot.n_tile_map = need;
    }
    a.tile_map = slot.tile_map;
    blok::launch_scatter_tiles(a, codes, stream);
    BLOK_HIP_TRY(ctx, hipGetLastError());
    return BLOK_OK;
}
}   // namespace blok_api
extern "C" {

int blok_hip_scatter_tile_frames_device(blok_hip_ctx* ctx, const void* gathered_dev, uint32_t n_ranks, size_t rank_stride_words, uint32_t tile,
                                        uint32_t max_records, uint32_t n_frames, void* out_frames_rgba_dev, void* tile_state_dev, void* hip_stream) {
    return scatter_frames(ctx, false, gathered_dev, nullptr, n_ranks, rank_stride_words, tile, max_records, n_frames, out_frames_rgba_dev, tile_state_dev, hip_stream);
}

uint32_t blok_hip_exchange_code_bits(const blok_hip_ctx* ctx) {
    return ctx && ctx->has_world && ctx->d_materials && (ctx->n_materials + 1u) * 8u <= 0xFFFFu ? 16u : 0u;
}

size_t blok_hip_compact_code_words(uint32_t tile, uint32_t n_tiles) { return 1u + static_cast<size_t>(n_tiles) * (1u + static_cast<size_t>(tile) * tile / 2u); }

int blok_hip_compact_hit_tile_frames_device(blok_hip_ctx* ctx, const void* hit_tiles_dev, uint32_t tile, uint32_t n_tiles, uint32_t n_frames,
                                            uint32_t frame_stride_tiles, void* out_words_dev, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!hit_tiles_dev || !out_words_dev || !tile || (tile & 1u) || !n_frames || (n_frames > 1 && frame_stride_tiles < n_tiles))
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad compact arguments");
    if (!blok_hip_exchange_code_bits(ctx)) return set_error(ctx, BLOK_ERR_UNSUPPORTED, "the material table is too large for 16-bit pixel codes");
    BLOK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    BLOK_HIP_TRY(ctx, hipMemsetAsync(out_words_dev, 0, n_frames * sizeof(uint32_t), stream));          // the count words
    blok::CompactHitArgs a{static_cast<const blok_hit*>(hit_tiles_dev), static_cast<uint32_t*>(out_words_dev), tile, n_tiles, n_frames,
                           static_cast<uint32_t>(ctx->n_materials), static_cast<size_t>(frame_stride_tiles) * tile * tile};
    blok::launch_compact_hit_tiles(a, stream);
    BLOK_HIP_TRY(ctx, hipGetLastError());
    return BLOK_OK;
}

int blok_hip_scatter_code_tile_frames_device(blok_hip_ctx* ctx, const void* gathered_dev, uint32_t n_ranks, size_t rank_stride_words, uint32_t tile,
                                             uint32_t max_records, uint32_t n_frames, void* out_frames_rgba_dev, void* tile_state_dev, void* hip_stream) {
    return scatter_frames(ctx, true, gathered_dev, nullptr, n_ranks, rank_stride_words, tile, max_records, n_frames, out_frames_rgba_dev, tile_state_dev, hip_stream);
}

int blok_hip_scatter_tiles_device(blok_hip_ctx* ctx, const void* gathered_dev, uint32_t n_ranks, size_t rank_stride_words,
                                  uint32_t tile, uint32_t max_records, void* out_frame_rgba_dev, void* hip_stream) {
    return blok_hip_scatter_tile_frames_device(ctx, gathered_dev, n_ranks, rank_stride_words, tile, max_records, 1, out_frame_rgba_dev, nullptr, hip_stream);
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
    if (!rect_inside(ctx, x0, y0, w, h)) return set_error(ctx, BLOK_ERR_INVALID_ARG, "rectangle outside the frame");
    const size_t n = static_cast<size_t>(w) * h;
    rc = ensure_frame(ctx, (n + 3) / 4);                       // n RGBA8 pixels fit in n/4 16-byte records
    if (rc != BLOK_OK) return rc;
    rc = blok_hip_trace_primary_device(ctx, cam, x0, y0, w, h, nullptr, ctx->d_frame, nullptr);
    if (rc != BLOK_OK) return rc;
    BLOK_HIP_TRY(ctx, hipMemcpy(out_rgba8_host, ctx->d_frame, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return BLOK_OK;
}

// Launch of the path kernel (pre-pass first) for either plane set; `planes` carries the output pointers and prev_view_proj.
static int launch_path_frame(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t spp,
                             uint32_t max_bounces, uint32_t frame_index, const blok::PathArgs& planes, void* hip_stream) {
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    if (!rect_inside(ctx, x0, y0, w, h) || !spp || !max_bounces)
        return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad path-trace arguments");
    if (!ctx->n_materials) return set_error(ctx, BLOK_ERR_INVALID_ARG, "path tracing needs a material table");
    blok::PathArgs p = planes;
    p.trace = base_args(ctx, cam);
    p.trace.x0 = x0; p.trace.y0 = y0; p.trace.w = w; p.trace.h = h;
    p.spp = spp; p.max_bounces = max_bounces; p.frame_count = frame_index;
    p.batch_kinds = ctx->ray_batching;
    p.resume_secondary = ctx->path_resume ? 1u : 0u;
    p.fine_beam = ctx->path_fine_beam ? 1u : 0u;
    if (ctx->sun_map_enabled && ctx->has_sun_map) {         // shadow rays stop at the last occluder of their column
        const blok::SunMapArgs& m = ctx->sun;
        p.sun_map = ctx->d_sun_map;
        for (int a = 0; a < 3; ++a) { p.sun_u[a] = m.u[a]; p.sun_v[a] = m.v[a]; }
        p.sun_u0 = m.u0; p.sun_v0 = m.v0; p.sun_inv_texel = 1.0f / m.texel; p.sun_nu = m.nu; p.sun_nv = m.nv;
    }
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (ctx->sun_event_pending && p.sun_map) {              // a patch of the map enqueued behind the latest rebuild (update_sun_map)
        if (hipEventQuery(ctx->sun_event) == hipSuccess) ctx->sun_event_pending = false;
        else { (void)hipGetLastError(); BLOK_HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->sun_event, 0)); }
    }
    uint32_t n_beams = 0;                                   // the primary rays of every sample start behind the beam pre-pass
    rc = prepare_beam(ctx, blok::RayMode::Rect, p.trace, stream, 0, &n_beams);
    if (rc != BLOK_OK) return rc;
    const uint32_t path_blocks = blok::rect_grid_blocks(w, h);
    // the bounce rounds' tail pool (path_core.h: shade_pixel): with the rounds taken sample by sample, two bounces or more, rays from the root
    p.tail_pool = nullptr;
    // (from four samples per pixel on: with fewer a wave's pool never fills, and the rays parked wait for the end: 1 spp 0.82 -> 0.93 ms, 4 spp 2.78 -> 2.54)
    if (ctx->ray_batching >= 3u && max_bounces >= 2u && spp >= 4u && !ctx->path_resume && blok::kBlock == 64) {
        const size_t bytes = static_cast<size_t>(path_blocks) * blok::kTailCapacity * sizeof(blok::TailRecord);
        auto& scratch = ctx->beam_buffers[stream];             // per launch stream: launches on two streams may be in flight together
        if (bytes > scratch.tail_pool_bytes) {
            if (scratch.tail_pool) { BLOK_HIP_TRY(ctx, hipStreamSynchronize(stream)); (void)hipFree(scratch.tail_pool); scratch.tail_pool = nullptr; scratch.tail_pool_bytes = 0; }
            BLOK_HIP_TRY(ctx, hipMalloc(&scratch.tail_pool, bytes));
            scratch.tail_pool_bytes = bytes;
        }
        p.tail_pool = static_cast<blok::TailRecord*>(scratch.tail_pool);
        p.tail_cap = ctx->tail_cap; p.tail_cap_parked = ctx->tail_cap_parked;
    }
    if (ctx->timing) BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, stream));
    if (n_beams) blok::launch_beam(blok::RayMode::Rect, p.trace, n_beams, stream);
    blok::launch_paths(p, path_blocks, stream);
    BLOK_HIP_TRY(ctx, hipGetLastError());
    if (ctx->timing) { BLOK_HIP_TRY(ctx, hipEventRecord(ctx->ev_end, stream)); ctx->timed = true; }
    return BLOK_OK;
}

int blok_hip_trace_paths_device(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0, uint32_t w,
                                uint32_t h, uint32_t spp, uint32_t max_bounces, uint32_t frame_index,
                                const blok_gbuffer* planes, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!planes) return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad path-trace arguments");
    blok::PathArgs p{};
    p.color = planes->color; p.world_pos = planes->world_pos;
    p.normal_roughness = planes->normal_roughness; p.albedo_metallic = planes->albedo_metallic;
    return launch_path_frame(ctx, cam, x0, y0, w, h, spp, max_bounces, frame_index, p, hip_stream);
}

int blok_hip_trace_paths_ref_device(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0, uint32_t w,
                                    uint32_t h, uint32_t spp, uint32_t max_bounces, uint32_t frame_index,
                                    const float prev_view_proj[16], const blok_gbuffer_ref* planes, void* hip_stream) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!planes || (planes->motion && !prev_view_proj)) return set_error(ctx, BLOK_ERR_INVALID_ARG, "bad path-trace arguments (a motion plane needs prevViewProj)");
    blok::PathArgs p{};
    p.color = planes->color; p.world_pos = planes->world_pos;
    p.normal_roughness_h = planes->normal_roughness; p.albedo_metallic_u8 = planes->albedo_metallic; p.motion_h = planes->motion;
    if (prev_view_proj) for (int k = 0; k < 16; ++k) p.prev_view_proj[k] = prev_view_proj[k];
    return launch_path_frame(ctx, cam, x0, y0, w, h, spp, max_bounces, frame_index, p, hip_stream);
}

int blok_hip_trace_paths(blok_hip_ctx* ctx, const blok_camera* cam, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                         uint32_t spp, uint32_t max_bounces, uint32_t frame_index, const blok_gbuffer* planes_host) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    if (!planes_host) return set_error(ctx, BLOK_ERR_INVALID_ARG, "null planes");
    int rc = check_trace(ctx, cam);
    if (rc != BLOK_OK) return rc;
    if (!rect_inside(ctx, x0, y0, w, h)) return set_error(ctx, BLOK_ERR_INVALID_ARG, "rectangle outside the frame");
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

int blok_hip_set_taa_jitter(blok_hip_ctx* ctx, const float jitter_px[2]) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    const float jx = jitter_px ? jitter_px[0] : 0.0f, jy = jitter_px ? jitter_px[1] : 0.0f;
    // the beam pre-pass grows a tile's frustum by one pixel; the path kernel's own sub-pixel jitter takes +-0.25 of it
    if (!(std::fabs(jx) <= 0.5f) || !(std::fabs(jy) <= 0.5f)) return set_error(ctx, BLOK_ERR_INVALID_ARG, "jitter must lie within +-0.5 pixel");
    ctx->jitter_px[0] = jx; ctx->jitter_px[1] = jy;
    return BLOK_OK;
}

int blok_hip_set_rt_taa_jitter(blok_hip_ctx* ctx, int enabled) {
    if (!ctx) return BLOK_ERR_INVALID_ARG;
    ctx->rt_taa_jitter = enabled != 0;
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
