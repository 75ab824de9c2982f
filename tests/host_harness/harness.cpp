// TEST INFRASTRUCTURE: runs the kernel's per-ray body on the CPU (one "lane" at a time) over a tree
// built by the product's host tree builder.  Never linked into the shipped libraries.
#define BLOK_TRACE_HOST_HARNESS 1
#include <cstdint>
// statistics build: events 0 iteration, 1 descend, 2 step, 3 ascend, per level
static thread_local uint64_t g_stat[8][8];      // [event][level]: 0 iteration, 1 descend, 2 step, 3 ascent, 4 walk begins, 5 walk resumed, 6 launch-pad level
static thread_local unsigned char* g_seq = nullptr;      // optional per-ray event log: 1 descend, 2 step (+level*4)
static thread_local uint32_t g_seq_len = 0, g_seq_cap = 0;
static thread_local uint32_t g_kind = 0;                 // kind of the ray being walked (path loop): 0 primary, 1 shadow, 2 bounce
static thread_local uint32_t g_resume = 0;                // PathArgs::resume_secondary of the path entries below (hh_set_path_resume)
#define BLOK_PATH_KIND(kind) do { g_kind = (kind); } while (0)
static thread_local uint64_t g_kstat[3][8];     // [kind of ray][event], path entries
#define BLOK_STAT(event, level) do { ++g_stat[event][level]; ++g_kstat[g_kind < 3 ? g_kind : 0][event]; \
    if (g_seq && (event == 1 || event == 2 || event == 4 || event == 6) && g_seq_len < g_seq_cap) g_seq[g_seq_len++] = (unsigned char)((event == 6 ? 3 : event) | ((level) << 2) | (event == 4 ? g_kind << 3 : 0u)); } while (0)
#include "trace_core.h"
#include "path_core.h"
#include "post_core.h"
#include "reference_world.h"

#include <vector>

using namespace blok;

namespace {
struct Harness {
    HostTree tree;
    std::vector<uint4> nodes;
};
}

extern "C" {

void* hh_build(const blok_svo_node* nodes, size_t n_nodes, const blok_sub_chunk* subs, size_t n_subs, const char** why) {
    static const char* none = "";
    *why = none;
    std::vector<VoxelRec> voxels;
    if (!extract_voxels(nodes, n_nodes, subs, n_subs, voxels, why)) return nullptr;
    auto* h = new Harness();
    if (!build_tree(voxels, h->tree, why)) { delete h; return nullptr; }
    h->nodes.resize(h->tree.nodes.size());
    std::memcpy(h->nodes.data(), h->tree.nodes.data(), h->nodes.size() * sizeof(uint4));
    return h;
}
void hh_free(void* h) { delete static_cast<Harness*>(h); }
const void* hh_tree_nodes(const void* h, size_t* count, int32_t* origin) {
    const Harness* H = static_cast<const Harness*>(h);
    *count = H->nodes.size();
    for (int i = 0; i < 3; ++i) origin[i] = H->tree.origin[i];
    return H->nodes.data();
}
uint32_t hh_levels(const void* h) { return static_cast<const Harness*>(h)->tree.levels; }
uint64_t hh_voxels(const void* h) { return static_cast<const Harness*>(h)->tree.n_voxels; }

static float g_jitter_clip[2] = {0.0f, 0.0f};
void hh_set_jitter_clip(float jx, float jy) { g_jitter_clip[0] = jx; g_jitter_clip[1] = jy; }

static TraceArgs make_args(const Harness* H) {
    TraceArgs a{};
    a.jitter_clip[0] = g_jitter_clip[0]; a.jitter_clip[1] = g_jitter_clip[1];
    a.voxel_size = 1.0f; a.inv_voxel_size = 1.0f;
    a.nodes = H->nodes.data();
    a.materials = H->tree.materials.data();
    for (int i = 0; i < 3; ++i) a.origin[i] = H->tree.origin[i];
    a.levels = H->tree.levels;
    a.tmin = BLOK_RAY_TMIN; a.tmax = BLOK_RAY_TMAX;
    return a;
}

void hh_trace_rays(const void* h, const blok_ray* rays, size_t n, blok_hit* out) {
    const Harness* H = static_cast<const Harness*>(h);
    TraceArgs a = make_args(H);
    std::vector<uint4> stack(size_t(kMaxLevels) * 2 * kBlock);
    for (size_t i = 0; i < n; ++i) {
        RayIn r{rays[i].org[0], rays[i].org[1], rays[i].org[2], rays[i].dir[0], rays[i].dir[1], rays[i].dir[2],
                rays[i].tmin, rays[i].tmax};
        trace_one(a, r, stack.data(), Sink{out + i, nullptr});
    }
}

// raygen.rgen loop of the kernel, on the CPU: four float4 planes of the full frame.
void hh_render_paths(const void* h, const blok_camera* cam, const blok_material* materials, uint32_t n_materials,
                     uint32_t width, uint32_t height, uint32_t spp, uint32_t max_bounces, uint32_t frame_index,
                     float* color, float* world_pos, float* normal_roughness, float* albedo_metallic) {
    const Harness* H = static_cast<const Harness*>(h);
    PathArgs p{};
    p.trace = make_args(H);
    p.trace.cam = *cam; p.trace.frame_w = width; p.trace.frame_h = height;
    p.trace.mat_table = materials; p.trace.n_materials = n_materials;
    p.resume_secondary = g_resume;
    p.spp = spp; p.max_bounces = max_bounces; p.frame_count = frame_index;
    p.color = color; p.world_pos = world_pos; p.normal_roughness = normal_roughness; p.albedo_metallic = albedo_metallic;
    std::vector<uint4> stack(size_t(kMaxLevels) * 2 * kBlock);
    std::vector<uint2> keep_lohi(size_t(kMaxLevels) * kBlock); std::vector<uint32_t> keep_base(size_t(kMaxLevels) * kBlock);
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) shade_pixel(p, x, y, size_t(y) * width + x, stack.data(), 0.0f, keep_lohi.data(), keep_base.data());
}

// Event log per ray (cap bytes each, zero padded) for an 8x8-tile wave simulation.
void hh_trace_primary_events(const void* h, const blok_camera* cam, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0,
                             uint32_t w, uint32_t hgt, uint32_t cap, unsigned char* events) {
    const Harness* H = static_cast<const Harness*>(h);
    TraceArgs a = make_args(H);
    a.cam = *cam; a.frame_w = width; a.frame_h = height;
    std::vector<uint4> stack(size_t(kMaxLevels) * 2 * kBlock);
    blok_hit tmp;
    for (uint32_t y = 0; y < hgt; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            g_seq = events + (size_t(y) * w + x) * cap; g_seq_len = 0; g_seq_cap = cap;
            const RayIn r = primary_ray(a, x0 + x, y0 + y);
            trace_one(a, r, stack.data(), Sink{&tmp, nullptr});
        }
    g_seq = nullptr;
}

// Same with a per-pixel start parameter (beam pre-pass experiments).
void hh_trace_rect_events(const void* h, const blok_camera* cam, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0,
                          uint32_t w, uint32_t hgt, const float* tstart, uint32_t cap, unsigned char* events) {
    const Harness* H = static_cast<const Harness*>(h);
    TraceArgs a = make_args(H);
    a.cam = *cam; a.frame_w = width; a.frame_h = height;
    std::vector<uint4> stack(size_t(kMaxLevels) * 2 * kBlock);
    blok_hit tmp;
    for (uint32_t y = 0; y < hgt; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            g_seq = events + (size_t(y) * w + x) * cap; g_seq_len = 0; g_seq_cap = cap;
            RayIn r = primary_ray(a, x0 + x, y0 + y);
            if (tstart) { const float t0 = tstart[size_t(y) * w + x]; if (t0 > 9.0e3f) continue; if (t0 > r.tmin) r.tmin = t0; }
            trace_one(a, r, stack.data(), Sink{&tmp, nullptr});
        }
    g_seq = nullptr;
}

void hh_tonemap(const float* hdr, uint32_t n, float exposure, float saturation_boost, int op, uint32_t* out) {
    TonemapArgs t{hdr, out, n, exposure, saturation_boost, op};
    for (uint32_t i = 0; i < n; ++i) out[i] = tonemap_pixel(t, i);
}

// Event log per PIXEL of the path loop (cap bytes each): 1 descend / 2 step events of every walk, a 3 before each walk.
void hh_render_paths_events(const void* h, const blok_camera* cam, const blok_material* materials, uint32_t n_materials,
                            uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t w, uint32_t hgt,
                            uint32_t spp, uint32_t max_bounces, uint32_t cap, unsigned char* events) {
    const Harness* H = static_cast<const Harness*>(h);
    PathArgs p{};
    p.trace = make_args(H);
    p.trace.cam = *cam; p.trace.frame_w = width; p.trace.frame_h = height;
    p.trace.mat_table = materials; p.trace.n_materials = n_materials;
    p.resume_secondary = g_resume;
    p.spp = spp; p.max_bounces = max_bounces; p.frame_count = 1;
    std::vector<uint4> stack(size_t(kMaxLevels) * 2 * kBlock);
    std::vector<uint2> keep_lohi(size_t(kMaxLevels) * kBlock); std::vector<uint32_t> keep_base(size_t(kMaxLevels) * kBlock);
    for (uint32_t y = 0; y < hgt; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            g_seq = events + (size_t(y) * w + x) * cap; g_seq_len = 0; g_seq_cap = cap;
            shade_pixel(p, x0 + x, y0 + y, 0, stack.data(), 0.0f, keep_lohi.data(), keep_base.data());
        }
    g_seq = nullptr;
}

// As hh_render_paths_events with a per-pixel start parameter for the primary rays (what the beam pre-pass provides) and the kind of
// every walk in its start marker: (byte & 3) == 0 starts a walk, kind = byte >> 3.
void hh_render_paths_events2(const void* h, const blok_camera* cam, const blok_material* materials, uint32_t n_materials,
                             uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t w, uint32_t hgt,
                             uint32_t spp, uint32_t max_bounces, const float* tstart, uint32_t cap, unsigned char* events) {
    const Harness* H = static_cast<const Harness*>(h);
    PathArgs p{};
    p.trace = make_args(H);
    p.trace.cam = *cam; p.trace.frame_w = width; p.trace.frame_h = height;
    p.trace.mat_table = materials; p.trace.n_materials = n_materials;
    p.resume_secondary = g_resume;
    p.spp = spp; p.max_bounces = max_bounces; p.frame_count = 1;
    std::vector<uint4> stack(size_t(kMaxLevels) * 2 * kBlock);
    std::vector<uint2> keep_lohi(size_t(kMaxLevels) * kBlock); std::vector<uint32_t> keep_base(size_t(kMaxLevels) * kBlock);
    for (uint32_t y = 0; y < hgt; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            g_seq = events + (size_t(y) * w + x) * cap; g_seq_len = 0; g_seq_cap = cap;
            shade_pixel(p, x0 + x, y0 + y, 0, stack.data(), tstart ? tstart[size_t(y) * w + x] : 0.0f, keep_lohi.data(), keep_base.data());
        }
    g_seq = nullptr; g_kind = 0;
}

// Per-ray iteration counts for a rectangle of a frame with an optional per-pixel start parameter (beam experiments).
void hh_trace_rect_stats(const void* h, const blok_camera* cam, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0,
                         uint32_t w, uint32_t hgt, const float* tstart, blok_hit* out, uint32_t* iters_per_ray) {
    const Harness* H = static_cast<const Harness*>(h);
    TraceArgs a = make_args(H);
    a.cam = *cam; a.frame_w = width; a.frame_h = height;
    std::vector<uint4> stack(size_t(kMaxLevels) * 2 * kBlock);
    std::memset(g_stat, 0, sizeof(g_stat));
    for (uint32_t y = 0; y < hgt; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            uint64_t before = 0;
            for (int l = 0; l < 8; ++l) before += g_stat[0][l];
            RayIn r = primary_ray(a, x0 + x, y0 + y);
            if (tstart && tstart[size_t(y) * w + x] > r.tmin) r.tmin = tstart[size_t(y) * w + x];
            trace_one(a, r, stack.data(), Sink{out + size_t(y) * w + x, nullptr});
            uint64_t after = 0;
            for (int l = 0; l < 8; ++l) after += g_stat[0][l];
            iters_per_ray[size_t(y) * w + x] = uint32_t(after - before);
        }
}

// As hh_trace_rect_stats with an optional per-pixel far bound as well (tmax).
void hh_trace_rect_stats2(const void* h, const blok_camera* cam, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0,
                          uint32_t w, uint32_t hgt, const float* tstart, const float* tfar, blok_hit* out, uint32_t* iters_per_ray) {
    const Harness* H = static_cast<const Harness*>(h);
    TraceArgs a = make_args(H);
    a.cam = *cam; a.frame_w = width; a.frame_h = height;
    std::vector<uint4> stack(size_t(kMaxLevels) * 2 * kBlock);
    std::memset(g_stat, 0, sizeof(g_stat));
    for (uint32_t y = 0; y < hgt; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            uint64_t before = 0;
            for (int l = 0; l < 8; ++l) before += g_stat[0][l];
            RayIn r = primary_ray(a, x0 + x, y0 + y);
            if (tstart && tstart[size_t(y) * w + x] > r.tmin) r.tmin = tstart[size_t(y) * w + x];
            if (tfar && tfar[size_t(y) * w + x] < r.tmax) r.tmax = tfar[size_t(y) * w + x];
            trace_one(a, r, stack.data(), Sink{out + size_t(y) * w + x, nullptr});
            uint64_t after = 0;
            for (int l = 0; l < 8; ++l) after += g_stat[0][l];
            iters_per_ray[size_t(y) * w + x] = uint32_t(after - before);
        }
}

void hh_stat_totals(uint64_t* totals) { std::memcpy(totals, g_stat, 5 * 8 * sizeof(uint64_t)); }      // events 0-4 (older scripts size their array for five)
void hh_stat_totals8(uint64_t* totals) { std::memcpy(totals, g_stat, sizeof(g_stat)); }
void hh_set_path_resume(uint32_t enabled) { g_resume = enabled; }
void hh_stat_reset() { std::memset(g_stat, 0, sizeof(g_stat)); std::memset(g_kstat, 0, sizeof(g_kstat)); }
void hh_stat_by_kind(uint64_t* totals) { std::memcpy(totals, g_kstat, sizeof(g_kstat)); }

// Per-ray iteration counts for a frame (row-major), plus the event totals [4][8].
void hh_trace_primary_stats(const void* h, const blok_camera* cam, uint32_t width, uint32_t height, blok_hit* out,
                            uint32_t* iters_per_ray, uint64_t* totals) {
    const Harness* H = static_cast<const Harness*>(h);
    TraceArgs a = make_args(H);
    a.cam = *cam; a.frame_w = width; a.frame_h = height;
    std::vector<uint4> stack(size_t(kMaxLevels) * 2 * kBlock);
    std::memset(g_stat, 0, sizeof(g_stat));
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            uint64_t before = 0;
            for (int l = 0; l < 8; ++l) before += g_stat[0][l];
            const RayIn r = primary_ray(a, x, y);
            trace_one(a, r, stack.data(), Sink{out + size_t(y) * width + x, nullptr});
            uint64_t after = 0;
            for (int l = 0; l < 8; ++l) after += g_stat[0][l];
            iters_per_ray[size_t(y) * width + x] = uint32_t(after - before);
        }
    std::memcpy(totals, g_stat, 5 * 8 * sizeof(uint64_t));
}

void hh_trace_primary(const void* h, const blok_camera* cam, uint32_t width, uint32_t height, blok_hit* out) {
    const Harness* H = static_cast<const Harness*>(h);
    TraceArgs a = make_args(H);
    a.cam = *cam; a.frame_w = width; a.frame_h = height;
    std::vector<uint4> stack(size_t(kMaxLevels) * 2 * kBlock);
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            const RayIn r = primary_ray(a, x, y);
            trace_one(a, r, stack.data(), Sink{out + size_t(y) * width + x, nullptr});
        }
}


// ---- image-space chain (post_core.h) on the CPU: the same ping-pong as blok_amd/csrc/hip/api.hip's denoiser -------------
struct HostPost {
    uint32_t w, h;
    std::vector<float> hist_color[2], moments[2], world_pos[2], unit_normals[2], variance, ping, pong, taa_hist[2];
    std::vector<uint16_t> hist_len[2], motion;
    int cur = 0, taa_cur = 0;
};
void* hh_post_new(uint32_t w, uint32_t h) {
    auto* P = new HostPost();
    P->w = w; P->h = h;
    const size_t n = size_t(w) * h;
    for (int k = 0; k < 2; ++k) {
        P->hist_color[k].assign(4 * n, 0.f); P->moments[k].assign(2 * n, 0.f); P->world_pos[k].assign(4 * n, 0.f);
        P->hist_len[k].assign(n, 0); P->unit_normals[k].assign(4 * n, 0.f); P->taa_hist[k].assign(4 * n, 0.f);
    }
    P->variance.assign(n, 0.f); P->ping.assign(4 * n, 0.f); P->pong.assign(4 * n, 0.f); P->motion.assign(2 * n, 0);
    return P;
}
void hh_post_free(void* p) { delete static_cast<HostPost*>(p); }
struct hh_settings { float temporal_alpha, moment_alpha, variance_clip_gamma, depth_threshold, normal_threshold, phi_color, phi_normal, phi_depth;
                     int atrous_iterations; float variance_boost; int min_history_length; };
void hh_post_denoise(void* p, const float* color, const float* world_pos, const float* normal_roughness, const float* motion_in,
                     const float* prev_view_proj, uint32_t frame_count, const hh_settings* S, float* out_color) {
    HostPost& P = *static_cast<HostPost*>(p);
    const int cur = P.cur, prev = cur ^ 1;
    PostFrame f{};
    f.w = P.w; f.h = P.h; f.frame_count = frame_count;
    std::memcpy(&f.s, S, sizeof(DenoiseSettings));
    static_assert(sizeof(DenoiseSettings) == sizeof(hh_settings), "settings layout");
    for (int k = 0; k < 16; ++k) f.prev_view_proj[k] = prev_view_proj[k];
    TemporalArgs t{};
    t.f = f; t.color = color; t.world_pos = world_pos; t.normal_roughness = normal_roughness; t.motion_in = motion_in;
    t.prev_color = P.hist_color[prev].data(); t.prev_moments = P.moments[prev].data(); t.prev_world_pos = P.world_pos[prev].data();
    t.prev_hist_len = P.hist_len[prev].data(); t.prev_unit_normals = P.unit_normals[prev].data();
    t.out_color = P.hist_color[cur].data(); t.out_moments = P.moments[cur].data(); t.hist_world_pos = P.world_pos[cur].data();
    t.out_hist_len = P.hist_len[cur].data(); t.unit_normals = P.unit_normals[cur].data(); t.motion = P.motion.data();
    for (uint32_t y = 0; y < P.h; ++y) for (uint32_t x = 0; x < P.w; ++x) temporal_pixel(t, int(x), int(y));
    VarianceArgs v{};
    v.f = f; v.color = P.hist_color[cur].data(); v.moments = P.moments[cur].data(); v.world_pos = P.world_pos[cur].data();
    v.hist_len = P.hist_len[cur].data(); v.unit_normals = P.unit_normals[cur].data(); v.variance = P.variance.data();
    for (uint32_t y = 0; y < P.h; ++y) for (uint32_t x = 0; x < P.w; ++x) variance_pixel(v, int(x), int(y));
    const float* in = P.hist_color[cur].data();
    for (int it = 0; it < S->atrous_iterations; ++it) {
        AtrousArgs a{};
        a.w = P.w; a.h = P.h; a.step = 1 << it; a.phi_color = S->phi_color; a.phi_depth = S->phi_depth;
        a.color = in; a.variance = P.variance.data(); a.world_pos = P.world_pos[cur].data(); a.unit_normals = P.unit_normals[cur].data();
        a.out = it == S->atrous_iterations - 1 ? out_color : ((it & 1) ? P.pong.data() : P.ping.data());
        for (uint32_t y = 0; y < P.h; ++y) for (uint32_t x = 0; x < P.w; ++x) atrous_pixel(a, int(x), int(y));
        in = a.out;
    }
    if (S->atrous_iterations == 0) std::memcpy(out_color, P.hist_color[cur].data(), P.hist_color[cur].size() * sizeof(float));
    P.cur = prev;
}
void hh_post_state(const void* p, float* history_color, float* moments, float* history_length, float* variance, float* motion) {
    const HostPost& P = *static_cast<const HostPost*>(p);
    const int last = P.cur ^ 1;
    const size_t n = size_t(P.w) * P.h;
    std::memcpy(history_color, P.hist_color[last].data(), 4 * n * sizeof(float));
    std::memcpy(moments, P.moments[last].data(), 2 * n * sizeof(float));
    std::memcpy(variance, P.variance.data(), n * sizeof(float));
    for (size_t i = 0; i < n; ++i) history_length[i] = h2f(P.hist_len[last][i]);
    for (size_t i = 0; i < 2 * n; ++i) motion[i] = h2f(P.motion[i]);
}
void hh_post_taa(void* p, const float* color, const float* motion_in, float feedback_min, float feedback_max, uint32_t frame_count, float* out) {
    HostPost& P = *static_cast<HostPost*>(p);
    if (motion_in) for (size_t i = 0; i < P.motion.size(); ++i) P.motion[i] = f2h(motion_in[i]);
    TaaArgs a{};
    a.w = P.w; a.h = P.h; a.frame_count = frame_count; a.feedback_min = feedback_min; a.feedback_max = feedback_max;
    a.color = color; a.history = P.taa_hist[P.taa_cur ^ 1].data(); a.motion = P.motion.data(); a.out = out; a.out_history = P.taa_hist[P.taa_cur].data();
    for (uint32_t y = 0; y < P.h; ++y) for (uint32_t x = 0; x < P.w; ++x) taa_pixel(a, int(x), int(y));
    P.taa_cur ^= 1;
}
void hh_post_sharpen(const uint32_t* in, uint32_t w, uint32_t h, float strength, uint32_t* out) {
    SharpenArgs a{w, h, strength, in, out};
    float lut[256];
    for (uint32_t t = 0; t < 256; ++t) lut[t] = unorm8_to_float(t);
    for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) sharpen_pixel(a, int(x), int(y), lut);
}
float hh_q16(float x) { return q16(x); }
uint16_t hh_f2h(float x) { return f2h(x); }

}
