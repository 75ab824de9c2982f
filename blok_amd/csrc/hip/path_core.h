// Per-pixel sample / bounce loop on top of walk(): the reference's ray-generation shader
// (assets/shaders/raygen.rgen:77-165 RNG, sampling, sky; :167-414 main) with hit.rchit / miss.rmiss /
// shadow.rmiss folded in.  One lane = one pixel.  Operation order follows the shader statement by statement;
// every float op is a single rounded op (no contraction), sqrt and divide are correctly rounded; sin / cos come
// from the device math library and the shader's pow(x, 5 | 8 | 128) are done by multiplication (both a few ulp
// from any other implementation — the tolerance the parity tests state).  Compiles for the host under BLOK_TRACE_HOST_HARNESS like trace_core.h.
//
// Deliberate, documented differences from the shader:
//  - the primary ray is formed from the camera basis (cuda_tracer.cu:276-282) instead of invProj/invView
//    (raygen.rgen:204-205): glm, which would fix those matrices' arithmetic, is absent from the reference tree;
//  - MAX_BOUNCES (raygen.rgen:211, = 2) and sampleCount (forced to 8 by the host, renderer_denoising.cpp:683)
//    are parameters, as BASELINE.json configs[4] needs 64 spp and deeper paths;
//  - planes are written as float4 (the reference narrows normal+roughness to RGBA16F and albedo+metallic to
//    RGBA8, raygen.rgen:57-58), at the moment the first hit is known rather than after the last sample (same values),
//    and motion vectors (raygen.rgen:409-413) are produced by the temporal pass (post_core.h) from the position plane.
#ifndef BLOK_PATH_CORE_H
#define BLOK_PATH_CORE_H

#include "trace_core.h"
#include "half_bits.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

// Optional hook of the host harness' statistics build: the kind of the ray about to be walked (0 primary, 1 shadow, 2 bounce).
#ifndef BLOK_PATH_KIND
#define BLOK_PATH_KIND(kind)
#endif

namespace blok {

struct PathArgs {
    TraceArgs trace;                 // tree, camera, frame size, rectangle (x0, y0, w, h)
    uint32_t spp, max_bounces, frame_count;
    float* color;                    // float4 per pixel of the rectangle, row-major; any plane may be null
    float* world_pos;
    float* normal_roughness;
    float* albedo_metallic;
    // "last occluder" map for the shadow rays (beam.h: prism_far): per texel of the plane perpendicular to the sun, the largest
    // sun-direction depth at which any voxel exists; null = none.  Texel (iu, iv) covers u0 + iu*texel <= u.p < u0 + (iu+1)*texel.
    const float* sun_map;
    float sun_u[3], sun_v[3];
    float sun_u0, sun_v0, sun_inv_texel;
    uint32_t sun_nu, sun_nv;
    uint32_t batch_kinds;            // 1: the wave walks one kind of ray at a time; 2: and the oldest sample first (see shade_pixel)
    uint32_t fine_beam;              // the path kernel searches a start parameter of its own for every wave tile (trace_kernels.hip: path_kernel)
    uint32_t resume_secondary;       // rays enter the walk from the ancestors of the pixel's latest hit where there is one (trace_core.h: walk_resume); 0 = always from the root
    // The same G-buffer in the reference's own image formats (raygen.rgen:55-59; renderer_denoising.cpp:110-170), each optional:
    // normal + roughness RGBA16F, albedo + metallic RGBA8 (unorm), motion vectors RG16F (raygen.rgen:150-155, 409-413; needs
    // prev_view_proj = FrameUBO::prevViewProj, column-major).  48 B/pixel with the two float4 planes instead of 64.
    uint16_t* normal_roughness_h;
    uint32_t* albedo_metallic_u8;
    uint16_t* motion_h;
    float prev_view_proj[16];
    // The bounce rounds' tail pool (shade_pixel): kTailCapacity records per wave of the launch, in global memory; null = off.  A bounce round
    // stops after tail_cap trips of the walk's loop, a round over parked rays after tail_cap_parked.
    struct TailRecord* tail_pool;
    uint32_t tail_cap, tail_cap_parked;
};

// A last-segment ray cut off in its round: the ray, the parameter to go on from, what its answer is worth to its pixel, and the lane that owns the pixel.
struct TailRecord { float ox, oy, oz, tcur, dx, dy, dz, wx, wy, wz; uint32_t owner, pad; };
struct TailAnswer { float x, y, z; uint32_t owner; };          // a parked ray's term of its pixel's sum, and the lane whose pixel it is (~0: none)
constexpr uint32_t kTailBatch = 64u, kTailCapacity = 128u;
#ifndef BLOK_TRACE_HOST_HARNESS
// a record is written by one lane and read by another later on: read past the vector L1, which may still hold the slot's previous record
__device__ __forceinline__ float tail_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t tail_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

struct V3 { float x, y, z; };
BLOK_DEV V3 v3(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
BLOK_DEV V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
BLOK_DEV V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
BLOK_DEV V3 vmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
BLOK_DEV V3 vscale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
BLOK_DEV V3 vdivs(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
BLOK_DEV V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }
BLOK_DEV float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
BLOK_DEV V3 vnormalize(V3 v) { return vdivs(v, rn_sqrt(vdot(v, v))); }
BLOK_DEV V3 vcross(V3 a, V3 b) { return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
BLOK_DEV V3 vmix(V3 a, V3 b, float t) { return vadd(vscale(a, 1.0f - t), vscale(b, t)); }
BLOK_DEV float max3f(V3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }

constexpr float kPi = 3.14159265359f;        // raygen.rgen:74
constexpr float kInvPi = 0.31830988618f;     // raygen.rgen:75

BLOK_DEV uint32_t pcg(uint32_t& state) {      // raygen.rgen:77-82
    const uint32_t old = state;
    state = old * 747796405u + 2891336453u;
    const uint32_t word = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
    return (word >> 22u) ^ word;
}
BLOK_DEV float random_float(uint32_t& state) { return static_cast<float>(pcg(state)) / 4294967295.0f; }   // :84-86
BLOK_DEV uint32_t init_rng(uint32_t px, uint32_t py, uint32_t width, uint32_t frame, uint32_t sample) {   // :92-99
    uint32_t seed = px + py * width;
    seed ^= frame * 747796405u;
    seed ^= sample * 1664525u;
    pcg(seed);
    pcg(seed);
    return seed;
}
// The shader's tangent frame (:108-110, :125-127: up = |n.z| < 0.999 ? z : x; tangent = normalize(cross(up, n)); bitangent = cross(n, tangent))
// for the normal of a voxel face, n = +-e_k exactly (hit.rchit:69-75): every product in cross(up, n) is 0 or +-1, the vector is a
// signed unit axis, its squared length is exactly 1, sqrt(1) = 1 and x / 1 = x bit for bit (signed zeros included) — the shader's
// normalize() is the identity here, and its square root and three divisions are not spent.
BLOK_DEV void tangent_frame_of_face(V3 n, V3& tangent, V3& bitangent) {
    const V3 up = fabsf(n.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    tangent = vcross(up, n);
    bitangent = vcross(n, tangent);
}
// (sample_cosine_hemisphere :101-113 and sample_ggx :115-130 are written out in shade_pixel, where their common tail runs once per wave.)
// pow(x, k) for the shader's integer exponents 5, 8, 128 by multiplication (<= 4 ulp from a correctly rounded
// pow; GLSL's own pow is exp2(y*log2(x)) at driver precision).  Far inside the stated colour tolerance, and
// ~10x fewer instructions than the library powf on a path where every miss sample evaluates two of them.
#ifdef BLOK_PATH_LIBM_POW
// host harness only: the shader's literal pow(x, k) through libm, to show that nothing BUT these three differs between the kernel body
// and the CPU restatement of the shader on one libm (tests/test_paths.py)
BLOK_DEV float pow5(float x) { return powf(x, 5.0f); }
BLOK_DEV float pow8(float x) { return powf(x, 8.0f); }
BLOK_DEV float pow128(float x) { return powf(x, 128.0f); }
#else
BLOK_DEV float pow5(float x) { const float x2 = x * x; return x2 * x2 * x; }
BLOK_DEV float pow8(float x) { const float x2 = x * x; const float x4 = x2 * x2; return x4 * x4; }
BLOK_DEV float pow128(float x) { float y = pow8(x); y = y * y; y = y * y; y = y * y; return y * y; }
#endif

BLOK_DEV V3 fresnel_schlick(float cos_theta, V3 f0) {                  // :132-134
    const float w = pow5(fmaxf(1.0f - cos_theta, 0.0f));
    return v3(f0.x + (1.0f - f0.x) * w, f0.y + (1.0f - f0.y) * w, f0.z + (1.0f - f0.z) * w);
}
BLOK_DEV V3 sun_direction() { return vnormalize(v3(0.5f, 0.8f, 0.3f)); }   // :142, :185
BLOK_DEV V3 sky_color(V3 dir) {                                         // :136-148
    const float t = 0.5f * (dir.y + 1.0f);
    const V3 sky = vmix(v3(0.8f, 0.85f, 0.95f), v3(0.4f, 0.6f, 0.9f), t);
    const float sun_dot = fmaxf(vdot(dir, sun_direction()), 0.0f);
    const V3 sun = vscale(vscale(v3(1.0f, 0.95f, 0.8f), pow128(sun_dot)), 5.0f);
    const V3 glow = vscale(vscale(v3(1.0f, 0.9f, 0.7f), pow8(sun_dot)), 0.3f);
    return vadd(vadd(sky, sun), glow);
}
BLOK_DEV bool is_emissive(V3 e) { return vdot(e, v3(1.0f, 1.0f, 1.0f)) > 0.01f; }                // :158-160
BLOK_DEV float luminance(V3 c) { return vdot(c, v3(0.2126f, 0.7152f, 0.0722f)); }                // :163-165

BLOK_DEV void store4(float* plane, size_t i, float a, float b, float c, float d) {
    if (!plane) return;
    float* p = plane + 4 * i;
    p[0] = a; p[1] = b; p[2] = c; p[3] = d;
}

BLOK_DEV uint32_t unorm8(float v);
// The narrow planes of one pixel: RGBA16F normal + roughness, RGBA8 albedo + metallic (imageStore conversions: round to nearest
// even binary16; clamp, scale, round for unorm8), RG16F motion = currentUV - prevUV (computeMotionVector, raygen.rgen:150-155) or
// 0 for the sky (:409-412).
BLOK_DEV void store_narrow(const PathArgs& P, size_t index, uint32_t px, uint32_t py, V3 pos, float depth, bool had_hit, V3 normal, float roughness,
                           V3 albedo, float metallic) {
    if (P.normal_roughness_h) {
        uint16_t* o = P.normal_roughness_h + 4 * index;
        o[0] = f2h(normal.x); o[1] = f2h(normal.y); o[2] = f2h(normal.z); o[3] = f2h(roughness);
    }
    if (P.albedo_metallic_u8) P.albedo_metallic_u8[index] = unorm8(albedo.x) | (unorm8(albedo.y) << 8) | (unorm8(albedo.z) << 16) | (unorm8(metallic) << 24);
    if (P.motion_h) {
        float mu = 0.0f, mv = 0.0f;
        if (had_hit && depth < 9999.0f) {
            const float* M = P.prev_view_proj;
            const float cx = ((M[0] * pos.x + M[4] * pos.y) + M[8] * pos.z) + M[12];
            const float cy = ((M[1] * pos.x + M[5] * pos.y) + M[9] * pos.z) + M[13];
            const float cw = ((M[3] * pos.x + M[7] * pos.y) + M[11] * pos.z) + M[15];
            const float cu = (static_cast<float>(px) + 0.5f) / static_cast<float>(P.trace.frame_w), cv = (static_cast<float>(py) + 0.5f) / static_cast<float>(P.trace.frame_h);
            mu = cu - ((cx / cw) * 0.5f + 0.5f); mv = cv - ((cy / cw) * 0.5f + 0.5f);
        }
        P.motion_h[2 * index] = f2h(mu); P.motion_h[2 * index + 1] = f2h(mv);
    }
}

// raygen.rgen:167-414 for pixel (px, py) of the full frame; `index` is its slot in the output planes.
//
// The shader's nested loops (samples x bounces, each bounce a radiance trace and possibly a shadow trace) are
// run as a state machine with ONE trace per iteration: a lane whose sample ended starts its next sample at once,
// and a lane waiting for its shadow ray traces it in the same walk() call in which its neighbours trace radiance
// rays.  Every lane still performs exactly the shader's sequence of operations for its pixel, so results do not
// depend on what the other lanes do; what changes is that the wave runs max-over-lanes of the TOTAL number of
// traces instead of the sum over samples of per-sample maxima, and that walk() exists once in the code.
// t0: conservative start parameter of this pixel's primary rays from the beam pre-pass (beam.h), kBeamNone = they all
// miss, 0 = none computed; it covers the sub-pixel jitter (+-0.25 pixel, the beam's frustum is grown by a whole pixel).
// stk: the lane's slot of the walk's LDS stack; keep_lohi / keep_base (null: every ray starts at the root): the lane's slots of a second,
// compact LDS area of levels - 2 entries (12 bytes each; the root needs none) that holds the ancestors of the pixel's anchor while the
// walks of other rays overwrite the stack.
// kResume: compiled with the anchor machinery (walk_resume); the build without it is the one without its register pressure.
// kSkyOnly: for a pixel of a tile whose frustum meets no voxel (the caller has checked t0 >= kBeamNone): only the sky path below is compiled, a
// function of a few registers — in the full one, values spilled at its head were written out by three quarters of a frame's waves for nothing.
template <bool kResume = true, bool kSkyOnly = false>
BLOK_DEV void shade_pixel(const PathArgs& P, uint32_t px, uint32_t py, size_t index, uint4* stk, float t0 = 0.0f, uint2* keep_lohi = nullptr, uint32_t* keep_base = nullptr,
                          [[maybe_unused]] TailRecord* pool = nullptr, [[maybe_unused]] TailAnswer* tail_results = nullptr) {
    const TraceArgs& A = P.trace;
    const blok_camera& cam = A.cam;
    const V3 cam_pos = v3(cam.pos[0], cam.pos[1], cam.pos[2]);
    const V3 cam_f = v3(cam.fwd[0], cam.fwd[1], cam.fwd[2]);
    const V3 cam_r = v3(cam.right[0], cam.right[1], cam.right[2]);
    const V3 cam_u = v3(cam.up[0], cam.up[1], cam.up[2]);
    // First-hit G-buffer (:173-181, :395-407).  Its values are final once sample 0's primary ray has been traced, so they are
    // written at that moment instead of being carried in ~16 registers through every sample: the planes of a pixel whose
    // primary ray misses are stored here, a first hit overwrites them below.
    {
        const V3 sky_pos = vadd(cam_pos, vscale(vnormalize(cam_f), 10000.0f));
        const V3 sky_albedo = sky_color(vnormalize(vsub(sky_pos, cam_pos)));
        store4(P.world_pos, index, sky_pos.x, sky_pos.y, sky_pos.z, 10000.0f);
        store4(P.normal_roughness, index, 0.0f, 1.0f, 0.0f, 0.0f);
        store4(P.albedo_metallic, index, sky_albedo.x, sky_albedo.y, sky_albedo.z, 0.0f);
        store_narrow(P, index, px, py, sky_pos, 10000.0f, false, v3(0.0f, 1.0f, 0.0f), 0.0f, sky_albedo, 0.0f);
    }
    V3 accumulated = v3(0, 0, 0);
    const V3 sun_dir = sun_direction();
    const V3 sun_radiance = v3(3.0f, 2.9f, 2.7f);

    // per-sample state
    uint32_t s = 0u, bounce = 0u, rng = 0u;
    V3 ray_org = cam_pos, ray_dir = cam_f, radiance = v3(0, 0, 0), throughput = v3(1, 1, 1);
    // surface state kept across the shadow trace
    bool shadow_phase = false;
    // the pixel's anchor (walk_resume): a reported voxel whose ancestors lie in the side area (and, while stack_is_anchor, on the stack)
    bool anchored = false, stack_is_anchor = false;
    WalkAnchor anchor;
    anchor.vx = anchor.vy = anchor.vz = 0; anchor.brick.lo = anchor.brick.hi = anchor.brick.base = 0u;
    V3 n = v3(0, 1, 0), albedo = v3(0, 0, 0), hit_pos = v3(0, 0, 0);
    float roughness = 0.0f, metallic = 0.0f, n_dot_l = 0.0f;

    auto begin_sample = [&]() {                                                              // :190-209
        rng = init_rng(px, py, A.frame_w, P.frame_count, s);
        float pcx, pcy;
        if (s == 0u) { pcx = static_cast<float>(px) + 0.5f; pcy = static_cast<float>(py) + 0.5f; }
        else {
            const float jx = random_float(rng) - 0.5f;
            const float jy = random_float(rng) - 0.5f;
            pcx = static_cast<float>(px) + 0.5f + jx * 0.5f;
            pcy = static_cast<float>(py) + 0.5f + jy * 0.5f;
        }
        float u, v;
        camera_plane_uv(A, pcx, pcy, u, v);                                                  // incl. the frame's TAA jitter
        ray_dir = vnormalize(vadd(vadd(cam_f, vscale(cam_r, u)), vscale(cam_u, v)));
        ray_org = cam_pos;
        radiance = v3(0, 0, 0); throughput = v3(1, 1, 1);
        bounce = 0u;
    };
    if (P.spp != 0u && P.max_bounces != 0u) begin_sample();

#ifdef BLOK_PATH_CLOCKS
    uint32_t kind_clocks[3] = {0u, 0u, 0u}, kind_rounds[3] = {0u, 0u, 0u}, kind_lanes[3] = {0u, 0u, 0u};
    const uint64_t wave_clock0 = __builtin_amdgcn_s_memtime();
#endif
    // A tile whose frustum meets no voxel (beam.h: the start parameter says "none"; sub-pixel jitter stays inside the grown frustum): every
    // primary ray of every sample misses, so a sample is its camera ray's sky colour — the same sums in the same order as below, without the
    // ray set-up (three divisions), the empty walk and the round's bookkeeping.  Three quarters of the benchmark frame's wave tiles.
    if (t0 >= kBeamNone && P.max_bounces != 0u) {
        while (s < P.spp) {
            radiance = vadd(radiance, vmul(throughput, sky_color(ray_dir)));                  // :232-235, miss.rmiss
            accumulated = vadd(accumulated, radiance);                                        // :379
            s += 1u;
            if (s < P.spp) begin_sample();
        }
    }
    if constexpr (!kSkyOnly) {
    // ---- the bounce rounds' tail pool (round 4) ----
    // Nearly half of a bounce round's trips run with eight lanes or fewer still walking (profiles/r04_paths_kind_clocks.txt): a few grazing rays.
    // With two bounces a bounce ray is its path's LAST segment — nothing follows from it but a term of the pixel's sum: throughput x sky colour if
    // it leaves the world, throughput x emission if the voxel it reports glows.  So such a round stops after tail_cap trips; a ray still walking is
    // PARKED — ray, the parameter to go on from, throughput, owner lane: a record in the wave's pool — and its lane ends the sample without that term.
    // When 64 rays are parked (or the wave has nothing else left) they get a round of their own, one per lane whoever owns them, each
    // walked again from the root with tmin = where it stood (exact: walk_loop), again cut off after tail_cap_parked trips and parked
    // again if need be: rays of a kind, long with long.  An answer goes to the owner's sum through LDS, the owners taking them in record
    // order (deterministic).  What changes for the pixel is the ORDER of its float sum (the parked term is added later), inside
    // tests/test_paths.py's tolerance by five orders of magnitude; the G-buffer is untouched.  The CPU harness runs without it.
    // The pool's two counters — parked records, answers waiting for their owners — live in LDS behind the answers and are read by every
    // lane at the top of every trip: lanes sit rounds out, and a count kept in a register would be stale in those.
    [[maybe_unused]] uint32_t pool_n = 0u, pickup_n = 0u;
#if !defined(BLOK_TRACE_HOST_HARNESS)
    const auto lane_id = []() { return static_cast<uint32_t>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); };      // (two instructions: not kept in a register across the walks)
    uint32_t* const tail_ctl = pool != nullptr ? reinterpret_cast<uint32_t*>(tail_results + kTailBatch) : nullptr;
    if (pool != nullptr) { tail_ctl[0] = 0u; tail_ctl[1] = 0u; }
#endif
    for (;;) {
        [[maybe_unused]] bool tail_round = false;
        [[maybe_unused]] uint32_t tail_rank = 0u, tail_count = 0u;
#if !defined(BLOK_TRACE_HOST_HARNESS)
        if (pool != nullptr) { pool_n = __builtin_amdgcn_readfirstlane(tail_ctl[0]); pickup_n = __builtin_amdgcn_readfirstlane(tail_ctl[1]); }      // (every lane reads the same words: scalar registers)
#endif
        if (!((s < P.spp || pool_n != 0u || pickup_n != 0u) && P.max_bounces != 0u)) break;
#if !defined(BLOK_TRACE_HOST_HARNESS)
        if (pool != nullptr) {
            if (pickup_n != 0u) {                                   // answers of the last parked round: every lane takes its own, in record order
                for (uint32_t k = 0; k < pickup_n; ++k) {
                    const TailAnswer e = tail_results[k];
                    if (e.owner == lane_id()) accumulated = vadd(accumulated, v3(e.x, e.y, e.z));
                }
                if (lane_id() == static_cast<uint32_t>(__builtin_ctzll(__ballot(true)))) tail_ctl[1] = 0u;
                pickup_n = 0u;
            }
            const uint64_t here = __ballot(true);
            const uint32_t workers = static_cast<uint32_t>(__builtin_popcountll(here));
            const bool busy_any = __ballot(s < P.spp) != 0ull;
            if (pool_n >= kTailBatch || (!busy_any && pool_n != 0u)) {
                tail_round = true;
                tail_count = pool_n < workers ? pool_n : workers;
                tail_rank = static_cast<uint32_t>(__builtin_popcountll(here & ((1ull << lane_id()) - 1ull)));
            }
        }
#endif
        if (!tail_round && s >= P.spp) continue;                    // nothing of its own left: this lane only helps with parked rays
#if !defined(BLOK_TRACE_HOST_HARNESS) && !defined(BLOK_PATH_NO_PHASES)
        if (!tail_round) {
            // The wave walks one kind of ray at a time — primary rays (coherent, short behind the beam pre-pass), then shadow
            // rays (parallel, short behind the last-occluder map), then bounce rays (long, incoherent): a lane whose pending ray
            // is of another kind sits the iteration out, instead of a few bounce rays stretching every iteration and the kinds
            // spoiling each other's coherence.  Per lane nothing changes (same rays, same random numbers, same order), so
            // results are identical.  Measured at 4K: 2 bounces 8.75 -> 7.28 ms per 8 spp, 64 spp 69.4 -> 58.8 ms; with one
            // bounce there are only short rays and taking turns costs (3.8 -> 5.3 ms), so it is not done there.
            if (P.batch_kinds != 0u && P.max_bounces > 1u) {
                const uint32_t kind = shadow_phase ? 1u : (bounce == 0u ? 0u : 2u);
                if (P.batch_kinds >= 2u) {
                    // ... and sample by sample: the OLDEST pending (sample, kind) of the wave goes first.  With "primary rays first" the
                    // pixels whose primary rays miss race through all their samples in rounds of their own while the others still owe
                    // the shadow and bounce rays of sample 0 — twice the primary rounds, each at partial occupancy; in step, every
                    // sample is one full primary round, one shadow round, one bounce round.
                    const uint32_t key = s * 4u + kind;
                    uint32_t oldest = __builtin_amdgcn_readfirstlane(key);
                    for (;;) {                                   // min over the active lanes: each turn strictly lowers it
                        const unsigned long long lower = __ballot(key < oldest);
                        if (lower == 0ull) break;
                        oldest = __builtin_amdgcn_readlane(key, static_cast<uint32_t>(__builtin_ctzll(lower)));
                    }
                    if (key != oldest) continue;
                } else {
                    const bool any_primary = __ballot(kind == 0u) != 0ull, any_shadow = __ballot(kind == 1u) != 0ull;
                    if (kind != (any_primary ? 0u : (any_shadow ? 1u : 2u))) continue;
                }
            }
        }
#endif
        RayIn r;
#if !defined(BLOK_TRACE_HOST_HARNESS)
        if (tail_round) {
            // a parked ray for each of the first tail_count lanes (the pool's top records); the others walk nothing (an empty interval) but stay
            // in the round: its counts are formed by ballots
            r.ox = r.oy = r.oz = 0.0f; r.dx = r.dy = r.dz = 1.0f; r.tmin = 0.0f; r.tmax = 0.0f;
            if (tail_rank < tail_count) {
                const TailRecord* my_record = pool + (pool_n - tail_count + tail_rank);
                r.ox = tail_load(&my_record->ox); r.oy = tail_load(&my_record->oy); r.oz = tail_load(&my_record->oz);
                r.dx = tail_load(&my_record->dx); r.dy = tail_load(&my_record->dy); r.dz = tail_load(&my_record->dz);
                r.tmin = tail_load(&my_record->tcur); r.tmax = 10000.0f;
            }
        } else
#endif
        if (shadow_phase) {
            const V3 so = vadd(hit_pos, vscale(n, 0.001f));                                   // :284
            r.ox = so.x; r.oy = so.y; r.oz = so.z;
            r.dx = sun_dir.x; r.dy = sun_dir.y; r.dz = sun_dir.z;
            r.tmin = 0.001f; r.tmax = 1000.0f;                                                // :294,:296
            if (P.sun_map) {
                // beyond the last occluder of this ray's column nothing can be hit: cap tmax there (an empty interval = an
                // immediate miss); the "any hit" answer is unchanged, see beam.h
                const float fu = (vdot(v3(P.sun_u[0], P.sun_u[1], P.sun_u[2]), so) - P.sun_u0) * P.sun_inv_texel;
                const float fv = (vdot(v3(P.sun_v[0], P.sun_v[1], P.sun_v[2]), so) - P.sun_v0) * P.sun_inv_texel;
                if (fu >= 0.0f && fv >= 0.0f && fu < static_cast<float>(P.sun_nu) && fv < static_cast<float>(P.sun_nv)) {
                    const float last = P.sun_map[static_cast<uint32_t>(fv) * P.sun_nu + static_cast<uint32_t>(fu)];
                    const float t_last = last - vdot(sun_dir, so);
                    const float cap = t_last + 0.05f + 1.0e-4f * fabsf(t_last);
                    r.tmax = cap > r.tmin ? fminf(r.tmax, cap) : 0.0f;
                }
            }
        } else {
            r.ox = ray_org.x; r.oy = ray_org.y; r.oz = ray_org.z;
            r.dx = ray_dir.x; r.dy = ray_dir.y; r.dz = ray_dir.z;
            r.tmin = 0.001f; r.tmax = 10000.0f;                                               // :225,:227
        }
        if (!tail_round && !shadow_phase && bounce == 0u && t0 > 0.0f) {         // primary ray behind the beam pre-pass (one walk call site)
            r.tmin = fmaxf(r.tmin, t0);
            if (t0 >= kBeamNone) r.tmax = 0.0f;                   // the tile's frustum meets no voxel: empty interval, immediate miss
        }
        BLOK_PATH_KIND(shadow_phase ? 1u : (bounce == 0u ? 0u : 2u));
#ifdef BLOK_PATH_CLOCKS
        const uint64_t kind_clock0 = __builtin_amdgcn_s_memtime();
        const uint32_t round_kind = shadow_phase ? 1u : (bounce == 0u ? 0u : 2u);
        const uint32_t round_lanes = static_cast<uint32_t>(__builtin_popcountll(__ballot(true)));
#endif
        HitInfo hit;
        hit.found = false; hit.t = -1.0f; hit.material = 0u; hit.face = 0xFFu; hit.vx = hit.vy = hit.vz = 0; hit.brick.lo = hit.brick.hi = hit.brick.base = 0u;
        [[maybe_unused]] bool cut_off = false;                    // the walk was stopped by its round's cap: the ray is parked (or parked again)
        [[maybe_unused]] float tcur_at_cut = 0.0f;
        {
            BLOK_STAT(4, 0);                       // a walk begins
            const WalkRay R = walk_ray(A, r.ox, r.oy, r.oz, safe_inv(r.dx), safe_inv(r.dy), safe_inv(r.dz));
            WalkState ws;
            // ONE loop for every kind of ray.  Whenever the pixel has an anchor — the latest reported voxel another ray starts from, its
            // ancestors in the side area — the walk is put into the state the walk from the root would reach (walk_resume: exact for ANY
            // anchor and any ray; it pays because a shadow / bounce ray starts next to the anchor and the next sample's primary ray, behind
            // the wave tile's own start parameter, not far from it); otherwise it starts at the root.
            bool resumed = false;
            if (kResume && anchored && P.resume_secondary != 0u) {
                if (!stack_is_anchor) {            // an earlier walk has overwritten the stack below the root's slot (which only ever holds the root)
                    for (uint32_t j = 0; j + 2u < A.levels; ++j) { const uint2 c = keep_lohi[j * kBlock]; stk[j * kBlock] = make_uint4(c.x, c.y, keep_base[j * kBlock], 0u); }
                    stack_is_anchor = true;
                }
                resumed = walk_resume(A, r, R, anchor, stk, ws);
            }
            // (the frame kernels' common start of a wave's primary rays, trace_core.h: walk_enter_wave, was tried here too: 44.7 -> 45.6 ms at 64 spp — 61 spilled
            // registers instead of 28 cost more than a round's shared descents save; profiles/r04_paths_start_ab.txt)
            if (!resumed) walk_enter(A, R, r.tmin, r.tmax, ws);
            if (ws.walking) stack_is_anchor = false;
#if !defined(BLOK_TRACE_HOST_HARNESS)
            // the round's cap: a round over parked rays; a bounce round all of whose rays are their paths' last segments (with room in the pool)
            uint32_t cap = 0xFFFFFFFFu;
            if (pool != nullptr) {
                if (tail_round) cap = P.tail_cap_parked;
                else if (__ballot(!(bounce != 0u && !shadow_phase && bounce + 1u >= P.max_bounces)) == 0ull && pool_n + kTailBatch <= kTailCapacity) cap = P.tail_cap;
            }
            walk_loop<true>(A, R, r.tmax, ws, stk, cap);
            cut_off = ws.walking;
#else
            walk_loop(A, R, r.tmax, ws, stk);
#endif
            if (ws.found) hit = walk_hit(A, r, R, ws);
            tcur_at_cut = ws.tCur;
            // a report becomes the anchor when it is a first hit or another ray of this path will start from it; its ancestors go to the side
            if (kResume && ws.found && !shadow_phase && keep_lohi != nullptr && (bounce == 0u || bounce + 1u < P.max_bounces)) {
                anchor.vx = hit.vx; anchor.vy = hit.vy; anchor.vz = hit.vz; anchor.brick = hit.brick;
                for (uint32_t j = 0; j + 2u < A.levels; ++j) { const uint4 c = stk[j * kBlock]; keep_lohi[j * kBlock] = make_uint2(c.x, c.y); keep_base[j * kBlock] = c.z; }
                anchored = true; stack_is_anchor = true;
            }
        }
#ifdef BLOK_PATH_CLOCKS
        {   // diagnostic build (scripts/r03/paths_kind_clocks.py): the round's clocks are booked by its first active lane, under the kind of that lane
            const uint32_t dt = static_cast<uint32_t>((__builtin_amdgcn_s_memtime() - kind_clock0) >> 4);
            if ((threadIdx.x & 63u) == static_cast<uint32_t>(__builtin_ctzll(__ballot(true)))) {
                kind_clocks[round_kind] += dt; kind_rounds[round_kind] += 1u; kind_lanes[round_kind] += round_lanes;
            }
        }
#endif

#if !defined(BLOK_TRACE_HOST_HARNESS)
        if (pool != nullptr) {
            const uint32_t first = static_cast<uint32_t>(__builtin_ctzll(__ballot(true)));
            const uint32_t my_lane = lane_id();
            if (tail_round) {
                // answers to LDS (owner in .w; none for a lane without a record or with a ray parked again), rays still walking back onto the pool
                const bool worker = tail_rank < tail_count;
                const TailRecord* my_record = pool + (pool_n - tail_count + (worker ? tail_rank : 0u));      // (formed again: not kept across the walk)
                float wx = 0.0f, wy = 0.0f, wz = 0.0f; uint32_t owner = 0xFFFFFFFFu;
                if (worker) { wx = tail_load(&my_record->wx); wy = tail_load(&my_record->wy); wz = tail_load(&my_record->wz); owner = tail_load(&my_record->owner); }
                const bool again = worker && cut_off;
                const uint64_t again_lanes = __ballot(again);
                V3 term = v3(0.0f, 0.0f, 0.0f);
                if (worker && !again) {
                    if (!hit.found) term = vmul(v3(wx, wy, wz), sky_color(v3(r.dx, r.dy, r.dz)));                   // :232-235
                    else {
                        const uint32_t id = hit.material < 65535u ? hit.material : 65535u;
                        const blok_material mat = A.mat_table[id < A.n_materials ? id : 0u];
                        const V3 emission = v3(mat.emission[0], mat.emission[1], mat.emission[2]);
                        if (is_emissive(emission)) term = vmul(v3(wx, wy, wz), emission);                              // :265-277
                    }
                }
                tail_results[tail_rank < kTailBatch ? tail_rank : 0u] = TailAnswer{term.x, term.y, term.z, worker && !again ? owner : 0xFFFFFFFFu};
                const uint32_t base = pool_n - tail_count;
                if (again) {
                    // (every worker has read its record: the loads above are complete before these stores in program order)
                    TailRecord* to = pool + base + static_cast<uint32_t>(__builtin_popcountll(again_lanes & ((1ull << my_lane) - 1ull)));
                    to->ox = r.ox; to->oy = r.oy; to->oz = r.oz; to->tcur = tcur_at_cut; to->dx = r.dx; to->dy = r.dy; to->dz = r.dz;
                    to->wx = wx; to->wy = wy; to->wz = wz; to->owner = owner;
                }
                if (my_lane == first) { tail_ctl[0] = base + static_cast<uint32_t>(__builtin_popcountll(again_lanes)); tail_ctl[1] = tail_count; }
                continue;
            }
            const uint64_t parked_lanes = __ballot(cut_off);
            if (cut_off) {
                TailRecord* to = pool + pool_n + static_cast<uint32_t>(__builtin_popcountll(parked_lanes & ((1ull << my_lane) - 1ull)));
                to->ox = r.ox; to->oy = r.oy; to->oz = r.oz; to->tcur = tcur_at_cut; to->dx = r.dx; to->dy = r.dy; to->dz = r.dz;
                to->wx = throughput.x; to->wy = throughput.y; to->wz = throughput.z; to->owner = my_lane;
            }
            if (parked_lanes != 0ull && my_lane == first) tail_ctl[0] = pool_n + static_cast<uint32_t>(__builtin_popcountll(parked_lanes));
        }
#endif
        bool end_sample = false, continue_path = false;
        if (shadow_phase) {
            shadow_phase = false;
            if (!hit.found) {                                                                 // not shadowed, :300-325
                const V3 diffuse = vscale(albedo, 1.0f - metallic);
                radiance = vadd(radiance, vscale(vscale(vmul(vmul(throughput, diffuse), sun_radiance), n_dot_l), kInvPi));
                if (roughness < 0.9f) {
                    const V3 h = vnormalize(vsub(sun_dir, ray_dir));
                    const float n_dot_h = fmaxf(vdot(n, h), 0.0f);
                    const float v_dot_h = fmaxf(vdot(vneg(ray_dir), h), 0.0f);
                    const float a = roughness * roughness;
                    const float a2 = a * a;
                    const float denom = n_dot_h * n_dot_h * (a2 - 1.0f) + 1.0f;
                    const float d = a2 / (kPi * denom * denom);
                    const V3 f0 = vmix(v3(0.04f, 0.04f, 0.04f), albedo, metallic);
                    const V3 f = fresnel_schlick(v_dot_h, f0);
                    radiance = vadd(radiance, vscale(vmul(vscale(vscale(vmul(throughput, f), d), 0.25f), sun_radiance), n_dot_l));
                }
            }
            continue_path = true;
        } else if (!hit.found) {                                                              // :232-235, miss.rmiss
            if (!cut_off) radiance = vadd(radiance, vmul(throughput, sky_color(ray_dir)));    // (a parked ray's term comes later, through the pool)
            end_sample = true;
        } else {
            // hit.rchit:58-75
            const uint32_t id = hit.material < 65535u ? hit.material : 65535u;
            const blok_material mat = A.mat_table[id < A.n_materials ? id : 0u];
            const V3 emission = v3(mat.emission[0], mat.emission[1], mat.emission[2]);
            // the voxel a path's LAST segment reports (bounce + 1 = the loop bound, :212) can give the pixel its emission and nothing else: no
            // shadow ray leaves it (:282, bounce 0 only), no further segment — its position, normal and surface parameters are not formed
            const bool last_segment = bounce != 0u && bounce + 1u >= P.max_bounces;
            if (!last_segment) {
                hit_pos = vadd(ray_org, vscale(ray_dir, hit.t));                              // :238
                n = v3(hit.face == 0u ? 1.0f : (hit.face == 1u ? -1.0f : 0.0f),
                       hit.face == 2u ? 1.0f : (hit.face == 3u ? -1.0f : 0.0f),
                       hit.face == 4u ? 1.0f : (hit.face == 5u ? -1.0f : 0.0f));
                albedo = v3(mat.albedo[0], mat.albedo[1], mat.albedo[2]);
                metallic = static_cast<float>((mat.flags >> 24) & 0xFFu) / 255.0f;
                roughness = fmaxf(static_cast<float>((mat.flags >> 16) & 0xFFu) / 255.0f, 0.04f);
                if (vdot(n, ray_dir) > 0.0f) n = vneg(n);                                     // :248-250
                if (bounce == 0u && s == 0u) {                                                // :253-263, :403-407 (reached once)
                    const V3 final_albedo = is_emissive(emission) ? emission : albedo;
                    store4(P.world_pos, index, hit_pos.x, hit_pos.y, hit_pos.z, hit.t);
                    store4(P.normal_roughness, index, n.x, n.y, n.z, roughness);
                    store4(P.albedo_metallic, index, final_albedo.x, final_albedo.y, final_albedo.z, metallic);
                    store_narrow(P, index, px, py, hit_pos, hit.t, true, n, roughness, final_albedo, metallic);
                }
            }
            if (is_emissive(emission)) {                                                      // :265-277
                radiance = vadd(radiance, vmul(throughput, emission));
                if (luminance(emission) > 5.0f || bounce > 0u) end_sample = true;
            }
            if (last_segment) end_sample = true;
            if (!end_sample) {
                n_dot_l = fmaxf(vdot(n, sun_dir), 0.0f);                                      // :280
                if (n_dot_l > 0.0f && bounce == 0u) shadow_phase = true;                      // next trace: the shadow ray
                else continue_path = true;
            }
        }

        // The path's last segment has been traced: what the shader still does with it — Russian roulette, the next direction, the throughput
        // (:329-376) — feeds a ray its loop bound (:212) never lets it trace, so none of it can reach `radiance`: the sample ends here.
        // (Round 4: with two bounces that is every bounce ray that reports a voxel, and two sincos + a handful of divisions each.)
        if (continue_path && bounce + 1u >= P.max_bounces) { continue_path = false; end_sample = true; }
        if (continue_path) {
            if (bounce > 0u) {                                                                // :329-335
                const float p = fminf(max3f(throughput), 0.95f);
                if (random_float(rng) > p) end_sample = true;
                else throughput = vdivs(throughput, p);
            }
            if (!end_sample) {
                const float ux = random_float(rng);                                           // :338
                const float uy = random_float(rng);
                const V3 f0 = vmix(v3(0.04f, 0.04f, 0.04f), albedo, metallic);                // :341-347
                const V3 view = vneg(ray_dir);
                const float n_dot_v = fmaxf(vdot(n, view), 0.001f);
                const V3 f = fresnel_schlick(n_dot_v, f0);
                float spec_w = (f.x + f.y + f.z) / 3.0f;
                spec_w = spec_w * (1.0f - metallic) + 1.0f * metallic;
                // One of the shader's two samplers per lane (sample_ggx :115-130, sample_cosine_hemisphere :101-113).  Both are "a point in the
                // normal's tangent frame, normalised": what differs is the point's polar part and which random number turns it.  The polar part is
                // formed per branch; the sine and cosine, the frame, the combination and the normalisation — the expensive, identical rest — once
                // for all lanes instead of once per branch (a wave nearly always holds lanes of both kinds).  Per lane the same operations as
                // the shader's, in the same order.
                const bool specular = random_float(rng) < spec_w;                             // :349
                float turn, k_plane, k_normal;
                if (specular) {                                                               // :116-123
                    const float a = fmaxf(roughness, 0.04f) * fmaxf(roughness, 0.04f);
                    const float a2 = a * a;
                    const float cos_theta = rn_sqrt((1.0f - uy) / (1.0f + (a2 - 1.0f) * uy));
                    turn = ux; k_plane = rn_sqrt(fmaxf(0.0f, 1.0f - cos_theta * cos_theta)); k_normal = cos_theta;
                } else {                                                                      // :102-106
                    turn = uy; k_plane = rn_sqrt(ux); k_normal = rn_sqrt(fmaxf(0.0f, 1.0f - ux));
                }
                const float phi = 2.0f * kPi * turn;
                const float lx = k_plane * cosf(phi), ly = k_plane * sinf(phi);
                V3 tangent, bitangent;
                tangent_frame_of_face(n, tangent, bitangent);
                const V3 sampled = vnormalize(vadd(vadd(vscale(tangent, lx), vscale(bitangent, ly)), vscale(n, k_normal)));
                if (specular) {                                                               // :349-360
                    const V3 h = sampled;
                    const V3 new_dir = vsub(ray_dir, vscale(h, 2.0f * vdot(h, ray_dir)));
                    if (vdot(new_dir, n) <= 0.0f) end_sample = true;
                    else {
                        const float h_dot_v = fmaxf(vdot(h, view), 0.0f);
                        const V3 fh = fresnel_schlick(h_dot_v, f0);
                        throughput = vmul(throughput, vdivs(fh, fmaxf(spec_w, 0.001f)));
                        ray_dir = new_dir;
                    }
                } else {                                                                      // :361-367
                    const V3 diffuse = vscale(albedo, 1.0f - metallic);
                    throughput = vmul(throughput, vdivs(diffuse, fmaxf(1.0f - spec_w, 0.001f)));
                    ray_dir = sampled;
                }
                if (!end_sample) {
                    const float max_t = max3f(throughput);                                    // :370-373
                    if (max_t > 10.0f) throughput = vscale(throughput, 10.0f / max_t);
                    ray_org = vadd(hit_pos, vscale(n, 0.002f));                               // :376
                    bounce += 1u;
                    if (bounce >= P.max_bounces) end_sample = true;                           // loop bound, :212
                }
            }
        }

        if (end_sample) {
            accumulated = vadd(accumulated, radiance);                                        // :379
            s += 1u;
            if (s < P.spp) begin_sample();
        }
    }

    }      // !kSkyOnly

#ifdef BLOK_PATH_CLOCKS
    if (A.debug_clocks) {                                   // [kind] clocks / 16, rounds, active lanes: summed over the wave, added once
        if ((threadIdx.x & 63u) == 0u)                      // [9]: the wave's clocks / 16 from its first sample to here; [10]: waves
            { (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(A.debug_clocks) + 9, static_cast<unsigned long long>((__builtin_amdgcn_s_memtime() - wave_clock0) >> 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(A.debug_clocks) + 10, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        for (int k = 0; k < 3; ++k) {
            uint32_t c = kind_clocks[k], n = kind_rounds[k], l = kind_lanes[k];
            for (int off = 32; off > 0; off >>= 1) { c += __shfl_down(c, off); n += __shfl_down(n, off); l += __shfl_down(l, off); }
            if ((threadIdx.x & 63u) == 0u) {
                (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(A.debug_clocks) + k * 3 + 0, static_cast<unsigned long long>(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(A.debug_clocks) + k * 3 + 1, static_cast<unsigned long long>(n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(A.debug_clocks) + k * 3 + 2, static_cast<unsigned long long>(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#endif
    V3 color = vdivs(accumulated, static_cast<float>(P.spp));                                 // :383
    const float max_val = max3f(color);                                                       // :386-389
    if (max_val > 100.0f) color = vscale(color, 100.0f / max_val);
    store4(P.color, index, color.x, color.y, color.z, 1.0f);                                  // :392
}

// ---------------------------------------------------------------------------------------------------
// tonemap.comp (reference assets/shaders/tonemap.comp:17-143): exposure, Khronos PBR Neutral (or the soft-clip
// "neutral" operator), post-tonemap saturation recovery, clamp, RGBA8.  Defaults of the reference:
// exposure 1.0, saturationBoost 1.15, operator 1 (renderer_postprocess.hpp:110-113).
struct TonemapArgs {
    const float* hdr;        // float4 per pixel
    uint32_t* ldr;           // RGBA8
    uint32_t n;
    float exposure, saturation_boost;
    int op;                  // 0 neutral soft clip, 1 Khronos PBR Neutral
};
BLOK_DEV float vlength(V3 v) { return rn_sqrt(vdot(v, v)); }
BLOK_DEV V3 khronos_pbr_neutral(V3 hdr) {                        // tonemap.comp:65-82
    const float start = 0.8f - 0.04f;
    const float desaturation = 0.15f;
    const float x = fminf(hdr.x, fminf(hdr.y, hdr.z));
    const float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
    hdr = v3(hdr.x - offset, hdr.y - offset, hdr.z - offset);
    const float peak = fmaxf(hdr.x, fmaxf(hdr.y, hdr.z));
    if (peak < start) return hdr;
    const float d = 1.0f - start;
    const float new_peak = 1.0f - d * d / (peak + d - start);
    hdr = vscale(hdr, new_peak / peak);
    const float g = 1.0f - 1.0f / (desaturation * (peak - new_peak) + 1.0f);
    return vmix(hdr, v3(new_peak, new_peak, new_peak), g);
}
BLOK_DEV V3 neutral_tonemap(V3 hdr) {                            // tonemap.comp:85-95
    const float peak = fmaxf(fmaxf(hdr.x, hdr.y), hdr.z);
    if (peak <= 1.0f) return hdr;
    const float compressed = 1.0f - expf(-(peak - 1.0f));
    return vscale(hdr, (1.0f + compressed) / peak);
}
BLOK_DEV V3 saturation_recovery(V3 ldr, V3 hdr, float boost) {   // tonemap.comp:43-61
    if (boost <= 1.0f) return ldr;
    const float hdr_luma = luminance(hdr);
    const float hdr_sat = hdr_luma > 0.0001f ? vlength(vsub(hdr, v3(hdr_luma, hdr_luma, hdr_luma))) / hdr_luma : 0.0f;
    const float ldr_luma = luminance(ldr);
    const float ldr_sat = ldr_luma > 0.0001f ? vlength(vsub(ldr, v3(ldr_luma, ldr_luma, ldr_luma))) / ldr_luma : 0.0f;
    if (ldr_sat > 0.0001f && ldr_luma > 0.01f) {
        const float ratio = fminf(hdr_sat / fmaxf(ldr_sat, 0.001f), 2.0f);
        const float recovery = 1.0f * (1.0f - (boost - 1.0f)) + ratio * (boost - 1.0f);
        return vmix(v3(ldr_luma, ldr_luma, ldr_luma), ldr, fminf(recovery, 1.5f));
    }
    return ldr;
}
BLOK_DEV uint32_t unorm8(float v) {                              // rgba8 imageStore: clamp, scale, round to nearest
    const float c = fminf(fmaxf(v, 0.0f), 1.0f);
    return static_cast<uint32_t>(c * 255.0f + 0.5f);
}
BLOK_DEV uint32_t tonemap_pixel(const TonemapArgs& T, uint32_t i) {   // tonemap.comp:97-143
    V3 hdr = v3(T.hdr[4 * i], T.hdr[4 * i + 1], T.hdr[4 * i + 2]);
    hdr = vscale(hdr, T.exposure);
    const V3 original = hdr;
    V3 ldr = T.op == 0 ? neutral_tonemap(hdr) : khronos_pbr_neutral(hdr);
    if (T.saturation_boost > 1.0f) ldr = saturation_recovery(ldr, original, T.saturation_boost);
    else if (T.saturation_boost < 1.0f && T.saturation_boost > 0.0f) {
        const float luma = luminance(ldr);
        ldr = vmix(v3(luma, luma, luma), ldr, T.saturation_boost);
    }
    return unorm8(ldr.x) | (unorm8(ldr.y) << 8) | (unorm8(ldr.z) << 16) | 0xFF000000u;
}

// ---------------------------------------------------------------------------------------------------
// Progressive accumulation + display transform of the reference's compute backend
// (blok/src/cuda_tracer.cu:372-386 accumulate, :209-216 ACES fit, :95-99 gamma 2.2 to 8 bits).
struct AccumArgs {
    const float* color;      // this frame's average radiance, float4 per pixel
    float* accum;            // xyz running sum, w frames accumulated
    uint32_t* rgba;          // may be null
    uint32_t n;
};
BLOK_DEV float aces_channel(float x) {                            // cuda_tracer.cu:209-216
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return fminf(fmaxf((x * (a * x + b)) / (x * (c * x + d) + e), 0.0f), 1.0f);
}
BLOK_DEV uint32_t to_srgb8(float x) {                             // cuda_tracer.cu:95-99
    x = fminf(fmaxf(x, 0.0f), 1.0f);
    const float g = powf(x, 1.0f / 2.2f);
    return static_cast<uint32_t>(g * 255.0f + 0.5f) & 0xFFu;
}
BLOK_DEV void accumulate_pixel(const AccumArgs& T, uint32_t i) {  // cuda_tracer.cu:372-386
    float sx = T.accum[4 * i], sy = T.accum[4 * i + 1], sz = T.accum[4 * i + 2], spp = T.accum[4 * i + 3];
    sx = sx + T.color[4 * i]; sy = sy + T.color[4 * i + 1]; sz = sz + T.color[4 * i + 2];
    spp += 1.0f;
    T.accum[4 * i] = sx; T.accum[4 * i + 1] = sy; T.accum[4 * i + 2] = sz; T.accum[4 * i + 3] = spp;
    if (!T.rgba) return;
    const float inv = 1.0f / spp;
    T.rgba[i] = to_srgb8(aces_channel(sx * inv)) | (to_srgb8(aces_channel(sy * inv)) << 8) |
                (to_srgb8(aces_channel(sz * inv)) << 16) | 0xFF000000u;
}

}  // namespace blok
#endif
