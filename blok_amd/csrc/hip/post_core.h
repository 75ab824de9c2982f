// Image-space chain behind the path tracer, one lane = one pixel:
//   temporal accumulation  assets/shaders/temporal_reproject.comp:195-316
//   variance estimation    assets/shaders/variance.comp:101-144
//   a-trous wavelet filter assets/shaders/atrous.comp:154-220            (x atrousIterations, step 1, 2, 4, ...)
//   TAA resolve            assets/shaders/taa.comp:109-220
//   sharpen                assets/shaders/sharpen.comp:19-74
// plus the motion vectors the ray-generation shader writes for them (raygen.rgen:150-155, 409-413), computed here
// from the first-hit position plane instead of inside the path kernel.
//
// HBM layout (60-150 algorithmic bytes per pixel and pass): history length and motion vectors, which the reference keeps in
// 16-bit float images, are stored as binary16 here too (half, half2), so a reader sees exactly the values the reference's
// images would hold.  Normals are different: every later use of the rgba16f normal image is normalize(xyz) (variance.comp:113,
// atrous.comp:172,191, temporal_reproject.comp:248 for the previous frame), 25 times per pixel and a-trous iteration, so the
// temporal pass narrows the path kernel's normal to binary16, normalises it ONCE and stores that unit normal as a float4
// plane (same values, none of the divides and square roots downstream).
// The temporal pass also copies the current geometry into the history slot (Denoiser::copyCurrentGeometryToHistory,
// renderer_denoising.cpp:833-866) and produces the motion vectors, in the same sweep.
//
// Arithmetic conventions where GLSL leaves them open (the CPU checker in the test suite states the same ones):
//   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z; no fused multiply-add; length = sqrt(dot); normalize(v) = v / length(v);
//   mix(a,b,t) = a*(1-t) + b*t; 16-bit float images hold round-to-nearest-even binary16 values; rgba8 stores
//   floor(clamp(x,0,1)*255 + 0.5); a linear sampler (clamp to edge) is an exact binary32 bilinear blend of the four texels
//   around uv*size - 0.5, and sampling at a texel centre returns that texel (atrous.comp, sharpen.comp).
// Every float op is a single rounded binary32 op in the shader's order, so compiled for the CPU (tests/host_harness) this
// file reproduces the checker bit for bit; on the GPU only expf in variance.comp's depth weight can differ (by ulps).
#ifndef BLOK_POST_CORE_H
#define BLOK_POST_CORE_H

#include "path_core.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace blok {


struct DenoiseSettings {           // = blok_denoise_settings (include/blok_hip.h), Denoiser::Settings renderer_denoising.hpp:49-66
    float temporal_alpha, moment_alpha, variance_clip_gamma, depth_threshold, normal_threshold, phi_color, phi_normal, phi_depth;
    int atrous_iterations;
    float variance_boost;
    int min_history_length;
};

struct PostFrame {                 // what every pass of one frame shares
    uint32_t w, h, frame_count;
    float prev_view_proj[16];      // column-major (GLM)
    DenoiseSettings s;
};

BLOK_DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
BLOK_DEV float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
BLOK_DEV float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
// float4 planes are read and written as one 16-byte access per pixel (planes are 16-byte aligned: hipMalloc / torch / numpy)
struct alignas(16) F4 { float x, y, z, w; };
BLOK_DEV F4 load4(const float* plane, size_t i) { return reinterpret_cast<const F4*>(plane)[i]; }
BLOK_DEV V3 xyz(F4 v) { return v3(v.x, v.y, v.z); }
BLOK_DEV V3 load3(const float* plane, size_t i) { return xyz(load4(plane, i)); }
BLOK_DEV V3 load3q(const float* plane, size_t i) { const F4 v = load4(plane, i); return v3(q16(v.x), q16(v.y), q16(v.z)); }
BLOK_DEV void store4(float* plane, size_t i, V3 c, float a) { F4 v; v.x = c.x; v.y = c.y; v.z = c.z; v.w = a; reinterpret_cast<F4*>(plane)[i] = v; }
BLOK_DEV V3 vmin3(V3 a, V3 b) { return v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
BLOK_DEV V3 vmax3(V3 a, V3 b) { return v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
BLOK_DEV V3 vsplat(float s) { return v3(s, s, s); }
BLOK_DEV V3 vdiv3(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
BLOK_DEV V3 vsqrt3(V3 a) { return v3(rn_sqrt(a.x), rn_sqrt(a.y), rn_sqrt(a.z)); }
BLOK_DEV float lum709(V3 c) { return vdot(c, v3(0.2126f, 0.7152f, 0.0722f)); }
BLOK_DEV V3 to_ycocg(V3 c) { return v3(0.25f * c.x + 0.5f * c.y + 0.25f * c.z, 0.5f * c.x - 0.5f * c.z, -0.25f * c.x + 0.5f * c.y - 0.25f * c.z); }
BLOK_DEV V3 from_ycocg(V3 c) { return v3(c.x + c.y - c.z, c.x + c.z, c.x - c.y - c.z); }

// linear sampler, clamp to edge, of a float4 plane (rgb)
BLOK_DEV V3 bilinear3(const float* plane, int w, int h, float u, float v) {
    const float fx = u * static_cast<float>(w) - 0.5f, fy = v * static_cast<float>(h) - 0.5f;
    const float bx = floorf(fx), by = floorf(fy);
    const float ax = fx - bx, ay = fy - by;
    const int x0 = clampi(static_cast<int>(bx), 0, w - 1), x1 = clampi(static_cast<int>(bx) + 1, 0, w - 1);
    const size_t r0 = static_cast<size_t>(clampi(static_cast<int>(by), 0, h - 1)) * w, r1 = static_cast<size_t>(clampi(static_cast<int>(by) + 1, 0, h - 1)) * w;
    const V3 top = vmix(load3(plane, r0 + x0), load3(plane, r0 + x1), ax), bottom = vmix(load3(plane, r1 + x0), load3(plane, r1 + x1), ax);
    return vmix(top, bottom, ay);
}

// prevNDC.xy * 0.5 + 0.5 of prevViewProj * vec4(p, 1)   (temporal_reproject.comp:108-113, raygen.rgen:150-154)
BLOK_DEV void project_prev(const float* M, V3 p, float& u, float& v) {
    const float cx = ((M[0] * p.x + M[4] * p.y) + M[8] * p.z) + M[12];
    const float cy = ((M[1] * p.x + M[5] * p.y) + M[9] * p.z) + M[13];
    const float cw = ((M[3] * p.x + M[7] * p.y) + M[11] * p.z) + M[15];
    u = (cx / cw) * 0.5f + 0.5f; v = (cy / cw) * 0.5f + 0.5f;
}

// ---------------------------------------------------------------------------------------------- temporal
struct TemporalArgs {
    PostFrame f;
    const float *color, *world_pos, *normal_roughness;      // this frame, float4 planes from the path kernel
    const float* motion_in;                                 // float2 per pixel or null (then computed from world_pos)
    // alternatives in the reference's image formats (path kernel: PathArgs::normal_roughness_h / motion_h): RGBA16F instead of
    // normal_roughness, RG16F instead of motion_in.  Same values: the float4 plane is narrowed to binary16 on read below.
    const uint16_t* normal_roughness_h;
    const uint16_t* motion_in_h;
    const float *prev_color, *prev_moments, *prev_world_pos; // history of the previous frame
    const uint16_t* prev_hist_len;                          // half
    const float* prev_unit_normals;                         // float4: normalize(binary16 normal) of the previous frame
    float *out_color, *out_moments, *hist_world_pos, *unit_normals;   // history of this frame
    uint16_t *out_hist_len, *motion;                        // half, half2
};

// normal + roughness of pixel i as the rgba16f image holds them
BLOK_DEV F4 load_nr16(const TemporalArgs& T, size_t i) {
    F4 r;
    if (T.normal_roughness_h) {
        const uint16_t* p = T.normal_roughness_h + 4 * i;
        r.x = h2f(p[0]); r.y = h2f(p[1]); r.z = h2f(p[2]); r.w = h2f(p[3]);
    } else {
        const F4 f = load4(T.normal_roughness, i);
        r.x = q16(f.x); r.y = q16(f.y); r.z = q16(f.z); r.w = q16(f.w);
    }
    return r;
}

BLOK_DEV void temporal_pixel(const TemporalArgs& T, int cx, int cy) {
    const int w = static_cast<int>(T.f.w), h = static_cast<int>(T.f.h);
    const size_t i = static_cast<size_t>(cy) * w + cx;
    const V3 current = load3(T.color, i);
    const F4 wp = load4(T.world_pos, i);
    const V3 world = xyz(wp);
    const float depth = wp.w;
    const float cu = (static_cast<float>(cx) + 0.5f) / static_cast<float>(w), cv = (static_cast<float>(cy) + 0.5f) / static_cast<float>(h);

    // geometry of this frame into the history slot; the unit normal of the binary16-narrowed normal, once for all passes
    const F4 nr = load_nr16(T, i);
    const V3 normal = vnormalize(v3(nr.x, nr.y, nr.z));
    store4(T.unit_normals, i, normal, nr.w);
    store4(T.hist_world_pos, i, world, depth);

    // motion vector (raygen.rgen:409-413), held as half2
    float mu = 0.0f, mv = 0.0f;
    if (T.motion_in_h) { mu = h2f(T.motion_in_h[2 * i]); mv = h2f(T.motion_in_h[2 * i + 1]); }
    else if (T.motion_in) { mu = T.motion_in[2 * i]; mv = T.motion_in[2 * i + 1]; }
    else if (depth < 9999.0f) { float pu, pv; project_prev(T.f.prev_view_proj, world, pu, pv); mu = cu - pu; mv = cv - pv; }
    const uint16_t hu = f2h(mu), hv = f2h(mv);
    T.motion[2 * i] = hu; T.motion[2 * i + 1] = hv;
    mu = h2f(hu); mv = h2f(hv);

    float pu, pv;                                                                         // :217-226
    if (rn_sqrt(mu * mu + mv * mv) > 0.0001f) { pu = cu - mu; pv = cv - mv; }
    else project_prev(T.f.prev_view_proj, world, pu, pv);

    V3 out = current;                                                                     // :228-232
    const float lum = lum709(current);
    float m1o = lum, m2o = lum * lum, hist_len = 1.0f;

    if (pu >= 0.0f && pu <= 1.0f && pv >= 0.0f && pv <= 1.0f && T.f.frame_count > 0u) {   // :235-237
        const V3 history = bilinear3(T.prev_color, w, h, pu, pv);
        const int px = clampi(static_cast<int>(pu * static_cast<float>(w)), 0, w - 1), py = clampi(static_cast<int>(pv * static_cast<float>(h)), 0, h - 1);
        const size_t p = static_cast<size_t>(py) * w + px;
        const F4 pwp = load4(T.prev_world_pos, p);
        const V3 prev_normal = load3(T.prev_unit_normals, p);
        const bool depth_ok = fabsf(depth - pwp.w) < T.f.s.depth_threshold * depth + 0.5f;              // :252-255
        const bool normal_ok = vdot(normal, prev_normal) > T.f.s.normal_threshold;                       // :258-259
        const bool pos_ok = vlength(vsub(world, xyz(pwp))) < 2.0f;                                       // :262-264
        if (depth_ok && normal_ok && pos_ok) {
            // neighbourhood statistics in YCoCg over the 3x3 pixels on the same surface (:120-193)
            // (the nine taps' planes are fetched before any is used: memory-level parallelism, as in variance_pixel)
            float tap_depth[9]; V3 tap_normal[9], tap_color[9];
#if defined(__clang__)
#pragma unroll
#endif
            for (int k = 0; k < 9; ++k) {
                const size_t s = static_cast<size_t>(clampi(cy + k / 3 - 1, 0, h - 1)) * w + clampi(cx + k % 3 - 1, 0, w - 1);
                tap_depth[k] = T.world_pos[4 * s + 3];
                { const F4 tn = load_nr16(T, s); tap_normal[k] = v3(tn.x, tn.y, tn.z); }
                tap_color[k] = load3(T.color, s);
            }
            V3 s1 = vsplat(0.0f), s2 = vsplat(0.0f), lo = vsplat(1e10f), hi = vsplat(-1e10f);
            float wsum = 0.0f;
#if defined(__clang__)
#pragma unroll
#endif
            for (int k = 0; k < 9; ++k) {
                const float dd = fabsf(depth - tap_depth[k]);
                const float nd = vdot(normal, tap_normal[k]);
                const float wgt = (dd < (depth * 0.02f + 0.1f) ? 1.0f : 0.0f) * (nd > 0.9f ? 1.0f : 0.0f);
                if (wgt > 0.0f) {
                    const V3 c = to_ycocg(tap_color[k]);
                    s1 = vadd(s1, vscale(c, wgt)); s2 = vadd(s2, vscale(vmul(c, c), wgt));
                    lo = vmin3(lo, c); hi = vmax3(hi, c);
                    wsum += wgt;
                }
            }
            V3 mean, sd;
            if (wsum > 0.0f) {
                mean = vdivs(s1, wsum);
                sd = vsqrt3(vmax3(vsub(vdivs(s2, wsum), vmul(mean, mean)), vsplat(0.0f)));
            } else { mean = to_ycocg(current); sd = vsplat(0.1f); lo = mean; hi = mean; }
            const float gamma = T.f.s.variance_clip_gamma;
            const V3 box_lo = vmax3(vsub(mean, vscale(sd, gamma)), vsub(lo, vsplat(0.05f)));
            const V3 box_hi = vmin3(vadd(mean, vscale(sd, gamma)), vadd(hi, vsplat(0.05f)));
            // clipToAABB (:92-106)
            const V3 hy = to_ycocg(history);
            const V3 centre = vscale(vadd(box_lo, box_hi), 0.5f), extent = vscale(vsub(box_hi, box_lo), 0.5f);
            const V3 off = vsub(hy, centre);
            const V3 unit = vdiv3(off, vmax3(extent, vsplat(0.0001f)));
            const float biggest = fmaxf(fmaxf(fabsf(unit.x), fabsf(unit.y)), fabsf(unit.z));
            const V3 clipped_y = biggest > 1.0f ? vadd(centre, vdivs(off, biggest)) : hy;
            const V3 clipped = vmax3(from_ycocg(clipped_y), vsplat(0.0f));
            // blend (:278-309)
            const float len = h2f(T.prev_hist_len[p]) + 1.0f;
            const float history_factor = 1.0f / fmaxf(len, 1.0f);
            float alpha = fmaxf(T.f.s.temporal_alpha, history_factor);
            const float lc = lum709(current), lh = lum709(clipped);
            const float ld = fabsf(lc - lh) / fmaxf(lc + lh + 0.01f, 0.01f);
            alpha = mixf(alpha, fminf(alpha + 0.2f, 0.5f), ld * 0.3f);
            alpha = clampf(alpha, T.f.s.temporal_alpha, 1.0f);
            out = vmix(clipped, current, alpha);
            const float ma = fmaxf(T.f.s.moment_alpha, history_factor);
            m1o = mixf(T.prev_moments[2 * p], lc, ma); m2o = mixf(T.prev_moments[2 * p + 1], lc * lc, ma);
            hist_len = fminf(len, 64.0f);
        }
    }
    store4(T.out_color, i, v3(clampf(out.x, 0.0f, 100.0f), clampf(out.y, 0.0f, 100.0f), clampf(out.z, 0.0f, 100.0f)), 1.0f);
    T.out_moments[2 * i] = clampf(m1o, 0.0f, 10000.0f); T.out_moments[2 * i + 1] = clampf(m2o, 0.0f, 10000.0f);
    T.out_hist_len[i] = f2h(hist_len);
}

// ---------------------------------------------------------------------------------------------- variance
struct VarianceArgs {
    PostFrame f;
    const float *color, *moments, *world_pos;    // temporal output colour + moments, geometry
    const uint16_t* hist_len;                    // half
    const float* unit_normals;                   // float4
    float* variance;
};

BLOK_DEV void variance_pixel(const VarianceArgs& A, int cx, int cy) {
    const int w = static_cast<int>(A.f.w), h = static_cast<int>(A.f.h);
    const size_t i = static_cast<size_t>(cy) * w + cx;
    const float m1 = A.moments[2 * i], m2 = A.moments[2 * i + 1];
    const float history = h2f(A.hist_len[i]);
    const float depth = A.world_pos[4 * i + 3];
    const V3 normal = load3(A.unit_normals, i);
    const float temporal_var = fmaxf(m2 - m1 * m1, 0.0f);
    // computeSpatialVariance :57-99.  Memory-level parallelism: the nine taps' three planes are fetched before any of them is
    // used (the shader's conditional colour fetch becomes an unconditional one whose value is only accumulated under the
    // shader's condition), otherwise every tap is a dependent round trip to L2 / HBM
    float tap_depth[9]; V3 tap_normal[9], tap_color[9];
#if defined(__clang__)
#pragma unroll
#endif
    for (int k = 0; k < 9; ++k) {
        const size_t s = static_cast<size_t>(clampi(cy + k / 3 - 1, 0, h - 1)) * w + clampi(cx + k % 3 - 1, 0, w - 1);
        tap_depth[k] = A.world_pos[4 * s + 3];
        tap_normal[k] = load3(A.unit_normals, s);
        tap_color[k] = load3(A.color, s);
    }
    float a1 = 0.0f, a2 = 0.0f, wsum = 0.0f;
#if defined(__clang__)
#pragma unroll
#endif
    for (int k = 0; k < 9; ++k) {
        const float dd = fabsf(depth - tap_depth[k]);
        const float nd = vdot(normal, tap_normal[k]);
        const float wgt = expf(-dd * dd / (0.5f * 0.5f)) * (nd > 0.9f ? 1.0f : 0.0f);
        if (wgt > 0.01f) {
            const float l = lum709(tap_color[k]);
            a1 += l * wgt; a2 += l * l * wgt; wsum += wgt;
        }
    }
    float spatial_var = 0.0f;
    if (wsum > 0.0f) { const float mean = a1 / wsum; spatial_var = fmaxf(a2 / wsum - mean * mean, 0.0f); }
    const float min_len = static_cast<float>(A.f.s.min_history_length > 4 ? A.f.s.min_history_length : 4);
    float hw = clampf((history - 1.0f) / min_len, 0.0f, 1.0f);
    hw = hw * hw;
    float var = mixf(spatial_var, temporal_var, hw);
    if (history < min_len) var *= mixf(A.f.s.variance_boost, 1.0f, history / min_len);
    A.variance[i] = fmaxf(var, 0.0001f);
}

// ---------------------------------------------------------------------------------------------- a-trous
struct AtrousArgs {
    uint32_t w, h;
    int step;
    float phi_color, phi_depth;
    const float *color, *variance, *world_pos, *unit_normals;
    float* out;
};

BLOK_DEV void atrous_pixel(const AtrousArgs& A, int cx, int cy) {
    const int w = static_cast<int>(A.w), h = static_cast<int>(A.h);
    const size_t i = static_cast<size_t>(cy) * w + cx;
    const V3 centre = load3(A.color, i);
    const F4 cwp = load4(A.world_pos, i);
    const float depth = cwp.w;
    if (depth > 9000.0f) { store4(A.out, i, centre, 1.0f); return; }                        // sky, :174-177
    const V3 pos = xyz(cwp);
    const V3 normal = load3(A.unit_normals, i);
    const float var = A.variance[i];
    // per-pixel constants of the weight functions
    const float sigma_c = 0.01f + A.phi_color * rn_sqrt(fmaxf(var, 0.0f));                  // :97-99
    const float denom_c = 2.0f * sigma_c * sigma_c + 1e-6f;
    const float sigma_d = A.phi_depth * static_cast<float>(A.step) + 0.1f;                  // :142
    const float denom_d = sigma_d * sigma_d + 1e-6f;
    V3 sum = vsplat(0.0f);
    float wsum = 0.0f;
    // 25 taps, row by row: the five taps' three planes are fetched together, then weighted (no early exits between the
    // loads; a tap the shader skips — sky sample :194-196, weight < 0.001 :207-209 — is simply not accumulated)
#if defined(__clang__)
#pragma unroll 1
#endif
    for (int row = 0; row < 5; ++row) {          // one row in flight: unrolling the rows too needs 255 VGPRs (one wave per SIMD)
        const int oy = row - 2;
        const size_t base = static_cast<size_t>(clampi(cy + oy * A.step, 0, h - 1)) * w;
        V3 tc[5], tn[5]; F4 tp[5];
#if defined(__clang__)
#pragma unroll
#endif
        for (int col = 0; col < 5; ++col) {
            const size_t s = base + clampi(cx + (col - 2) * A.step, 0, w - 1);
            tc[col] = load3(A.color, s); tp[col] = load4(A.world_pos, s); tn[col] = load3(A.unit_normals, s);
        }
        const float ky = oy == 0 ? 1.0f : ((oy == 1 || oy == -1) ? 2.0f / 3.0f : 1.0f / 6.0f);
#if defined(__clang__)
#pragma unroll
#endif
        for (int col = 0; col < 5; ++col) {
            const int ox = col - 2;
            const V3 sc = tc[col];
            const float sdepth = tp[col].w;
            const float kx = ox == 0 ? 1.0f : ((ox == 1 || ox == -1) ? 2.0f / 3.0f : 1.0f / 6.0f);
            const V3 diff = vsub(centre, sc);
            const float wc = 1.0f / (1.0f + vdot(diff, diff) / denom_c);                    // fastExp, :78-80, :101
            const float nd = fmaxf(vdot(normal, tn[col]), 0.0f);                            // :105-122
            const float t = (nd - 0.9f) / (1.0f - 0.9f);
            const float wn = nd < 0.9f ? 0.0f : t * t;
            const float dd = fabsf(depth - sdepth);                                         // :125-152
            const float plane = fabsf(vdot(vsub(xyz(tp[col]), pos), normal));
            const float dist = fmaxf(dd * 0.1f, plane);
            const float wd = dist > sigma_d * 2.0f ? 0.0f : 1.0f / (1.0f + dist * dist / denom_d);
            const float wgt = kx * ky * wc * wn * wd;
            if (!(sdepth > 9000.0f) && !(wgt < 0.001f)) {
                sum = vadd(sum, vscale(sc, wgt));
                wsum += wgt;
            }
        }
    }
    const V3 res = wsum > 0.01f ? vdivs(sum, wsum) : centre;
    store4(A.out, i, vmax3(res, vsplat(0.0f)), 1.0f);
}

// ---------------------------------------------------------------------------------------------- TAA
struct TaaArgs {
    uint32_t w, h, frame_count;
    float feedback_min, feedback_max;
    const float *color, *history;
    const uint16_t* motion;          // half2
    float *out, *out_history;
};

BLOK_DEV void taa_pixel(const TaaArgs& A, int px, int py) {
    const int w = static_cast<int>(A.w), h = static_cast<int>(A.h);
    const size_t i = static_cast<size_t>(py) * w + px;
    const float u = (static_cast<float>(px) + 0.5f) / static_cast<float>(w), v = (static_cast<float>(py) + 0.5f) / static_cast<float>(h);
    const F4 cur4 = load4(A.color, i);
    const V3 current = xyz(cur4);
    const float mx = h2f(A.motion[2 * i]), my = h2f(A.motion[2 * i + 1]);
    const float pu = u - mx, pv = v - my;
    const bool valid = pu >= 0.0f && pu <= 1.0f && pv >= 0.0f && pv <= 1.0f;
    const V3 history = bilinear3(A.history, w, h, pu, pv);
    V3 lo = vsplat(1e10f), hi = vsplat(-1e10f), s1 = vsplat(0.0f), s2 = vsplat(0.0f);
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const V3 c = to_ycocg(load3(A.color, static_cast<size_t>(clampi(py + dy, 0, h - 1)) * w + clampi(px + dx, 0, w - 1)));
            lo = vmin3(lo, c); hi = vmax3(hi, c); s1 = vadd(s1, c); s2 = vadd(s2, vmul(c, c));
        }
    const V3 mean = vdivs(s1, 9.0f);
    const V3 sd = vsqrt3(vmax3(vsub(vdivs(s2, 9.0f), vmul(mean, mean)), vsplat(0.0f)));
    const V3 hy = to_ycocg(history);
    const V3 box_lo = vmax3(vsub(mean, vscale(sd, 1.5f)), lo), box_hi = vmin3(vadd(mean, vscale(sd, 1.5f)), hi);     // varianceClip :95-107
    const V3 centre = vscale(vadd(box_hi, box_lo), 0.5f), extent = vscale(vsub(box_hi, box_lo), 0.5f);
    const V3 off = vsub(hy, centre);
    const V3 ts = vdiv3(v3(fabsf(extent.x), fabsf(extent.y), fabsf(extent.z)), vmax3(v3(fabsf(off.x), fabsf(off.y), fabsf(off.z)), vsplat(0.0001f)));
    const float t = clampf(fminf(fminf(ts.x, ts.y), ts.z), 0.0f, 1.0f);
    const V3 clipped_y = vadd(centre, vscale(off, t));
    const V3 clipped = from_ycocg(clipped_y);
    const float vx = mx * static_cast<float>(w), vy = my * static_cast<float>(h);
    float feedback = mixf(A.feedback_max, A.feedback_min, clampf(rn_sqrt(vx * vx + vy * vy) / 10.0f, 0.0f, 1.0f));
    if (!valid || A.frame_count == 0u) feedback = 0.0f;
    feedback *= 1.0f - clampf(vlength(vsub(clipped_y, hy)) * 2.0f, 0.0f, 0.5f);
    const V3 result = vmix(current, clipped, feedback);
    const V3 sharpened = vadd(current, vscale(vsub(current, from_ycocg(mean)), 0.1f));
    store4(A.out, i, vmix(sharpened, clipped, feedback), cur4.w);
    store4(A.out_history, i, result, 1.0f);
}

// ---------------------------------------------------------------------------------------------- sharpen
struct SharpenArgs {
    uint32_t w, h;
    float strength;
    const uint32_t* in;      // rgba8
    uint32_t* out;
};
// unorm8 -> float is n / 255 correctly rounded: 27 IEEE divides per pixel if done in place, so the 256 quotients come from a
// table (LDS on the device, filled with one divide per lane of the workgroup)
BLOK_DEV float unorm8_to_float(uint32_t n) { return static_cast<float>(n) / 255.0f; }
BLOK_DEV V3 unpack_rgb8(const float* lut, uint32_t p) { return v3(lut[p & 0xFFu], lut[(p >> 8) & 0xFFu], lut[(p >> 16) & 0xFFu]); }
BLOK_DEV void sharpen_pixel(const SharpenArgs& A, int x, int y, const float* lut) {
    const int w = static_cast<int>(A.w), h = static_cast<int>(A.h);
    const int xl = clampi(x - 1, 0, w - 1), xr = clampi(x + 1, 0, w - 1);
    const size_t r0 = static_cast<size_t>(clampi(y - 1, 0, h - 1)) * w, r1 = static_cast<size_t>(y) * w, r2 = static_cast<size_t>(clampi(y + 1, 0, h - 1)) * w;
    const V3 a = unpack_rgb8(lut, A.in[r0 + xl]), b = unpack_rgb8(lut, A.in[r0 + x]), c = unpack_rgb8(lut, A.in[r0 + xr]);
    const V3 d = unpack_rgb8(lut, A.in[r1 + xl]), e = unpack_rgb8(lut, A.in[r1 + x]), f = unpack_rgb8(lut, A.in[r1 + xr]);
    const V3 g = unpack_rgb8(lut, A.in[r2 + xl]), hh = unpack_rgb8(lut, A.in[r2 + x]), k = unpack_rgb8(lut, A.in[r2 + xr]);
    const V3 corners = vscale(vadd(vadd(vadd(a, c), g), k), 1.0f), cross = vscale(vadd(vadd(vadd(b, d), f), hh), 2.0f);
    const V3 blur = vdivs(vadd(vadd(corners, cross), vscale(e, 4.0f)), 16.0f);
    const V3 res = vadd(e, vscale(vsub(e, blur), A.strength * 3.0f));
    A.out[r1 + x] = unorm8(res.x) | (unorm8(res.y) << 8) | (unorm8(res.z) << 16) | 0xFF000000u;
}

}  // namespace blok
#endif
