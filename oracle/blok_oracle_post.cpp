// TEST INFRASTRUCTURE ONLY — part of liboracle.so; never linked into or called by the product libraries.
//
// CPU restatement of the reference's image-space chain behind the path tracer, statement by
// statement, each function citing the shader lines it follows:
//   denoiser   assets/shaders/temporal_reproject.comp, variance.comp, atrous.comp
//              (order and ping-pong: blok/src/renderer_denoising.cpp:714-776, 520-536; defaults renderer_denoising.hpp:49-66)
//   post       assets/shaders/taa.comp, sharpen.comp (order: blok/src/renderer_postprocess.cpp:505-556; defaults .hpp:102-118)
//   motion     assets/shaders/raygen.rgen:150-155, 409-413
//
// PARITY PIN: the reference holds no test, fixture or golden image for these shaders and they cannot run here
// (Vulkan compute) — this section is "parity unpinned": a literal restatement checked only against itself and
// against hand-computed cases in tests/test_post.py.
//
// Where GLSL leaves the arithmetic to the implementation this file fixes it, and the product follows the same choices:
//   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z, no fused multiply-add anywhere (-ffp-contract=off);
//   length(v) = sqrt(dot(v,v)), normalize(v) = v / length(v), mix(a,b,t) = a*(1-t) + b*t, exp = expf;
//   16-bit float images (rgba16f normals+roughness, rg16f motion vectors, r16f history length) hold
//   round-to-nearest-even binary16 values; rgba8 stores floor(clamp(x,0,1)*255 + 0.5);
//   a linear sampler (clamp to edge) is an exact binary32 bilinear blend of the four texels around uv*size - 0.5;
//   sampling at a texel centre returns that texel (atrous.comp, sharpen.comp).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace {

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 splat(float s) { return {s, s, s}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a / length(a); }
inline V3 vmin(V3 a, V3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
inline V3 vabs(V3 a) { return {std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }
inline V3 vsqrt(V3 a) { return {std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)}; }
inline float clampf(float x, float lo, float hi) { return std::fmin(std::fmax(x, lo), hi); }
inline V3 vclamp(V3 a, float lo, float hi) { return {clampf(a.x, lo, hi), clampf(a.y, lo, hi), clampf(a.z, lo, hi)}; }
inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline V3 mix3(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }
inline int clampi(int v, int lo, int hi) { return std::min(std::max(v, lo), hi); }

// binary32 -> binary16 (round to nearest even) -> binary32
float q16(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = x & 0x80000000u;
    x &= 0x7FFFFFFFu;
    uint32_t out;
    if (x >= 0x7F800000u) out = x;                                  // inf / nan unchanged
    else if (x >= 0x477FF000u) out = 0x7F800000u;                   // rounds to >= 65520 -> inf
    else if (x < 0x33000001u) out = 0;                              // below half of the smallest subnormal (2^-25): zero
    else {
        const int e = int(x >> 23) - 127;
        const int drop = e < -14 ? 13 + (-14 - e) : 13;             // mantissa bits that do not fit
        const uint32_t m = (x & 0x007FFFFFu) | 0x00800000u;         // explicit leading one
        const uint32_t lsb = 1u << drop, half = lsb >> 1;
        uint32_t r = m & ~(lsb - 1u);
        const uint32_t rem = m & (lsb - 1u);
        if (rem > half || (rem == half && (r & lsb))) r += lsb;
        // rebuild the float value r * 2^(e-23)
        const float v = std::ldexp(static_cast<float>(r), e - 23);
        std::memcpy(&out, &v, 4);
    }
    out |= sign;
    float g; std::memcpy(&g, &out, 4);
    return g;
}

struct Img4 {                                   // rgba32f image, row-major
    const float* p; int w, h;
    V4 load(int x, int y) const { const float* q = p + 4 * (size_t(y) * w + x); return {q[0], q[1], q[2], q[3]}; }
    V3 rgb(int x, int y) const { const V4 v = load(x, y); return {v.x, v.y, v.z}; }
    // linear sampler, clamp to edge
    V3 sample(float u, float v) const {
        const float fx = u * float(w) - 0.5f, fy = v * float(h) - 0.5f;
        const float x0f = std::floor(fx), y0f = std::floor(fy);
        const float ax = fx - x0f, ay = fy - y0f;
        const int x0 = clampi(int(x0f), 0, w - 1), x1 = clampi(int(x0f) + 1, 0, w - 1);
        const int y0 = clampi(int(y0f), 0, h - 1), y1 = clampi(int(y0f) + 1, 0, h - 1);
        const V3 top = mix3(rgb(x0, y0), rgb(x1, y0), ax), bot = mix3(rgb(x0, y1), rgb(x1, y1), ax);
        return mix3(top, bot, ay);
    }
};
// rgba16f view of a float4 plane: every component read through binary16
struct Img4h {
    const float* p; int w, h;
    V4 load(int x, int y) const { const float* q = p + 4 * (size_t(y) * w + x); return {q16(q[0]), q16(q[1]), q16(q[2]), q16(q[3])}; }
};

inline float luminance(V3 c) { return dot(c, V3{0.2126f, 0.7152f, 0.0722f}); }      // temporal_reproject.comp:72-74 (and the others)

V3 RGBToYCoCg(V3 rgb) {                                                                // temporal_reproject.comp:76-82, taa.comp:61-67
    return {0.25f * rgb.x + 0.5f * rgb.y + 0.25f * rgb.z,
            0.5f * rgb.x - 0.5f * rgb.z,
            -0.25f * rgb.x + 0.5f * rgb.y - 0.25f * rgb.z};
}
V3 YCoCgToRGB(V3 c) {                                                                  // temporal_reproject.comp:84-90, taa.comp:70-79
    return {c.x + c.y - c.z, c.x + c.z, c.x - c.y - c.z};
}

}  // namespace

extern "C" {

struct orc_denoise_settings {                      // Denoiser::Settings, renderer_denoising.hpp:49-66
    float temporalAlpha, momentAlpha, varianceClipGamma, depthThreshold, normalThreshold, phiColor, phiNormal, phiDepth;
    int atrousIterations;
    float varianceBoost;
    int minHistoryLength;
};

float orc_q16(float x) { return q16(x); }

// raygen.rgen:150-155, 409-413: motion vector of every pixel from the first-hit position plane (xyz, w = depth).
// prevViewProj is column-major (GLM).  Output rg (unquantised; the image is rg16f, readers quantise).
void orc_motion_vectors(const float* worldPos, int w, int h, const float* prevViewProj, float* motion) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float* wp = worldPos + 4 * (size_t(y) * w + x);
            float mx = 0.0f, my = 0.0f;
            if (wp[3] < 9999.0f) {                                                     // hadFirstHit && firstHitDepth < 9999, :410
                const float cu = (float(x) + 0.5f) / float(w), cv = (float(y) + 0.5f) / float(h);
                const float* M = prevViewProj;
                const float cx = ((M[0] * wp[0] + M[4] * wp[1]) + M[8] * wp[2]) + M[12];
                const float cy = ((M[1] * wp[0] + M[5] * wp[1]) + M[9] * wp[2]) + M[13];
                const float cw = ((M[3] * wp[0] + M[7] * wp[1]) + M[11] * wp[2]) + M[15];
                mx = cu - ((cx / cw) * 0.5f + 0.5f);                                   // :151-154
                my = cv - ((cy / cw) * 0.5f + 0.5f);
            }
            motion[2 * (size_t(y) * w + x)] = mx; motion[2 * (size_t(y) * w + x) + 1] = my;
        }
}

// temporal_reproject.comp main() :195-316 with computeNeighborhoodStatistics :120-193, clipToAABB :92-106, worldToPrevUV :108-113
void orc_temporal(const float* inColor, const float* inWorldPosition, const float* inNormalRoughness, const float* inMotion,
                  const float* prevHistoryColor, const float* prevMoments, const float* prevHistoryLength,
                  const float* prevWorldPosition, const float* prevNormalRoughness, int w, int h, uint32_t frameCount,
                  const float* prevViewProj, const orc_denoise_settings* S, float* outColor, float* outMoments, float* outHistoryLength) {
    const Img4 color{inColor, w, h}, wpos{inWorldPosition, w, h}, hist{prevHistoryColor, w, h}, pwpos{prevWorldPosition, w, h};
    const Img4h nrm{inNormalRoughness, w, h}, pnrm{prevNormalRoughness, w, h};
    for (int cy = 0; cy < h; ++cy)
        for (int cx = 0; cx < w; ++cx) {
            const size_t idx = size_t(cy) * w + cx;
            const V3 currentColor = color.rgb(cx, cy);
            const V4 worldPosData = wpos.load(cx, cy);
            const V4 normalRoughnessData = nrm.load(cx, cy);
            const V3 worldPos = {worldPosData.x, worldPosData.y, worldPosData.z};
            const float depth = worldPosData.w;
            const V3 normal = normalize(V3{normalRoughnessData.x, normalRoughnessData.y, normalRoughnessData.z});
            const V2 motionVector = {q16(inMotion[2 * idx]), q16(inMotion[2 * idx + 1])};

            const V2 currentUV = {(float(cx) + 0.5f) / float(w), (float(cy) + 0.5f) / float(h)};
            V2 prevUV;
            if (std::sqrt(motionVector.x * motionVector.x + motionVector.y * motionVector.y) > 0.0001f) {
                prevUV = {currentUV.x - motionVector.x, currentUV.y - motionVector.y};
            } else {                                                                     // worldToPrevUV
                const float* M = prevViewProj;
                const float px = ((M[0] * worldPos.x + M[4] * worldPos.y) + M[8] * worldPos.z) + M[12];
                const float py = ((M[1] * worldPos.x + M[5] * worldPos.y) + M[9] * worldPos.z) + M[13];
                const float pw = ((M[3] * worldPos.x + M[7] * worldPos.y) + M[11] * worldPos.z) + M[15];
                prevUV = {(px / pw) * 0.5f + 0.5f, (py / pw) * 0.5f + 0.5f};
            }

            V3 outputColor = currentColor;
            const float lum = luminance(currentColor);
            V2 outputMoments = {lum, lum * lum};
            float outputHistoryLength = 1.0f;

            const bool validReprojection = prevUV.x >= 0.0f && prevUV.x <= 1.0f && prevUV.y >= 0.0f && prevUV.y <= 1.0f && frameCount > 0;
            if (validReprojection) {
                const V3 historyColor = hist.sample(prevUV.x, prevUV.y);
                int pcx = int(prevUV.x * float(w)), pcy = int(prevUV.y * float(h));
                pcx = clampi(pcx, 0, w - 1); pcy = clampi(pcy, 0, h - 1);
                const V4 prevWorldPosData = pwpos.load(pcx, pcy);
                const V4 prevNormalData = pnrm.load(pcx, pcy);
                const float prevDepth = prevWorldPosData.w;
                const V3 prevNormal = normalize(V3{prevNormalData.x, prevNormalData.y, prevNormalData.z});

                const float absoluteDepthThreshold = S->depthThreshold * depth + 0.5f;
                const float depthDiff = std::fabs(depth - prevDepth);
                const bool depthValid = depthDiff < absoluteDepthThreshold;
                const float normalDot = dot(normal, prevNormal);
                const bool normalValid = normalDot > S->normalThreshold;
                const V3 prevWorldPos = {prevWorldPosData.x, prevWorldPosData.y, prevWorldPosData.z};
                const float worldPosDiff = length(worldPos - prevWorldPos);
                const bool worldPosValid = worldPosDiff < 2.0f;

                if (depthValid && normalValid && worldPosValid) {
                    const size_t pidx = size_t(pcy) * w + pcx;
                    const V2 prevMomentsData = {prevMoments[2 * pidx], prevMoments[2 * pidx + 1]};
                    const float prevHistLen = q16(prevHistoryLength[pidx]);

                    // computeNeighborhoodStatistics(coord, normal, depth, ...)
                    V3 m1 = splat(0.0f), m2 = splat(0.0f), minVal = splat(1e10f), maxVal = splat(-1e10f);
                    float totalWeight = 0.0f;
                    for (int dy = -1; dy <= 1; dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            const int sx = clampi(cx + dx, 0, w - 1), sy = clampi(cy + dy, 0, h - 1);
                            const V4 sampleWorldPos = wpos.load(sx, sy);
                            const V4 sampleNormalData = nrm.load(sx, sy);
                            const float sampleDepth = sampleWorldPos.w;
                            const V3 sampleNormal = {sampleNormalData.x, sampleNormalData.y, sampleNormalData.z};     // not normalised, :138
                            const float dd = std::fabs(depth - sampleDepth);
                            const float nd = dot(normal, sampleNormal);
                            const float depthWeight = dd < (depth * 0.02f + 0.1f) ? 1.0f : 0.0f;
                            const float normalWeight = nd > 0.9f ? 1.0f : 0.0f;
                            const float weight = depthWeight * normalWeight;
                            if (weight > 0.0f) {
                                const V3 sampleYCoCg = RGBToYCoCg(color.rgb(sx, sy));
                                m1 = m1 + sampleYCoCg * weight;
                                m2 = m2 + sampleYCoCg * sampleYCoCg * weight;
                                minVal = vmin(minVal, sampleYCoCg);
                                maxVal = vmax(maxVal, sampleYCoCg);
                                totalWeight += weight;
                            }
                        }
                    V3 mean, stdDev;
                    if (totalWeight > 0.0f) {
                        mean = m1 / totalWeight;
                        const V3 variance = vmax(m2 / totalWeight - mean * mean, splat(0.0f));
                        stdDev = vsqrt(variance);
                    } else {
                        mean = RGBToYCoCg(color.rgb(cx, cy));
                        stdDev = splat(0.1f);
                        minVal = mean; maxVal = mean;
                    }
                    const float gamma = S->varianceClipGamma;
                    V3 minC = mean - gamma * stdDev, maxC = mean + gamma * stdDev;
                    minC = vmax(minC, minVal - splat(0.05f));
                    maxC = vmin(maxC, maxVal + splat(0.05f));

                    const V3 historyYCoCg = RGBToYCoCg(historyColor);
                    V3 clippedYCoCg;                                                     // clipToAABB
                    {
                        const V3 center = 0.5f * (minC + maxC);
                        const V3 extents = 0.5f * (maxC - minC);
                        const V3 offset = historyYCoCg - center;
                        const V3 unitOffset = offset / vmax(extents, splat(0.0001f));
                        const float maxComponent = std::fmax(std::fmax(std::fabs(unitOffset.x), std::fabs(unitOffset.y)), std::fabs(unitOffset.z));
                        clippedYCoCg = maxComponent > 1.0f ? center + offset / maxComponent : historyYCoCg;
                    }
                    V3 clippedHistory = YCoCgToRGB(clippedYCoCg);
                    clippedHistory = vmax(clippedHistory, splat(0.0f));

                    const float historyLen = prevHistLen + 1.0f;
                    float alpha = S->temporalAlpha;
                    const float historyFactor = 1.0f / std::fmax(historyLen, 1.0f);
                    alpha = std::fmax(alpha, historyFactor);
                    const float lumCurrent = luminance(currentColor);
                    const float lumHistory = luminance(clippedHistory);
                    const float lumDiff = std::fabs(lumCurrent - lumHistory) / std::fmax(lumCurrent + lumHistory + 0.01f, 0.01f);
                    alpha = mixf(alpha, std::fmin(alpha + 0.2f, 0.5f), lumDiff * 0.3f);
                    alpha = clampf(alpha, S->temporalAlpha, 1.0f);
                    outputColor = mix3(clippedHistory, currentColor, alpha);

                    const float newLum = luminance(currentColor);
                    const V2 currentMoments = {newLum, newLum * newLum};
                    const float momentAlpha = std::fmax(S->momentAlpha, historyFactor);
                    outputMoments = {mixf(prevMomentsData.x, currentMoments.x, momentAlpha), mixf(prevMomentsData.y, currentMoments.y, momentAlpha)};
                    outputHistoryLength = std::fmin(historyLen, 64.0f);
                }
            }
            outputColor = vclamp(outputColor, 0.0f, 100.0f);
            outputMoments = {clampf(outputMoments.x, 0.0f, 10000.0f), clampf(outputMoments.y, 0.0f, 10000.0f)};
            outColor[4 * idx] = outputColor.x; outColor[4 * idx + 1] = outputColor.y; outColor[4 * idx + 2] = outputColor.z; outColor[4 * idx + 3] = 1.0f;
            outMoments[2 * idx] = outputMoments.x; outMoments[2 * idx + 1] = outputMoments.y;
            outHistoryLength[idx] = q16(outputHistoryLength);
        }
}

// variance.comp main() :101-144 with computeSpatialVariance :57-99
void orc_variance(const float* inColor, const float* inMoments, const float* inHistoryLength, const float* inWorldPosition,
                  const float* inNormalRoughness, int w, int h, const orc_denoise_settings* S, float* outVariance) {
    const Img4 color{inColor, w, h}, wpos{inWorldPosition, w, h};
    const Img4h nrm{inNormalRoughness, w, h};
    for (int cy = 0; cy < h; ++cy)
        for (int cx = 0; cx < w; ++cx) {
            const size_t idx = size_t(cy) * w + cx;
            const V2 moments = {inMoments[2 * idx], inMoments[2 * idx + 1]};
            const float historyLength = q16(inHistoryLength[idx]);
            const V4 worldPosData = wpos.load(cx, cy);
            const V4 normalData = nrm.load(cx, cy);
            const float centerDepth = worldPosData.w;
            const V3 centerNormal = normalize(V3{normalData.x, normalData.y, normalData.z});
            const float temporalVariance = std::fmax(moments.y - moments.x * moments.x, 0.0f);

            float m1 = 0.0f, m2 = 0.0f, totalWeight = 0.0f;                            // computeSpatialVariance
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const int sx = clampi(cx + dx, 0, w - 1), sy = clampi(cy + dy, 0, h - 1);
                    const V4 sampleWorldPos = wpos.load(sx, sy);
                    const V4 sampleNormalData = nrm.load(sx, sy);
                    const float sampleDepth = sampleWorldPos.w;
                    const V3 sampleNormal = normalize(V3{sampleNormalData.x, sampleNormalData.y, sampleNormalData.z});
                    const float depthDiff = std::fabs(centerDepth - sampleDepth);
                    const float normalDot = dot(centerNormal, sampleNormal);
                    const float depthWeight = std::exp(-depthDiff * depthDiff / (0.5f * 0.5f));
                    const float normalWeight = normalDot > 0.9f ? 1.0f : 0.0f;
                    const float weight = depthWeight * normalWeight;
                    if (weight > 0.01f) {
                        const float lum = luminance(color.rgb(sx, sy));
                        m1 += lum * weight;
                        m2 += lum * lum * weight;
                        totalWeight += weight;
                    }
                }
            float spatialVariance = 0.0f;
            if (totalWeight > 0.0f) {
                const float mean = m1 / totalWeight;
                spatialVariance = std::fmax(m2 / totalWeight - mean * mean, 0.0f);
            }
            const float minHistLen = float(std::max(S->minHistoryLength, 4));
            float historyWeight = clampf((historyLength - 1.0f) / minHistLen, 0.0f, 1.0f);
            historyWeight = historyWeight * historyWeight;
            float variance = mixf(spatialVariance, temporalVariance, historyWeight);
            if (historyLength < minHistLen) {
                const float boostFactor = mixf(S->varianceBoost, 1.0f, historyLength / minHistLen);
                variance *= boostFactor;
            }
            variance = std::fmax(variance, 0.0001f);
            outVariance[idx] = variance;
        }
}

// atrous.comp main() :154-220 with the weight functions :58-152; one iteration with push constants (stepSize, phi*)
void orc_atrous(const float* inColor, const float* inVariance, const float* inWorldPosition, const float* inNormalRoughness,
                int w, int h, int stepSize, float phiColor, float phiNormal, float phiDepth, float* outColor) {
    (void)phiNormal;                                                                     // pushed but unused by the shader
    const Img4 color{inColor, w, h}, wpos{inWorldPosition, w, h};
    const Img4h nrm{inNormalRoughness, w, h};
    const float kernel[3] = {1.0f, 2.0f / 3.0f, 1.0f / 6.0f};
    auto fastExp = [](float x) { return 1.0f / (1.0f + x); };
    for (int cy = 0; cy < h; ++cy)
        for (int cx = 0; cx < w; ++cx) {
            const size_t idx = size_t(cy) * w + cx;
            const V3 centerColor = color.rgb(cx, cy);
            const V4 centerWorldPosData = wpos.load(cx, cy);
            const V4 centerNormalData = nrm.load(cx, cy);
            const float centerVariance = inVariance[idx];
            const V3 centerWorldPos = {centerWorldPosData.x, centerWorldPosData.y, centerWorldPosData.z};
            const float centerDepth = centerWorldPosData.w;
            const V3 centerNormal = normalize(V3{centerNormalData.x, centerNormalData.y, centerNormalData.z});
            V3 outputColor;
            if (centerDepth > 9000.0f) {
                outputColor = centerColor;                                               // sky: stored as is, :174-177
            } else {
                V3 sumColor = splat(0.0f);
                float sumWeight = 0.0f;
                for (int i = 0; i < 25; i++) {
                    const int ox = i % 5 - 2, oy = i / 5 - 2;                            // offsets[i], :60-66
                    const int sx = clampi(cx + ox * stepSize, 0, w - 1), sy = clampi(cy + oy * stepSize, 0, h - 1);
                    const V3 sampleColor = color.rgb(sx, sy);
                    const V4 sampleWorldPosData = wpos.load(sx, sy);
                    const V4 sampleNormalData = nrm.load(sx, sy);
                    const V3 sampleWorldPos = {sampleWorldPosData.x, sampleWorldPosData.y, sampleWorldPosData.z};
                    const float sampleDepth = sampleWorldPosData.w;
                    const V3 sampleNormal = normalize(V3{sampleNormalData.x, sampleNormalData.y, sampleNormalData.z});
                    if (sampleDepth > 9000.0f) continue;
                    const float kernelWeight = kernel[std::abs(ox)] * kernel[std::abs(oy)];
                    float colorWeight;
                    {
                        const V3 diff = centerColor - sampleColor;
                        const float colorDistSq = dot(diff, diff);
                        const float baseSigma = 0.01f;
                        const float varianceSigma = phiColor * std::sqrt(std::fmax(centerVariance, 0.0f));
                        const float sigma = baseSigma + varianceSigma;
                        colorWeight = fastExp(colorDistSq / (2.0f * sigma * sigma + 1e-6f));
                    }
                    float normalWeight;
                    {
                        const float dotProduct = std::fmax(dot(centerNormal, sampleNormal), 0.0f);
                        const float threshold = 0.9f;
                        if (dotProduct < threshold) normalWeight = 0.0f;
                        else { const float t = (dotProduct - threshold) / (1.0f - threshold); normalWeight = t * t; }
                    }
                    float depthWeight;
                    {
                        const float depthDiff = std::fabs(centerDepth - sampleDepth);
                        const V3 posDiff = sampleWorldPos - centerWorldPos;
                        const float planeDistance = std::fabs(dot(posDiff, centerNormal));
                        const float effectiveDistance = std::fmax(depthDiff * 0.1f, planeDistance);
                        const float sigma = phiDepth * float(stepSize) + 0.1f;
                        if (effectiveDistance > sigma * 2.0f) depthWeight = 0.0f;
                        else depthWeight = fastExp(effectiveDistance * effectiveDistance / (sigma * sigma + 1e-6f));
                    }
                    const float weight = kernelWeight * colorWeight * normalWeight * depthWeight;
                    if (weight < 0.001f) continue;
                    sumColor = sumColor + sampleColor * weight;
                    sumWeight += weight;
                }
                outputColor = sumWeight > 0.01f ? sumColor / sumWeight : centerColor;
                outputColor = vmax(outputColor, splat(0.0f));
            }
            outColor[4 * idx] = outputColor.x; outColor[4 * idx + 1] = outputColor.y; outColor[4 * idx + 2] = outputColor.z; outColor[4 * idx + 3] = 1.0f;
        }
}

// taa.comp main() :109-220 with clipToAABB :82-92 and varianceClip :95-107
void orc_taa(const float* currentColor, const float* previousHistory, const float* motionVectors, int w, int h, uint32_t frameCount,
             float feedbackMin, float feedbackMax, float* outputColor, float* outputHistory) {
    const Img4 cur{currentColor, w, h}, hist{previousHistory, w, h};
    for (int py = 0; py < h; ++py)
        for (int px = 0; px < w; ++px) {
            const size_t idx = size_t(py) * w + px;
            const V2 uv = {(float(px) + 0.5f) / float(w), (float(py) + 0.5f) / float(h)};
            const V4 currentSample = cur.load(px, py);
            const V3 current = {currentSample.x, currentSample.y, currentSample.z};
            const V2 motion = {q16(motionVectors[2 * idx]), q16(motionVectors[2 * idx + 1])};
            const V2 prevUV = {uv.x - motion.x, uv.y - motion.y};
            const bool validHistory = prevUV.x >= 0.0f && prevUV.x <= 1.0f && prevUV.y >= 0.0f && prevUV.y <= 1.0f;
            const V3 history = hist.sample(prevUV.x, prevUV.y);
            V3 neighborhoodMin = splat(1e10f), neighborhoodMax = splat(-1e10f), neighborhoodSum = splat(0.0f), neighborhoodSumSq = splat(0.0f);
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const int sx = clampi(px + dx, 0, w - 1), sy = clampi(py + dy, 0, h - 1);
                    const V3 s = RGBToYCoCg(cur.rgb(sx, sy));
                    neighborhoodMin = vmin(neighborhoodMin, s);
                    neighborhoodMax = vmax(neighborhoodMax, s);
                    neighborhoodSum = neighborhoodSum + s;
                    neighborhoodSumSq = neighborhoodSumSq + s * s;
                }
            const V3 mean = neighborhoodSum / 9.0f;
            const V3 variance = (neighborhoodSumSq / 9.0f) - (mean * mean);
            const V3 stdDev = vsqrt(vmax(variance, splat(0.0f)));
            const V3 historyYCoCg = RGBToYCoCg(history);
            V3 clippedHistoryYCoCg;                                                      // varianceClip -> clipToAABB
            {
                const float gamma = 1.5f;
                V3 minC = mean - gamma * stdDev, maxC = mean + gamma * stdDev;
                minC = vmax(minC, neighborhoodMin);
                maxC = vmin(maxC, neighborhoodMax);
                const V3 center = 0.5f * (maxC + minC);
                const V3 extents = 0.5f * (maxC - minC);
                const V3 offset = historyYCoCg - center;
                const V3 ts = vabs(extents) / vmax(vabs(offset), splat(0.0001f));
                const float t = clampf(std::fmin(std::fmin(ts.x, ts.y), ts.z), 0.0f, 1.0f);
                clippedHistoryYCoCg = center + offset * t;
            }
            const V3 clippedHistory = YCoCgToRGB(clippedHistoryYCoCg);
            const float mvx = motion.x * float(w), mvy = motion.y * float(h);
            const float velocityLength = std::sqrt(mvx * mvx + mvy * mvy);
            const float velocityFactor = clampf(velocityLength / 10.0f, 0.0f, 1.0f);
            float feedback = mixf(feedbackMax, feedbackMin, velocityFactor);
            if (!validHistory || frameCount == 0) feedback = 0.0f;
            const float clipDist = length(clippedHistoryYCoCg - historyYCoCg);
            feedback *= 1.0f - clampf(clipDist * 2.0f, 0.0f, 0.5f);
            const V3 result = mix3(current, clippedHistory, feedback);
            const V3 blurred = mean;
            const V3 sharpened = current + 0.1f * (current - YCoCgToRGB(blurred));
            const V3 outputResult = mix3(sharpened, clippedHistory, feedback);
            outputColor[4 * idx] = outputResult.x; outputColor[4 * idx + 1] = outputResult.y; outputColor[4 * idx + 2] = outputResult.z; outputColor[4 * idx + 3] = currentSample.w;
            outputHistory[4 * idx] = result.x; outputHistory[4 * idx + 1] = result.y; outputHistory[4 * idx + 2] = result.z; outputHistory[4 * idx + 3] = 1.0f;
        }
}

// sharpen.comp main() :19-74: rgba8 in (sampled at texel centres, clamp to edge), rgba8 out
void orc_sharpen(const uint32_t* inputImage, int w, int h, float sharpenStrength, uint32_t* outputImage) {
    auto texel = [&](int x, int y) {
        const uint32_t p = inputImage[size_t(clampi(y, 0, h - 1)) * w + clampi(x, 0, w - 1)];
        return V3{float(p & 0xFFu) / 255.0f, float((p >> 8) & 0xFFu) / 255.0f, float((p >> 16) & 0xFFu) / 255.0f};
    };
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const V3 a = texel(x - 1, y - 1), b = texel(x, y - 1), c = texel(x + 1, y - 1);
            const V3 d = texel(x - 1, y), e = texel(x, y), f = texel(x + 1, y);
            const V3 g = texel(x - 1, y + 1), hh = texel(x, y + 1), i = texel(x + 1, y + 1);
            const V3 blur = (1.0f * (a + c + g + i) + 2.0f * (b + d + f + hh) + 4.0f * e) / 16.0f;
            const V3 detail = e - blur;
            const float intensity = sharpenStrength * 3.0f;
            V3 result = e + detail * intensity;
            result = vclamp(result, 0.0f, 1.0f);
            auto unorm = [](float v) { return uint32_t(v * 255.0f + 0.5f); };
            outputImage[size_t(y) * w + x] = unorm(result.x) | (unorm(result.y) << 8) | (unorm(result.z) << 16) | 0xFF000000u;
        }
}

}  // extern "C"
