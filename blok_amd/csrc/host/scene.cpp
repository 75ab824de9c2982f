// Camera basis (mirror of blok::Camera) and the synthetic benchmark scene G(N, seed).
//
// Camera: reference blok/include/camera.hpp:25-42 builds forward/right/up from yaw/pitch with
// glm (absent from the reference tree: external/glm is an empty submodule), and the reference's
// compute backend reduces a Camera to {pos, forward, right, up, tan(fov/2), aspect}
// (reference blok/src/cuda_tracer.cu:404-415).  That basis is this backend's camera contract.
//
// Scene: not reference behaviour.  Integer-only, seedable, identical on every host.
#include "blok_world.h"
#include "../common/taa_jitter.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

struct V3 { float x, y, z; };
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline V3 unit(V3 v) {
    const float len = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return {v.x / len, v.y / len, v.z / len};
}
constexpr float kDegToRad = 0.01745329251994329576923690768489f;  // glm::radians factor

// ---- scene -------------------------------------------------------------------------------
inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
inline uint32_t hash3(uint32_t x, uint32_t y, uint32_t z, uint32_t s) {
    return fmix32(x * 0x9E3779B1u ^ y * 0x85EBCA77u ^ z * 0xC2B2AE3Du ^ s);
}

struct Terrain {
    uint32_t n, seed;
    std::vector<int32_t> h;  // n*n heights, index x + z*n
    Terrain(uint32_t n_, uint32_t seed_) : n(n_), seed(seed_), h(static_cast<size_t>(n_) * n_) {
        for (uint32_t z = 0; z < n; ++z)
            for (uint32_t x = 0; x < n; ++x) h[x + static_cast<size_t>(z) * n] = eval(x, z);
    }
    // 4 octaves of fixed-point bilinear value noise, weights 8:4:2:1, cells n/4 .. n/32
    int32_t eval(uint32_t x, uint32_t z) const {
        uint64_t acc = 0;
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t cell = std::max(1u, n >> (2 + k));
            const uint32_t ix = x / cell, iz = z / cell;
            const uint64_t fx = x % cell, fz = z % cell;
            const uint64_t v00 = hash3(ix, k, iz, seed) & 0xFFFFu, v10 = hash3(ix + 1, k, iz, seed) & 0xFFFFu;
            const uint64_t v01 = hash3(ix, k, iz + 1, seed) & 0xFFFFu, v11 = hash3(ix + 1, k, iz + 1, seed) & 0xFFFFu;
            const uint64_t top = v00 * (cell - fx) + v10 * fx;
            const uint64_t bot = v01 * (cell - fx) + v11 * fx;
            const uint64_t val = (top * (cell - fz) + bot * fz) / (static_cast<uint64_t>(cell) * cell);
            acc += val * (8u >> k);
        }
        const uint64_t fbm = acc / 15u;  // 0..65535
        return static_cast<int32_t>(n / 8 + ((fbm * (3ull * n / 8)) >> 16));
    }
    int32_t at(int32_t x, int32_t z) const {
        if (x < 0 || z < 0 || x >= static_cast<int32_t>(n) || z >= static_cast<int32_t>(n)) return -1;
        return h[static_cast<size_t>(x) + static_cast<size_t>(z) * n];
    }
};

template <class Emit>
uint64_t generate(uint32_t n, uint32_t seed, Emit&& emit) {
    const Terrain terrain(n, seed);
    const int32_t N = static_cast<int32_t>(n);
    auto material = [&](int32_t x, int32_t y, int32_t z) {
        return 1u + hash3(static_cast<uint32_t>(x) >> 4, static_cast<uint32_t>(y) >> 4,
                          static_cast<uint32_t>(z) >> 4, seed) % 255u;
    };
    uint64_t count = 0;
    // terrain shell: solid iff y <= H(x,z); kept iff a 6-neighbour is empty (outside = empty
    // on the four sides, solid below y = 0)
    for (int32_t z = 0; z < N; ++z)
        for (int32_t x = 0; x < N; ++x) {
            const int32_t top = std::min(terrain.at(x, z), N - 1);
            const int32_t low = std::min({terrain.at(x - 1, z), terrain.at(x + 1, z),
                                          terrain.at(x, z - 1), terrain.at(x, z + 1)});
            const int32_t from = std::max(0, std::min(top, low + 1));
            for (int32_t y = from; y <= top; ++y) { emit(x, y, z, material(x, y, z)); ++count; }
        }
    // 64 hashed spheres, shell only
    const int32_t rmin = std::max(1, N / 128), rmax = std::max(2, N / 24);
    for (uint32_t i = 0; i < 64; ++i) {
        const int32_t cx = static_cast<int32_t>(hash3(i, 1, 0, seed) % n);
        const int32_t cz = static_cast<int32_t>(hash3(i, 2, 0, seed) % n);
        const int32_t cy = N / 2 + static_cast<int32_t>(hash3(i, 3, 0, seed) % std::max(1u, 3 * n / 8));
        const int32_t r = rmin + static_cast<int32_t>(hash3(i, 4, 0, seed) % static_cast<uint32_t>(rmax - rmin + 1));
        const int64_t r2 = static_cast<int64_t>(r) * r;
        auto inside = [&](int32_t x, int32_t y, int32_t z) {
            const int64_t dx = x - cx, dy = y - cy, dz = z - cz;
            return dx * dx + dy * dy + dz * dz <= r2;
        };
        for (int32_t z = std::max(0, cz - r); z <= std::min(N - 1, cz + r); ++z)
            for (int32_t y = std::max(0, cy - r); y <= std::min(N - 1, cy + r); ++y)
                for (int32_t x = std::max(0, cx - r); x <= std::min(N - 1, cx + r); ++x) {
                    if (!inside(x, y, z)) continue;
                    if (inside(x - 1, y, z) && inside(x + 1, y, z) && inside(x, y - 1, z) &&
                        inside(x, y + 1, z) && inside(x, y, z - 1) && inside(x, y, z + 1))
                        continue;
                    emit(x, y, z, material(x, y, z));
                    ++count;
                }
    }
    return count;
}

}  // namespace

extern "C" {

int blok_camera_from_yaw_pitch(const float pos[3], float yaw_deg, float pitch_deg, float fov_deg,
                               uint32_t width, uint32_t height, blok_camera* out) {
    if (!pos || !out || !width || !height) return BLOK_ERR_INVALID_ARG;
    const float yaw = yaw_deg * kDegToRad, pitch = pitch_deg * kDegToRad;
    const V3 f = unit({std::cos(yaw) * std::cos(pitch), std::sin(pitch), std::sin(yaw) * std::cos(pitch)});
    const V3 r = unit(cross(f, {0.0f, 1.0f, 0.0f}));
    const V3 u = unit(cross(r, f));
    out->pos[0] = pos[0]; out->pos[1] = pos[1]; out->pos[2] = pos[2];
    out->fwd[0] = f.x; out->fwd[1] = f.y; out->fwd[2] = f.z;
    out->right[0] = r.x; out->right[1] = r.y; out->right[2] = r.z;
    out->up[0] = u.x; out->up[1] = u.y; out->up[2] = u.z;
    // reference blok/src/cuda_tracer.cu:405-406 (note its 3.14159f)
    out->aspect = static_cast<float>(width) / static_cast<float>(height);
    out->tan_half_fov = std::tan(0.5f * fov_deg * 3.14159f / 180.0f);
    return BLOK_OK;
}

int blok_camera_look_at(const float pos[3], const float target[3], float fov_deg,
                        uint32_t width, uint32_t height, blok_camera* out) {
    if (!pos || !target) return BLOK_ERR_INVALID_ARG;
    const float dx = target[0] - pos[0], dy = target[1] - pos[1], dz = target[2] - pos[2];
    const float len = std::sqrt(dx * dx + dy * dy + dz * dz);
    if (!(len > 0.0f)) return BLOK_ERR_INVALID_ARG;
    float pitch = std::asin(dy / len) / kDegToRad;
    pitch = std::min(89.0f, std::max(-89.0f, pitch));       // reference camera.hpp:77-78
    const float yaw = std::atan2(dz, dx) / kDegToRad;
    return blok_camera_from_yaw_pitch(pos, yaw, pitch, fov_deg, width, height, out);
}

// ---- Camera matrices and the TAA jitter sequence (reference blok/include/camera.hpp:49-59, blok/src/renderer_postprocess.cpp:208-268).
// glm is not in the reference tree (external/glm is an empty submodule); these are glm's published formulas for
// lookAt (right-handed), perspective (right-handed, depth 0..1: camera.hpp:9 defines GLM_FORCE_DEPTH_ZERO_TO_ONE) and a
// cofactor inverse, in float, column-major (M[col * 4 + row]).

void blok_camera_view(const blok_camera* c, float M[16]) {
    if (!c || !M) return;
    // glm::lookAt(pos, pos + f, up): f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f)
    const V3 f = unit({c->fwd[0], c->fwd[1], c->fwd[2]});
    const V3 s = unit(cross(f, {c->up[0], c->up[1], c->up[2]}));
    const V3 u = cross(s, f);
    const V3 e = {c->pos[0], c->pos[1], c->pos[2]};
    auto dot = [](V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    const float m[16] = {s.x, u.x, -f.x, 0.0f,  s.y, u.y, -f.y, 0.0f,  s.z, u.z, -f.z, 0.0f,  -dot(s, e), -dot(u, e), dot(f, e), 1.0f};
    for (int i = 0; i < 16; ++i) M[i] = m[i];
}

void blok_camera_projection(const blok_camera* c, float z_near, float z_far, float M[16]) {
    if (!c || !M) return;
    // glm::perspectiveRH_ZO(fovy, aspect, near, far) with p[1][1] *= -1 (camera.hpp:57, "vulkan requirement")
    for (int i = 0; i < 16; ++i) M[i] = 0.0f;
    M[0] = 1.0f / (c->aspect * c->tan_half_fov);
    M[5] = -(1.0f / c->tan_half_fov);
    M[10] = z_far / (z_near - z_far);
    M[11] = -1.0f;
    M[14] = -(z_far * z_near) / (z_far - z_near);
}

int blok_mat4_inverse(const float m[16], float out[16]) {
    if (!m || !out) return BLOK_ERR_INVALID_ARG;
    double inv[16];
    const double a[16] = {m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]};
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (det == 0.0) return BLOK_ERR_INVALID_ARG;
    for (int i = 0; i < 16; ++i) out[i] = static_cast<float>(inv[i] / det);
    return BLOK_OK;
}

void blok_taa_jitter(uint32_t frame_index, float out_px[2]) {
    if (out_px) blok::taa_jitter_px(frame_index, out_px);
}

void blok_taa_jitter_clip(const float jitter_px[2], uint32_t width, uint32_t height, float out_clip[2]) {
    if (!jitter_px || !out_clip || !width || !height) return;
    out_clip[0] = (2.0f * jitter_px[0]) / static_cast<float>(width);        // getJitterClipSpace, :234-241
    out_clip[1] = (2.0f * jitter_px[1]) / static_cast<float>(height);
}

void blok_jittered_projection(const float proj[16], const float jitter_px[2], uint32_t width, uint32_t height, float out[16]) {
    if (!proj || !jitter_px || !out) return;
    float clip[2] = {0.0f, 0.0f};
    blok_taa_jitter_clip(jitter_px, width, height, clip);
    for (int i = 0; i < 16; ++i) out[i] = proj[i];
    out[8] += clip[0];              // jitteredProj[2][0], :264
    out[9] += clip[1];              // jitteredProj[2][1], :265
}

int blok_scene_generate(blok_world* w, uint32_t n, uint32_t seed, uint64_t* out_n_voxels) {
    if (!w || n < 16 || n > 4096 || (n & (n - 1))) return BLOK_ERR_INVALID_ARG;
    std::vector<int32_t> xyz;
    std::vector<uint32_t> mats;
    const size_t batch = 1u << 20;
    xyz.reserve(3 * batch);
    mats.reserve(batch);
    int rc = BLOK_OK;
    const uint64_t count = generate(n, seed, [&](int32_t x, int32_t y, int32_t z, uint32_t m) {
        xyz.push_back(x); xyz.push_back(y); xyz.push_back(z);
        mats.push_back(m);
        if (mats.size() == batch) {
            if (rc == BLOK_OK) rc = blok_world_set_voxels(w, xyz.data(), mats.data(), mats.size());
            xyz.clear(); mats.clear();
        }
    });
    if (rc == BLOK_OK && !mats.empty()) rc = blok_world_set_voxels(w, xyz.data(), mats.data(), mats.size());
    if (out_n_voxels) *out_n_voxels = count;  // writes, including terrain/sphere overlaps
    return rc;
}

int blok_scene_generate_dense(uint32_t n, uint32_t seed, uint32_t* ids, uint64_t* out_n_voxels) {
    if (!ids || n < 16 || n > 1024 || (n & (n - 1))) return BLOK_ERR_INVALID_ARG;     // 1024^3 ids = 4 GiB
    const size_t total = static_cast<size_t>(n) * n * n;
    std::fill(ids, ids + total, 0u);
    generate(n, seed, [&](int32_t x, int32_t y, int32_t z, uint32_t m) {
        ids[static_cast<size_t>(x) + static_cast<size_t>(y) * n + static_cast<size_t>(z) * n * n] = m;
    });
    if (out_n_voxels) {
        uint64_t filled = 0;
        for (size_t i = 0; i < total; ++i) filled += ids[i] != 0u;
        *out_n_voxels = filled;
    }
    return BLOK_OK;
}

int blok_scene_materials(uint32_t seed, blok_material* out) {
    if (!out) return BLOK_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < 256; ++i) {
        const uint32_t h = hash3(i, 77, 0, seed);
        blok_material m{};
        m.albedo[0] = static_cast<float>(h & 0xFFu) / 255.0f;
        m.albedo[1] = static_cast<float>((h >> 8) & 0xFFu) / 255.0f;
        m.albedo[2] = static_cast<float>((h >> 16) & 0xFFu) / 255.0f;
        // roughness 0.5 -> 127, metallic 0, type diffuse, alpha 1 -> 15, specular 0.5 -> 127
        // (bit layout: reference blok/include/material.hpp:100-106)
        m.flags = (0u << 24) | (127u << 16) | (0u << 12) | (15u << 8) | 127u;
        m.emission[0] = m.emission[1] = m.emission[2] = 0.0f;
        m.ior = 0.0f;
        out[i] = m;
    }
    return BLOK_OK;
}

int blok_scene_camera(uint32_t n, uint32_t seed, int pose, uint32_t width, uint32_t height, blok_camera* out) {
    if (!out || n < 16) return BLOK_ERR_INVALID_ARG;
    const float N = static_cast<float>(n);
    if (pose == 0) {
        const float pos[3] = {-0.35f * N, 0.85f * N, -0.35f * N};
        const float target[3] = {0.5f * N, 0.25f * N, 0.5f * N};
        return blok_camera_look_at(pos, target, 60.0f, width, height, out);
    }
    if (pose == 1) {
        Terrain probe(std::min(n, 4096u), seed);
        const float ground = static_cast<float>(probe.at(static_cast<int32_t>(n / 2), static_cast<int32_t>(n / 8)));
        const float pos[3] = {0.5f * N + 0.37f, ground + 4.25f, 0.125f * N + 0.41f};
        return blok_camera_from_yaw_pitch(pos, 90.0f, -5.0f, 60.0f, width, height, out);
    }
    if (pose == 2) {
        const float pos[3] = {0.5f * N + 0.29f, 1.5f * N, 0.5f * N + 0.31f};
        return blok_camera_from_yaw_pitch(pos, 0.0f, -89.0f, 60.0f, width, height, out);
    }
    return BLOK_ERR_INVALID_ARG;
}

}  // extern "C"
