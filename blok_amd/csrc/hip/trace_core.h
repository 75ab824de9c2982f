// Per-ray walk of the 64-tree: the body of the gfx950 trace kernel (see trace_kernels.h for the
// semantics).  Written against a handful of HIP device intrinsics; tests/host_harness.cpp compiles the
// same text for the host (BLOK_TRACE_HOST_HARNESS) to run it under sanitizers and to debug without a
// GPU.  The shipped library never builds or calls the host form.
#ifndef BLOK_TRACE_CORE_H
#define BLOK_TRACE_CORE_H

#include "trace_kernels.h"

#ifdef BLOK_TRACE_HOST_HARNESS
#define BLOK_DEV inline
#else
#define BLOK_DEV __device__ __forceinline__
#endif

// Optional event hook, defined only by the host harness' statistics build.
#ifndef BLOK_STAT
#define BLOK_STAT(event, level)
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace blok {

// One IEEE-754 binary32 operation each, round-to-nearest-even, never fused.  (HIP's __fmul_rn & co
// are plain operators on this toolchain and __fsqrt_rn is the *native* approximate square root, so
// they are not used; `/` and __builtin_sqrtf are correctly rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.)
BLOK_DEV float rn_add(float a, float b) { return a + b; }
BLOK_DEV float rn_sub(float a, float b) { return a - b; }
BLOK_DEV float rn_mul(float a, float b) { return a * b; }
BLOK_DEV float rn_div(float a, float b) { return a / b; }
BLOK_DEV float rn_sqrt(float a) { return __builtin_sqrtf(a); }

// intersect.rint:79
BLOK_DEV float safe_inv(float d) {
    return rn_div(1.0f, fabsf(d) < 1e-6f ? 1e-6f : d);
}

// RGBA8 of a first hit through hit.rchit's material fetch (hit.rchit:58-67): albedo of
// materials[min(id, 65535)] scaled by a fixed per-face factor; a constant sky colour on a miss.
constexpr uint32_t kSkyRgba = 0xFF000000u | (230u << 16) | (200u << 8) | 160u;
BLOK_DEV uint32_t shade_rgba(const blok_material* table, uint32_t n_materials, uint32_t material_id, uint32_t face) {
    const uint32_t id = material_id < 65535u ? material_id : 65535u;
    float r = 1.0f, g = 0.0f, b = 1.0f;                              // out-of-table ids show magenta
    if (id < n_materials) { r = table[id].albedo[0]; g = table[id].albedo[1]; b = table[id].albedo[2]; }
    const float k = face == 2u ? 1.0f : (face < 2u ? 0.8f : (face == 3u ? 0.4f : 0.6f));
    const uint32_t R = static_cast<uint32_t>(fminf(r * k, 1.0f) * 255.0f + 0.5f);
    const uint32_t G = static_cast<uint32_t>(fminf(g * k, 1.0f) * 255.0f + 0.5f);
    const uint32_t B = static_cast<uint32_t>(fminf(b * k, 1.0f) * 255.0f + 0.5f);
    return 0xFF000000u | (B << 16) | (G << 8) | R;
}

// Where a lane's results go; either pointer may be null.
struct Sink { blok_hit* hit; uint32_t* rgba; };

BLOK_DEV void write_miss(const Sink& dst) {
    // t = -1 (miss.rmiss:25-27), material 0, voxel 0, face 0xFF, hit 0
    if (dst.hit) *reinterpret_cast<uint4*>(dst.hit) = make_uint4(0xBF800000u, 0u, 0u, 0x00FF0000u);
    if (dst.rgba) *dst.rgba = kSkyRgba;
}

struct NodeRec { uint32_t lo, hi, base; };

BLOK_DEV bool mask_bit(const NodeRec& n, uint32_t bit) {
    const uint32_t word = bit < 32u ? n.lo : n.hi;
    return (word >> (bit & 31u)) & 1u;
}
BLOK_DEV uint32_t mask_rank(const NodeRec& n, uint32_t bit) {
    // number of set bits below `bit`
    const uint32_t below_lo = bit < 32u ? (n.lo & ((1u << bit) - 1u)) : n.lo;
    const uint32_t below_hi = bit < 32u ? 0u : (n.hi & ((1u << (bit & 31u)) - 1u));
    return __popc(below_lo) + __popc(below_hi);
}

struct RayIn { float ox, oy, oz, dx, dy, dz, tmin, tmax; };


// Two-bit digit of a (non-negative) tree coordinate at bit offset `shift`: one v_bfe_u32.
BLOK_DEV uint32_t digit2(uint32_t q, uint32_t shift) {
#ifdef BLOK_TRACE_HOST_HARNESS
    return (q >> shift) & 3u;
#else
    return __builtin_amdgcn_ubfe(q, shift, 2u);
#endif
}

// The walk runs in MIRRORED tree coordinates: on an axis the ray travels in the negative direction the
// coordinate is reflected (q = W - p), so in q-space every ray travels towards +q on every axis, the far
// plane of a cell is always q + size and a step is always +size.  A mirrored plane q maps back to the
// world plane  base + sgn * q  (sgn = +1: base = origin; sgn = -1: base = origin + W) before T is
// evaluated, so T is the same canonical function of the same world integer as without mirroring, and
// a node's child bit is the mirrored digit triple XOR a per-ray constant.
//
// A mirrored coordinate is held as the binary32 number  f = 2^23 + q  (0 <= q <= W <= 2^14): its mantissa
// field IS q (ulp = 1 throughout [2^23, 2^24)), so the digit of a level is one v_bfe of the float's bits, the
// level of a crossed boundary one v_ffbl, re-aligning to a coarser cell one v_and, and stepping by a cell size
// an exact float add.  The world plane is  fma(sgn, f, base - sgn * 2^23)  — one fused operation whose exact
// result, the integer base + sgn * q, is representable, hence equal to float(int plane) of the integer
// formulation bit for bit — so a plane costs three full-rate float instructions where integer coordinates cost
// v_mad_i32_i24 + v_cvt_f32_i32 (both half-rate on gfx950) + sub + mul.
constexpr float kCoordBias = 8388608.0f;            // 2^23
constexpr uint32_t kCoordBits = 0x4B000000u;        // bits of 2^23

struct Axis {
    float o, inv;     // ray origin component, safe inverse direction (intersect.rint:79)
    float sgn, c;     // mirrored coordinate f = 2^23 + q  ->  world plane fma(sgn, f, c), c = base - sgn * 2^23
};

BLOK_DEV float exact_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// T(a, q) for the mirrored plane held in f: t = fl(fl(float(world plane) - o) * inv), intersect.rint:48-49,179-180.
BLOK_DEV float plane_t(const Axis& a, float f) {
    return rn_mul(rn_sub(exact_fma(a.sgn, f, a.c), a.o), a.inv);
}

// Cell size 4^lvl as a float: 2^(2 lvl).
BLOK_DEV float cell_size(uint32_t lvl) { return __uint_as_float(0x3F800000u + (lvl << 24)); }

// One axis of "enter a node" whose near corner is f and whose children have size s: how many of the three
// interior planes have T <= tS (binary search, T is monotone in q), i.e. which child slab holds the ray at tS,
// and the T of that slab's far plane.  t_far comes in as the node's own far plane.  s2 = 2 s, s3 = 3 s.
BLOK_DEV void enter_axis(const Axis& a, float& f, float& t_far, float s, float s2, float s3, float tS) {
    const float m2 = plane_t(a, f + s2);
    const bool g = m2 <= tS;
    const float mq = plane_t(a, f + (g ? s3 : s));
    const bool g2 = mq <= tS;
    const float inner = g ? t_far : m2;          // far plane if mq is already crossed
    t_far = g2 ? inner : mq;
    f += (g ? s2 : 0.0f) + (g2 ? s : 0.0f);
}

// What closest-hit sees of a procedural hit (intersect.rint:138-141, hit.rchit:58-74), in registers.
struct HitInfo {
    bool found;
    float t;
    uint32_t material, face;
    int vx, vy, vz;          // world voxel
    NodeRec brick;           // the node record of the voxel's 4x4x4 brick (what walk_resume needs besides the LDS stack)
};

// What a walk needs of its ray: the three axes (origin, safe inverse direction, mirroring) and the child-bit constant.
struct WalkRay {
    Axis ax, ay, az;
    uint32_t mirror;
};
// Where a walk is: the current cell (mirrored min corner, level, parent node), the T of its far planes, the ray parameter at which it
// was entered, and — once `found` — the reported voxel (cell, node, bit, tCur).
struct WalkState {
    float fx, fy, fz, tFx, tFy, tFz, tCur, size;
    uint32_t lvl, bit;
    NodeRec node;
    bool walking, found;     // walking: the loop still has work for this lane
};

// ox, oy, oz and the safe inverse direction (intersect.rint:79) -> the axes.
BLOK_DEV WalkRay walk_ray(const TraceArgs& A, float ox, float oy, float oz, float ix, float iy, float iz) {
    WalkRay R;
    const int W = 1 << (2 * A.levels);
    R.ax.o = ox; R.ay.o = oy; R.az.o = oz;
    R.ax.inv = ix; R.ay.inv = iy; R.az.inv = iz;
    const bool negx = !(ix > 0.0f), negy = !(iy > 0.0f), negz = !(iz > 0.0f);
    // world plane = (base + sgn q) * voxel_size with a power-of-two voxel_size: sgn and c carry the factor, every product exact
    const float vs = A.voxel_size;
    R.ax.sgn = negx ? -vs : vs; R.ay.sgn = negy ? -vs : vs; R.az.sgn = negz ? -vs : vs;
    // c = (base -+ 2^23) * voxel_size: an integer below 2^24 in magnitude times a power of two, exact
    R.ax.c = (static_cast<float>(A.origin[0] + (negx ? W : 0)) + (negx ? kCoordBias : -kCoordBias)) * vs;
    R.ay.c = (static_cast<float>(A.origin[1] + (negy ? W : 0)) + (negy ? kCoordBias : -kCoordBias)) * vs;
    R.az.c = (static_cast<float>(A.origin[2] + (negz ? W : 0)) + (negz ? kCoordBias : -kCoordBias)) * vs;
    R.mirror = (negx ? 3u : 0u) | (negy ? 12u : 0u) | (negz ? 48u : 0u);
    return R;
}

// The walk's start for the interval [tmin, tmax): world box, then the root's start cell.  Not walking = the interval misses the box.
BLOK_DEV void walk_enter(const TraceArgs& A, const WalkRay& R, float tmin, float tmax, WalkState& s) {
    const uint32_t L = A.levels;
    // world box: near planes q = 0, far planes q = W (T is monotone in q, so no min/max is needed)
    const float fW = kCoordBias + static_cast<float>(1 << (2 * L));
    s.tFx = plane_t(R.ax, fW); s.tFy = plane_t(R.ay, fW); s.tFz = plane_t(R.az, fW);
    s.tCur = fmaxf(fmaxf(fmaxf(plane_t(R.ax, kCoordBias), plane_t(R.ay, kCoordBias)), plane_t(R.az, kCoordBias)), tmin);
    s.found = false; s.bit = 0u;
    s.fx = kCoordBias; s.fy = kCoordBias; s.fz = kCoordBias;   // mirrored min corner of the current cell (2^23 + q)
    s.lvl = L - 1;                        // current cells have size 4^lvl; `node` is their parent
    s.size = cell_size(s.lvl);
    s.node.lo = s.node.hi = s.node.base = 0u;
    s.walking = s.tCur < fminf(fminf(fminf(s.tFx, s.tFy), s.tFz), tmax);
    if (!s.walking) return;
    const uint4 q = A.nodes[0];
    s.node.lo = q.x; s.node.hi = q.y; s.node.base = q.z;
    const float s2 = s.size + s.size, s3 = s2 + s.size;
    enter_axis(R.ax, s.fx, s.tFx, s.size, s2, s3, s.tCur);
    enter_axis(R.ay, s.fy, s.tFy, s.size, s2, s3, s.tCur);
    enter_axis(R.az, s.fz, s.tFz, s.size, s2, s3, s.tCur);
}

// The loop: from the state walk_enter left to the first reported voxel, the end of the interval or the world's far side.
// kCapped: at most `cap` trips; a lane cut off leaves with `walking` still set and tCur = the parameter at which it entered the cell it stands in —
// a walk of the same ray from the root with tmin = that tCur reports what this one would have (every voxel before it has an empty clipped
// interval, every one after it is entered at or after tCur: the argument of the beam pre-pass).  path_core.h: the bounce rounds' tail pool.
template <bool kCapped = false>
BLOK_DEV void walk_loop(const TraceArgs& A, const WalkRay& R, const float tmax, WalkState& s, uint4* stk, [[maybe_unused]] const uint32_t cap = 0u) {
    const uint32_t L = A.levels;
    // Invariant: tCur starts at max(world entry, tmin) and never decreases — a cell's far planes are never
    // before the plane through which it was entered (T is monotone along each axis and the start cell of a
    // node only counts planes with T <= tCur as crossed) — so max(tCur, tmin) == tCur throughout and the
    // reported t of a voxel, max(entry, tmin), is tCur itself.
    float fx = s.fx, fy = s.fy, fz = s.fz, tFx = s.tFx, tFy = s.tFy, tFz = s.tFz, tCur = s.tCur, size = s.size;
    uint32_t lvl = s.lvl, bit = s.bit;
    NodeRec node = s.node;
    bool found = false;
    const bool walking = s.walking;
    [[maybe_unused]] uint32_t trips = 0u;
    [[maybe_unused]] bool cut = false;
    while (walking) {
        if constexpr (kCapped) { if (trips >= cap) { cut = true; break; } ++trips; }
        BLOK_STAT(0, lvl);
        const uint32_t shift = 2 * lvl;
        bit = (digit2(__float_as_uint(fx), shift) | (digit2(__float_as_uint(fy), shift) << 2) | (digit2(__float_as_uint(fz), shift) << 4)) ^ R.mirror;
        const bool occupied = mask_bit(node, bit);
        if (occupied && lvl != 0) {
            // descend: remember the node we are leaving, fetch the child, pick its start cell
            BLOK_STAT(1, lvl);
            stk[(lvl - 1) * kBlock] = make_uint4(node.lo, node.hi, node.base, 0u);     // node of level lvl+1
            const uint4 c = A.nodes[node.base + mask_rank(node, bit)];
            node.lo = c.x; node.hi = c.y; node.base = c.z;
            lvl -= 1;
            size *= 0.25f;
            const float s2 = size + size, s3 = s2 + size;
            enter_axis(R.ax, fx, tFx, size, s2, s3, tCur);     // tCur >= tmin always (see the invariant above the loop)
            enter_axis(R.ay, fy, tFy, size, s2, s3, tCur);
            enter_axis(R.az, fz, tFz, size, s2, s3, tCur);
            continue;
        }
        const float tExit = fminf(fminf(tFx, tFy), tFz);
        if (occupied) {
            // a filled voxel: reported iff its clipped interval is non-empty (intersect.rint:189-193)
            if (tCur < fminf(tExit, tmax)) { found = true; break; }
        }
        // step: cross the nearest far plane (x, then y, then z on ties)
        BLOK_STAT(2, lvl);
        tCur = tExit;
        if (!(tCur < tmax)) break;
        const bool sx = tFx == tExit;
        const bool sy = !sx && tFy == tExit;
        const bool sz = !sx && !sy;
        fx += sx ? size : 0.0f; fy += sy ? size : 0.0f; fz += sz ? size : 0.0f;
        // the stepped coordinate is now a multiple of 4^k for the level k whose cell boundary was crossed (its mantissa
        // field is q > 0, so the lowest set bit of the float's bits is the lowest set bit of q)
        const uint32_t up = static_cast<uint32_t>(__ffs(static_cast<int>(__float_as_uint(sx ? fx : (sy ? fy : fz)))) - 1) >> 1;
        if (up != lvl) {
            BLOK_STAT(3, lvl);
            if (up >= L) break;                                    // left the world box
            lvl = up;
            size = cell_size(lvl);
            const uint32_t keep = ~((1u << (2 * up)) - 1u);        // clears mantissa bits only: the exponent field stays
            fx = __uint_as_float(__float_as_uint(fx) & keep); fy = __uint_as_float(__float_as_uint(fy) & keep); fz = __uint_as_float(__float_as_uint(fz) & keep);
            const uint4 c = stk[(lvl - 1) * kBlock];               // node of level lvl+1
            node.lo = c.x; node.hi = c.y; node.base = c.z;
        }
        tFx = plane_t(R.ax, fx + size); tFy = plane_t(R.ay, fy + size); tFz = plane_t(R.az, fz + size);
    }
    s.fx = fx; s.fy = fy; s.fz = fz; s.tCur = tCur; s.bit = bit; s.node = node; s.found = found;
    if constexpr (kCapped) s.walking = cut; else s.walking = false;
}

// The reported voxel of a finished walk: intersect.rint:136-141, hit.rchit:58-74.  r: the ray itself (origin, direction).
BLOK_DEV HitInfo walk_hit(const TraceArgs& A, const RayIn& r, const WalkRay& R, const WalkState& s) {
    HitInfo out;
    const int W = 1 << (2 * A.levels);
    const float vs = A.voxel_size;
    const bool negx = !(R.ax.inv > 0.0f), negy = !(R.ay.inv > 0.0f), negz = !(R.az.inv > 0.0f);
    const float tc = s.tCur;                                          // = max(entry, tmin), intersect.rint:189,141
    const uint32_t material = A.materials[s.node.base + mask_rank(s.node, s.bit)];
    // world voxel = mirrored cell un-mirrored: base + sgn * q - (negative ? 1 : 0)
    const int qx = static_cast<int>(__float_as_uint(s.fx) & 0x7FFFFFu), qy = static_cast<int>(__float_as_uint(s.fy) & 0x7FFFFFu), qz = static_cast<int>(__float_as_uint(s.fz) & 0x7FFFFFu);
    const int vx = negx ? A.origin[0] + W - qx - 1 : A.origin[0] + qx;
    const int vy = negy ? A.origin[1] + W - qy - 1 : A.origin[1] + qy;
    const int vz = negz ? A.origin[2] + W - qz - 1 : A.origin[2] + qz;
    const float hx = rn_add(r.ox, rn_mul(r.dx, tc));
    const float hy = rn_add(r.oy, rn_mul(r.dy, tc));
    const float hz = rn_add(r.oz, rn_mul(r.dz, tc));
    // leaf centre = (voxel + 1/2) * voxel_size: exact, and the value the shader reaches by halving from the sub-chunk centre
    const float ex = rn_sub(hx, rn_mul(rn_add(static_cast<float>(vx), 0.5f), vs));
    const float ey = rn_sub(hy, rn_mul(rn_add(static_cast<float>(vy), 0.5f), vs));
    const float ez = rn_sub(hz, rn_mul(rn_add(static_cast<float>(vz), 0.5f), vs));
    const float gx = fabsf(ex), gy = fabsf(ey), gz = fabsf(ez);
    uint32_t face;                                                   // getHitFace, intersect.rint:58-68
    if (gx >= gy && gx >= gz) face = ex > 0.0f ? 0u : 1u;
    else if (gy >= gz)        face = ey > 0.0f ? 2u : 3u;
    else                      face = ez > 0.0f ? 4u : 5u;
    out.found = true; out.t = tc; out.material = material; out.face = face;
    out.vx = vx; out.vy = vy; out.vz = vz;
    out.brick = s.node;
    return out;
}

// ---- secondary rays: a walk entered from the previous hit's ancestors instead of the root --------------------------------------------
// A shadow or bounce ray starts a hair off the voxel the lane has just walked to (raygen.rgen:284 hit + N * 0.001, :376 hit + N * 0.002), and the
// LDS stack still holds that voxel's ancestors — node records, valid under any ray's mirroring.  walk_resume puts the new ray's walk
// into the state the walk from the root would be in after its first descents, and then lets it leave at once every ancestor that holds
// nothing the ray can still reach:
//
//  1. START VOXEL.  The walk from the root enters every node at the child slab that holds the ray at tS = max(world entry, tmin): per axis the
//     number of interior planes with T <= tS (enter_axis).  T is monotone in q, so level by level this selects the ancestors of ONE
//     voxel: per axis the q with T(q) <= tS < T(q + 1).  Here q is guessed from the point o + d tS and VERIFIED with the canonical T
//     (two plane evaluations per axis, one more after a correction by one): the verified q is the root walk's, whatever the guess
//     was.  T(0) <= T(q) <= tS on every axis then also says that the ray is inside the world box at tS = tmin, as the root walk's
//     tCur = max(T(0)..., tmin) requires.  No verified q (the origin outside the box, a guess off by more than one): the caller walks from the root.
//  2. COMMON ANCESTOR.  Above the lowest level m at which the start voxel and the reported voxel h share a node, the start voxel's
//     ancestors are h's — all occupied (h is a filled voxel), so the root walk descends through exactly them, pushing the records the
//     stack already holds, without changing tCur; it arrives at level m in node N_m with the far planes of the start voxel's level-m cell.
//     That state is formed here directly (N_0 = the brick record kept from the hit, N_m = stack slot m - 1).
//  3. LAUNCH PAD.  In mirrored coordinates every ray travels towards +q on all three axes: what it can still reach inside a node is the box
//     of children >= its current child.  If no occupied child of N lies in that box the ray leaves N through N's far planes without
//     a report — by the same (axis, T) the last of the single steps would take, and its fine position is dropped on the ascent anyway (the argument
//     of round 3's leave-the-node-early experiment, profiles/r03_skip_ahead_estimate.txt; there it ran inside the loop and cost more
//     than it saved, here it runs once per ray, where it pays most: a ray that leaves a surface spends most of its iterations climbing out
//     of the levels it starts at the bottom of).  So the walk is entered at the lowest ancestor that does hold something ahead, with
//     its own cell left at once like an empty one; no such ancestor up to the root: the ray reports nothing.
struct WalkAnchor { int vx, vy, vz; NodeRec brick; };      // the reported voxel (world lattice) and its brick record

// the children of a node a ray can still reach from child `bit` (un-mirrored digits x | y << 2 | z << 4): >= on an axis it travels
// positively, <= on a mirrored one
BLOK_DEV unsigned long long ahead_box(uint32_t bit, uint32_t mirror) {
    const uint32_t wx = bit & 3u, wy = (bit >> 2) & 3u, wz = bit >> 4;
    const uint32_t nx = (mirror & 3u) ? (0xFu >> (3u - wx)) : ((0xFu << wx) & 0xFu);
    const uint32_t ny = (mirror & 12u) ? (0xFFFFu >> (4u * (3u - wy))) : ((0xFFFFu << (4u * wy)) & 0xFFFFu);
    const uint32_t xy = (nx * 0x11111111u) & (ny * 0x00010001u);
    const unsigned long long az = (mirror & 48u) ? (~0ull >> (16u * (3u - wz))) : (~0ull << (16u * wz));
    return ((static_cast<unsigned long long>(xy) << 32) | xy) & az;
}

// One axis of step 1: f = 2^23 + q with T(q) <= tS < T(q + 1), t_far = T(q + 1); false if no such q next to the guess or outside [0, W).
BLOK_DEV bool resume_axis(const Axis& a, bool neg, float point, float inv_vs, int origin, int W, float tS, float& f, float& t_far) {
    const float w = floorf(point * inv_vs) - static_cast<float>(origin);          // the guess, in tree coordinates (any float will do)
    const float u = neg ? static_cast<float>(W - 1) - w : w;
    if (!(u >= 0.0f && u <= static_cast<float>(W - 1))) return false;             // NaN included
    f = kCoordBias + u;
    float t_lo = plane_t(a, f);
    t_far = plane_t(a, f + 1.0f);
    if (t_lo > tS) { f -= 1.0f; t_far = t_lo; t_lo = plane_t(a, f); }
    else if (t_far <= tS) { f += 1.0f; t_lo = t_far; t_far = plane_t(a, f + 1.0f); }
    return t_lo <= tS && tS < t_far && f >= kCoordBias && f <= kCoordBias + static_cast<float>(W - 1);
}

// false: nothing was set up, walk_enter must be called.  true: `s` is ready for walk_loop (or says that there is nothing to walk).
BLOK_DEV bool walk_resume(const TraceArgs& A, const RayIn& r, const WalkRay& R, const WalkAnchor& anchor, const uint4* stk, WalkState& s) {
    const uint32_t L = A.levels;
    const int W = 1 << (2 * L);
    const float tS = r.tmin;
    const bool negx = !(R.ax.inv > 0.0f), negy = !(R.ay.inv > 0.0f), negz = !(R.az.inv > 0.0f);
    const float inv_vs = rn_div(1.0f, A.voxel_size);
    float fx, fy, fz, tFx, tFy, tFz;
    bool ok = resume_axis(R.ax, negx, __builtin_fmaf(r.dx, tS, r.ox), inv_vs, A.origin[0], W, tS, fx, tFx);
    ok = resume_axis(R.ay, negy, __builtin_fmaf(r.dy, tS, r.oy), inv_vs, A.origin[1], W, tS, fy, tFy) && ok;
    ok = resume_axis(R.az, negz, __builtin_fmaf(r.dz, tS, r.oz), inv_vs, A.origin[2], W, tS, fz, tFz) && ok;
    if (!ok) { BLOK_STAT(7, 0); return false; }
    BLOK_STAT(5, 0);                       // a walk begins from an anchor
    s.found = false; s.bit = 0u; s.tCur = tS;
    s.walking = tS < r.tmax;               // the root walk's  tCur < min(world far planes, tmax): the far planes are beyond T(q + 1) > tS
    s.fx = fx; s.fy = fy; s.fz = fz; s.tFx = tFx; s.tFy = tFy; s.tFz = tFz; s.lvl = 0u; s.size = 1.0f; s.node = anchor.brick;
    if (!s.walking) return true;
    // step 2: the lowest level at which start voxel and anchor share a node
    const uint32_t qx = __float_as_uint(fx) & 0x7FFFFFu, qy = __float_as_uint(fy) & 0x7FFFFFu, qz = __float_as_uint(fz) & 0x7FFFFFu;
    const uint32_t ux = negx ? static_cast<uint32_t>(W - 1) - qx : qx, uy = negy ? static_cast<uint32_t>(W - 1) - qy : qy, uz = negz ? static_cast<uint32_t>(W - 1) - qz : qz;
    const uint32_t hx = static_cast<uint32_t>(anchor.vx - A.origin[0]), hy = static_cast<uint32_t>(anchor.vy - A.origin[1]), hz = static_cast<uint32_t>(anchor.vz - A.origin[2]);
    const uint32_t diff = (ux ^ hx) | (uy ^ hy) | (uz ^ hz);
    uint32_t lvl = diff ? static_cast<uint32_t>(31 - __clz(static_cast<int>(diff))) >> 1 : 0u;
    if (lvl >= L) return false;            // (an anchor outside the tree: never, but then from the root)
    NodeRec node = anchor.brick;
    if (lvl != 0u) { const uint4 c = stk[(lvl - 1) * kBlock]; node.lo = c.x; node.hi = c.y; node.base = c.z; }
    // step 3: climb while nothing of the node lies ahead
    bool climbed = false;
#ifndef BLOK_RESUME_LAUNCH_PAD
#define BLOK_RESUME_LAUNCH_PAD 1
#endif
    for (; BLOK_RESUME_LAUNCH_PAD;) {
        const uint32_t shift = 2 * lvl;
        const uint32_t bit = (digit2(qx, shift) | (digit2(qy, shift) << 2) | (digit2(qz, shift) << 4)) ^ R.mirror;
        unsigned long long ahead = ((static_cast<unsigned long long>(node.hi) << 32) | node.lo) & ahead_box(bit, R.mirror);
        if (climbed) ahead &= ~(1ull << bit);          // the child it came out of holds nothing ahead: established one level down
        if (ahead != 0ull) break;
        BLOK_STAT(6, lvl);                 // a level left on the launch pad
        if (lvl + 1u >= L) { s.walking = false; return true; }      // nothing ahead in the whole tree
        lvl += 1u; climbed = true;
        const uint4 c = stk[(lvl - 1) * kBlock];
        node.lo = c.x; node.hi = c.y; node.base = c.z;
    }
    if (lvl != 0u) {
        const uint32_t keep = ~((1u << (2 * lvl)) - 1u);
        s.size = cell_size(lvl);
        s.fx = __uint_as_float(__float_as_uint(fx) & keep); s.fy = __uint_as_float(__float_as_uint(fy) & keep); s.fz = __uint_as_float(__float_as_uint(fz) & keep);
        s.tFx = plane_t(R.ax, s.fx + s.size); s.tFy = plane_t(R.ay, s.fy + s.size); s.tFz = plane_t(R.az, s.fz + s.size);
    }
    if (climbed) {
        // the cell the walk stands in (the child it climbed out of) holds nothing ahead: left through its nearest far plane exactly as the
        // loop leaves an empty cell (the step of walk_loop, once, here — a flag inside the loop would cost every iteration of every walk)
        BLOK_STAT(2, lvl);
        const float tExit = fminf(fminf(s.tFx, s.tFy), s.tFz);
        s.tCur = tExit;
        if (!(tExit < r.tmax)) { s.walking = false; return true; }
        const bool sx = s.tFx == tExit;
        const bool sy = !sx && s.tFy == tExit;
        const bool sz = !sx && !sy;
        s.fx += sx ? s.size : 0.0f; s.fy += sy ? s.size : 0.0f; s.fz += sz ? s.size : 0.0f;
        const uint32_t up = static_cast<uint32_t>(__ffs(static_cast<int>(__float_as_uint(sx ? s.fx : (sy ? s.fy : s.fz)))) - 1) >> 1;
        if (up != lvl) {
            BLOK_STAT(3, lvl);
            if (up >= L) { s.walking = false; return true; }      // left the world box
            lvl = up;
            s.size = cell_size(lvl);
            const uint32_t keep = ~((1u << (2 * up)) - 1u);
            s.fx = __uint_as_float(__float_as_uint(s.fx) & keep); s.fy = __uint_as_float(__float_as_uint(s.fy) & keep); s.fz = __uint_as_float(__float_as_uint(s.fz) & keep);
            const uint4 c = stk[(lvl - 1) * kBlock];
            node.lo = c.x; node.hi = c.y; node.base = c.z;
        }
        s.tFx = plane_t(R.ax, s.fx + s.size); s.tFy = plane_t(R.ay, s.fy + s.size); s.tFz = plane_t(R.az, s.fz + s.size);
    }
    s.lvl = lvl; s.node = node;
    return true;
}

#ifndef BLOK_TRACE_HOST_HARNESS
// ---- primary rays: the wave's COMMON prefix of descents, once per wave --------------------------------------------------------------------
// The 64 rays of a wave tile start in nearly the same place (behind their beam tile's start parameter), so their walks begin with the same
// descents from the root — each a dependent node load and ~90 vector instructions, executed by all 64 lanes for one answer.  Here every lane
// finds its verified start voxel (walk_resume, step 1), the wave agrees on the lowest level m at which the 64 start voxels still share a cell
// (ballots), and ONE scalar chain walks from the root through the shared cells of levels L-1 .. m — uniform node loads, mask test and rank on
// the scalar unit — until a shared cell is empty or the cells part; the nodes passed go to every lane's stack slots, and each lane takes up the
// walk at its own cell of the level reached, in the state the walk from the root would be in (same argument as walk_resume: that walk descends
// through the cells that contain its start voxel for as long as they are occupied).  Measured on the benchmark frame's live wave tiles
// (scripts/r04/common_prefix_estimate.py): 2.2 / 4.0 / 2.4 of the 5 descents are shared in poses A / B / C.  false: the wave starts at the
// root as before (a lane without a verified start voxel, cells that part at the root).
BLOK_DEV bool walk_enter_wave(const TraceArgs& A, const RayIn& r, const WalkRay& R, uint4* stk, WalkState& s) {
    const uint32_t L = A.levels;
    const int W = 1 << (2 * L);
    const float tS = r.tmin;
    const bool negx = !(R.ax.inv > 0.0f), negy = !(R.ay.inv > 0.0f), negz = !(R.az.inv > 0.0f);
    float fx, fy, fz, tFx, tFy, tFz;
    bool ok = resume_axis(R.ax, negx, __builtin_fmaf(r.dx, tS, r.ox), A.inv_voxel_size, A.origin[0], W, tS, fx, tFx);
    ok = resume_axis(R.ay, negy, __builtin_fmaf(r.dy, tS, r.oy), A.inv_voxel_size, A.origin[1], W, tS, fy, tFy) && ok;
    ok = resume_axis(R.az, negz, __builtin_fmaf(r.dz, tS, r.oz), A.inv_voxel_size, A.origin[2], W, tS, fz, tFz) && ok;
    ok = ok && tS < r.tmax;
    if (__ballot(!ok) != 0ull) return false;
    const uint32_t qx = __float_as_uint(fx) & 0x7FFFFFu, qy = __float_as_uint(fy) & 0x7FFFFFu, qz = __float_as_uint(fz) & 0x7FFFFFu;
    const uint32_t ux = negx ? static_cast<uint32_t>(W - 1) - qx : qx, uy = negy ? static_cast<uint32_t>(W - 1) - qy : qy, uz = negz ? static_cast<uint32_t>(W - 1) - qz : qz;
    const uint32_t ux0 = __builtin_amdgcn_readfirstlane(ux), uy0 = __builtin_amdgcn_readfirstlane(uy), uz0 = __builtin_amdgcn_readfirstlane(uz);
    const uint32_t diff = (ux ^ ux0) | (uy ^ uy0) | (uz ^ uz0);
    const uint32_t mine = diff ? (static_cast<uint32_t>(31 - __clz(static_cast<int>(diff))) >> 1) + 1u : 0u;      // cells of 4^mine voxels hold this lane's start voxel and the first lane's
    uint32_t m = 0u;                                                       // ... of 4^m: all of them
    for (uint32_t k = 0; k < L; ++k) if (__ballot(mine > k) != 0ull) m = k + 1u;
    if (m >= L) return false;                                              // they part at the root: nothing to share
    // the scalar chain: the nodes through the scalar cache (they are read-only while frames run; the compiler, which cannot know that next to the
    // kernel's stores, would make them vector loads of one address: 64 lanes' worth of latency for a wave-uniform record) — a load from the
    // constant address space at a wave-uniform index is an s_load_dwordx4
    typedef uint32_t Words4 __attribute__((ext_vector_type(4)));
    auto uniform_node = [&](uint32_t index, uint32_t& lo, uint32_t& hi, uint32_t& base) {
        const Words4 v = reinterpret_cast<const __attribute__((address_space(4))) Words4*>(reinterpret_cast<uintptr_t>(A.nodes))[__builtin_amdgcn_readfirstlane(index)];
        lo = v.x; hi = v.y; base = v.z;
    };
    uint32_t n_lo, n_hi, n_base;
    uniform_node(0u, n_lo, n_hi, n_base);
    uint32_t lvl = L - 1u;
    for (;;) {
        if (lvl < m) break;                                                // from here on every lane has its own cell
        const uint32_t shift = 2u * lvl;
        const uint32_t bit = ((ux0 >> shift) & 3u) | (((uy0 >> shift) & 3u) << 2) | (((uz0 >> shift) & 3u) << 4);      // un-mirrored digits: the node's own bit order
        const uint32_t word = bit < 32u ? n_lo : n_hi;
        if (!((word >> (bit & 31u)) & 1u) || lvl == 0u) break;             // an empty shared cell (the walk steps on from it), or the start voxel itself
        BLOK_STAT(1, lvl);
        stk[(lvl - 1u) * kBlock] = make_uint4(n_lo, n_hi, n_base, 0u);
        const uint32_t below_lo = bit < 32u ? (n_lo & ((1u << bit) - 1u)) : n_lo;
        const uint32_t below_hi = bit < 32u ? 0u : (n_hi & ((1u << (bit & 31u)) - 1u));
        const uint32_t child = n_base + static_cast<uint32_t>(__builtin_popcount(below_lo)) + static_cast<uint32_t>(__builtin_popcount(below_hi));
        uniform_node(child, n_lo, n_hi, n_base);
        lvl -= 1u;
    }
    s.found = false; s.bit = 0u; s.tCur = tS; s.walking = true;
    s.lvl = lvl; s.size = cell_size(lvl);
    s.node.lo = n_lo; s.node.hi = n_hi; s.node.base = n_base;
    if (lvl != 0u) {
        const uint32_t keep = ~((1u << (2 * lvl)) - 1u);
        s.fx = __uint_as_float(__float_as_uint(fx) & keep); s.fy = __uint_as_float(__float_as_uint(fy) & keep); s.fz = __uint_as_float(__float_as_uint(fz) & keep);
        s.tFx = plane_t(R.ax, s.fx + s.size); s.tFy = plane_t(R.ay, s.fy + s.size); s.tFz = plane_t(R.az, s.fz + s.size);
    } else { s.fx = fx; s.fy = fy; s.fz = fz; s.tFx = tFx; s.tFy = tFy; s.tFz = tFz; }
    return true;
}
#endif

// Walks one ray.  `stk` points at this lane's slot of the LDS node stack (stride kBlock entries between
// levels; slot l-2 holds the node of level l on the current path).
//
// Tried in round 3 and removed (profiles/r03_paths_straggler_split_ab.txt, r03_walk_order_experiment*.txt): cutting the rest
// of a wave's last few rays into parameter segments for its idle lanes (exact: a ray's answer is the hit of its first segment that
// reports one; all parity tests passed) — primary rays are long or short tile by tile, not ray by ray, so there is nobody idle to
// help, and in the path loop the restarts cost more than the shorter rounds gave back (64 spp: 46.9 -> 51.2 ms); raising a wave's issue
// priority with its age (a long wave is a chain of dependent instructions: no priority shortens it); running the descend path only when
// enough lanes want it (waiting lanes stretch the critical path: -7 to -14 %); the path loop's shadow ray and bounce ray of a hit as a
// PAIR walked in one loop, a lane going on to its second ray while others are still on their first (profiles/
// r03_paths_shadow_bounce_pair_ab.txt: same frames, 46.3 -> 72 ms — two rays' state on top of the path state spills inside the loop, and
// the loop body as a function over a state struct alone costs the primary kernels a fifth: the loop keeps its state in locals).
BLOK_DEV HitInfo walk(const TraceArgs& A, const RayIn& r, uint4* stk) {
    BLOK_STAT(4, 0);                       // a walk begins
    HitInfo out;
    out.found = false; out.t = -1.0f; out.material = 0u; out.face = 0xFFu; out.vx = out.vy = out.vz = 0;
    out.brick.lo = out.brick.hi = out.brick.base = 0u;
    const WalkRay R = walk_ray(A, r.ox, r.oy, r.oz, safe_inv(r.dx), safe_inv(r.dy), safe_inv(r.dz));
    WalkState s;
    walk_enter(A, R, r.tmin, r.tmax, s);
    walk_loop(A, R, r.tmax, s, stk);
    if (s.found) out = walk_hit(A, r, R, s);
    return out;
}

// As walk(), the wave's rays entered together (walk_enter_wave): for the primary rays of a wave tile.  Every active lane of the wave must call it.
BLOK_DEV HitInfo walk_wave(const TraceArgs& A, const RayIn& r, uint4* stk) {
#ifdef BLOK_TRACE_HOST_HARNESS
    return walk(A, r, stk);
#else
    HitInfo out;
    out.found = false; out.t = -1.0f; out.material = 0u; out.face = 0xFFu; out.vx = out.vy = out.vz = 0;
    out.brick.lo = out.brick.hi = out.brick.base = 0u;
    const WalkRay R = walk_ray(A, r.ox, r.oy, r.oz, safe_inv(r.dx), safe_inv(r.dy), safe_inv(r.dz));
    WalkState s;
    if (!walk_enter_wave(A, r, R, stk, s)) walk_enter(A, R, r.tmin, r.tmax, s);
    walk_loop(A, R, r.tmax, s, stk);
    if (s.found) out = walk_hit(A, r, R, s);
    return out;
#endif
}

// Walks one ray and writes its 16-byte record and/or RGBA8 pixel.  kWave: through walk_wave (the coherent primary rays of a wave tile).
template <bool kWave = false>
BLOK_DEV void trace_one(const TraceArgs& A, const RayIn& r, uint4* stk, const Sink& dst) {
    const HitInfo h = kWave ? walk_wave(A, r, stk) : walk(A, r, stk);
    if (!h.found) { write_miss(dst); return; }
    uint4 rec;
    rec.x = __float_as_uint(h.t);
    rec.y = h.material;
    rec.z = (static_cast<uint32_t>(h.vx) & 0xFFFFu) | (static_cast<uint32_t>(h.vy) << 16);
    rec.w = (static_cast<uint32_t>(h.vz) & 0xFFFFu) | (h.face << 16) | (1u << 24);
    if (dst.hit) *reinterpret_cast<uint4*>(dst.hit) = rec;
    if (dst.rgba) *dst.rgba = shade_rgba(A.mat_table, A.n_materials, h.material, h.face);
}

// Camera-plane coordinates of the ray through continuous pixel (pcx, pcy): reference blok/src/cuda_tracer.cu:276-282,
// algebraically assets/shaders/raygen.rgen:201-205 — ndc = 2 uv - 1 shifted by the TAA jitter the reference puts into
// proj[2][0..1] (x_ndc = P00 x/-z - jx_clip, so the ray of NDC d is the un-jittered ray of d + jitter_clip; y alike, then the
// Vulkan flip).  A zero jitter adds an exact 0: bit-identical to the un-jittered form.
BLOK_DEV void camera_plane_uv(const TraceArgs& A, float pcx, float pcy, float& u, float& v) {
    const blok_camera& c = A.cam;
    const float qx = rn_div(pcx, static_cast<float>(A.frame_w));
    const float qy = rn_div(pcy, static_cast<float>(A.frame_h));
    u = rn_mul(rn_mul(rn_add(rn_sub(rn_mul(2.0f, qx), 1.0f), A.jitter_clip[0]), c.tan_half_fov), c.aspect);
    v = rn_mul(rn_sub(rn_sub(1.0f, rn_mul(2.0f, qy)), A.jitter_clip[1]), c.tan_half_fov);
}

// Primary ray of pixel (x, y), tmin/tmax raygen.rgen:225,227.
BLOK_DEV RayIn primary_ray(const TraceArgs& A, uint32_t x, uint32_t y) {
    const blok_camera& c = A.cam;
    float u, v;
    camera_plane_uv(A, rn_add(static_cast<float>(x), 0.5f), rn_add(static_cast<float>(y), 0.5f), u, v);
    const float dx = rn_add(rn_add(c.fwd[0], rn_mul(c.right[0], u)), rn_mul(c.up[0], v));
    const float dy = rn_add(rn_add(c.fwd[1], rn_mul(c.right[1], u)), rn_mul(c.up[1], v));
    const float dz = rn_add(rn_add(c.fwd[2], rn_mul(c.right[2], u)), rn_mul(c.up[2], v));
    const float len = rn_sqrt(rn_add(rn_add(rn_mul(dx, dx), rn_mul(dy, dy)), rn_mul(dz, dz)));
    RayIn r;
    r.ox = c.pos[0]; r.oy = c.pos[1]; r.oz = c.pos[2];
    r.dx = rn_div(dx, len); r.dy = rn_div(dy, len); r.dz = rn_div(dz, len);
    r.tmin = A.tmin; r.tmax = A.tmax;
    return r;
}

}  // namespace blok
#endif
