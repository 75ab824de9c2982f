// Beam pre-pass: a conservative start parameter for all primary rays of a tile of pixels (32x32 by default).
//
// No reference counterpart: the reference lets Vulkan RT hardware cull sub-chunk boxes per ray
// (reference blok/src/renderer_raytracing.cpp:15-254).  Here one wave walks the 64-tree ONCE for the whole
// tile, cooperatively — lane i tests child i of the current node against the tile's frustum — and finds a
// lower bound t0 on the parameter of any filled voxel any ray of the tile can report (beam_kernel, a few
// thousand waves per 4K frame).  The per-ray walk then runs with tmin' = max(tmin, t0).  That cannot change a record: a voxel is reported iff
// max(entry, tmin) < min(exit, tmax) with t = max(entry, tmin) (trace_kernels.h), and raising tmin to a value
// below every reportable voxel's entry leaves both the set of reported voxels and their t untouched.  If no
// cell of the tree meets the frustum, every ray of the tile misses: beam_kernel writes those pixels itself and the
// tile's trace waves exit on their first instruction.
//
// Conservative by construction, not by tuning:
//   * the frustum is the tile's pixel rectangle grown by one pixel on every side (covers the pixel centres, the
//     path kernel's sub-pixel jitter and the |d_a| < 1e-6 direction clamp of the walk) and each side plane is
//     pushed outward by kBeamSlack voxels (covers float rounding of the plane tests, ~1e-3 at 16384^3);
//   * a cell is culled only if its farthest corner is strictly behind a side plane (NaNs never cull);
//   * the bound is the smallest depth, along the tile's central direction c, of the nearest corner of any
//     surviving non-empty cell: for a unit direction d and a point p = o + t d of that cell,
//     t = (p - o).c / (d.c) >= (p - o).c because d.c <= 1; it is then reduced by kBeamSlack and 1e-4 relative
//     (|d| = 1 within float rounding).
#ifndef BLOK_BEAM_H
#define BLOK_BEAM_H

#include "trace_kernels.h"

namespace blok {

#ifndef BLOK_BEAM_STOP_LEVEL
#define BLOK_BEAM_STOP_LEVEL 1      // finest cells examined = children of a node of this level (1: voxels, 2: 4^3 bricks)
#endif
constexpr float kBeamSlack = 0.05f;
// Node visits one search may spend; TraceArgs::beam_budget overrides.  Searches average 35 visits on the benchmark frame, but the
// pre-pass lasts as long as its LONGEST wave (all of them are resident at once; a visit is a chain of ~650 dependent cycles):
// 77 us for a 13 us average.  A long search first coarsens (below), and one that runs out stops where it is with a valid, less
// tight answer (the end of beam_search), so the budget trades the pre-pass's tail against the walk's start: measured at 4K over
// 1024^3 (scripts/beam_budget_sweep.py, profiles/r02_beam_budget_sweep.txt), launch pair alone / three frames in flight, poses
// A, B, C: unlimited 0.259 / 0.186, 0.308 / 0.245, 0.290 / 0.192 ms; 256: 0.251 / 0.186, 0.305 / 0.248, 0.269 / 0.192;
// 128: 0.239 / 0.189, 0.317 / 0.275, 0.248 / 0.193 — below 256 the grazing pose pays more in the walk than the pre-pass saves.
constexpr uint32_t kBeamMaxVisits = 256u;
#ifndef BLOK_BEAM_COARSEN1
#define BLOK_BEAM_COARSEN1 3      // eighths of the budget after which a search stops at bricks ...
#endif
#ifndef BLOK_BEAM_COARSEN2
#define BLOK_BEAM_COARSEN2 5      // ... and at 16^3 cells (8 = never)
#endif
constexpr uint32_t kBeamCoarsen1 = BLOK_BEAM_COARSEN1, kBeamCoarsen2 = BLOK_BEAM_COARSEN2;

struct BeamVec { float x, y, z; };

// un-normalised direction through continuous pixel coordinates (px, py); 1 ulp reciprocals are fine here, the
// one-pixel margin is eight orders of magnitude larger
__device__ __forceinline__ BeamVec beam_dir(const blok_camera& c, float px, float py, float inv_w, float inv_h) {
    const float u = (2.0f * (px * inv_w) - 1.0f) * c.tan_half_fov * c.aspect;
    const float v = (1.0f - 2.0f * (py * inv_h)) * c.tan_half_fov;
    return {__builtin_fmaf(c.up[0], v, __builtin_fmaf(c.right[0], u, c.fwd[0])), __builtin_fmaf(c.up[1], v, __builtin_fmaf(c.right[1], u, c.fwd[1])),
            __builtin_fmaf(c.up[2], v, __builtin_fmaf(c.right[2], u, c.fwd[2]))};
}
__device__ __forceinline__ float beam_dot(BeamVec n, float x, float y, float z) { return __builtin_fmaf(n.x, x, __builtin_fmaf(n.y, y, n.z * z)); }
__device__ __forceinline__ BeamVec beam_unit(BeamVec a) {
    const float k = __builtin_amdgcn_rsqf(beam_dot(a, a.x, a.y, a.z));
    return {a.x * k, a.y * k, a.z * k};
}
__device__ __forceinline__ uint32_t beam_uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float beam_lane(float v, uint32_t lane) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane)); }
__device__ __forceinline__ BeamVec beam_lane(BeamVec v, uint32_t lane) { return {beam_lane(v.x, lane), beam_lane(v.y, lane), beam_lane(v.z, lane)}; }

// The cooperative depth-first search both pre-passes share.  The region is the intersection of four half-spaces
// n_k . (p - ref) + c_k >= 0 (each pushed out by kBeamSlack; a cell is culled only if its farthest corner is strictly
// behind one); the result is the smallest value, over the filled voxels that meet the region, of mid . (nearest corner - ref),
// or kBeamNone.  Must be called by all 64 lanes of the wave.
template <bool kClampAtZero>
__device__ __forceinline__ float beam_search(const TraceArgs& A, BeamVec ref, BeamVec n0, BeamVec n1, BeamVec n2, BeamVec n3,
                                             float c0, float c1, float c2, float c3, BeamVec mid, uint32_t lane, float initial_best = kBeamNone,
                                             uint32_t* visits_out = nullptr) {
    // lanes are children in front-to-back order for the central direction: mirrored child index
    const uint32_t mirror = beam_uniform((mid.x < 0.0f ? 0x03u : 0u) | (mid.y < 0.0f ? 0x0Cu : 0u) | (mid.z < 0.0f ? 0x30u : 0u));
    const uint32_t child = lane ^ mirror;
    const float cx = static_cast<float>(child & 3u), cy = static_cast<float>((child >> 2) & 3u), cz = static_cast<float>(child >> 4);
    // Plane values are carried down the tree instead of being recomputed from coordinates: p_k is the value of plane k at the
    // current node's min corner (wave-uniform), a_k the per-lane step to a child's min corner in units of the child size, so a
    // child's value is g_k = p_k + s a_k, the value at its farthest (side planes) / nearest (depth) corner g_k + s far_k, and
    // descending into child j is p_k <- readlane(g_k, j).  The drift over <= 7 levels is a few ulp of the coordinate
    // magnitude (<= 1e-2 voxel at 16384^3), far inside kBeamSlack and the depth margin.
    const float a0 = beam_dot(n0, cx, cy, cz), a1 = beam_dot(n1, cx, cy, cz), a2 = beam_dot(n2, cx, cy, cz), a3 = beam_dot(n3, cx, cy, cz);
    const float a4 = beam_dot(mid, cx, cy, cz);
    auto far_of = [](BeamVec n) { return (n.x > 0.0f ? n.x : 0.0f) + (n.y > 0.0f ? n.y : 0.0f) + (n.z > 0.0f ? n.z : 0.0f); };
    const float far0 = far_of(n0), far1 = far_of(n1), far2 = far_of(n2), far3 = far_of(n3);
    const float near4 = (mid.x < 0.0f ? mid.x : 0.0f) + (mid.y < 0.0f ? mid.y : 0.0f) + (mid.z < 0.0f ? mid.z : 0.0f);
    const bool child_hi = child >= 32u;
    const uint32_t child_bit = 1u << (child & 31u);

    const uint32_t root_level = A.levels;
    uint32_t level = root_level, node = 0;
    const float ox = static_cast<float>(A.origin[0]) - ref.x, oy = static_cast<float>(A.origin[1]) - ref.y, oz = static_cast<float>(A.origin[2]) - ref.z;
    float p0 = beam_dot(n0, ox, oy, oz) + c0, p1 = beam_dot(n1, ox, oy, oz) + c1, p2 = beam_dot(n2, ox, oy, oz) + c2, p3 = beam_dot(n3, ox, oy, oz) + c3;
    float p4 = beam_dot(mid, ox, oy, oz);
    float best = initial_best;                         // cells not nearer than this are never visited
    uint32_t stk_node = 0, stk_lo = 0, stk_hi = 0;     // lane l holds the entry of level l
    float stk_p0 = 0.0f, stk_p1 = 0.0f, stk_p2 = 0.0f, stk_p3 = 0.0f, stk_p4 = 0.0f;
    bool fresh = true;
    uint64_t cand = 0;
    // every wave reaches the exit: the search is a finite tree walk, and a visit budget bounds it even for a frustum whose
    // planes cull nothing (degenerate inputs): running out is answered with a lower bound over what is left, never with "none"
    const uint32_t budget0 = A.beam_budget ? A.beam_budget : kBeamMaxVisits;
    uint32_t budget = budget0;
    // A long search coarsens before it is cut off: past kBeamCoarsen1/8 of its budget it stops at the 4^3 bricks (their nearest
    // corners stand for their voxels), past kBeamCoarsen2/8 at the 16^3 cells — still a front-to-back search with a valid bound,
    // with the brick visits (most of a search) gone.  Only kClampAtZero searches (the others want an exact answer).
    const uint32_t coarse1 = kClampAtZero ? budget0 - budget0 * kBeamCoarsen1 / 8u : 0u, coarse2 = kClampAtZero ? budget0 - budget0 * kBeamCoarsen2 / 8u : 0u;
    // The loop, as two: a node entered for the first time is evaluated and its nearest candidate descended into (the outer loop); a node
    // without candidates sends the search back up, where what is left of the parent's candidates is looked at again (the inner loop).  One
    // unit of the budget per evaluation, first or repeated, as when this was a single loop with a `fresh` flag — written apart because in
    // the single loop the eight stack registers and the five plane values lived in two register sets copied into each other on every trip.
    typedef uint32_t BeamWords4 __attribute__((ext_vector_type(4)));
    const auto* const tree = reinterpret_cast<const __attribute__((address_space(4))) BeamWords4*>(reinterpret_cast<uintptr_t>(A.nodes));
    bool exhausted = false;
    // the node record through the scalar cache: `node` is wave-uniform and the tree is read-only while frames run, which the compiler
    // cannot know next to the kernels' stores — left alone it issues a vector load of one address and three v_readfirstlane
    // (profiles/r04_search_scalar_loads_ab.txt: the frame alone 0.191 -> 0.182 ms).  A child's record is asked for as soon as its index
    // is known, before the stack and the plane values are dealt with: the load's way through the scalar cache runs beside that work
    // (worth 0.4 % of the frame alone, profiles/r04_search_two_loops_ab.txt)
    BeamWords4 entered = tree[beam_uniform(node)];
    for (;;) {
        // ---- a node entered for the first time
        if (budget == 0u) { exhausted = true; fresh = true; break; }
        const uint32_t spent_before = budget0 - budget;                                // evaluations before this one
        --budget;
        BeamWords4 rec = entered;
        uint32_t shift = 2u * (level - 1u);
        float s = static_cast<float>(1u << shift);
        float g4 = __builtin_fmaf(s, a4, p4);
        float depth = __builtin_fmaf(s, near4, g4);                                    // lower bound of the child's depth
        float g0 = __builtin_fmaf(s, a0, p0), g1 = __builtin_fmaf(s, a1, p1), g2 = __builtin_fmaf(s, a2, p2), g3 = __builtin_fmaf(s, a3, p3);
        {
            // behind any side plane?  fminf drops NaNs, so a NaN never culls (as with four separate comparisons)
            const bool outside = fminf(fminf(__builtin_fmaf(s, far0, g0), __builtin_fmaf(s, far1, g1)),
                                       fminf(__builtin_fmaf(s, far2, g2), __builtin_fmaf(s, far3, g3))) < -kBeamSlack;
            const bool filled = ((child_hi ? rec.y : rec.x) & child_bit) != 0u;
            cand = __ballot(filled && !outside && !(depth >= best));
            const uint32_t left = budget0 - spent_before;                              // the budget as the single loop saw it during this evaluation
            const uint32_t stop_level = left < coarse2 ? BLOK_BEAM_STOP_LEVEL + 2u : (left < coarse1 ? BLOK_BEAM_STOP_LEVEL + 1u : BLOK_BEAM_STOP_LEVEL);
            if (level <= stop_level) {
                while (cand) {
                    const uint32_t j = static_cast<uint32_t>(__builtin_ctzll(cand));
                    const float dj = beam_lane(depth, j);
                    if (kClampAtZero) best = dj >= 0.0f ? fminf(best, dj) : 0.0f;      // negative or NaN: start at the ray origin
                    else best = dj == dj ? fminf(best, dj) : -kBeamNone;               // NaN: the most conservative answer
                    cand &= cand - 1u;
                    cand &= __ballot(!(depth >= best));
                }
            }
        }
        // ---- nothing (left) to descend into: up, and the parent's remaining candidates against the bound as it is now
        bool finished = false, revisited = false;
        while (cand == 0u) {
            if (level == root_level) { finished = true; break; }
            ++level;
            node = __builtin_amdgcn_readlane(stk_node, level);
            cand = static_cast<uint64_t>(__builtin_amdgcn_readlane(stk_lo, level)) |
                   (static_cast<uint64_t>(__builtin_amdgcn_readlane(stk_hi, level)) << 32);
            p0 = beam_lane(stk_p0, level); p1 = beam_lane(stk_p1, level); p2 = beam_lane(stk_p2, level); p3 = beam_lane(stk_p3, level);
            p4 = beam_lane(stk_p4, level);
            if (budget == 0u) { exhausted = true; fresh = false; break; }
            --budget;
            rec = tree[beam_uniform(node)];
            shift = 2u * (level - 1u);
            s = static_cast<float>(1u << shift);
            g4 = __builtin_fmaf(s, a4, p4);
            depth = __builtin_fmaf(s, near4, g4);
            cand &= __ballot(!(depth >= best));
            revisited = true;
        }
        if (finished || exhausted) break;
        // ---- into the nearest candidate
        const uint32_t j = static_cast<uint32_t>(__builtin_ctzll(cand));
        cand &= cand - 1u;
        const uint32_t parent = node;
        {
            const uint32_t mlo = rec.x, mhi = rec.y, base = rec.z;
            const uint32_t cj = j ^ mirror;
            const uint32_t below_lo = cj < 32u ? (mlo & ((1u << cj) - 1u)) : mlo;
            const uint32_t below_hi = cj < 32u ? 0u : (mhi & ((1u << (cj & 31u)) - 1u));
            node = base + __builtin_popcount(below_lo) + __builtin_popcount(below_hi);
            entered = tree[beam_uniform(node)];
        }
        if (lane == level) {
            stk_node = parent; stk_lo = static_cast<uint32_t>(cand); stk_hi = static_cast<uint32_t>(cand >> 32);
            stk_p0 = p0; stk_p1 = p1; stk_p2 = p2; stk_p3 = p3; stk_p4 = p4;
        }
        if (revisited) {                                                               // the side values of a revisited node were not formed
            g0 = __builtin_fmaf(s, a0, p0); g1 = __builtin_fmaf(s, a1, p1); g2 = __builtin_fmaf(s, a2, p2); g3 = __builtin_fmaf(s, a3, p3);
        }
        p0 = beam_lane(g0, j); p1 = beam_lane(g1, j); p2 = beam_lane(g2, j); p3 = beam_lane(g3, j); p4 = beam_lane(g4, j);
        --level;
    }
    if (visits_out) *visits_out = budget0 - budget - (exhausted ? 0u : 1u);            // diagnostics: node visits this search spent (the last evaluation of a finished search not counted, as before)
    if (exhausted) {
        if constexpr (!kClampAtZero) return -kBeamNone;
        // Out of visits.  Every filled voxel the search has not seen lies in a cell that is still pending: the node just entered
        // (fresh), or the candidates left at this level and at every level on the stack.  A cell's `depth` is a lower bound for
        // everything inside it, so min(best, depth of every pending cell) is a valid — only less tight — answer: the search stops
        // where it is instead of giving up, which bounds the pre-pass's longest wave (its duration) by the budget.
        float lb = best;
        auto take = [&](float d) { lb = d >= 0.0f ? fminf(lb, d) : 0.0f; };          // negative or NaN: start at the ray origin
        uint32_t l = level;
        if (fresh) {
            take(__builtin_fmaf(static_cast<float>(1u << (2u * level)), near4, p4));   // the whole node: its nearest corner
            l = level + 1u;                                                            // its parent's entry is on the stack
        }
        for (; l <= root_level; ++l) {
            uint64_t pending; float q4;
            if (l == level) { pending = cand; q4 = p4; }                               // revisited node: what is left of its candidates
            else {
                pending = static_cast<uint64_t>(__builtin_amdgcn_readlane(stk_lo, l)) | (static_cast<uint64_t>(__builtin_amdgcn_readlane(stk_hi, l)) << 32);
                q4 = beam_lane(stk_p4, l);
            }
            const float sl = static_cast<float>(1u << (2u * (l - 1u)));
            const float dl = __builtin_fmaf(sl, near4, __builtin_fmaf(sl, a4, q4));    // this lane's child of that node
            while (pending) {
                take(beam_lane(dl, static_cast<uint32_t>(__builtin_ctzll(pending))));
                pending &= pending - 1u;
            }
        }
        return lb;
    }
    return best;
}

// Pixel rectangle [px_lo, px_hi] x [py_lo, py_hi] in continuous pixel coordinates of the frame (pixel x covers
// [x, x+1]).  Must be called by all 64 lanes of the wave.  Returns kBeamNone or a start parameter >= 0.
// depth_limit: only voxels nearer than this (along the tile's central direction) are looked for; when none is found the
// answer is the limit itself (everything the rays can report lies at or beyond it) instead of kBeamNone.
__device__ __forceinline__ float beam_start(const TraceArgs& A, float px_lo, float py_lo, float px_hi, float py_hi, uint32_t lane,
                                            float depth_limit = kBeamNone, uint32_t* visits_out = nullptr) {
    const float inv_w = __builtin_amdgcn_rcpf(static_cast<float>(A.frame_w)), inv_h = __builtin_amdgcn_rcpf(static_cast<float>(A.frame_h));
    const BeamVec mid = beam_unit(beam_dir(A.cam, 0.5f * (px_lo + px_hi), 0.5f * (py_lo + py_hi), inv_w, inv_h));
    // lane k < 4 builds side plane k through corners k and k+1 of the grown rectangle (corner i: x high for i = 1, 2;
    // y high for i = 2, 3), oriented towards the central direction; the four planes are then read back wave-wide
    const uint32_t k0 = lane & 3u, k1 = (lane + 1u) & 3u;
    const BeamVec p = beam_dir(A.cam, (k0 == 1u || k0 == 2u) ? px_hi + 1.0f : px_lo - 1.0f, k0 >= 2u ? py_hi + 1.0f : py_lo - 1.0f, inv_w, inv_h);
    const BeamVec q = beam_dir(A.cam, (k1 == 1u || k1 == 2u) ? px_hi + 1.0f : px_lo - 1.0f, k1 >= 2u ? py_hi + 1.0f : py_lo - 1.0f, inv_w, inv_h);
    BeamVec side = beam_unit({p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - p.y * q.x});
    if (beam_dot(side, mid.x, mid.y, mid.z) < 0.0f) side = {-side.x, -side.y, -side.z};
    const BeamVec n0 = beam_lane(side, 0), n1 = beam_lane(side, 1), n2 = beam_lane(side, 2), n3 = beam_lane(side, 3);

    // the search runs in voxel units (tree coordinates): the camera position is scaled into them (a power of two: exact), the
    // directions are unit-free, and the depth found, less its margins, goes back to the rays' world parameter
    const float iv = A.inv_voxel_size;
    const float best = beam_search<true>(A, {A.cam.pos[0] * iv, A.cam.pos[1] * iv, A.cam.pos[2] * iv}, n0, n1, n2, n3, 0.0f, 0.0f, 0.0f, 0.0f, mid, lane,
                                         depth_limit >= kBeamNone ? kBeamNone : depth_limit * iv, visits_out);
    if (best >= kBeamNone) return kBeamNone;
    return fmaxf(best * (1.0f - 1.0e-4f) - 2.0f * kBeamSlack, 0.0f) * A.voxel_size;
}

// ---- "last occluder" map for the shadow rays of the path kernel ------------------------------------------------------
// All shadow rays share one direction (the shader's constant sun, raygen.rgen:142,185), so "the farthest point along the
// sun direction at which any voxel exists" is a function of the two coordinates perpendicular to it.  One wave per texel
// of that plane searches the prism {u_lo <= u.p <= u_hi, v_lo <= v.p <= v_hi} (grown by kBeamSlack) for the largest s-depth
// of a filled voxel's far corner.  A shadow ray starting at o then needs tmax' = D(texel of o) - s.o (+ margin) only: every
// voxel it can report is entered before that parameter, so lowering tmax changes no "any hit" answer (trace_kernels.h: a
// voxel is reported iff max(entry, tmin) < min(exit, tmax)); rays above the last occluder skip the walk altogether.
struct SunFrame { BeamVec u, v, s; };      // orthonormal; s = the sun direction of path_core.h
__device__ __forceinline__ float prism_far(const TraceArgs& A, const SunFrame& F, float u_lo, float u_hi, float v_lo, float v_hi, uint32_t lane) {
    const BeamVec nu = {-F.u.x, -F.u.y, -F.u.z}, nv = {-F.v.x, -F.v.y, -F.v.z}, back = {-F.s.x, -F.s.y, -F.s.z};
    const float best = beam_search<false>(A, {0.0f, 0.0f, 0.0f}, F.u, nu, F.v, nv, -u_lo, u_hi, -v_lo, v_hi, back, lane);
    return best >= kBeamNone ? -kBeamNone : -best;          // = max over voxels of s . (far corner)
}

}  // namespace blok
#endif
