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

// T(a, p): t at which the ray crosses the plane  axis_a = p  (p a world integer).
BLOK_DEV float plane_t(int p_world, float o, float inv) {
    return rn_mul(rn_sub(static_cast<float>(p_world), o), inv);
}

// intersect.rint:79
BLOK_DEV float safe_inv(float d) {
    return rn_div(1.0f, fabsf(d) < 1e-6f ? 1e-6f : d);
}

BLOK_DEV void write_miss(blok_hit* dst) {
    // t = -1 (miss.rmiss:25-27), material 0, voxel 0, face 0xFF, hit 0
    *reinterpret_cast<uint4*>(dst) = make_uint4(0xBF800000u, 0u, 0u, 0x00FF0000u);
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

// One axis of "enter a node": which of the node's 4 child slabs contains the ray at tS, and the T of
// that slab's far plane.  p = node corner (local), s = child size = 1 << shift.
BLOK_DEV void enter_axis(int& p, float& t_far, int org, float o, float inv, bool pos,
                                           uint32_t shift, float tS) {
    const int base = p + org;
    const float m2 = plane_t(base + (2 << shift), o, inv);          // middle plane (ray order j = 2)
    const bool g = m2 <= tS;
    const int jq = g ? 3 : 1;                                        // next plane to test, ray order
    const float mq = plane_t(base + ((pos ? jq : 4 - jq) << shift), o, inv);
    const bool g2 = mq <= tS;
    const int n = (g ? 2 : 0) + (g2 ? 1 : 0);                        // interior planes already crossed
    t_far = g ? (g2 ? t_far : mq) : (g2 ? m2 : mq);
    p += (pos ? n : 3 - n) << shift;
}

struct RayIn { float ox, oy, oz, dx, dy, dz, tmin, tmax; };

// Walks one ray; writes the 16-byte record.  `stk` points at this lane's slot of the LDS node stack
// (stride kBlock entries between levels).
BLOK_DEV void trace_one(const TraceArgs& A, const RayIn& r, uint4* stk, blok_hit* dst) {
    const float ix = safe_inv(r.dx), iy = safe_inv(r.dy), iz = safe_inv(r.dz);
    const bool posx = ix > 0.0f, posy = iy > 0.0f, posz = iz > 0.0f;
    const int orgx = A.origin[0], orgy = A.origin[1], orgz = A.origin[2];
    const uint32_t L = A.levels;
    const int world = 1 << (2 * L);

    // world box
    float tFx, tFy, tFz, tCur;
    {
        const float x0 = plane_t(orgx, r.ox, ix), x1 = plane_t(orgx + world, r.ox, ix);
        const float y0 = plane_t(orgy, r.oy, iy), y1 = plane_t(orgy + world, r.oy, iy);
        const float z0 = plane_t(orgz, r.oz, iz), z1 = plane_t(orgz + world, r.oz, iz);
        const float t_in = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
        tFx = fmaxf(x0, x1); tFy = fmaxf(y0, y1); tFz = fmaxf(z0, z1);
        tCur = fmaxf(t_in, r.tmin);
        if (!(tCur < fminf(fminf(fminf(tFx, tFy), tFz), r.tmax))) { write_miss(dst); return; }
    }

    int px = 0, py = 0, pz = 0;          // min corner of the current cell, tree-local voxel units
    NodeRec node;                         // node whose children are the cells of level `lvl`
    {
        const uint4 q = A.nodes[0];
        node.lo = q.x; node.hi = q.y; node.base = q.z;
    }
    uint32_t lvl = L - 1;
    {
        const uint32_t shift = 2 * lvl;
        const float tS = tCur;            // already >= tmin
        enter_axis(px, tFx, orgx, r.ox, ix, posx, shift, tS);
        enter_axis(py, tFy, orgy, r.oy, iy, posy, shift, tS);
        enter_axis(pz, tFz, orgz, r.oz, iz, posz, shift, tS);
    }

    for (uint32_t guard = 0; guard < (1u << 20); ++guard) {
        BLOK_STAT(0, lvl);
        const uint32_t shift = 2 * lvl;
        const uint32_t bit = ((px >> shift) & 3) | (((py >> shift) & 3) << 2) | (((pz >> shift) & 3) << 4);
        if (mask_bit(node, bit)) {
            if (lvl == 0) {
                const float tc = fmaxf(tCur, r.tmin);
                const float lim = fminf(fminf(fminf(tFx, tFy), tFz), r.tmax);
                if (tc < lim) {
                    // reported: intersect.rint:136-141, hit.rchit:58-74
                    const uint32_t material = A.materials[node.base + mask_rank(node, bit)];
                    const int vx = px + orgx, vy = py + orgy, vz = pz + orgz;
                    const float hx = rn_add(r.ox, rn_mul(r.dx, tc));
                    const float hy = rn_add(r.oy, rn_mul(r.dy, tc));
                    const float hz = rn_add(r.oz, rn_mul(r.dz, tc));
                    const float ex = rn_sub(hx, rn_add(static_cast<float>(vx), 0.5f));
                    const float ey = rn_sub(hy, rn_add(static_cast<float>(vy), 0.5f));
                    const float ez = rn_sub(hz, rn_add(static_cast<float>(vz), 0.5f));
                    const float ax = fabsf(ex), ay = fabsf(ey), az = fabsf(ez);
                    uint32_t face;
                    if (ax >= ay && ax >= az) face = ex > 0.0f ? 0u : 1u;
                    else if (ay >= az)        face = ey > 0.0f ? 2u : 3u;
                    else                      face = ez > 0.0f ? 4u : 5u;
                    uint4 rec;
                    rec.x = __float_as_uint(tc);
                    rec.y = material;
                    rec.z = (static_cast<uint32_t>(vx) & 0xFFFFu) | (static_cast<uint32_t>(vy) << 16);
                    rec.w = (static_cast<uint32_t>(vz) & 0xFFFFu) | (face << 16) | (1u << 24);
                    *reinterpret_cast<uint4*>(dst) = rec;
                    return;
                }
            } else {
                // descend: remember the node we are leaving, fetch the child, pick its start cell
                BLOK_STAT(1, lvl);
                stk[(lvl - 1) * kBlock] = make_uint4(node.lo, node.hi, node.base, 0u);   // slot of level lvl+1
                const uint4 q = A.nodes[node.base + mask_rank(node, bit)];
                node.lo = q.x; node.hi = q.y; node.base = q.z;
                lvl -= 1;
                const uint32_t cs = 2 * lvl;
                const float tS = fmaxf(tCur, r.tmin);
                enter_axis(px, tFx, orgx, r.ox, ix, posx, cs, tS);
                enter_axis(py, tFy, orgy, r.oy, iy, posy, cs, tS);
                enter_axis(pz, tFz, orgz, r.oz, iz, posz, cs, tS);
                continue;
            }
        }
        // step to the next cell of the merge sequence: cross the nearest far plane (x, then y, then z on ties)
        BLOK_STAT(2, lvl);
        const bool sx = tFx <= tFy && tFx <= tFz;
        const bool sy = !sx && tFy <= tFz;
        tCur = sx ? tFx : (sy ? tFy : tFz);
        if (!(tCur < r.tmax)) break;
        const int size = 1 << shift;
        const int pold = sx ? px : (sy ? py : pz);
        const bool pos = sx ? posx : (sy ? posy : posz);
        const int pnew = pos ? pold + size : pold - size;
        if (pnew < 0 || pnew >= world) break;
        if (sx) px = pnew; else if (sy) py = pnew; else pz = pnew;
        const uint32_t crossed = static_cast<uint32_t>(pold ^ pnew) >> (shift + 2);
        if (crossed != 0u) {
            // left the parent node: climb to the level whose cell boundary was crossed
            BLOK_STAT(3, lvl);
            const uint32_t k = ((31u - __clz(crossed)) >> 1) + 1u;
            lvl += k;
            const uint32_t ns = 2 * lvl;
            const int keep = ~((1 << ns) - 1);
            px &= keep; py &= keep; pz &= keep;
            const uint4 q = stk[(lvl - 1) * kBlock];                 // node of level lvl+1
            node.lo = q.x; node.hi = q.y; node.base = q.z;
            const int far = 1 << ns;
            tFx = plane_t(px + orgx + (posx ? far : 0), r.ox, ix);
            tFy = plane_t(py + orgy + (posy ? far : 0), r.oy, iy);
            tFz = plane_t(pz + orgz + (posz ? far : 0), r.oz, iz);
        } else {
            const float tn = plane_t(pnew + (sx ? orgx : (sy ? orgy : orgz)) + (pos ? size : 0),
                                     sx ? r.ox : (sy ? r.oy : r.oz), sx ? ix : (sy ? iy : iz));
            if (sx) tFx = tn; else if (sy) tFy = tn; else tFz = tn;
        }
    }
    write_miss(dst);
}

// Primary ray of pixel (x, y): reference blok/src/cuda_tracer.cu:276-282 with zero jitter
// (algebraically assets/shaders/raygen.rgen:201-205), tmin/tmax raygen.rgen:225,227.
BLOK_DEV RayIn primary_ray(const TraceArgs& A, uint32_t x, uint32_t y) {
    const blok_camera& c = A.cam;
    const float qx = rn_div(rn_add(static_cast<float>(x), 0.5f), static_cast<float>(A.frame_w));
    const float qy = rn_div(rn_add(static_cast<float>(y), 0.5f), static_cast<float>(A.frame_h));
    const float u = rn_mul(rn_mul(rn_sub(rn_mul(2.0f, qx), 1.0f), c.tan_half_fov), c.aspect);
    const float v = rn_mul(rn_sub(1.0f, rn_mul(2.0f, qy)), c.tan_half_fov);
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
