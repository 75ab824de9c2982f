// Dense-grid path for BASELINE.json configs[1] ("256^3 dense grid, 1920x1080 primary-ray DDA, coalesced HBM, no SVO"): the
// uploaded id grid itself, kept in HBM in 8x8x8-cell tiles (2 KiB of ids each, so the 64 rays of a wave that meet neighbouring
// cells read the same few lines), plus one occupancy BIT per tile staged into LDS (256^3 -> 32^3 bits = 4 KiB) — the only
// acceleration structure.  One lane = one ray, walking the same T-sorted merge sequence of integer planes as the tree kernel
// (trace_kernels.h), two levels only: whole tiles whose bit is clear are stepped over, cells of occupied tiles one by one
// (Amanatides-Woo with every T evaluated from its integer plane, never accumulated), first filled cell whose clipped interval is
// non-empty wins.  Hence the records equal the tree kernel's and the reference's bit for bit.  No reference counterpart for
// the kernel (the reference has no DDA); the data it walks is Chunk::materialIds in bulk (reference
// blok/include/chunk.hpp:35-36).
#include "dense_kernels.h"
#include "trace_core.h"

namespace blok {

namespace {

// tiles the ids of an [nz][ny][nx] grid; cells beyond the grid inside the last tiles are 0
__global__ __launch_bounds__(256) void dense_tile_kernel(const uint32_t* ids, uint32_t nx, uint32_t ny, uint32_t nz, uint32_t tx, uint32_t ty, uint32_t tz,
                                                          uint32_t* tiled, uint32_t* tile_bits) {
    const uint32_t tile = blockIdx.x;                    // one workgroup of 256 lanes per tile: 2 cells per lane
    if (tile >= tx * ty * tz) return;
    const uint32_t bx = tile % tx, by = (tile / tx) % ty, bz = tile / (tx * ty);
    uint32_t any = 0;
    for (uint32_t c = threadIdx.x; c < 512u; c += 256u) {
        const uint32_t x = bx * 8u + (c & 7u), y = by * 8u + ((c >> 3) & 7u), z = bz * 8u + (c >> 6);
        const uint32_t id = (x < nx && y < ny && z < nz) ? ids[(static_cast<size_t>(z) * ny + y) * nx + x] : 0u;
        tiled[static_cast<size_t>(tile) * 512u + c] = id;
        any |= id;
    }
    if (__syncthreads_or(any != 0u) && threadIdx.x == 0) atomicOr(tile_bits + (tile >> 5), 1u << (tile & 31u));
}

struct DAxis {
    float o, inv, sgn, c;      // as trace_core.h: Axis
    float f;                   // mirrored coordinate 2^23 + q of the current cell / tile corner
    float tF;                  // T of its far plane
    uint32_t n;                // padded extent (multiple of 8) on this axis
    bool neg;
};

__device__ __forceinline__ float dplane(const DAxis& a, float f) { return rn_mul(rn_sub(exact_fma(a.sgn, f, a.c), a.o), a.inv); }

// the slab [q, q + step) of `count` slabs starting at a.f that holds the ray at tS: counts interior planes with T <= tS by
// bisection (T is monotone in q); sets a.f and a.tF (t_far comes in as the far plane of the whole span)
__device__ __forceinline__ void denter(DAxis& a, uint32_t count, float step, float tS) {
    uint32_t lo = 0u, hi = count - 1u;                   // the answer lies in [lo, hi]
    float t_hi = a.tF;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        const float t = dplane(a, a.f + static_cast<float>(mid) * step);
        if (t <= tS) lo = mid; else { hi = mid - 1u; t_hi = t; }
    }
    a.f += static_cast<float>(lo) * step;
    a.tF = t_hi;
}

__global__ __launch_bounds__(64) void dense_kernel(const DenseArgs D) {
    extern __shared__ uint32_t lds_bits[];
    const TraceArgs& A = D.trace;
    const uint32_t lane = threadIdx.x;
    const bool bits_in_lds = D.bit_words <= kDenseLdsWords;
    if (bits_in_lds) {
        for (uint32_t i = lane; i < D.bit_words; i += 64u) lds_bits[i] = D.tile_bits[i];
        __syncthreads();
    }
    const uint32_t bx_count = (A.w + kWaveW - 1u) / kWaveW;
    const uint32_t bx = blockIdx.x % bx_count, by = blockIdx.x / bx_count;
    const uint32_t rx = bx * kWaveW + lane % kWaveW, ry = by * kWaveH + lane / kWaveW;
    if (rx >= A.w || ry >= A.h) return;
    const size_t out_index = static_cast<size_t>(ry) * A.w + rx;
    const Sink sink{A.out ? A.out + out_index : nullptr, A.out_rgba ? A.out_rgba + out_index : nullptr};
    const RayIn r = primary_ray(A, A.x0 + rx, A.y0 + ry);

    DAxis ax[3];
    const float org[3] = {r.ox, r.oy, r.oz}, dir[3] = {r.dx, r.dy, r.dz};
    const uint32_t dims[3] = {D.tx * 8u, D.ty * 8u, D.tz * 8u};
    float t_in = r.tmin, t_out = r.tmax;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        DAxis& x = ax[a];
        x.o = org[a]; x.inv = safe_inv(dir[a]); x.neg = !(x.inv > 0.0f); x.n = dims[a];
        x.sgn = x.neg ? -1.0f : 1.0f;
        x.c = static_cast<float>(A.origin[a] + (x.neg ? static_cast<int>(x.n) : 0)) + (x.neg ? kCoordBias : -kCoordBias);
        x.f = kCoordBias;
        x.tF = dplane(x, kCoordBias + static_cast<float>(x.n));
        t_in = fmaxf(t_in, dplane(x, kCoordBias));
        t_out = fminf(t_out, x.tF);
    }
    if (!(t_in < t_out)) { write_miss(sink); return; }
    float tCur = t_in;
    // tile level first: the tile that holds the ray at tCur
#pragma unroll
    for (int a = 0; a < 3; ++a) denter(ax[a], ax[a].n / 8u, 8.0f, tCur);
    uint32_t lvl = 1u;
    bool found = false;
    uint32_t id = 0u;
    for (;;) {
        // un-mirrored cell coordinates of the current corner
        uint32_t cell[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint32_t q = __float_as_uint(ax[a].f) & 0x7FFFFFu;
            cell[a] = ax[a].neg ? ax[a].n - (lvl ? 8u : 1u) - q : q;
        }
        const uint32_t tile = (cell[0] >> 3) + D.tx * ((cell[1] >> 3) + D.ty * (cell[2] >> 3));
        if (lvl == 1u) {
            const uint32_t word = bits_in_lds ? lds_bits[tile >> 5] : D.tile_bits[tile >> 5];
            if ((word >> (tile & 31u)) & 1u) {
                // occupied tile: the cell inside it that holds the ray at tCur
#pragma unroll
                for (int a = 0; a < 3; ++a) denter(ax[a], 8u, 1.0f, tCur);
                lvl = 0u;
                continue;
            }
        } else {
            id = D.tiled[static_cast<size_t>(tile) * 512u + ((cell[0] & 7u) | ((cell[1] & 7u) << 3) | ((cell[2] & 7u) << 6))];
        }
        const float tExit = fminf(fminf(ax[0].tF, ax[1].tF), ax[2].tF);
        if (lvl == 0u && id != 0u) {
            if (tCur < fminf(tExit, r.tmax)) { found = true; break; }          // the canonical predicate (trace_kernels.h)
        }
        // step across the nearest far plane (x, then y, then z on ties)
        tCur = tExit;
        if (!(tCur < r.tmax)) break;
        const int s = ax[0].tF == tExit ? 0 : (ax[1].tF == tExit ? 1 : 2);
        const float size = lvl ? 8.0f : 1.0f;
        float fs = 0.0f; uint32_t ns = 0u;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (a == s) { ax[a].f += size; fs = ax[a].f; ns = ax[a].n; }
        }
        const uint32_t qs = __float_as_uint(fs) & 0x7FFFFFu;
        if (qs >= ns) break;                                                   // left the grid
        if (lvl == 0u && (qs & 7u) == 0u) {
            // crossed into another tile: back to tile level, corner aligned to the tile
            lvl = 1u;
#pragma unroll
            for (int a = 0; a < 3; ++a) ax[a].f = __uint_as_float(__float_as_uint(ax[a].f) & ~7u);
        }
        const float far = lvl ? 8.0f : 1.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) ax[a].tF = dplane(ax[a], ax[a].f + far);
    }
    if (!found) { write_miss(sink); return; }
    int v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int q = static_cast<int>(__float_as_uint(ax[a].f) & 0x7FFFFFu);
        v[a] = ax[a].neg ? A.origin[a] + static_cast<int>(ax[a].n) - q - 1 : A.origin[a] + q;
    }
    const float hx = rn_add(r.ox, rn_mul(r.dx, tCur)), hy = rn_add(r.oy, rn_mul(r.dy, tCur)), hz = rn_add(r.oz, rn_mul(r.dz, tCur));
    const float ex = rn_sub(hx, rn_add(static_cast<float>(v[0]), 0.5f)), ey = rn_sub(hy, rn_add(static_cast<float>(v[1]), 0.5f)),
                ez = rn_sub(hz, rn_add(static_cast<float>(v[2]), 0.5f));
    const float gx = fabsf(ex), gy = fabsf(ey), gz = fabsf(ez);
    uint32_t face;                                                             // getHitFace, intersect.rint:58-68
    if (gx >= gy && gx >= gz) face = ex > 0.0f ? 0u : 1u;
    else if (gy >= gz)        face = ey > 0.0f ? 2u : 3u;
    else                      face = ez > 0.0f ? 4u : 5u;
    uint4 rec;
    rec.x = __float_as_uint(tCur);
    rec.y = id;
    rec.z = (static_cast<uint32_t>(v[0]) & 0xFFFFu) | (static_cast<uint32_t>(v[1]) << 16);
    rec.w = (static_cast<uint32_t>(v[2]) & 0xFFFFu) | (face << 16) | (1u << 24);
    if (sink.hit) *reinterpret_cast<uint4*>(sink.hit) = rec;
    if (sink.rgba) *sink.rgba = shade_rgba(A.mat_table, A.n_materials, id, face);
}

}  // namespace

void launch_dense_tile(const uint32_t* ids, uint32_t nx, uint32_t ny, uint32_t nz, uint32_t tx, uint32_t ty, uint32_t tz, uint32_t* tiled,
                       uint32_t* tile_bits, hipStream_t stream) {
    hipLaunchKernelGGL(dense_tile_kernel, dim3(tx * ty * tz), dim3(256), 0, stream, ids, nx, ny, nz, tx, ty, tz, tiled, tile_bits);
}

void launch_dense(const DenseArgs& args, hipStream_t stream) {
    const uint32_t blocks = ((args.trace.w + kWaveW - 1u) / kWaveW) * ((args.trace.h + kWaveH - 1u) / kWaveH);
    if (!blocks) return;
    const size_t lds = (args.bit_words <= kDenseLdsWords ? args.bit_words : 0u) * sizeof(uint32_t);
    hipLaunchKernelGGL(dense_kernel, dim3(blocks), dim3(64), lds, stream, args);
}

}  // namespace blok
