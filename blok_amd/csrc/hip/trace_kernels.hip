// See trace_kernels.h for the semantics this file implements.
// Compiled with -ffp-contract=off; the float ops that define results additionally use the
// explicitly rounded intrinsics so that no flag can fuse them.
#include "trace_kernels.h"
#include "trace_core.h"
#include "path_core.h"
#include "beam.h"

namespace blok {

namespace {

// Tile (bx, by) of workgroup b for a rectangle of bx_count x by_count tiles; false if the workgroup has no tile.
__device__ __forceinline__ bool block_to_tile(uint32_t b, uint32_t grid, uint32_t bx_count, uint32_t by_count,
                                              uint32_t& bx, uint32_t& by) {
#if BLOK_XCD_MAP == 0
    (void)grid;
    bx = b % bx_count; by = b / bx_count;
    return by < by_count;
#else
    const uint32_t vb = (b & 7u) * (grid >> 3) + (b >> 3);        // XCD k owns virtual blocks [k*grid/8, (k+1)*grid/8)
#if BLOK_XCD_MAP == 1
    bx = vb % bx_count; by = vb / bx_count;
    return by < by_count;
#else
    const uint32_t super_x = (bx_count + kSuper - 1u) / kSuper;
    const uint32_t st = vb / (kSuper * kSuper), in = vb % (kSuper * kSuper);
    // Morton order inside the supertile: even bits -> x, odd bits -> y
    uint32_t mx = in & 0x55u, my = (in >> 1) & 0x55u;
    mx = (mx | (mx >> 1)) & 0x33u; mx = (mx | (mx >> 2)) & 0x0Fu;
    my = (my | (my >> 1)) & 0x33u; my = (my | (my >> 2)) & 0x0Fu;
    bx = (st % super_x) * kSuper + mx; by = (st / super_x) * kSuper + my;
    return bx < bx_count && by < by_count;
#endif
#endif
}

template <RayMode MODE>
__global__ __launch_bounds__(kBlock) void trace_kernel(const TraceArgs A) {
    extern __shared__ uint4 lds_stack[];       // [levels-1][kBlock]
    const uint32_t tid = threadIdx.x;
    uint4* stk = lds_stack + tid;

    if constexpr (MODE == RayMode::Rays) {
        const uint32_t i = blockIdx.x * kBlock + tid;
        if (i >= A.n_rays) return;
        const blok_ray ray = A.rays[i];
        RayIn r{ray.org[0], ray.org[1], ray.org[2], ray.dir[0], ray.dir[1], ray.dir[2], ray.tmin, ray.tmax};
        trace_one(A, r, stk, Sink{A.out ? A.out + i : nullptr, A.out_rgba ? A.out_rgba + i : nullptr});
        return;
    } else {
        // a block is a 16x16 pixel tile, a wave an 8x8 sub-tile (coherent rays per wave)
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        const uint32_t lx = (wave & 1u) * kWaveW + (lane % kWaveW);
        const uint32_t ly = (wave >> 1) * kWaveH + (lane / kWaveW);
        // where this wave's 8x8 pixels start inside the block (wave-uniform)
        const uint32_t wave_x = __builtin_amdgcn_readfirstlane((wave & 1u) * kWaveW), wave_y = __builtin_amdgcn_readfirstlane((wave >> 1) * kWaveH);
        uint32_t x, y;
        float t0 = 0.0f;                           // start parameter of this wave's beam tile (beam.h); kBeamNone: the pre-pass
        size_t out_index;                          // has already written the tile's pixels as misses, nothing left to do
        bool inside;
        if constexpr (MODE == RayMode::Rect) {
            const uint32_t bx_count = (A.w + kTileW - 1u) / kTileW, by_count = (A.h + kTileH - 1u) / kTileH;
            uint32_t bx, by;
            if (!block_to_tile(blockIdx.x, gridDim.x, bx_count, by_count, bx, by)) return;
            if (A.beam) {
                t0 = A.beam[((by * kTileH + wave_y) / A.beam_tile) * A.beam_bx + (bx * kTileW + wave_x) / A.beam_tile];
                if (t0 >= kBeamNone) return;
            }
            const uint32_t rx = bx * kTileW + lx, ry = by * kTileH + ly;
            inside = rx < A.w && ry < A.h;
            x = A.x0 + rx; y = A.y0 + ry;
            out_index = static_cast<size_t>(ry) * A.w + rx;
        } else {
            const uint32_t per_side = A.tile / kTileW;                  // blocks per tile row
            const uint32_t per_tile = per_side * (A.tile / kTileH);
            const uint32_t local_tile = blockIdx.x / per_tile, sub = blockIdx.x % per_tile;
            const uint32_t global_tile = A.rank + local_tile * A.n_ranks;
            const uint32_t tx = global_tile % A.tiles_x, ty = global_tile / A.tiles_x;
            const uint32_t ix = (sub % per_side) * kTileW + lx, iy = (sub / per_side) * kTileH + ly;
            x = tx * A.tile + ix; y = ty * A.tile + iy;
            out_index = static_cast<size_t>(local_tile) * A.tile * A.tile + static_cast<size_t>(iy) * A.tile + ix;
            inside = x < A.frame_w && y < A.frame_h;
            if (global_tile >= A.tiles_total) {                        // whole workgroup: padding tile of the last round
                write_miss(Sink{A.out ? A.out + out_index : nullptr, A.out_rgba ? A.out_rgba + out_index : nullptr});
                return;
            }
            if (A.beam) {
                const uint32_t beams_per_side = A.tile / A.beam_tile;
                t0 = A.beam[(local_tile * beams_per_side + ((sub / per_side) * kTileH + wave_y) / A.beam_tile) * beams_per_side +
                            ((sub % per_side) * kTileW + wave_x) / A.beam_tile];
                if (t0 >= kBeamNone) return;
            }
        }
        const Sink sink{A.out ? A.out + out_index : nullptr, A.out_rgba ? A.out_rgba + out_index : nullptr};
        if (!inside) {
            if constexpr (MODE == RayMode::Tiles) write_miss(sink);     // the tile buffer is dense
            return;
        }
        RayIn r = primary_ray(A, x, y);
        r.tmin = fmaxf(r.tmin, t0);
        trace_one(A, r, stk, sink);
    }
}

// One wave per beam tile: TraceArgs::beam[tile] = conservative start parameter of the tile's rays, or kBeamNone.
template <RayMode MODE>
__global__ __launch_bounds__(64) void beam_kernel(const TraceArgs A, const uint32_t n_beam_tiles) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (b >= n_beam_tiles) return;
    const uint32_t B = A.beam_tile;
    uint32_t px, py, px_end, py_end;                                   // frame pixels [px, px_end) x [py, py_end)
    [[maybe_unused]] uint32_t tile_x0 = 0, tile_y0 = 0;                // Tiles: frame origin of the screen tile and its slot
    [[maybe_unused]] size_t tile_base = 0;                             // in the rank's dense tile buffer
    if constexpr (MODE == RayMode::Rect) {
        px = A.x0 + (b % A.beam_bx) * B; py = A.y0 + (b / A.beam_bx) * B;
        px_end = min(px + B, A.x0 + A.w); py_end = min(py + B, A.y0 + A.h);
    } else {
        const uint32_t per_side = A.tile / B, per_tile = per_side * per_side;
        const uint32_t local_tile = b / per_tile, sub = b % per_tile;
        const uint32_t global_tile = A.rank + local_tile * A.n_ranks;
        if (global_tile >= A.tiles_total) { if (lane == 0) A.beam[b] = kBeamNone; return; }
        tile_x0 = (global_tile % A.tiles_x) * A.tile; tile_y0 = (global_tile / A.tiles_x) * A.tile;
        tile_base = static_cast<size_t>(local_tile) * A.tile * A.tile;
        px = tile_x0 + (sub % per_side) * B; py = tile_y0 + (sub / per_side) * B;
        px_end = px + B; py_end = py + B;
    }
    const float t0 = beam_start(A, static_cast<float>(px), static_cast<float>(py), static_cast<float>(px_end), static_cast<float>(py_end), lane);
    if (lane == 0) A.beam[b] = t0;
    if (t0 >= kBeamNone && (A.out || A.out_rgba)) {
        // no ray of this tile can hit anything: its pixels are written here, 64 at a time, and the tile's trace waves exit at once
        const uint32_t tw = px_end - px, n = tw * (py_end - py);
        for (uint32_t i = lane; i < n; i += 64u) {
            const uint32_t x = px + i % tw, y = py + i / tw;
            size_t out_index;
            if constexpr (MODE == RayMode::Rect) out_index = static_cast<size_t>(y - A.y0) * A.w + (x - A.x0);
            else out_index = tile_base + static_cast<size_t>(y - tile_y0) * A.tile + (x - tile_x0);
            write_miss(Sink{A.out ? A.out + out_index : nullptr, A.out_rgba ? A.out_rgba + out_index : nullptr});
        }
    }
}

__global__ __launch_bounds__(64) void sun_map_kernel(const SunMapArgs a) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (b >= a.nu * a.nv) return;
    const SunFrame f{{a.u[0], a.u[1], a.u[2]}, {a.v[0], a.v[1], a.v[2]}, {a.s[0], a.s[1], a.s[2]}};
    const float u_lo = a.u0 + static_cast<float>(b % a.nu) * a.texel, v_lo = a.v0 + static_cast<float>(b / a.nu) * a.texel;
    const float last = prism_far(a.trace, f, u_lo, u_lo + a.texel, v_lo, v_lo + a.texel, lane);
    if (lane == 0) a.map[b] = last;
}

// raygen.rgen main(): one lane per pixel of the rectangle, same 16x16 / 8x8 pixel mapping as the trace kernel.
#ifndef BLOK_PATH_WAVES
#define BLOK_PATH_WAVES 6        // waves per SIMD the path kernel is compiled for (register budget 512 / waves): 4 (109 VGPRs) 74.2 ms, 5 70.7, 6 69.9, 8 71.1 at 4K 64 spp
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(BLOK_PATH_WAVES, BLOK_PATH_WAVES))) void path_kernel(const PathArgs P) {
    extern __shared__ uint4 lds_stack[];
    const TraceArgs& A = P.trace;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t lx = (wave & 1u) * kWaveW + (lane % kWaveW);
    const uint32_t ly = (wave >> 1) * kWaveH + (lane / kWaveW);
    const uint32_t bx_count = (A.w + kTileW - 1u) / kTileW, by_count = (A.h + kTileH - 1u) / kTileH;
    uint32_t bx, by;
    if (!block_to_tile(blockIdx.x, gridDim.x, bx_count, by_count, bx, by)) return;
    const uint32_t rx = bx * kTileW + lx, ry = by * kTileH + ly;
    if (rx >= A.w || ry >= A.h) return;
    float t0 = 0.0f;
    if (A.beam) t0 = A.beam[__builtin_amdgcn_readfirstlane(((ry - lane / kWaveW) / A.beam_tile) * A.beam_bx + (rx - lane % kWaveW) / A.beam_tile)];
    shade_pixel(P, A.x0 + rx, A.y0 + ry, static_cast<size_t>(ry) * A.w + rx, lds_stack + tid, t0);
}

__global__ __launch_bounds__(256) void tonemap_kernel(const TonemapArgs T) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < T.n) T.ldr[i] = tonemap_pixel(T, i);
}

__global__ __launch_bounds__(256) void accumulate_kernel(const AccumArgs T) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < T.n) accumulate_pixel(T, i);
}

template <typename Elem>
__global__ __launch_bounds__(256) void untile_kernel(const UntileArgs U) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= U.frame_w * U.frame_h) return;
    const uint32_t x = i % U.frame_w, y = i / U.frame_w;
    const uint32_t g = (y / U.tile) * U.tiles_x + x / U.tile;
    const uint32_t rank = g % U.n_ranks, local = g / U.n_ranks;
    const size_t src = (static_cast<size_t>(rank) * U.tiles_per_rank_max + local) * U.tile * U.tile +
                       static_cast<size_t>(y % U.tile) * U.tile + (x % U.tile);
    static_cast<Elem*>(U.frame)[i] = static_cast<const Elem*>(U.gathered)[src];
}

}  // namespace

void launch_trace(RayMode mode, const TraceArgs& args, uint32_t n_blocks, hipStream_t stream) {
    if (n_blocks == 0) return;
    const size_t lds = static_cast<size_t>(args.levels > 1 ? args.levels - 1 : 1) * kBlock * sizeof(uint4);
    switch (mode) {
        case RayMode::Rect:  hipLaunchKernelGGL(trace_kernel<RayMode::Rect>,  dim3(n_blocks), dim3(kBlock), lds, stream, args); break;
        case RayMode::Tiles: hipLaunchKernelGGL(trace_kernel<RayMode::Tiles>, dim3(n_blocks), dim3(kBlock), lds, stream, args); break;
        case RayMode::Rays:  hipLaunchKernelGGL(trace_kernel<RayMode::Rays>,  dim3(n_blocks), dim3(kBlock), lds, stream, args); break;
    }
}

uint32_t beam_tiles(RayMode mode, const TraceArgs& a, uint32_t tiles_of_rank) {
    if (mode == RayMode::Rect) return ((a.w + a.beam_tile - 1u) / a.beam_tile) * ((a.h + a.beam_tile - 1u) / a.beam_tile);
    return tiles_of_rank * (a.tile / a.beam_tile) * (a.tile / a.beam_tile);
}

void launch_beam(RayMode mode, const TraceArgs& args, uint32_t n_beam_tiles, hipStream_t stream) {
    if (n_beam_tiles == 0 || mode == RayMode::Rays) return;
    if (mode == RayMode::Rect) hipLaunchKernelGGL(beam_kernel<RayMode::Rect>, dim3(n_beam_tiles), dim3(64), 0, stream, args, n_beam_tiles);
    else hipLaunchKernelGGL(beam_kernel<RayMode::Tiles>, dim3(n_beam_tiles), dim3(64), 0, stream, args, n_beam_tiles);
}

void launch_sun_map(const SunMapArgs& args, hipStream_t stream) {
    const uint32_t n = args.nu * args.nv;
    if (n) hipLaunchKernelGGL(sun_map_kernel, dim3(n), dim3(64), 0, stream, args);
}

void launch_paths(const PathArgs& args, uint32_t n_blocks, hipStream_t stream) {
    if (n_blocks == 0) return;
    const uint32_t levels = args.trace.levels;
    const size_t lds = static_cast<size_t>(levels > 1 ? levels - 1 : 1) * kBlock * sizeof(uint4);
    hipLaunchKernelGGL(path_kernel, dim3(n_blocks), dim3(kBlock), lds, stream, args);
}

void launch_tonemap(const TonemapArgs& args, hipStream_t stream) {
    if (args.n) hipLaunchKernelGGL(tonemap_kernel, dim3((args.n + 255u) / 256u), dim3(256), 0, stream, args);
}

void launch_accumulate(const AccumArgs& args, hipStream_t stream) {
    if (args.n) hipLaunchKernelGGL(accumulate_kernel, dim3((args.n + 255u) / 256u), dim3(256), 0, stream, args);
}

void launch_untile(const UntileArgs& args, hipStream_t stream) {
    const uint32_t n = args.frame_w * args.frame_h;
    if (args.elem_bytes == 16) hipLaunchKernelGGL(untile_kernel<uint4>, dim3((n + 255u) / 256u), dim3(256), 0, stream, args);
    else hipLaunchKernelGGL(untile_kernel<uint32_t>, dim3((n + 255u) / 256u), dim3(256), 0, stream, args);
}

}  // namespace blok
