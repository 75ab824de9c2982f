// Longest-first scheduling of the walk's workgroups (tile_order.h): a radix sort of the tiles by the clocks their waves
// spent in the previous frame of the same launch geometry.  No reference counterpart (the reference hands scheduling to the
// Vulkan driver, renderer_raytracing.cpp:666-685).
#include "tile_order.h"

#include <hipcub/hipcub.hpp>

namespace blok {

namespace {
__global__ __launch_bounds__(256) void iota_kernel(uint32_t* v, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) v[i] = i;
}
__device__ __forceinline__ uint32_t sort_key(uint32_t cost) { return (cost >> 8) & 0xFFFFu; }      // the bits the sort looks at
__global__ __launch_bounds__(256) void order_finish_kernel(const uint32_t* order, const uint32_t* cost_sorted, uint32_t n, uint32_t* rank_of, uint32_t* live_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    rank_of[order[i]] = i;
    const bool live = sort_key(cost_sorted[i]) != 0u;                     // descending: the live entries are a prefix; its last one reports its length
    if (live && (i + 1u == n || sort_key(cost_sorted[i + 1u]) == 0u)) *live_out = i + 1u;
    if (i == 0u && !live) *live_out = 0u;
}
// ---- a camera in motion: counting sort of DILATED costs, three small launches behind the frame -------------------------------------
// The key of a tile is the largest cost within `radius` tiles of it — the heavy tiles of the next frame are NEAR the heavy tiles of this
// one (silhouettes, grazing rays), not on them: ordering by the undilated costs of even a quarter of a degree ago is no better than
// row-major, by the dilated ones within a few per cent of the frame's own (profiles/r03_moving_order_experiment.txt) — reduced to one of
// 64 classes, four to the octave of clocks (a finer order of dilated costs gains nothing).  Sorted by class, heaviest first, within a
// class by 16x16-tile block of the screen and position in it: (1) every block dilates its tiles through LDS and counts its classes,
// (2) one workgroup scans the 64 x blocks counts (laid out heaviest class first, so the scan IS the order) and leaves the length of the
// live prefix, (3) every block writes its tiles to their places, and the inverse permutation with them.  Deterministic; reads the live
// cost buffer once per use (frames on other streams may be writing it: any mixture of old and new costs is as good a key).
constexpr uint32_t kClassBlock = 16u, kClasses = 64u, kMaxRadius = 8u;
__device__ __forceinline__ uint32_t cost_class(uint32_t m) {          // 0 = nothing walked near here; 1..63 by quarter octaves from 256 clocks up
    if (m == 0u) return 0u;
    const uint32_t q = __float_as_uint(static_cast<float>(m)) >> 21;     // exponent and two mantissa bits
    return q <= 540u ? 1u : (q - 539u > 63u ? 63u : q - 539u);
}
// flat index of (class, block) in the count table: heaviest class first
__device__ __forceinline__ uint32_t count_slot(uint32_t cls, uint32_t block, uint32_t n_blocks) { return (kClasses - 1u - cls) * n_blocks + block; }

__global__ __launch_bounds__(256) void order_class_kernel(const uint32_t* cost, uint8_t* cls_out, uint32_t* counts, uint32_t tiles_x, uint32_t tiles_y, uint32_t radius) {
    __shared__ uint32_t region[(kClassBlock + 2u * kMaxRadius) * (kClassBlock + 2u * kMaxRadius)];
    __shared__ uint32_t rows[(kClassBlock + 2u * kMaxRadius) * kClassBlock];
    __shared__ uint32_t hist[kClasses];
    const uint32_t side = kClassBlock + 2u * radius;
    const uint32_t blocks_x = (tiles_x + kClassBlock - 1u) / kClassBlock;
    const uint32_t bx = blockIdx.x % blocks_x, by = blockIdx.x / blocks_x;
    const int32_t ox = static_cast<int32_t>(bx * kClassBlock) - static_cast<int32_t>(radius), oy = static_cast<int32_t>(by * kClassBlock) - static_cast<int32_t>(radius);
    if (threadIdx.x < kClasses) hist[threadIdx.x] = 0u;
    for (uint32_t i = threadIdx.x; i < side * side; i += 256u) {
        const int32_t x = ox + static_cast<int32_t>(i % side), y = oy + static_cast<int32_t>(i / side);
        region[i] = (x >= 0 && y >= 0 && x < static_cast<int32_t>(tiles_x) && y < static_cast<int32_t>(tiles_y)) ? cost[static_cast<uint32_t>(y) * tiles_x + static_cast<uint32_t>(x)] : 0u;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < side * kClassBlock; i += 256u) {          // along x
        const uint32_t col = i % kClassBlock, row = i / kClassBlock;
        uint32_t m = 0u;
        for (uint32_t k = 0; k <= 2u * radius; ++k) m = max(m, region[row * side + col + k]);
        rows[i] = m;
    }
    __syncthreads();
    const uint32_t lx = threadIdx.x % kClassBlock, ly = threadIdx.x / kClassBlock;
    const uint32_t tx = bx * kClassBlock + lx, ty = by * kClassBlock + ly;
    if (tx < tiles_x && ty < tiles_y) {                                          // along y
        uint32_t m = 0u;
        for (uint32_t k = 0; k <= 2u * radius; ++k) m = max(m, rows[(ly + k) * kClassBlock + lx]);
        const uint32_t c = cost_class(m);
        cls_out[ty * tiles_x + tx] = static_cast<uint8_t>(c);
        atomicAdd(&hist[c], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kClasses) counts[count_slot(threadIdx.x, blockIdx.x, gridDim.x)] = hist[threadIdx.x];
}

// (2) workgroup c scans the block counts of class c in place (exclusive: where block b's tiles of the class start inside the class) and
// leaves the class total; it also reduces its share of the frame's beam tiles to a partial (count, sum, sum of squares) of their inverse
// start parameters — depth_out[c * 3 ..], summed on the host (api.hip): no workgroup waits for another, nothing is added atomically.
__global__ __launch_bounds__(256) void order_rows_kernel(uint32_t* counts, uint32_t n_blocks, uint32_t* totals,
                                                         const float* beam, const unsigned long long* slots, uint32_t serial, uint32_t n_beams, float* depth_out) {
    __shared__ uint32_t wave_sum[4];
    __shared__ float part[3][4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* const row = counts + static_cast<size_t>(kClasses - 1u - blockIdx.x) * n_blocks;      // heaviest class first (count_slot)
    uint32_t carry = 0u;
    for (uint32_t base = 0; base < n_blocks; base += 256u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? row[i] : 0u;
        uint32_t incl = v;
        for (uint32_t off = 1u; off < 64u; off <<= 1) { const uint32_t u = __shfl_up(incl, off); if (lane >= off) incl += u; }
        __syncthreads();                                                         // (the previous round's wave_sum has been read)
        if (lane == 63u) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wave; ++w) before += wave_sum[w];
        if (i < n_blocks) row[i] = before + incl - v;
        carry += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
    const uint32_t share = (n_beams + kClasses - 1u) / kClasses, b0 = blockIdx.x * share, b1 = min(b0 + share, n_beams);
    float cnt = 0.0f, s1 = 0.0f, s2 = 0.0f;
    for (uint32_t i = b0 + threadIdx.x; i < b1; i += 256u) {
        float t;
        if (slots) { const unsigned long long v = slots[i]; t = static_cast<uint32_t>(v >> 32) == serial ? __uint_as_float(static_cast<uint32_t>(v)) : 3.0e38f; }
        else t = beam[i];
        if (t < 1.0e38f) { const float inv = 1.0f / fmaxf(t, 1.0f); cnt += 1.0f; s1 += inv; s2 += inv * inv; }
    }
    for (int off = 32; off > 0; off >>= 1) { cnt += __shfl_down(cnt, off); s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
    if (lane == 0u) { part[0][wave] = cnt; part[1][wave] = s1; part[2][wave] = s2; }
    __syncthreads();
    if (threadIdx.x < 3u) depth_out[blockIdx.x * 3u + threadIdx.x] = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
}

// (3) every block writes its tiles to their places: class base (the totals of the heavier classes) + the block's start inside the class +
// the tile's rank among the block's tiles of that class; the inverse permutation with them.  Block 0 leaves the length of the live prefix.
__global__ __launch_bounds__(256) void order_scatter_kernel(const uint8_t* cls_in, const uint32_t* offsets, const uint32_t* totals, uint32_t* order, uint32_t* rank_of,
                                                            uint32_t* live_out, uint32_t tiles_x, uint32_t tiles_y) {
    __shared__ uint32_t wave_count[4][kClasses];
    __shared__ uint32_t class_base[kClasses];
    const uint32_t blocks_x = (tiles_x + kClassBlock - 1u) / kClassBlock;
    const uint32_t bx = blockIdx.x % blocks_x, by = blockIdx.x / blocks_x;
    const uint32_t tx = bx * kClassBlock + threadIdx.x % kClassBlock, ty = by * kClassBlock + threadIdx.x / kClassBlock;
    const bool valid = tx < tiles_x && ty < tiles_y;
    const uint32_t tile = ty * tiles_x + tx;
    const uint32_t cls = valid ? cls_in[tile] : 0xFFu;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    wave_count[wave][lane] = 0u;                                                 // kClasses == 64 == lanes
    if (wave == 0u) {
        // lane l stands for class 63 - l: an exclusive scan over the lanes = the tiles of all heavier classes
        const uint32_t mine = totals[kClasses - 1u - lane];
        uint32_t incl = mine;
        for (uint32_t off = 1u; off < 64u; off <<= 1) { const uint32_t u = __shfl_up(incl, off); if (lane >= off) incl += u; }
        class_base[kClasses - 1u - lane] = incl - mine;
        if (blockIdx.x == 0u && lane == 63u) *live_out = incl - mine;           // class 0 comes last: everything in front of it walked, or lies within `radius` tiles of a tile that did
    }
    uint32_t rank_in_wave = 0u;
    unsigned long long remaining = __ballot(valid);
    while (remaining) {                                                          // one round per class present in the wave
        const uint32_t c = __builtin_amdgcn_readlane(cls, static_cast<int>(__builtin_ctzll(remaining)));
        const unsigned long long m = __ballot(valid && cls == c);
        if (valid && cls == c) rank_in_wave = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
        if (lane == 0u) wave_count[wave][c] = static_cast<uint32_t>(__builtin_popcountll(m));
        remaining &= ~m;
    }
    __syncthreads();
    if (!valid) return;
    uint32_t at = class_base[cls] + offsets[count_slot(cls, blockIdx.x, gridDim.x)] + rank_in_wave;
    for (uint32_t w = 0; w < wave; ++w) at += wave_count[w][cls];
    order[at] = tile;
    rank_of[tile] = at;
}

}  // namespace

size_t tile_order_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                                        static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), static_cast<int>(n), 8, 24);
    return bytes;
}

hipError_t launch_iota(uint32_t* v, uint32_t n, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(iota_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, v, n);
    return hipGetLastError();
}

// order_out = tile indices by descending cost.  Sixteen key bits (8..23 of the clock count: 256-clock resolution up to 16 M
// clocks) in two radix passes.  `cost` must not change while the sort runs (a radix sort reads its keys more than once): the caller
// hands in a snapshot of the live cost buffer (api.hip: order_after_launch).
hipError_t launch_tile_order_sort(const uint32_t* cost, uint32_t* cost_sorted_scratch, const uint32_t* iota, uint32_t* order_out, void* temp,
                                  size_t temp_bytes, uint32_t n, hipStream_t stream) {
    return hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_bytes, cost, cost_sorted_scratch, iota, order_out, static_cast<int>(n), 8, 24, stream);
}

size_t tile_order_class_sort_bytes(uint32_t tiles_x, uint32_t tiles_y) {
    const size_t blocks = static_cast<size_t>((tiles_x + kClassBlock - 1u) / kClassBlock) * ((tiles_y + kClassBlock - 1u) / kClassBlock);
    return (blocks + 1u) * kClasses * sizeof(uint32_t) + static_cast<size_t>(tiles_x) * tiles_y;      // the count table, the class totals, then a class byte per tile
}

size_t tile_order_class_sort_bytes_max(uint32_t n) {
    // blocks <= (tx/16 + 1)(ty/16 + 1) <= n/256 + n/8 + 1 for tx ty <= n: 256 bytes of counts per block, one class byte per tile
    return static_cast<size_t>(n) * 35u + 4096u;
}

hipError_t launch_tile_order_class_sort(const uint32_t* cost, uint32_t tiles_x, uint32_t tiles_y, uint32_t radius, void* scratch, uint32_t* order_out, uint32_t* rank_of,
                                        uint32_t* live_out, const float* beam, const unsigned long long* slots, uint32_t serial, uint32_t n_beams, float* depth_out,
                                        hipStream_t stream) {
    if (!tiles_x || !tiles_y) return hipSuccess;
    if (radius > kMaxRadius) radius = kMaxRadius;
    const uint32_t blocks = ((tiles_x + kClassBlock - 1u) / kClassBlock) * ((tiles_y + kClassBlock - 1u) / kClassBlock);
    uint32_t* counts = static_cast<uint32_t*>(scratch);
    uint32_t* totals = counts + static_cast<size_t>(blocks) * kClasses;
    uint8_t* cls = reinterpret_cast<uint8_t*>(totals + kClasses);
    hipLaunchKernelGGL(order_class_kernel, dim3(blocks), dim3(256), 0, stream, cost, cls, counts, tiles_x, tiles_y, radius);
    hipLaunchKernelGGL(order_rows_kernel, dim3(kClasses), dim3(256), 0, stream, counts, blocks, totals, beam, slots, serial, n_beams, depth_out);
    hipLaunchKernelGGL(order_scatter_kernel, dim3(blocks), dim3(256), 0, stream, cls, counts, totals, order_out, rank_of, live_out, tiles_x, tiles_y);
    return hipGetLastError();
}

hipError_t launch_tile_order_finish(const uint32_t* order, const uint32_t* cost_sorted, uint32_t n, uint32_t* rank_of, uint32_t* live_out, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(order_finish_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, order, cost_sorted, n, rank_of, live_out);
    return hipGetLastError();
}

}  // namespace blok
