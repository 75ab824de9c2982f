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

hipError_t launch_tile_order_finish(const uint32_t* order, const uint32_t* cost_sorted, uint32_t n, uint32_t* rank_of, uint32_t* live_out, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(order_finish_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, order, cost_sorted, n, rank_of, live_out);
    return hipGetLastError();
}

}  // namespace blok
