// Longest-first order of the walk's workgroups (tile_order.hip).
#ifndef BLOK_TILE_ORDER_H
#define BLOK_TILE_ORDER_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace blok {
size_t tile_order_temp_bytes(uint32_t n);
hipError_t launch_iota(uint32_t* v, uint32_t n, hipStream_t stream);
hipError_t launch_tile_order_sort(const uint32_t* cost, uint32_t* cost_sorted_scratch, const uint32_t* iota, uint32_t* order_out, void* temp,
                                  size_t temp_bytes, uint32_t n, hipStream_t stream);
}  // namespace blok
#endif
