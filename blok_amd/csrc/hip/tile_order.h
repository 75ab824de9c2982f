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
// Behind the sort: rank_of[tile] = the tile's position in `order`, and *live_out = how many leading entries of the order have a
// non-zero sort key (tiles whose wave walked last time) — the prefix a joint launch dispatches walk waves for.  live_out may be
// pinned host memory.
hipError_t launch_tile_order_finish(const uint32_t* order, const uint32_t* cost_sorted, uint32_t n, uint32_t* rank_of, uint32_t* live_out, hipStream_t stream);
}  // namespace blok
#endif
