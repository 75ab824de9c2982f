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
// A camera in motion: order_out = the tiles by descending class of their DILATED cost (the largest within `radius` <= 8 tiles, 64 classes,
// four to the octave), rank_of its inverse, *live_out = how many leading entries lie within `radius` tiles of a tile that walked, and
// depth_out[64][3] = partial (count, sum, sum of squares) of 1 / start parameter over the frame's live beam tiles, to be added up by the
// reader (beam: n_beams floats, or — slots non-null — 64-bit words `serial << 32 | float bits` as a joint launch publishes them).
// live_out and depth_out may be pinned host memory.  Three small launches; scratch: tile_order_class_sort_bytes().
constexpr uint32_t kOrderDepthPartials = 64;
size_t tile_order_class_sort_bytes(uint32_t tiles_x, uint32_t tiles_y);
size_t tile_order_class_sort_bytes_max(uint32_t n_tiles);      // ... for any grid of at most n_tiles tiles
hipError_t launch_tile_order_class_sort(const uint32_t* cost, uint32_t tiles_x, uint32_t tiles_y, uint32_t radius, void* scratch, uint32_t* order_out, uint32_t* rank_of,
                                        uint32_t* live_out, const float* beam, const unsigned long long* slots, uint32_t serial, uint32_t n_beams, float* depth_out,
                                        hipStream_t stream);
}  // namespace blok
#endif
