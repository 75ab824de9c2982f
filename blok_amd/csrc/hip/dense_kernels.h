// Dense-grid two-level DDA (dense_kernels.hip): the tiled id grid and its tile occupancy bits.
#ifndef BLOK_DENSE_KERNELS_H
#define BLOK_DENSE_KERNELS_H
#include "trace_kernels.h"

namespace blok {

constexpr uint32_t kDenseLdsWords = 8192u;     // tile occupancy bits staged into LDS when they fit 32 KiB (grids up to 512^3 ... 64^3 tiles)

struct DenseArgs {
    TraceArgs trace;               // camera, frame, rectangle, outputs, material table; origin = grid corner (voxelSize 1)
    const uint32_t* tiled;         // ids, 8x8x8-cell tiles of 512 words: tile (tx, ty, tz) row-major, cell x | y << 3 | z << 6 inside
    const uint32_t* tile_bits;     // one bit per tile: any id != 0
    uint32_t tx, ty, tz;           // tiles per axis
    uint32_t bit_words;
};

void launch_dense_tile(const uint32_t* ids, uint32_t nx, uint32_t ny, uint32_t nz, uint32_t tx, uint32_t ty, uint32_t tz, uint32_t* tiled,
                       uint32_t* tile_bits, hipStream_t stream);
void launch_dense(const DenseArgs& args, hipStream_t stream);

}  // namespace blok
#endif
