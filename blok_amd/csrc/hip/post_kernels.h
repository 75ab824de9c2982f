// Launchers of the image-space chain (post_core.h); one lane = one pixel, 64x4-pixel workgroups (one wave per row
// segment: coalesced 16-B-per-lane plane reads).
#ifndef BLOK_POST_KERNELS_H
#define BLOK_POST_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blok {
struct TemporalArgs;
struct VarianceArgs;
struct AtrousArgs;
struct TaaArgs;
struct SharpenArgs;
void launch_temporal(const TemporalArgs& a, hipStream_t stream);
void launch_variance(const VarianceArgs& a, hipStream_t stream);
void launch_atrous(const AtrousArgs& a, hipStream_t stream);
void launch_taa(const TaaArgs& a, hipStream_t stream);
void launch_sharpen(const SharpenArgs& a, hipStream_t stream);
// half / half2 planes -> float planes (state download for tests and tools)
void launch_widen(const uint16_t* src, float* dst, size_t n, hipStream_t stream);
}  // namespace blok
#endif
