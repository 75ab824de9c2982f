// Kernels of the image-space chain: see post_core.h for the per-pixel bodies and the HBM layout.
// Compiled with -ffp-contract=off like the trace kernels.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "post_kernels.h"
#include "trace_kernels.h"
#include "post_core.h"

namespace blok {
namespace {

constexpr uint32_t kPostBx = 64, kPostBy = 4;

template <class Args, void (*Body)(const Args&, int, int)>
__global__ __launch_bounds__(kPostBx * kPostBy) void pixel_kernel(const Args a, const uint32_t w, const uint32_t h) {
    const uint32_t x = blockIdx.x * kPostBx + threadIdx.x, y = blockIdx.y * kPostBy + threadIdx.y;
    if (x < w && y < h) Body(a, static_cast<int>(x), static_cast<int>(y));
}

__global__ __launch_bounds__(kPostBx * kPostBy) void sharpen_kernel(const SharpenArgs a) {
    __shared__ float lut[256];
    static_assert(kPostBx * kPostBy == 256, "one table entry per lane");
    const uint32_t t = threadIdx.y * kPostBx + threadIdx.x;
    lut[t] = unorm8_to_float(t);
    __syncthreads();
    const uint32_t x = blockIdx.x * kPostBx + threadIdx.x, y = blockIdx.y * kPostBy + threadIdx.y;
    if (x < a.w && y < a.h) sharpen_pixel(a, static_cast<int>(x), static_cast<int>(y), lut);
}

__global__ __launch_bounds__(256) void widen_kernel(const uint16_t* src, float* dst, size_t n) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x;
    if (i < n) dst[i] = h2f(src[i]);
}

template <class Args, void (*Body)(const Args&, int, int)>
void launch_pixels(const Args& a, uint32_t w, uint32_t h, hipStream_t stream) {
    if (!w || !h) return;
    hipLaunchKernelGGL((pixel_kernel<Args, Body>), dim3((w + kPostBx - 1) / kPostBx, (h + kPostBy - 1) / kPostBy), dim3(kPostBx, kPostBy), 0, stream, a, w, h);
}

}  // namespace

void launch_temporal(const TemporalArgs& a, hipStream_t s) { launch_pixels<TemporalArgs, temporal_pixel>(a, a.f.w, a.f.h, s); }
void launch_variance(const VarianceArgs& a, hipStream_t s) { launch_pixels<VarianceArgs, variance_pixel>(a, a.f.w, a.f.h, s); }
void launch_atrous(const AtrousArgs& a, hipStream_t s) { launch_pixels<AtrousArgs, atrous_pixel>(a, a.w, a.h, s); }
void launch_taa(const TaaArgs& a, hipStream_t s) { launch_pixels<TaaArgs, taa_pixel>(a, a.w, a.h, s); }
void launch_sharpen(const SharpenArgs& a, hipStream_t s) {
    if (a.w && a.h) hipLaunchKernelGGL(sharpen_kernel, dim3((a.w + kPostBx - 1) / kPostBx, (a.h + kPostBy - 1) / kPostBy), dim3(kPostBx, kPostBy), 0, s, a);
}
void launch_widen(const uint16_t* src, float* dst, size_t n, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(widen_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream, src, dst, n);
}

}  // namespace blok
