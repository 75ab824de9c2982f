// TEST INFRASTRUCTURE: lets the per-ray kernel body (blok_amd/csrc/hip/trace_core.h) compile for the
// host, for sanitizer runs and GPU-less debugging.  Build with -ffp-contract=off.
#ifndef BLOK_HOST_HARNESS_SHIMS_H
#define BLOK_HOST_HARNESS_SHIMS_H
#include <cmath>
#include <cstdint>
#include <cstring>

struct uint4 { uint32_t x, y, z, w; };
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct uint2 { uint32_t x, y; };
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
typedef void* hipStream_t;

inline int __mul24(int a, int b) { return a * b; }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __popc(uint32_t v) { return __builtin_popcount(v); }
inline uint32_t __clz(uint32_t v) { return v ? static_cast<uint32_t>(__builtin_clz(v)) : 32u; }
inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
using std::fabs;
#endif
