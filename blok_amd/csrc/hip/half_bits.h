// binary16 <-> binary32 (round to nearest even), bit patterns in uint16_t: what a 16-bit float image holds.
#ifndef BLOK_HALF_BITS_H
#define BLOK_HALF_BITS_H
#include "trace_core.h"
#ifndef BLOK_TRACE_HOST_HARNESS
#include <hip/hip_fp16.h>
#endif

namespace blok {

#ifdef BLOK_TRACE_HOST_HARNESS
inline uint16_t f2h(float f) {
    uint32_t x = __float_as_uint(f);
    const uint16_t sign = static_cast<uint16_t>((x >> 16) & 0x8000u);
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u);
    if (x >= 0x477FF000u) return sign | 0x7C00u;
    if (x < 0x33000001u) return sign;
    const int e = static_cast<int>(x >> 23) - 127;
    const uint32_t m = (x & 0x007FFFFFu) | 0x00800000u;
    const int drop = e < -14 ? 13 + (-14 - e) : 13;
    uint32_t r = m >> drop;
    const uint32_t rem = m & ((1u << drop) - 1u), half = 1u << (drop - 1);
    if (rem > half || (rem == half && (r & 1u))) ++r;
    // r counts units of 2^(e-23+drop); normal halves: exponent field = e + 15 with the leading one of r at bit 10
    if (e < -14) return sign | static_cast<uint16_t>(r);                        // subnormal (or rounds up into the first normal)
    return sign | static_cast<uint16_t>((static_cast<uint32_t>(e + 15) << 10) + (r - 0x400u));   // mantissa carry bumps the exponent
}
inline float h2f(uint16_t h) {
    const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 31u, m = h & 0x3FFu;
    if (e == 31u) return __uint_as_float(sign | 0x7F800000u | (m << 13));
    if (e == 0u) { const float v = static_cast<float>(m) * 5.9604644775390625e-8f; return sign ? -v : v; }      // m * 2^-24
    return __uint_as_float(sign | ((e + 112u) << 23) | (m << 13));
}
#else
__device__ __forceinline__ uint16_t f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }
__device__ __forceinline__ float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
#endif
BLOK_DEV float q16(float f) { return h2f(f2h(f)); }

}  // namespace blok
#endif
