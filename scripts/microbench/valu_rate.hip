// Micro-benchmark: sustained issue rate of the VALU instructions the trace kernel uses, at 8 waves per SIMD
// (throughput) and 1 wave per SIMD (latency-ish), relative to v_fma_f32.  Four independent destination
// registers per instruction kind; cycles from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ONE(asmtext) asm volatile(asmtext : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(m) : "vcc");
#define R8(asmtext) ONE(asmtext) ONE(asmtext) ONE(asmtext) ONE(asmtext) ONE(asmtext) ONE(asmtext) ONE(asmtext) ONE(asmtext)
#define BODY(asmtext) \
    for (int it = 0; it < iters; ++it) { R8(asmtext) R8(asmtext) R8(asmtext) R8(asmtext) R8(asmtext) R8(asmtext) R8(asmtext) R8(asmtext) }
#define FOUR(ins, ops) ins " %0, " ops "\n " ins " %1, " ops "\n " ins " %2, " ops "\n " ins " %3, " ops

template <int KIND>
__global__ __launch_bounds__(512) void k(int iters, unsigned long long* out, float* sink, unsigned long long m) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = 1.0001f, f = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (KIND == 0) BODY("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5")
    if (KIND == 1) BODY("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4")
    if (KIND == 2) BODY("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3")
    if (KIND == 3) BODY("v_mad_i32_i24 %0, %0, %4, %5\n v_mad_i32_i24 %1, %1, %4, %5\n v_mad_i32_i24 %2, %2, %4, %5\n v_mad_i32_i24 %3, %3, %4, %5")
    if (KIND == 4) BODY("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %5, vcc\n v_cndmask_b32_e32 %3, %3, %5, vcc")
    if (KIND == 5) BODY("v_cndmask_b32_e64 %0, %0, %4, %6\n v_cndmask_b32_e64 %1, %1, %4, %6\n v_cndmask_b32_e64 %2, %2, %5, %6\n v_cndmask_b32_e64 %3, %3, %5, %6")
    if (KIND == 6) BODY("v_bfe_u32 %0, %0, %4, 2\n v_bfe_u32 %1, %1, %4, 2\n v_bfe_u32 %2, %2, %4, 2\n v_bfe_u32 %3, %3, %4, 2")
    if (KIND == 7) BODY("v_cmp_le_f32_e32 vcc, %0, %4\n v_cmp_le_f32_e32 vcc, %1, %4\n v_cmp_le_f32_e32 vcc, %2, %4\n v_cmp_le_f32_e32 vcc, %3, %4")
    if (KIND == 8) BODY("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4")
    if (KIND == 9) BODY("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3")
    if (KIND == 10) BODY("v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5")
    if (KIND == 11) BODY("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4")
    if (KIND == 12) BODY("v_lshl_add_u32 %0, %0, %4, %5\n v_lshl_add_u32 %1, %1, %4, %5\n v_lshl_add_u32 %2, %2, %4, %5\n v_lshl_add_u32 %3, %3, %4, %5")
    if (KIND == 13) BODY("v_lshlrev_b32 %0, %4, %0\n v_lshlrev_b32 %1, %4, %1\n v_lshlrev_b32 %2, %4, %2\n v_lshlrev_b32 %3, %4, %3")
    if (KIND == 14) BODY("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4")
    if (KIND == 15) BODY("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4")
    if (KIND == 16) BODY("v_cmp_le_f32_e64 s[20:21], %0, %4\n v_cmp_le_f32_e64 s[20:21], %1, %4\n v_cmp_le_f32_e64 s[22:23], %2, %4\n v_cmp_le_f32_e64 s[22:23], %3, %4")
    if (KIND == 17) BODY("v_bcnt_u32_b32 %0, %0, 0\n v_bcnt_u32_b32 %1, %1, 0\n v_bcnt_u32_b32 %2, %2, 0\n v_bcnt_u32_b32 %3, %3, 0")
    if (KIND == 18) BODY("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %5\n v_mov_b32 %3, %5")
    if (KIND == 19) BODY("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %4, %1, %5\n v_bfi_b32 %2, %4, %2, %5\n v_bfi_b32 %3, %4, %3, %5")
    if (KIND == 20) BODY("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5")
    if (KIND == 21) BODY("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5")
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a + b + c + d == 12345.678f) sink[0] = a;
}

static double base8 = 0, base1 = 0;
template <int KIND> void run(const char* name) {
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, 4096 * 8); hipMalloc(&d_sink, 4);
    const int iters = 200;
    double res[2];
    int idx = 0;
    for (int waves_per_simd : {1, 8}) {
        const int threads = waves_per_simd == 1 ? 256 : 512;
        const int blocks = 256 * (waves_per_simd == 1 ? 1 : 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, iters, d_out, d_sink, 0x5555555555555555ull);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        if (KIND == 0) printf("  [calibration, %d waves/SIMD: kernel %.4f ms wall]\n", waves_per_simd, ms);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), d_out, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        if (KIND == 0) printf("  [calibration: %.0f ticks per wave -> %.1f ticks/us]\n", avg, avg / (ms * 1000.0));
        res[idx++] = avg / (double(iters) * 64 * 4);
    }
    if (KIND == 0) { base1 = res[0]; base8 = res[1]; }
    printf("%-24s 1 wave/SIMD: %6.2f ticks/instr (%.2fx fma)   8 waves/SIMD: %6.2f ticks/instr/wave (%.2fx fma)\n", name, res[0], res[0] / base1, res[1], res[1] / base8);
    hipFree(d_out); hipFree(d_sink);
}

int main() {
    run<0>("v_fma_f32"); run<1>("v_mul_f32"); run<15>("v_sub_f32"); run<11>("v_max_f32"); run<10>("v_min3_f32"); run<20>("v_med3_f32");
    run<2>("v_cvt_f32_i32"); run<3>("v_mad_i32_i24"); run<8>("v_add_u32"); run<21>("v_add3_u32"); run<14>("v_and_b32");
    run<13>("v_lshlrev_b32 (var)"); run<12>("v_lshl_add_u32"); run<6>("v_bfe_u32"); run<19>("v_bfi_b32"); run<9>("v_ffbl_b32"); run<17>("v_bcnt_u32_b32");
    run<18>("v_mov_b32"); run<4>("v_cndmask_b32 e32 vcc"); run<5>("v_cndmask_b32 e64 sgpr"); run<7>("v_cmp_le_f32 e32 vcc"); run<16>("v_cmp_le_f32 e64 sgpr");
    return 0;
}
