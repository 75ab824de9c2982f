// Micro-benchmark, second list (round 4): sustained issue rate per SIMD of further VALU instructions — the candidates for replacing the walk
// loop's half-rate selects / compares / bit-field ops, and the packed-f32 forms the compiler's SLP pass puts into the loop.  8 waves per
// SIMD (throughput) and 1 wave per SIMD; four independent destinations per kind; cycles from s_memtime, relative to v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

#define ONE(asmtext) asm volatile(asmtext : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(m) : "vcc");
#define ONE2(asmtext) asm volatile(asmtext : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pe), "v"(pf));
#define R8(M, asmtext) M(asmtext) M(asmtext) M(asmtext) M(asmtext) M(asmtext) M(asmtext) M(asmtext) M(asmtext)
#define BODY(asmtext) \
    for (int it = 0; it < iters; ++it) { R8(ONE, asmtext) R8(ONE, asmtext) R8(ONE, asmtext) R8(ONE, asmtext) R8(ONE, asmtext) R8(ONE, asmtext) R8(ONE, asmtext) R8(ONE, asmtext) }
#define BODY2(asmtext) \
    for (int it = 0; it < iters; ++it) { R8(ONE2, asmtext) R8(ONE2, asmtext) R8(ONE2, asmtext) R8(ONE2, asmtext) R8(ONE2, asmtext) R8(ONE2, asmtext) R8(ONE2, asmtext) R8(ONE2, asmtext) }
#define U4(ins, rest) ins " %0, %0, " rest "\n " ins " %1, %1, " rest "\n " ins " %2, %2, " rest "\n " ins " %3, %3, " rest
#define U1(ins) ins " %0, %0\n " ins " %1, %1\n " ins " %2, %2\n " ins " %3, %3"

template <int KIND>
__global__ __launch_bounds__(512) void k(int iters, unsigned long long* out, float* sink, unsigned long long m) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = 1.0001f, f = 0.5f;
    float2v pa = {a, b}, pb = {b, c}, pc = {c, d}, pd = {d, a}, pe = {e, e}, pf = {f, f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (KIND == 0) BODY("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5")
    if (KIND == 1) BODY(U4("v_or_b32", "%4"))
    if (KIND == 2) BODY(U4("v_xor_b32", "%4"))
    if (KIND == 3) BODY(U4("v_sub_u32", "%4"))
    if (KIND == 4) BODY("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3")
    if (KIND == 5) BODY("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3")
    if (KIND == 6) BODY("v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3")
    if (KIND == 7) BODY(U4("v_and_or_b32", "%4, %5"))
    if (KIND == 8) BODY(U4("v_or3_b32", "%4, %5"))
    if (KIND == 9) BODY(U4("v_min_f32", "%4"))
    if (KIND == 10) BODY(U1("v_cvt_u32_f32"))
    if (KIND == 11) BODY(U1("v_cvt_f32_u32"))
    if (KIND == 12) BODY(U4("v_mul_u32_u24", "%4"))
    if (KIND == 13) BODY(U4("v_mul_lo_u32", "%4"))
    if (KIND == 14) BODY(U4("v_mad_u32_u24", "%4, %5"))
    if (KIND == 15) BODY(U4("v_lshl_or_b32", "%4, %5"))
    if (KIND == 16) BODY(U4("v_add_lshl_u32", "%4, %5"))
    if (KIND == 17) BODY(U1("v_floor_f32"))
    if (KIND == 18) BODY(U1("v_fract_f32"))
    if (KIND == 19) BODY(U4("v_perm_b32", "%4, %5"))
    if (KIND == 20) BODY(U4("v_alignbit_b32", "%4, %5"))
    if (KIND == 21) BODY(U4("v_fmac_f32", "%4"))
    if (KIND == 22) BODY(U4("v_fmac_f32", "%4"))
    if (KIND == 23) BODY2(U4("v_pk_fma_f32", "%4, %5"))
    if (KIND == 24) BODY2(U4("v_pk_mul_f32", "%4"))
    if (KIND == 25) BODY2(U4("v_pk_add_f32", "%4"))
    if (KIND == 26) BODY2("v_pk_mov_b32 %0, %0, %4\n v_pk_mov_b32 %1, %1, %4\n v_pk_mov_b32 %2, %2, %4\n v_pk_mov_b32 %3, %3, %4")
    if (KIND == 27) BODY("v_cmp_le_f32_e64 s[20:21], %0, %4\n v_cndmask_b32_e64 %0, %0, %5, s[20:21]\n v_cmp_le_f32_e64 s[22:23], %1, %4\n v_cndmask_b32_e64 %1, %1, %5, s[22:23]")
    if (KIND == 28) BODY("v_cmp_le_f32_e32 vcc, %0, %4\n v_cndmask_b32_e32 %0, %0, %5, vcc\n v_cmp_le_f32_e32 vcc, %1, %4\n v_cndmask_b32_e32 %1, %1, %5, vcc")
    if (KIND == 29) BODY("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_and_b32 %2, %0, %5\n v_and_b32 %3, %1, %5")
    if (KIND == 30) BODY("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5")
    if (KIND == 31) BODY("v_bitop3_b32 %0, %0, %4, %5 bitop3:0xca\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0xca\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0xca\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0xca")
    if (KIND == 32) BODY("v_cmpx_le_f32_e32 %0, %4\n v_cmpx_le_f32_e32 %1, %4\n v_cmpx_le_f32_e32 %2, %4\n v_cmpx_le_f32_e32 %3, %4\n s_mov_b64 exec, -1")
    if (KIND == 33) BODY("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4")
    if (KIND == 34) BODY("v_add_f32_e64 %0, %0, %4 clamp\n v_add_f32_e64 %1, %1, %4 clamp\n v_add_f32_e64 %2, %2, %4 clamp\n v_add_f32_e64 %3, %3, %4 clamp")
    if (KIND == 35) BODY("v_add_f32_e64 %0, %0, -%4\n v_add_f32_e64 %1, %1, -%4\n v_add_f32_e64 %2, |%2|, %4\n v_add_f32_e64 %3, |%3|, %4")
    if (KIND == 36) BODY("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
    if (KIND == 37) BODY("v_sub_u32 %0, %0, %4\n v_ashrrev_i32 %1, 31, %0\n v_and_b32 %2, %1, %5\n v_add_u32 %3, %3, %2")
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a + b + c + d + pa.x + pb.y + pc.x + pd.y == 12345.678f) sink[0] = a;
}

static double base8 = 0, base1 = 0;
template <int KIND> void run(const char* name, int per_body = 4) {
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, 4096 * 8); hipMalloc(&d_sink, 4);
    const int iters = 200;
    double res[2];
    int idx = 0;
    for (int waves_per_simd : {1, 8}) {
        const int threads = waves_per_simd == 1 ? 256 : 512;
        const int blocks = 256 * (waves_per_simd == 1 ? 1 : 4);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, iters, d_out, d_sink, 0x5555555555555555ull);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), d_out, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        res[idx++] = avg / (double(iters) * 64 * per_body);
    }
    if (KIND == 0) { base1 = res[0]; base8 = res[1]; }
    printf("%-44s 1 wave/SIMD: %6.2f ticks/instr (%.2fx fma)   8 waves/SIMD: %6.2f ticks/instr/wave (%.2fx fma)\n", name, res[0], res[0] / base1, res[1], res[1] / base8);
    hipFree(d_out); hipFree(d_sink);
}

int main() {
    run<0>("v_fma_f32"); run<33>("v_add_f32"); run<34>("v_add_f32 e64 clamp"); run<35>("v_add_f32 e64 neg/abs"); run<22>("v_fmac_f32"); run<9>("v_min_f32"); run<30>("v_max3_f32");
    run<1>("v_or_b32"); run<2>("v_xor_b32"); run<3>("v_sub_u32"); run<4>("v_lshlrev_b32 (const)"); run<5>("v_lshrrev_b32 (const)"); run<6>("v_ashrrev_i32 (const)");
    run<7>("v_and_or_b32"); run<8>("v_or3_b32"); run<15>("v_lshl_or_b32"); run<16>("v_add_lshl_u32"); run<31>("v_bitop3_b32");
    run<10>("v_cvt_u32_f32"); run<11>("v_cvt_f32_u32"); run<17>("v_floor_f32"); run<18>("v_fract_f32");
    run<12>("v_mul_u32_u24"); run<13>("v_mul_lo_u32"); run<14>("v_mad_u32_u24"); run<19>("v_perm_b32"); run<20>("v_alignbit_b32");
    run<23>("v_pk_fma_f32 (two results each)"); run<24>("v_pk_mul_f32 (two results each)"); run<25>("v_pk_add_f32 (two results each)"); run<26>("v_pk_mov_b32");
    run<27>("v_cmp e64 + v_cndmask e64 (pairs, dependent)"); run<28>("v_cmp e32 + v_cndmask e32 vcc (pairs)"); run<32>("v_cmpx_le_f32 e32 (+ s_mov exec per 4)");
    run<29>("sub, sub, and, and (dependent pairs)"); run<37>("sub, ashr, and, add (dependent chain)"); run<36>("v_mov_b32 dpp quad_perm");
    return 0;
}
