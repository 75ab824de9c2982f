// Same-address atomic throughput on MI355X: N waves (one per workgroup) each issue K fetch-adds (lane 0, returning), to
//   one      one counter for the whole chip
//   xcd      8 counters, workgroup b uses counter b % 8     (round-robin dispatch: one XCD per counter)
//   mixed    8 counters, workgroup b uses counter (b / 8) % 8 (every XCD hits every counter)
//   cu       one counter per 32 workgroups ( ~ per CU group)
//   private  one counter per workgroup (no contention: the latency-bound baseline)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, bool RET>
__global__ void k(unsigned* ctr, int iters) {
    if (threadIdx.x != 0) return;
    const unsigned b = blockIdx.x;
    unsigned idx = MODE == 0 ? 0u : MODE == 1 ? b % 8u : MODE == 2 ? (b / 8u) % 8u : MODE == 3 ? b / 32u : b;
    unsigned* p = ctr + idx * 32u;                     // 128-byte lines
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        if (RET) acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else (void)__hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 0xFFFFFFFFu) ctr[1] = acc;
}
template <int MODE, bool RET>
void run(const char* name, unsigned* ctr, int waves, int iters) {
    hipMemset(ctr, 0, 8192 * 128);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, RET>), dim3(waves), dim3(64), 0, 0, ctr, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, RET>), dim3(waves), dim3(64), 0, 0, ctr, iters);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double n = double(waves) * iters;
    printf("%-8s %-9s waves %5d x %3d: %8.3f ms  %8.1f M atomics/s total  %7.1f ns per atomic per counter-stream\n", name, RET ? "returning" : "no-return", waves, iters, ms, n / ms / 1e3,
           ms * 1e6 / (n / (MODE == 0 ? 1 : MODE == 1 || MODE == 2 ? 8 : MODE == 3 ? waves / 32 : waves)));
}
int main() {
    unsigned* ctr; hipMalloc(&ctr, 8192 * 128);
    for (int waves : {1024, 8192}) {
        run<0, true>("one", ctr, waves, 16); run<0, false>("one", ctr, waves, 16);
        run<1, true>("xcd", ctr, waves, 16); run<1, false>("xcd", ctr, waves, 16);
        run<2, true>("mixed", ctr, waves, 16); run<2, false>("mixed", ctr, waves, 16);
        run<3, true>("cu", ctr, waves, 16);
        run<4, true>("private", ctr, waves, 16);
    }
    return 0;
}
