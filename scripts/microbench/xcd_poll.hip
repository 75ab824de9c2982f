// Which polling primitive sees a value published by a wave on ANOTHER XCD (own L2 each) within one kernel?
// Producers (even workgroups) wait ~30 us, then publish flag[k]; consumers (odd workgroups, next XCD in the round-robin
// dispatch) have read flag[k] before that and poll it with a bounded budget.  Prints, per (publish, poll) pair, how many
// of the consumers saw the value and the mean number of polls.  Nothing here can hang: every loop is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum Pub { PubStoreRelaxed, PubStoreRelease, PubExchange };
enum Poll { PollLoadRelaxed, PollLoadAcquire, PollCas, PollFetchOr };

template <int PUB, int POLL>
__global__ void k(unsigned long long* flag, unsigned* polls_out, int n_pairs, int budget) {
    const int pair = blockIdx.x >> 1;
    if (pair >= n_pairs || threadIdx.x != 0) return;
    unsigned long long* f = flag + pair * 16;       // one 128-byte line per pair
    if ((blockIdx.x & 1) == 0) {
        for (int i = 0; i < 600; ++i) __builtin_amdgcn_s_sleep(127);       // ~600 * 8k clocks / 2.4 GHz ~ a few tens of us
        if (PUB == PubStoreRelaxed) __hip_atomic_store(f, 42ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (PUB == PubStoreRelease) __hip_atomic_store(f, 42ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (PUB == PubExchange) (void)__hip_atomic_exchange(f, 42ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        unsigned n = 0;
        unsigned long long v = 0;
        for (; n < (unsigned)budget; ++n) {
            if (POLL == PollLoadRelaxed) v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (POLL == PollLoadAcquire) v = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (POLL == PollCas) { unsigned long long e = 7ull; __hip_atomic_compare_exchange_strong(f, &e, 7ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = e; }
            if (POLL == PollFetchOr) v = __hip_atomic_fetch_or(f, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == 42ull) break;
            __builtin_amdgcn_s_sleep(32);
        }
        polls_out[pair] = v == 42ull ? n + 1 : 0xFFFFFFFFu;
    }
}

template <int PUB, int POLL>
void run(const char* name, unsigned long long* flag, unsigned* polls, int n_pairs) {
    hipMemset(flag, 0, n_pairs * 128);
    hipMemset(polls, 0, n_pairs * 4);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<PUB, POLL>), dim3(2 * n_pairs), dim3(64), 0, 0, flag, polls, n_pairs, 20000);
    hipEventRecord(b);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h(n_pairs);
    hipMemcpy(h.data(), polls, n_pairs * 4, hipMemcpyDeviceToHost);
    int seen = 0; double sum = 0;
    for (unsigned v : h) if (v != 0xFFFFFFFFu) { ++seen; sum += v; }
    printf("%-34s seen %4d / %d  mean polls %8.1f  kernel %.3f ms  (%s)\n", name, seen, n_pairs, seen ? sum / seen : 0.0, ms, hipGetErrorString(e));
}

int main() {
    const int n_pairs = 512;
    unsigned long long* flag; unsigned* polls;
    hipMalloc(&flag, n_pairs * 128); hipMalloc(&polls, n_pairs * 4);
    run<PubStoreRelaxed, PollLoadRelaxed>("store relaxed / load relaxed", flag, polls, n_pairs);
    run<PubStoreRelaxed, PollLoadAcquire>("store relaxed / load acquire", flag, polls, n_pairs);
    run<PubStoreRelaxed, PollCas>("store relaxed / cas", flag, polls, n_pairs);
    run<PubStoreRelaxed, PollFetchOr>("store relaxed / fetch_or 0", flag, polls, n_pairs);
    run<PubStoreRelease, PollLoadRelaxed>("store release / load relaxed", flag, polls, n_pairs);
    run<PubStoreRelease, PollLoadAcquire>("store release / load acquire", flag, polls, n_pairs);
    run<PubExchange, PollLoadRelaxed>("exchange / load relaxed", flag, polls, n_pairs);
    run<PubExchange, PollLoadAcquire>("exchange / load acquire", flag, polls, n_pairs);
    run<PubExchange, PollCas>("exchange / cas", flag, polls, n_pairs);
    return 0;
}
