// Development micro-benchmark: 64-bit atomic max on scattered words by memory scope, and the ISA
// each scope compiles to.  Device-scope atomics execute at the memory side on gfx950 (the per-XCD
// L2s are not coherent with each other); do narrower scopes stay in the XCD's L2 and run faster?
// hipcc --offload-arch=gfx950 -O3 tools/atomic_scope.hip -o build/atomic_scope && build/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int SCOPE>
__global__ __launch_bounds__(256) void k(unsigned long long *buf, uint32_t nwords, int iters) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t s = wave * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const uint32_t r = (s ^ (lane * 2246822519u)) * 3266489917u;
        const unsigned long long key = ((unsigned long long)(s | 1u) << 32) | lane;
        __hip_atomic_fetch_max(&buf[(r >> 3) % nwords], key, __ATOMIC_RELAXED, SCOPE);
    }
}

template <int SCOPE>
void run(const char *name, unsigned long long *buf, uint32_t nwords) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 4096, iters = 64;
    k<SCOPE><<<blocks, 256>>>(buf, nwords, 4);
    (void)hipEventRecord(e0);
    k<SCOPE><<<blocks, 256>>>(buf, nwords, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters;
    printf("%-12s scope, 64 random words per instruction: %.3f ms  %.1f G atomics/s\n", name, ms, n / ms / 1e6);
}

int main() {
    for (uint32_t nwords : {5616u * 3744u, 1u << 20}) {
        unsigned long long *buf;
        (void)hipMalloc(&buf, (size_t)nwords * 8); (void)hipMemset(buf, 0, (size_t)nwords * 8);
        printf("buffer of %u words (%.0f MB)\n", nwords, nwords * 8 / 1e6);
        run<__HIP_MEMORY_SCOPE_SYSTEM>("system", buf, nwords);
        run<__HIP_MEMORY_SCOPE_AGENT>("agent", buf, nwords);
        run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup", buf, nwords);
        run<__HIP_MEMORY_SCOPE_WAVEFRONT>("wavefront", buf, nwords);
        (void)hipFree(buf);
    }
    return 0;
}
