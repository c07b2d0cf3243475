// Development: rocPRIM radix_sort_pairs of (uint32 key, uint32 payload) at the size of to_geotiff's cell sort (11.7 M pairs,
// 27-bit keys): the library's tuned gfx950 configuration (8 bits per pass: four passes) against onesweep configurations with
// 9 bits per pass (three passes).   hipcc --offload-arch=gfx950 -O2 -o /tmp/sort_rate tools/sort_rate.hip && /tmp/sort_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <rocprim/rocprim.hpp>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class Config>
static int run(const char *name, unsigned *k, unsigned *ks, unsigned *v, unsigned *vs, size_t n, unsigned bits, const unsigned *k_host_sorted) {
    size_t tmp = 0;
    CK((rocprim::radix_sort_pairs<Config>(nullptr, tmp, k, ks, v, vs, n, 0u, bits, (hipStream_t)0)));
    void *t;
    CK(hipMalloc(&t, tmp));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a, 0));
        CK((rocprim::radix_sort_pairs<Config>(t, tmp, k, ks, v, vs, n, 0u, bits, (hipStream_t)0)));
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    std::vector<unsigned> out(n);
    CK(hipMemcpy(out.data(), ks, n * 4, hipMemcpyDeviceToHost));
    bool ok = true;
    for (size_t i = 0; i < n; ++i) if (out[i] != k_host_sorted[i]) { ok = false; break; }
    printf("%-34s %u bits: %.3f ms (temporary storage %.1f MB) %s\n", name, bits, best, tmp / 1e6, ok ? "sorted" : "WRONG");
    CK(hipFree(t));
    return 0;
}

template <class Config>
static int run_keys64(const char *name, unsigned long long *k, unsigned long long *ks, size_t n, unsigned bits) {
    size_t tmp = 0;
    CK((rocprim::radix_sort_keys<Config>(nullptr, tmp, k, ks, n, 0u, bits, (hipStream_t)0)));
    void *t;
    CK(hipMalloc(&t, tmp));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a, 0));
        CK((rocprim::radix_sort_keys<Config>(t, tmp, k, ks, n, 0u, bits, (hipStream_t)0)));
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    std::vector<unsigned long long> out(n);
    CK(hipMemcpy(out.data(), ks, n * 8, hipMemcpyDeviceToHost));
    bool ok = true;
    for (size_t i = 1; i < n; ++i) if (out[i] < out[i - 1]) { ok = false; break; }
    printf("64-bit keys, %-22s %u bits: %.3f ms %s\n", name, bits, best, ok ? "sorted" : "WRONG");
    CK(hipFree(t));
    return 0;
}

template <unsigned BS, unsigned IPT, unsigned BITS, rocprim::block_radix_rank_algorithm ALG = rocprim::block_radix_rank_algorithm::match>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, BITS, ALG>>;

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : 11674978;
    std::vector<unsigned> hk(n), hv(n);
    unsigned s = 12345u;
    // clustered like the frame: runs of equal keys of random length 1..64 next to each other in the input
    for (size_t i = 0; i < n;) {
        s = s * 1664525u + 1013904223u;
        const unsigned key = (s >> 5) & ((1u << 27) - 1u);
        s = s * 1664525u + 1013904223u;
        size_t len = 1 + (s >> 26);
        for (size_t j = 0; j < len && i < n; ++j, ++i) { hk[i] = key; hv[i] = (unsigned)i; }
    }
    std::vector<unsigned> sorted(hk);
    std::sort(sorted.begin(), sorted.end());
    unsigned *k, *ks, *v, *vs;
    CK(hipMalloc(&k, n * 4)); CK(hipMalloc(&ks, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&vs, n * 4));
    CK(hipMemcpy(k, hk.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, hv.data(), n * 4, hipMemcpyHostToDevice));
    run<rocprim::default_config>("default (tuned gfx950)", k, ks, v, vs, n, 27, sorted.data());
    run<Cfg<1024, 8, 8>>("1024 x 8, 8 bits, match", k, ks, v, vs, n, 27, sorted.data());
    run<Cfg<1024, 8, 9>>("1024 x 8, 9 bits, match", k, ks, v, vs, n, 27, sorted.data());
    run<Cfg<1024, 7, 9>>("1024 x 7, 9 bits, match", k, ks, v, vs, n, 27, sorted.data());
    run<Cfg<1024, 9, 9>>("1024 x 9, 9 bits, match", k, ks, v, vs, n, 27, sorted.data());
    run<Cfg<1024, 10, 9>>("1024 x 10, 9 bits, match", k, ks, v, vs, n, 27, sorted.data());
    run<Cfg<1024, 6, 9>>("1024 x 6, 9 bits, match", k, ks, v, vs, n, 27, sorted.data());
    run<Cfg<512, 8, 9>>("512 x 8, 9 bits, match", k, ks, v, vs, n, 27, sorted.data());
    run<Cfg<1024, 8, 10>>("1024 x 8, 10 bits, match", k, ks, v, vs, n, 27, sorted.data());
    run<Cfg<1024, 8, 10>>("1024 x 8, 10 bits, match", k, ks, v, vs, n, 30, sorted.data());
    run<rocprim::default_config>("default (tuned gfx950)", k, ks, v, vs, n, 30, sorted.data());
    run<Cfg<1024, 8, 9>>("1024 x 8, 9 bits, match", k, ks, v, vs, n, 30, sorted.data());
    run<Cfg<1024, 8, 9>>("1024 x 8, 9 bits, match", k, ks, v, vs, n, 18, sorted.data());
    run<rocprim::default_config>("default (tuned gfx950)", k, ks, v, vs, n, 18, sorted.data());
    // the composite keys of a median whose bands are not bytes: cell : value
    std::vector<unsigned long long> h64(n);
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h64[i] = ((unsigned long long)hk[i] << 32) | s; }
    unsigned long long *k64, *k64s;
    CK(hipMalloc(&k64, n * 8)); CK(hipMalloc(&k64s, n * 8));
    CK(hipMemcpy(k64, h64.data(), n * 8, hipMemcpyHostToDevice));
    run_keys64<rocprim::default_config>("default", k64, k64s, n, 59);
    run_keys64<Cfg<1024, 8, 9>>("1024 x 8, 9 bits", k64, k64s, n, 59);
    run_keys64<Cfg<1024, 6, 9>>("1024 x 6, 9 bits", k64, k64s, n, 59);
    run_keys64<Cfg<1024, 4, 9>>("1024 x 4, 9 bits", k64, k64s, n, 59);
    run_keys64<Cfg<512, 8, 9>>("512 x 8, 9 bits", k64, k64s, n, 59);
    run_keys64<Cfg<1024, 6, 10>>("1024 x 6, 10 bits", k64, k64s, n, 59);
    return 0;
}
