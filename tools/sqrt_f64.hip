// Development: the lean float64 square root of the population kernels (v_rsq_f64 + one Goldschmidt step + one
// correction) against the IEEE expansion of __builtin_sqrt: how many results differ, and by how many ulps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ double lean_sqrt(double a) {
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, a);
    g = __builtin_fma(d, h, g);
    return __builtin_amdgcn_class(a, 0x260) ? a : g;       // -0, +0, +inf
}
__global__ void k(unsigned long long seed, int lo_exp, int hi_exp, long long n, unsigned long long *out) {
    unsigned long long diff = 0, maxulp = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned long long s = seed + (unsigned long long)i * 0x9E3779B97F4A7C15ull;
        s ^= s >> 30; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 27; s *= 0x94D049BB133111EBull; s ^= s >> 31;
        const unsigned long long mant = s & 0xFFFFFFFFFFFFFull;
        const int e = lo_exp + (int)((s >> 52) % (unsigned)(hi_exp - lo_exp + 1));
        const unsigned long long bits = ((unsigned long long)(e + 1023) << 52) | mant;
        double a;
        memcpy(&a, &bits, 8);
        const double x = lean_sqrt(a), y = __builtin_sqrt(a);
        unsigned long long bx, by;
        memcpy(&bx, &x, 8); memcpy(&by, &y, 8);
        const unsigned long long u = bx > by ? bx - by : by - bx;
        diff += u != 0;
        maxulp = u > maxulp ? u : maxulp;
    }
    atomicAdd(&out[0], diff);
    atomicMax(&out[1], maxulp);
}
int main() {
    unsigned long long *out, h[2];
    (void)hipMalloc(&out, 16);
    for (auto range : {std::pair<int, int>{-40, 40}, std::pair<int, int>{-1000, 1000}, std::pair<int, int>{0, 1}}) {
        (void)hipMemset(out, 0, 16);
        const long long n = 1ll << 30;
        k<<<4096, 256>>>(12345, range.first, range.second, n, out);
        (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("exponents %d..%d: %lld inputs, %llu results differ from the IEEE square root, largest difference %llu ulp\n", range.first,
               range.second, n, h[0], h[1]);
    }
    return 0;
}
