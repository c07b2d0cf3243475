// Development: is  y0 = v_rcp_f32(x); e = fma(-x, y0, 1); y = fma(y0, e, y0)  the correctly rounded
// reciprocal 1.0f / x on this chip?  Exhaustive over every float32 mantissa, for a set of exponents
// (scaling by a power of two is exact while nothing is subnormal, so the mantissa decides).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/rcp_exact.hip -o /tmp/rcp_exact && /tmp/rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float fast1(float x) {
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    return __builtin_fmaf(y0, e, y0);
}
__device__ __forceinline__ float fast2(float x) {
    const float y1 = fast1(x);
    const float e = __builtin_fmaf(-x, y1, 1.0f);
    return __builtin_fmaf(y1, e, y1);
}

__global__ void check(int exponent, unsigned long long *bad) {
    const unsigned m = blockIdx.x * blockDim.x + threadIdx.x;      // 2^23 mantissas
    const float x = __uint_as_float(((unsigned)(127 + exponent) << 23) | m);
    const float ref = 1.0f / x;
    if (__builtin_amdgcn_rcpf(x) != ref) atomicAdd(&bad[0], 1ull);
    if (fast1(x) != ref) atomicAdd(&bad[1], 1ull);
    if (fast2(x) != ref) atomicAdd(&bad[2], 1ull);
}

int main() {
    unsigned long long *bad, h[3];
    hipMalloc(&bad, 24);
    for (int e : {0, 1, 2, 5, 10, 13, 20, 23, 24, 30, 31, 40, 60}) {
        hipMemset(bad, 0, 24);
        hipLaunchKernelGGL(check, dim3((1u << 23) / 256), dim3(256), 0, 0, e, bad);
        hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost);
        printf("x in [2^%d, 2^%d): v_rcp_f32 differs from IEEE for %llu of 8388608 mantissas, rcp+1 Newton %llu, rcp+2 Newton %llu\n",
               e, e + 1, h[0], h[1], h[2]);
    }
    return 0;
}
