// Development micro-benchmark 2: v_fma_f32 with 1, 2 or 3 VGPR source operands (uniform
// coefficients in SGPRs vs VGPRs), 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *coef, int iters, float a, float b) {
    float x[8], p[8], q[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = threadIdx.x * 1e-3f + i;
        p[i] = coef[threadIdx.x * 16 + i];         // per-lane -> VGPR
        q[i] = coef[threadIdx.x * 16 + 8 + i];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);          // v, s, s(literal)
                if (MODE == 1) x[i] = __builtin_fmaf(x[i], p[i], b);       // v, v, s
                if (MODE == 2) x[i] = __builtin_fmaf(x[i], p[i], q[i]);    // v, v, v
                if (MODE == 3) x[i] = __builtin_fmaf(x[i], p[(i + r) & 7], q[(i + 2 * r + 1) & 7]);
                if (MODE == 4) x[i] = __builtin_fmaf(p[i], q[i], x[i]);    // fmac form
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int blocks, const float *coef) {
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<MODE><<<blocks, 256>>>(out, coef, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, coef, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)blocks * 4 * iters * 32 / 1024.0;
    printf("%-22s blocks=%5d  %.3f ms  -> %.2f cycles/instr @2.4GHz\n", name, blocks, ms, 2.4e9 * ms * 1e-3 / per_simd);
    hipFree(out);
}

int main() {
    float *coef;
    hipMalloc(&coef, 256 * 16 * 4);
    hipMemset(coef, 0, 256 * 16 * 4);
    for (int blocks : {1024, 2048}) {
        run<0>("fma v,s,s", blocks, coef);
        run<1>("fma v,v,s", blocks, coef);
        run<2>("fma v,v,v", blocks, coef);
        run<3>("fma v,v,v rotating", blocks, coef);
        run<4>("fmac v,v,acc", blocks, coef);
    }
    return 0;
}
