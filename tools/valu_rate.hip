// Development micro-benchmark: sustained rate of v_fma_f32 vs v_pk_fma_f32 vs v_rcp_f32 on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o build/valu_rate && build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    float x[8];
    float2v y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = float2v{x[i], x[i] + 1}; }
    float2v av{a, a}, bv{b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
                if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], av, bv);
                if (MODE == 2) x[i] = __builtin_amdgcn_rcpf(x[i]);
                if (MODE == 3) x[i] = __builtin_amdgcn_sqrtf(x[i]);
                if (MODE == 4) x[i] = x[i] * a;
                if (MODE == 5) y[i] = y[i] * av;
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int blocks, double ops_per_instr) {
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<MODE><<<blocks, 256>>>(out, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * 4 * iters * 32;            // wave-instructions
    double per_simd = instr / 1024.0;
    printf("%-14s blocks=%5d  %.3f ms  %.2f G wave-instr/s/SIMD -> %.2f cycles/instr @2.4GHz  %.1f Tops/s\n", name, blocks, ms,
           per_simd / ms / 1e6, 2.4e9 * ms * 1e-3 / per_simd, instr * 64 * ops_per_instr / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int blocks : {1024, 2048, 4096}) {
        run<0>("v_fma_f32", blocks, 2);
        run<1>("v_pk_fma_f32", blocks, 4);
        run<2>("v_rcp_f32", blocks, 1);
        run<3>("v_sqrt_f32", blocks, 1);
        run<4>("v_mul_f32", blocks, 1);
        run<5>("v_pk_mul_f32", blocks, 2);
    }
    return 0;
}
