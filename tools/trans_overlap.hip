// Development micro-benchmark: do the quarter-rate transcendentals (v_rcp_f32, v_sqrt_f32) overlap with
// v_fma_f32 issue on gfx950, or do they occupy the same issue slots?  MODE 0: 32 independent v_fma per
// iteration; MODE 1: 4 independent v_rcp + 4 v_sqrt... per iteration; MODE 2: both interleaved.  Also
// packed FP32 (v_pk_fma_f32) and v_fmac vs v_fma with three distinct VGPR sources.
// hipcc --offload-arch=gfx950 -O3 tools/trans_overlap.hip -o build/trans_overlap && build/trans_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    float x[8], t[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; t[i] = 1.5f + threadIdx.x * 1e-3f + i; p[i] = f32x2{x[i], t[i]}; }
    const f32x2 pa = {a, a}, pb = {b, b};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = (i & 1) ? __builtin_amdgcn_rcpf(t[i]) + 1.0f : __builtin_amdgcn_sqrtf(t[i]) + 1.0f;   // 8 trans + 8 adds
        }
        if (MODE == 3) {      // 16 packed FMAs = 32 FMAs
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], pa, pb);
        }
        if (MODE == 4) {      // 32 FMAs with three VGPR sources each
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], t[(i + r) & 7], t[(i + 3) & 7]);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + t[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
float run(const char *name, int blocks) {
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
    const double waves_per_simd = blocks * 4 / 1024.0;
    printf("%-34s blocks=%5d  %.3f ms  = %.1f cycles per iteration per wave-slot @2.4GHz\n", name, blocks, ms,
           2.4e9 * ms * 1e-3 / iters / waves_per_simd);
    hipFree(out);
    return ms;
}

int main() {
    for (int blocks : {1024, 2048}) {
        const float f = run<0>("32 v_fma (1 VGPR + 2 SGPR src)", blocks);
        const float m = run<1>("8 trans + 8 add", blocks);
        const float b = run<2>("both interleaved", blocks);
        printf("  -> both / max = %.2f, both / sum = %.2f\n", b / (f > m ? f : m), b / (f + m));
        run<3>("16 v_pk_fma_f32 (= 32 fma)", blocks);
        run<4>("32 v_fma (3 VGPR src)", blocks);
    }
    return 0;
}
