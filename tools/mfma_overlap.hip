// Development micro-benchmark: does v_mfma_f32_16x16x4_f32 overlap with v_fma_f32 on gfx950?
// Three kernels with the same loop count: 32 independent v_fma per iteration (MODE 0), 4
// independent MFMAs per iteration (MODE 1), both interleaved (MODE 2).  If the pipes are
// separate, time(2) ~ max(time(0), time(1)); if they share issue or datapath, ~ the sum.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap.hip -o build/mfma_overlap && build/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    float x[8];
    f32x4 acc[4];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float ma = threadIdx.x * 1e-4f, mb = 1.0f + threadIdx.x * 1e-5f;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc[i], 0, 0, 0);
        }
        if (MODE != 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
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
    printf("%-22s blocks=%5d  %.3f ms  = %.1f cycles per iteration per wave-slot @2.4GHz\n", name, blocks, ms,
           2.4e9 * ms * 1e-3 / iters / waves_per_simd);
    hipFree(out);
    return ms;
}

int main() {
    for (int blocks : {1024, 2048}) {
        const float f = run<0>("32 v_fma", blocks);
        const float m = run<1>("4 mfma_16x16x4_f32", blocks);
        const float b = run<2>("both interleaved", blocks);
        printf("  -> both / max = %.2f, both / sum = %.2f\n", b / (f > m ? f : m), b / (f + m));
    }
    return 0;
}
