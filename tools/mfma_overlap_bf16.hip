// Development micro-benchmark: does the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16) run beside
// v_fma_f32 on gfx950?  MODE 0: 32 independent v_fma per iteration; MODE 1: NM independent MFMAs per iteration;
// MODE 2: both in one loop.  Separate pipes: time(2) ~ max; shared issue or datapath: ~ sum.
// (tools/mfma_overlap.hip showed the f32-input MFMA does NOT overlap with v_fma_f32.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int KIND, int NM>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    float x[8];
    f32x4 acc4[4];
    f32x16 acc16[2];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 4; ++i) acc4[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0;
    bf16x8 ma, mb;
    for (int i = 0; i < 8; ++i) { ma[i] = (__bf16)(threadIdx.x * 1e-4f + i); mb[i] = (__bf16)(1.0f + i * 1e-2f); }
    for (int it = 0; it < iters; ++it) {
        if (MODE != 0) {
            if (KIND == 0) {
#pragma unroll
                for (int i = 0; i < NM; ++i) acc4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, mb, acc4[i & 3], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < NM; ++i) acc16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma, mb, acc16[i & 1], 0, 0, 0);
            }
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
    for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc16[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int KIND, int NM>
float run(const char *name, int blocks) {
    float *out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    k<MODE, KIND, NM><<<blocks, 256>>>(out, 100, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    k<MODE, KIND, NM><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks * 4 / 1024.0;
    printf("%-34s blocks=%5d  %.3f ms  = %.1f cycles per iteration per wave-slot @2.4GHz\n", name, blocks, ms,
           2.4e9 * ms * 1e-3 / iters / waves_per_simd);
    (void)hipFree(out);
    return ms;
}

template <int KIND, int NM>
void trio(const char *mname, int blocks) {
    const float f = run<0, KIND, NM>("32 v_fma_f32", blocks);
    const float m = run<1, KIND, NM>(mname, blocks);
    const float b = run<2, KIND, NM>("both in one loop", blocks);
    printf("  -> both / max = %.2f, both / sum = %.2f\n", b / (f > m ? f : m), b / (f + m));
}

int main() {
    for (int blocks : {1024, 2048}) {
        trio<0, 1>("1 mfma_16x16x32_bf16", blocks);
        trio<0, 2>("2 mfma_16x16x32_bf16", blocks);
        trio<0, 4>("4 mfma_16x16x32_bf16", blocks);
        trio<1, 1>("1 mfma_32x32x16_bf16", blocks);
        trio<1, 2>("2 mfma_32x32x16_bf16", blocks);
    }
    return 0;
}
