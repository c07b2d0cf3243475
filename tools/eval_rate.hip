// Development micro-benchmark 3: pure arithmetic throughput of group_loss_sum<float,HUBER,V>
// (no LDS staging, no reductions): pose record in registers, inputs perturbed per iteration.
#include "alp_point_kernels.h"
namespace alp { void set_error(const char *, ...) {} int fail(int c, const char *, ...) { return c; } }
using namespace alp;

template <int V, bool SREC>
__global__ __launch_bounds__(256) void k(float *out, const float *rec, int iters, unsigned long long *clk) {
    float r[32];
    for (int i = 0; i < 32; ++i) r[i] = SREC ? rec[i] : rec[i + (threadIdx.x & 1) * 0];
    if (!SREC) for (int i = 0; i < 32; ++i) r[i] += threadIdx.x * 1e-9f;      // force VGPR residency
    float qx[V], qy[V], qz[V], uo[V], vo[V];
    bool ok[V];
    NormCoords<float, V> none;
    for (int j = 0; j < V; ++j) {
        qx[j] = 1000.f + threadIdx.x + j; qy[j] = 0.3f * threadIdx.x; qz[j] = 10.f * j;
        uo[j] = 2000.f - 2807.5f; vo[j] = 1500.f - 1871.5f;      // observations minus the image centre, as pop_group hands them over
        ok[j] = true;
    }
    float acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        acc += group_loss_sum<float, ALP_LOSS_HUBER, V, false, false>(r, qx, qy, qz, none, uo, vo, ok, 10.0f);
#pragma unroll
        for (int j = 0; j < V; ++j) qy[j] += 0.001f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int V, bool SREC>
void run(const char *name, int blocks, const float *rec) {
    float *out; unsigned long long *clk, hclk[2];
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    k<V, SREC><<<blocks, 256>>>(out, rec, 10, clk);
    hipEventRecord(e0);
    k<V, SREC><<<blocks, 256>>>(out, rec, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_evals_per_simd = (double)blocks * 4 * iters * V / 1024.0;
    hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    double ghz = (double)hclk[0] / (double)hclk[1] * 0.1;
    printf("%-26s blocks=%5d  %.3f ms  %.1f Gevals/s  clock %.2f GHz  %.1f real cycles/wave-eval\n", name, blocks, ms,
           (double)blocks * 256 * iters * V / ms / 1e6, ghz, ghz * 1e9 * ms * 1e-3 / wave_evals_per_simd);
    hipFree(out);
}

int main() {
    float h[32] = {0};
    h[1] = -1.3e-3f; h[3] = 0.1f; h[6] = -2.1e-3f; h[7] = 0.05f; h[8] = 1e-3f; h[11] = 0.2f;
    for (int i = 12; i < 18; ++i) h[i] = 0.01f;
    h[18] = h[19] = 2.f; for (int i = 20; i < 26; ++i) h[i] = 0.001f;
    h[26] = 2807.5f; h[27] = 1871.5f; h[28] = -h[26]; h[29] = -h[27];
    float *rec; hipMalloc(&rec, 128); hipMemcpy(rec, h, 128, hipMemcpyHostToDevice);
    for (int blocks : {1024, 2048}) {
        run<1, false>("V=1 rec in VGPR", blocks, rec);
        run<2, false>("V=2 rec in VGPR", blocks, rec);
        run<4, false>("V=4 rec in VGPR", blocks, rec);
        run<8, false>("V=8 rec in VGPR", blocks, rec);
        run<4, true>("V=4 rec in SGPR", blocks, rec);
        run<8, true>("V=8 rec in SGPR", blocks, rec);
    }
    return 0;
}
