// Development lab for the population-evaluation kernel: times several tilings of
// popeval_kernel on random data.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude
//   -Ialproj_amd/csrc tools/popeval_lab.hip -o build/popeval_lab
#include "alp_point_kernels.h"

#include <cstdlib>
#include <vector>

namespace alp {       // the lab links without alp_core
void set_error(const char *, ...) {}
int fail(int c, const char *, ...) { return c; }
}
using namespace alp;

static float frand(float a, float b) { return a + (b - a) * (float)rand() / RAND_MAX; }

template <typename Cfg>
void run(const char *name, int blocks_per_cu, const float *x, const float *y, const float *z, const float *uo,
         const float *vo, int64_t n, const PoseRec<float> *cands, int P, double *partials) {
    const int nblk = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((popeval_kernel<float, ALP_LOSS_HUBER, Cfg>), dim3(nblk), dim3(256), 0, 0, x, y, z, uo, vo,
                           n, cands, P, 10.0f, partials);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    double chk = 0;
    std::vector<double> h((size_t)nblk * P);
    hipMemcpy(h.data(), partials, h.size() * 8, hipMemcpyDeviceToHost);
    for (double v : h) chk += v;
    printf("%-28s blk/CU=%d  %.3f ms  %.1f Gevals/s  (%s) checksum %.6e\n", name, blocks_per_cu, best,
           (double)n * P / best / 1e6, hipGetErrorString(e), chk / n / P);
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int P = argc > 2 ? atoi(argv[2]) : 256;
    const int64_t npad = round_up(n, 1024);
    std::vector<float> hx(npad), hy(npad), hz(npad), hu(npad), hv(npad);
    srand(1);
    for (int64_t i = 0; i < n; ++i) {
        hz[i] = frand(100, 4000);                 // depth along +x (camera looks along x)
        hx[i] = hz[i];
        hy[i] = frand(-0.7f, 0.7f) * hz[i];
        float zz = frand(-0.45f, 0.45f) * hz[i];
        hz[i] = zz;
        hu[i] = frand(0, 5616); hv[i] = frand(0, 3744);
    }
    std::vector<PoseRec<float>> hc(P);
    for (int c = 0; c < P; ++c) {
        float *r = hc[c].v;
        for (int i = 0; i < POSE_WORDS; ++i) r[i] = 0;
        // x1 = -y/x*1.3, y1 = -z/x*2.1, Z = x   (+ small perturbation per candidate)
        r[1] = -1.3f + frand(-0.05f, 0.05f); r[6] = -2.1f + frand(-0.05f, 0.05f); r[8] = 1.0f;
        r[3] = frand(-1, 1); r[7] = frand(-1, 1); r[11] = frand(-1, 1);
        for (int i = 12; i < 18; ++i) r[i] = frand(-0.02f, 0.02f);
        r[18] = 2.0f; r[19] = 2.0f;
        for (int i = 20; i < 26; ++i) r[i] = frand(-0.002f, 0.002f);
        r[26] = 2807.5f; r[27] = 1871.5f;
    }
    float *x, *y, *z, *uo, *vo; PoseRec<float> *cands; double *partials;
    hipMalloc(&x, npad * 4); hipMalloc(&y, npad * 4); hipMalloc(&z, npad * 4);
    hipMalloc(&uo, npad * 4); hipMalloc(&vo, npad * 4);
    hipMalloc(&cands, sizeof(PoseRec<float>) * P);
    hipMalloc(&partials, sizeof(double) * 256 * 8 * P);
    hipMemcpy(x, hx.data(), npad * 4, hipMemcpyHostToDevice);
    hipMemcpy(y, hy.data(), npad * 4, hipMemcpyHostToDevice);
    hipMemcpy(z, hz.data(), npad * 4, hipMemcpyHostToDevice);
    hipMemcpy(uo, hu.data(), npad * 4, hipMemcpyHostToDevice);
    hipMemcpy(vo, hv.data(), npad * 4, hipMemcpyHostToDevice);
    hipMemcpy(cands, hc.data(), sizeof(PoseRec<float>) * P, hipMemcpyHostToDevice);
#define RUN(V, TC, MW, B) run<PopCfgT<float, V, TC, MW>>("V=" #V " TC=" #TC " minw=" #MW, B, x, y, z, uo, vo, n, cands, P, partials)
    RUN(1, 256, 1, 4);
    RUN(2, 256, 1, 4);
    RUN(4, 256, 1, 4);
    RUN(8, 256, 1, 4);
    RUN(4, 128, 1, 4);
    RUN(4, 128, 1, 8);
    RUN(4, 128, 2, 8);
    RUN(8, 128, 1, 8);
    RUN(8, 128, 3, 6);
    RUN(8, 256, 2, 4);
    RUN(8, 256, 3, 4);
    RUN(16, 256, 2, 2);
    RUN(16, 128, 2, 4);
    return 0;
}
