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

namespace alp {
// MEASURED AND NOT KEPT (round 2): 20 M points x 2048 candidates 895 G evaluations/s against 870 for K2
// (+2.8 %), 10 M x 256 625 against 848 (one candidate group: too few workgroups).  The per-candidate
// overheads this tiling removes (LDS record reads, DPP reductions, barriers) were not what bounds K2:
// both tilings run at the issue rate of the same ~48 vector instructions per evaluation.
// ------------------------------------------------------------------ K2': population evaluation, lane = candidate
// The transposed tiling of K2 for float32 point sets: a LANE owns CPL candidate poses, whose folded
// records stay in its registers for the whole kernel, and the wave walks the points of its stripe
// together -- the point coordinates are wave-uniform (scalar loads, scalar-register operands of the
// vector instructions).  What K2 pays per candidate and group of points disappears: no staging of
// records in LDS and no re-reading them (6 ds_read_b128 per candidate and group), no cross-lane
// reduction (6 DPP adds + the float64 read-modify-write of an LDS slot per candidate and group), no
// barriers; a lane simply accumulates ITS candidates' losses: float32 over CHUNK points, then float64.
// Same per-evaluation arithmetic (group_loss_sum), fixed summation order -> bitwise reproducible.
// grid = (stripes of the points) x (groups of 256 * CPL candidates); partials[stripe][candidate].
template <int LOSS, int V, int CPL, int CHUNK = 32>
__global__ __launch_bounds__(256) void popeval_lc_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                         const float *__restrict__ z, const float *__restrict__ uo,
                                                         const float *__restrict__ vo, int64_t n, int64_t stripe,
                                                         const PoseRec<float> *__restrict__ cands, int P, float f_scale,
                                                         double *__restrict__ partials) {
    static_assert(CHUNK % V == 0, "CHUNK must be a multiple of V");
    float r[CPL][32];
    int cid[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        cid[k] = (int)blockIdx.y * 256 * CPL + k * 256 + (int)threadIdx.x;
        const float4 *src = reinterpret_cast<const float4 *>(cands[cid[k] < P ? cid[k] : 0].v);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 t = src[q];
            r[k][4 * q + 0] = t.x; r[k][4 * q + 1] = t.y; r[k][4 * q + 2] = t.z; r[k][4 * q + 3] = t.w;
        }
    }
    const float c0 = cands[0].v[26], c1 = cands[0].v[27];      // identical in every record of a call
    const int64_t beg = (int64_t)blockIdx.x * stripe;
    const int64_t end = beg + stripe < n ? beg + stripe : n;
    double acc64[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) acc64[k] = 0.0;
    const NormCoords<float, V> none = {};
    const NormCoords<float, 1> none1 = {};
    int64_t i = beg;
    while (i + V <= end) {
        float acc[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) acc[k] = 0.0f;
        const int64_t stop = (i + CHUNK <= end) ? i + CHUNK : i + (end - i) / V * V;
        for (; i < stop; i += V) {
            float qx[V], qy[V], qz[V], uoc[V], voc[V];
            bool ok[V];
#pragma unroll
            for (int j = 0; j < V; ++j) {          // wave-uniform addresses: scalar loads
                qx[j] = x[i + j]; qy[j] = y[i + j]; qz[j] = z[i + j];
                uoc[j] = uo[i + j] - c0;
                voc[j] = vo[i + j] - c1;
                ok[j] = true;
            }
#pragma unroll
            for (int k = 0; k < CPL; ++k)
                acc[k] += group_loss_sum<float, LOSS, V, false, false>(r[k], qx, qy, qz, none, uoc, voc, ok, f_scale);
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k) acc64[k] += (double)acc[k];
    }
    if (i < end) {                                   // fewer than V points left
        float acc[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) acc[k] = 0.0f;
        for (; i < end; ++i) {
            const float qx[1] = {x[i]}, qy[1] = {y[i]}, qz[1] = {z[i]}, uoc[1] = {uo[i] - c0}, voc[1] = {vo[i] - c1};
            const bool ok[1] = {true};
#pragma unroll
            for (int k = 0; k < CPL; ++k)
                acc[k] += group_loss_sum<float, LOSS, 1, false, false>(r[k], qx, qy, qz, none1, uoc, voc, ok, f_scale);
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k) acc64[k] += (double)acc[k];
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k)
        if (cid[k] < P) partials[(int64_t)blockIdx.x * P + cid[k]] = acc64[k];
}

}  // namespace alp

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

template <int V, int CPL, int CHUNK>
void run_lc(const char *name, int64_t stripe, const float *x, const float *y, const float *z, const float *uo,
            const float *vo, int64_t n, const PoseRec<float> *cands, int P, double *partials) {
    const int nstripes = (int)((n + stripe - 1) / stripe), ngroups = (P + 256 * CPL - 1) / (256 * CPL);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((popeval_lc_kernel<ALP_LOSS_HUBER, V, CPL, CHUNK>), dim3(nstripes, ngroups), dim3(256), 0, 0, x, y, z, uo,
                           vo, n, stripe, cands, P, 10.0f, partials);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    double chk = 0;
    std::vector<double> h((size_t)nstripes * P);
    hipMemcpy(h.data(), partials, h.size() * 8, hipMemcpyDeviceToHost);
    for (double v : h) chk += v;
    printf("%-28s stripe=%lld  %.3f ms  %.1f Gevals/s  (%s) checksum %.6e\n", name, (long long)stripe, best,
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
    hipMalloc(&partials, sizeof(double) * (size_t)(256 * 64 + n / 4096 + 16) * P);   // the largest grid below
    hipMemcpy(x, hx.data(), npad * 4, hipMemcpyHostToDevice);
    hipMemcpy(y, hy.data(), npad * 4, hipMemcpyHostToDevice);
    hipMemcpy(z, hz.data(), npad * 4, hipMemcpyHostToDevice);
    hipMemcpy(uo, hu.data(), npad * 4, hipMemcpyHostToDevice);
    hipMemcpy(vo, hv.data(), npad * 4, hipMemcpyHostToDevice);
    hipMemcpy(cands, hc.data(), sizeof(PoseRec<float>) * P, hipMemcpyHostToDevice);
#define RUNLC(V, CPL, CH, S) run_lc<V, CPL, CH>("lane=cand V=" #V " CPL=" #CPL " chunk=" #CH, S, x, y, z, uo, vo, n, cands, P, partials)
    RUNLC(4, 1, 32, 16384);
    RUNLC(6, 1, 48, 16384);
    RUNLC(8, 1, 32, 16384);
    RUNLC(4, 2, 32, 16384);
    RUNLC(6, 2, 48, 16384);
    RUNLC(2, 4, 32, 16384);
    RUNLC(4, 1, 32, 65536);
    RUNLC(4, 2, 32, 65536);
#define RUN(V, TC, MW, B) run<PopCfgT<float, V, TC, MW>>("V=" #V " TC=" #TC " minw=" #MW, B, x, y, z, uo, vo, n, cands, P, partials)
    RUN(4, 128, 1, 8);
    RUN(6, 128, 4, 16);
    RUN(6, 128, 4, 64);
    RUN(8, 128, 3, 6);
    return 0;
}
