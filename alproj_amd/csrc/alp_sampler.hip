// libalproj_hip.so -- the candidate sampler of the CMA-ES generation loop on the device.
//
// The reference draws its candidates one by one from the third-party package `cmaes`
// (`optimizer.ask()`, src/alproj/optimize.py:420-421; pinned cmaes==0.12.0, requirements.txt:14, absent
// from the reference checkout): x = m + sigma * B * diag(D) * z with z ~ N(0, I); a draw outside the box is
// re-drawn up to n_max_resampling times, then one more draw is clipped to the box (documented at
// optimize.py:381-384).  With the reference's default sigma = 1.0 on [0, 1]-normalised parameters nearly
// every draw is infeasible, so the first generations cost n_max_resampling + 1 draws per candidate:
// 4.3 M normal deviates at population 2048 / D = 21 -- 16 ms of numpy per generation next to a 219 ms
// evaluation kernel, and more than the 3.5 ms kernel itself at population 256 on 10 M vertices.
//
// Here: one wave per candidate, one lane per TRY.  Lane t draws try t (Philox4x32-10 keyed by seed,
// counter = (block of the draw, try, candidate, generation): any rank, any launch shape gives the same
// numbers), forms B diag(D) z in float64 and tests the box; the lowest feasible try of the wave wins --
// exactly the first feasible draw of the sequential procedure -- and if none of the n_max_resampling
// tries is feasible, try number n_max_resampling is clipped.  PARITY UNPINNED like every CMA sampler here
// (the reference seeds nothing; its trajectory is not reproducible): what is kept is the procedure.
#include "alp_internal.h"

#include <cmath>

namespace alp {

constexpr int SAMPLER_MAX_D = 32;

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// D standard normal deviates of (try, candidate, generation): Box-Muller on 53-bit uniforms
template <int DMAX>
__device__ __forceinline__ void draw_normals(double (&z)[DMAX], int D, unsigned tr, unsigned cand, unsigned gen, unsigned k0, unsigned k1) {
#pragma unroll
    for (int j = 0; j < DMAX; j += 2) {
        if (j < D) {                 // predicated, so that the loop unrolls and z[] lives in registers
            unsigned w[4];
            philox4x32_10((unsigned)(j >> 1), tr, cand, gen, k0, k1, w);
            const double u1 = ((double)(((unsigned long long)w[0] << 21) | (w[1] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
            const double u2 = ((double)(((unsigned long long)w[2] << 21) | (w[3] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
            const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586476925286766559 * u2;
            z[j] = rad * cos(ang);
            if (j + 1 < DMAX) z[j + 1] = (j + 1 < D) ? rad * sin(ang) : 0.0;
        }
    }
}

struct SamplerArgs {
    double mean[SAMPLER_MAX_D], lower[SAMPLER_MAX_D], upper[SAMPLER_MAX_D];
    double sigma;
    int D, n_max, bounded;
    unsigned k0, k1, gen;
};

// x_i = mean_i + sigma * (BD z)_i for one draw; returns whether it lies inside the box
// (BD is zero-padded to DMAX columns per row on the device, so the inner loop needs no bound)
template <int DMAX, bool KEEP>
__device__ __forceinline__ bool make_x(const SamplerArgs &a, const double *__restrict__ BD, const double (&z)[DMAX], double (&x)[DMAX]) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < DMAX; ++i) {
        if (i < a.D) {
            double y = 0.0;
#pragma unroll
            for (int j = 0; j < DMAX; ++j) y = fma(BD[i * DMAX + j], z[j], y);
            const double xi = a.mean[i] + a.sigma * y;
            if (KEEP) x[i] = xi;
            if (a.bounded) ok = ok && xi >= a.lower[i] && xi <= a.upper[i];
        }
    }
    return ok;
}

template <int DMAX>
__global__ __launch_bounds__(256) void cma_sample_kernel(SamplerArgs a, const double *__restrict__ BD, long long P,
                                                         double *__restrict__ x_out, int *__restrict__ tries_out) {
    const long long cand = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (cand >= P) return;                         // whole waves leave together
    const int lane = (int)(threadIdx.x & 63);
    double z[DMAX], x[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; ++j) z[j] = 0.0;
    int accepted = -1;
    for (int base = 0; base < a.n_max && accepted < 0; base += 64) {
        const int t = base + lane;
        bool ok = false;
        if (t < a.n_max) {
            draw_normals<DMAX>(z, a.D, (unsigned)t, (unsigned)cand, a.gen, a.k0, a.k1);
            ok = make_x<DMAX, false>(a, BD, z, x);
        }
        const unsigned long long m = __ballot(ok);
        if (m) accepted = base + __ffsll((long long)m) - 1;      // the first feasible draw of the sequence
    }
    const int t_final = accepted >= 0 ? accepted : a.n_max;    // none feasible: one more draw, clipped
    if (lane == (t_final & 63)) {
        draw_normals<DMAX>(z, a.D, (unsigned)t_final, (unsigned)cand, a.gen, a.k0, a.k1);
        make_x<DMAX, true>(a, BD, z, x);
#pragma unroll
        for (int i = 0; i < DMAX; ++i)
            if (i < a.D) {
                double xi = x[i];
                if (accepted < 0 && a.bounded) xi = fmin(fmax(xi, a.lower[i]), a.upper[i]);   // np.clip
                x_out[cand * a.D + i] = xi;
            }
        if (tries_out) tries_out[cand] = t_final;
    }
}

}  // namespace alp

using namespace alp;

extern "C" int alp_cma_sample(const double *mean, double sigma, const double *BD, const double *lower, const double *upper,
                              int D, int64_t P, int n_max_resampling, uint64_t seed, uint64_t generation, double *x_out,
                              int32_t *tries_out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(mean && BD && x_out, "NULL argument");
    ALP_REQUIRE(D >= 1 && D <= SAMPLER_MAX_D, "D must be in [1, 32]");
    ALP_REQUIRE(P >= 1 && P <= (1 << 24), "P out of range");
    ALP_REQUIRE(n_max_resampling >= 0 && n_max_resampling <= (1 << 20), "n_max_resampling out of range");
    ALP_REQUIRE((lower == nullptr) == (upper == nullptr), "lower and upper must both be given or both be NULL");
    ALP_REQUIRE(sigma > 0, "sigma must be positive");
    SamplerArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < D; ++i) {
        a.mean[i] = mean[i];
        a.lower[i] = lower ? lower[i] : -INFINITY;
        a.upper[i] = upper ? upper[i] : INFINITY;
    }
    a.sigma = sigma;
    a.D = D;
    a.n_max = lower ? n_max_resampling : 1;       // unbounded: the first draw is always accepted
    if (a.n_max < 1) a.n_max = 0;
    a.bounded = lower != nullptr;
    a.k0 = (unsigned)seed;
    a.k1 = (unsigned)(seed >> 32);
    a.gen = (unsigned)generation;
    const int dmax = D <= 12 ? 12 : (D <= 24 ? 24 : 32);
    const size_t bd_bytes = (size_t)dmax * dmax * sizeof(double), x_bytes = (size_t)P * D * sizeof(double),
                 t_bytes = (size_t)P * sizeof(int);
    char *dev = nullptr;
    if (int rc = scratch_reserve(round_up((int64_t)bd_bytes, 256) + round_up((int64_t)x_bytes, 256) + t_bytes, (void **)&dev)) return rc;
    double *bd_dev = (double *)dev, *x_dev = (double *)(dev + round_up((int64_t)bd_bytes, 256));
    int *t_dev = (int *)((char *)x_dev + round_up((int64_t)x_bytes, 256));
    hipStream_t st = ctx().stream;
    double bd_pad[SAMPLER_MAX_D * SAMPLER_MAX_D];      // rows of dmax columns, zero-padded
    memset(bd_pad, 0, sizeof(bd_pad));
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) bd_pad[i * dmax + j] = BD[i * D + j];
    ALP_HIP(hipMemcpyAsync(bd_dev, bd_pad, bd_bytes, hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)((P * 64 + 255) / 256);
    int *tries_dev = tries_out ? t_dev : nullptr;
    if (dmax == 12) hipLaunchKernelGGL(cma_sample_kernel<12>, dim3(grid), dim3(256), 0, st, a, bd_dev, (long long)P, x_dev, tries_dev);
    else if (dmax == 24) hipLaunchKernelGGL(cma_sample_kernel<24>, dim3(grid), dim3(256), 0, st, a, bd_dev, (long long)P, x_dev, tries_dev);
    else hipLaunchKernelGGL(cma_sample_kernel<32>, dim3(grid), dim3(256), 0, st, a, bd_dev, (long long)P, x_dev, tries_dev);
    ALP_HIP(hipGetLastError());
    ALP_HIP(hipMemcpyAsync(x_out, x_dev, x_bytes, hipMemcpyDeviceToHost, st));
    if (tries_out) ALP_HIP(hipMemcpyAsync(tries_out, t_dev, t_bytes, hipMemcpyDeviceToHost, st));
    ALP_HIP(hipStreamSynchronize(st));
    return ALP_OK;
}
