// libalproj_hip.so -- device-resident point sets and the three kernels that run on them:
//   project_kernel   forward projection of every point with ONE pose (HBM-bound stream)
//   popeval_kernel   P candidate poses x every point -> per-candidate loss sums
//                    (VALU-bound; candidates staged in LDS, wave64 DPP reductions)
//   residual_kernel  observed - projected, interleaved (least-squares path)
//
// Reference arithmetic: src/alproj/optimize.py  project :122-155, _distort :98-120,
// rmse :157-178, huber_loss :181-212, compute_residuals :215-237, and the generation loop
// of CMAOptimizer.optimize :418-424.  The pose-dependent scalars are folded on the host in
// float64 (alp_core.hip: fold_pose); everything per point happens here.
//
// Data layout in HBM: structure-of-arrays planes x[], y[], z[] (local coordinates =
// absolute - origin), observed uo[], vo[] and projected u[], v[] (pixels), all of one
// element type T (float: 20 B/vertex streamed per projection pass; double: 40 B/vertex).
// Planes are padded to a multiple of 1024 elements so that 16-byte vector accesses and
// whole-workgroup tiles never leave the allocation.
#include "alp_internal.h"

#include <cmath>
#include <vector>

namespace alp {

// ------------------------------------------------------------------ element-type helpers
template <typename T> struct Num;
template <> struct Num<float> {
    using vec = float4;
    static constexpr int VEC = 4;
    static __device__ __forceinline__ float rcp(float a) { return __builtin_amdgcn_rcpf(a); }
    static __device__ __forceinline__ float sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
};
template <> struct Num<double> {
    using vec = double2;
    static constexpr int VEC = 2;
    static __device__ __forceinline__ double rcp(double a) { return 1.0 / a; }
    static __device__ __forceinline__ double sqrt(double a) { return __builtin_sqrt(a); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
};

// Pose -> normalised distorted image coordinates (x1_d, y1_d) of one point q (local coords).
// r[] is the folded pose record (alp_internal.h); wave-uniform.
//   optimize.py:144-149 (rigid transform, K, perspective divide, u-mirror) are in rows 0..11;
//   optimize.py:105-116 is the rest.  Quirks Q1 (tangential product form), Q3 (centre/scale),
//   Q8 (a1/a2 on the y ratio only) are kept.  Q2: r2 is formed as x1^2+y1^2 (float) or as
//   sqrt(.)^2 (double, like the reference), r4 = r2*r2.
template <typename T>
__device__ __forceinline__ void project_norm(const T *r, T qx, T qy, T qz, T &xd, T &yd) {
    using N = Num<T>;
    const T zc = N::fma(r[8], qx, N::fma(r[9], qy, N::fma(r[10], qz, r[11])));
    const T xn = N::fma(r[0], qx, N::fma(r[1], qy, N::fma(r[2], qz, r[3])));
    const T yn = N::fma(r[4], qx, N::fma(r[5], qy, N::fma(r[6], qz, r[7])));
    const T iz = N::rcp(zc);
    const T x1 = xn * iz;
    const T y1 = yn * iz;
    const T xx = x1 * x1;
    const T yy = y1 * y1;
    T r2 = xx + yy;
    if constexpr (sizeof(T) == 8) {
        const T rr = N::sqrt(r2);
        r2 = rr * rr;
    }
    const T r4 = r2 * r2;
    const T tn = N::fma(N::fma(r[14], r2, r[13]), r2, r[12]);   // k1 + k2 r2 + k3 r4
    const T td = N::fma(N::fma(r[17], r2, r[16]), r2, r[15]);   // k4 + k5 r2 + k6 r4
    const T nx = N::fma(tn, r2, (T)1);
    const T dx = N::fma(td, r2, (T)1);
    const T ny = N::fma(tn, r2, r[18]);                          // 1 + a1 + ...
    const T dy = N::fma(td, r2, r[19]);                          // 1 + a2 + ...
    const T xy = x1 * y1;
    T ax = x1 * (nx * N::rcp(dx));
    ax = N::fma(r[20], xy, ax);            // 2 p1 x y
    ax = N::fma(r[21], r2 * xx, ax);       // p2 (r2 * 2 * x^2)
    ax = N::fma(r[22], r2, ax);            // s1 r2
    ax = N::fma(r[23], r4, ax);            // s2 r4
    T ay = y1 * (ny * N::rcp(dy));
    ay = N::fma(r[20], xy, ay);
    ay = N::fma(r[21], r2 * yy, ay);
    ay = N::fma(r[24], r2, ay);            // s3 r2
    ay = N::fma(r[25], r4, ay);            // s4 r4
    xd = ax;
    yd = ay;
}

// normalised -> pixels (optimize.py:117-118)
template <typename T>
__device__ __forceinline__ void to_pixels(const T *r, T xd, T yd, T &u, T &v) {
    u = Num<T>::fma(xd, r[26], r[26]);
    v = Num<T>::fma(yd, r[27], r[27]);
}

template <typename T> __device__ __forceinline__ T &vget(typename Num<T>::vec &a, int i);
template <> __device__ __forceinline__ float &vget<float>(float4 &a, int i) { return (&a.x)[i]; }
template <> __device__ __forceinline__ double &vget<double>(double2 &a, int i) { return (&a.x)[i]; }

// ------------------------------------------------------------------ K1: forward projection
// One pose, every point.  16-byte loads from the three coordinate planes, 16-byte stores to
// the two pixel planes: 12 + 8 = 20 B/vertex (float), 40 B/vertex (double).  The pose
// record is a kernel argument (lives in SGPRs).
template <typename T>
__global__ __launch_bounds__(256) void project_kernel(const T *__restrict__ x, const T *__restrict__ y,
                                                      const T *__restrict__ z, T *__restrict__ u,
                                                      T *__restrict__ v, int64_t nvec, PoseRec<T> pose) {
    using Vt = typename Num<T>::vec;
    const Vt *x4 = reinterpret_cast<const Vt *>(x);
    const Vt *y4 = reinterpret_cast<const Vt *>(y);
    const Vt *z4 = reinterpret_cast<const Vt *>(z);
    Vt *u4 = reinterpret_cast<Vt *>(u);
    Vt *v4 = reinterpret_cast<Vt *>(v);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        Vt qx = x4[i], qy = y4[i], qz = z4[i];
        Vt ou, ov;
#pragma unroll
        for (int k = 0; k < Num<T>::VEC; ++k) {
            T xd, yd;
            project_norm<T>(pose.v, vget<T>(qx, k), vget<T>(qy, k), vget<T>(qz, k), xd, yd);
            to_pixels<T>(pose.v, xd, yd, vget<T>(ou, k), vget<T>(ov, k));
        }
        u4[i] = ou;
        v4[i] = ov;
    }
}

// ------------------------------------------------------------------ K3: residual vector
// out[2i] = uo - u, out[2i+1] = vo - v  (optimize.py:233-236), float64 output.
template <typename T>
__global__ __launch_bounds__(256) void residual_kernel(const T *__restrict__ x, const T *__restrict__ y,
                                                       const T *__restrict__ z, const T *__restrict__ uo,
                                                       const T *__restrict__ vo, double2 *__restrict__ out,
                                                       int64_t n, PoseRec<T> pose) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T xd, yd, u, v;
        project_norm<T>(pose.v, x[i], y[i], z[i], xd, yd);
        to_pixels<T>(pose.v, xd, yd, u, v);
        out[i] = make_double2((double)(uo[i] - u), (double)(vo[i] - v));
    }
}

// ------------------------------------------------------------------ wave64 sum
// DPP butterfly inside each row of 16 lanes, then row_bcast15 / row_bcast31: lane 63 ends
// up with the sum of all 64 lanes.  Fixed order -> bitwise reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_f<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += dpp_f<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0x141, 0xf>(v);   // row_half_mirror
    v += dpp_f<0x140, 0xf>(v);   // row_mirror       -> every lane: sum of its row of 16
    v += dpp_f<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
    v += dpp_f<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3 -> lane 63: total
    return v;
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ------------------------------------------------------------------ K2: population evaluation
// grid.x persistent workgroups, each owning a contiguous stripe of the points.  For each
// tile of TC candidate poses: the workgroup stages the tile's folded records in LDS, then
// walks its stripe in groups of 256*V points held in registers; for every candidate the
// V losses of a lane are summed, the wave is reduced with DPP, and lane 63 accumulates
// into a per-wave float64 slot in LDS.  The tile ends with one float64 row of TC partial
// sums per workgroup (deterministic: no atomics).
template <typename T> struct PopCfg;
template <> struct PopCfg<float> { static constexpr int TC = 256; static constexpr int V = 8; };
template <> struct PopCfg<double> { static constexpr int TC = 128; static constexpr int V = 4; };

template <typename T, int LOSS>
__device__ __forceinline__ T point_loss(const T *r, T qx, T qy, T qz, T uo, T vo, T f_scale, T half_f2) {
    using N = Num<T>;
    T xd, yd, u, v;
    project_norm<T>(r, qx, qy, qz, xd, yd);
    to_pixels<T>(r, xd, yd, u, v);
    const T du = uo - u;
    const T dv = vo - v;
    const T d2 = N::fma(dv, dv, du * du);
    const T dist = N::sqrt(d2);
    if constexpr (LOSS == ALP_LOSS_MEAN_DIST) {
        return dist;                                              // optimize.py:176
    } else {
        // optimize.py:207-211: r <= f ? 0.5 r^2 : f (r - 0.5 f)
        const T quad = (T)0.5 * (sizeof(T) == 8 ? dist * dist : d2);
        const T lin = N::fma(f_scale, dist, -half_f2);
        // NaN: (NaN <= f) is false -> linear branch -> NaN propagates like np.where does
        return (dist <= f_scale) ? quad : lin;
    }
}

template <typename T, int LOSS, int V, bool MASKED>
__device__ __forceinline__ void pop_group(const T *__restrict__ x, const T *__restrict__ y,
                                          const T *__restrict__ z, const T *__restrict__ uo,
                                          const T *__restrict__ vo, int64_t base, int64_t end,
                                          const PoseRec<T> *s_c, double *s_sum_wave, int tc,
                                          T f_scale, T half_f2) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    T qx[V], qy[V], qz[V], ou[V], ov[V];
    bool ok[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        int64_t i = base + (int64_t)j * 256 + tid;
        ok[j] = MASKED ? (i < end) : true;
        if (MASKED && !ok[j]) i = base;       // any valid point; its loss is discarded
        qx[j] = x[i]; qy[j] = y[i]; qz[j] = z[i]; ou[j] = uo[i]; ov[j] = vo[i];
    }
    for (int c = 0; c < tc; ++c) {
        T r[28];
        const typename Num<T>::vec *rv = reinterpret_cast<const typename Num<T>::vec *>(s_c[c].v);
#pragma unroll
        for (int k = 0; k < 28 / Num<T>::VEC; ++k) {
            typename Num<T>::vec t = rv[k];
#pragma unroll
            for (int e = 0; e < Num<T>::VEC; ++e) r[k * Num<T>::VEC + e] = vget<T>(t, e);
        }
        T acc = 0;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const T l = point_loss<T, LOSS>(r, qx[j], qy[j], qz[j], ou[j], ov[j], f_scale, half_f2);
            acc += (MASKED && !ok[j]) ? (T)0 : l;
        }
        acc = wave_sum_to_lane63(acc);
        if (lane == 63) s_sum_wave[c] += (double)acc;
    }
}

template <typename T, int LOSS>
__global__ __launch_bounds__(256) void popeval_kernel(const T *__restrict__ x, const T *__restrict__ y,
                                                      const T *__restrict__ z, const T *__restrict__ uo,
                                                      const T *__restrict__ vo, int64_t n,
                                                      const PoseRec<T> *__restrict__ cands, int P,
                                                      T f_scale, double *__restrict__ partials) {
    constexpr int TC = PopCfg<T>::TC;
    constexpr int V = PopCfg<T>::V;
    __shared__ PoseRec<T> s_c[TC];
    __shared__ double s_sum[4][TC];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const T half_f2 = (T)0.5 * f_scale * f_scale;

    // stripe of this workgroup: multiples of 256 points so that only the last stripe is ragged
    const int64_t rows = (n + 255) / 256;
    const int64_t rows_per = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t beg = (int64_t)blockIdx.x * rows_per * 256;
    const int64_t end = (beg + rows_per * 256 < n) ? beg + rows_per * 256 : n;

    for (int c0 = 0; c0 < P; c0 += TC) {
        const int tc = (P - c0 < TC) ? (P - c0) : TC;
        __syncthreads();
        {   // stage tc records (16-byte vectors) and clear the accumulators
            using Vt = typename Num<T>::vec;
            const Vt *src = reinterpret_cast<const Vt *>(cands + c0);
            Vt *dst = reinterpret_cast<Vt *>(s_c);
            const int nv = tc * (int)(sizeof(PoseRec<T>) / sizeof(Vt));
            for (int i = tid; i < nv; i += 256) dst[i] = src[i];
            for (int i = tid; i < 4 * TC; i += 256) (&s_sum[0][0])[i] = 0.0;
        }
        __syncthreads();
        int64_t base = beg;
        for (; base + 256 * V <= end; base += 256 * V)
            pop_group<T, LOSS, V, false>(x, y, z, uo, vo, base, end, s_c, s_sum[wave], tc, f_scale, half_f2);
        for (; base < end; base += 256)
            pop_group<T, LOSS, 1, true>(x, y, z, uo, vo, base, end, s_c, s_sum[wave], tc, f_scale, half_f2);
        __syncthreads();
        if (tid < tc)
            partials[(int64_t)blockIdx.x * P + c0 + tid] =
                ((s_sum[0][tid] + s_sum[1][tid]) + s_sum[2][tid]) + s_sum[3][tid];
    }
}

// sums[c] = sum over workgroups of partials[b][c] (fixed order); sums[P] = local point count.
// One workgroup handles 32 candidates x 8 row-groups.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *__restrict__ partials,
                                                              int nblk, int P, double n_local,
                                                              double *__restrict__ sums) {
    __shared__ double s[8][32];
    const int cl = threadIdx.x & 31;
    const int g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double acc = 0.0;
    if (c < P)
        for (int b = g; b < nblk; b += 8) acc += partials[(int64_t)b * P + c];
    s[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < P) {
        double t = s[0][cl];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += s[k][cl];
        sums[c] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[P] = n_local;
}

// Stand-alone loss of two interleaved (n x 2) float64 arrays: one float64 partial per workgroup.
template <int LOSS>
__global__ __launch_bounds__(256) void loss_uv_kernel(const double2 *__restrict__ obs,
                                                      const double2 *__restrict__ prj, int64_t n,
                                                      double f_scale, double *__restrict__ partials) {
    __shared__ double s[4];
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double2 o = obs[i], q = prj[i];
        const double du = o.x - q.x, dv = o.y - q.y;
        const double r = __builtin_sqrt(du * du + dv * dv);
        if constexpr (LOSS == ALP_LOSS_MEAN_DIST) acc += r;
        else acc += (r <= f_scale) ? 0.5 * (r * r) : f_scale * (r - 0.5 * f_scale);
    }
    acc = wave_sum_to_lane63(acc);
    if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((s[0] + s[1]) + s[2]) + s[3];
}

// ------------------------------------------------------------------ upload helpers
// AoS (n x C, TIn) chunk -> SoA planes of T, subtracting origin in float64 first.
template <typename TIn, typename T, int C>
__global__ __launch_bounds__(256) void aos_to_planes_kernel(const TIn *__restrict__ src, int64_t count,
                                                            int64_t dst_off, double o0, double o1, double o2,
                                                            T *__restrict__ p0, T *__restrict__ p1,
                                                            T *__restrict__ p2) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        p0[dst_off + i] = (T)((double)src[i * C + 0] - o0);
        p1[dst_off + i] = (T)((double)src[i * C + 1] - o1);
        if constexpr (C == 3) p2[dst_off + i] = (T)((double)src[i * C + 2] - o2);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_strided_kernel(const T *__restrict__ u, const T *__restrict__ v,
                                                             int64_t first, int64_t stride, int64_t count,
                                                             double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        out[i] = (double)u[first + i * stride];
        out[count + i] = (double)v[first + i * stride];
    }
}

}  // namespace alp

using namespace alp;

// ------------------------------------------------------------------ the handle
struct alp_points {
    int64_t n = 0;
    int64_t n_pad = 0;
    int precision = ALP_F32;
    double origin[3] = {0, 0, 0};
    void *x = nullptr, *y = nullptr, *z = nullptr;
    void *uo = nullptr, *vo = nullptr;
    void *u = nullptr, *v = nullptr;
    bool projected = false;
    // population-evaluation scratch
    int64_t cand_cap = 0;
    void *cand_dev = nullptr;
    void *cand_host = nullptr;     // pinned
    double *partials = nullptr;
    int64_t partials_cap = 0;
    double *sums_dev = nullptr;    // cand_cap + 1
    double *sums_host = nullptr;   // pinned, cand_cap + 1
    int64_t pending_P = 0;
    size_t esize() const { return precision == ALP_F64 ? 8 : 4; }
};

namespace {

int alloc_plane(void **p, int64_t n_pad, size_t es) {
    ALP_HIP(hipMalloc(p, (size_t)n_pad * es));
    ALP_HIP(hipMemsetAsync(*p, 0, (size_t)n_pad * es, ctx().stream));
    return ALP_OK;
}

template <typename TIn, typename T, int C>
int upload_columns(const TIn *host, int64_t n, const double o[3], T *p0, T *p1, T *p2) {
    const int64_t CH = 4 << 20;   // points per staging chunk
    const int64_t ch = n < CH ? (n > 0 ? n : 1) : CH;
    TIn *stage = nullptr;
    ALP_HIP(hipMalloc((void **)&stage, (size_t)ch * C * sizeof(TIn)));
    int rc = ALP_OK;
    for (int64_t off = 0; off < n; off += ch) {
        const int64_t cnt = (n - off < ch) ? (n - off) : ch;
        hipError_t e = hipMemcpyAsync(stage, host + off * C, (size_t)cnt * C * sizeof(TIn),
                                      hipMemcpyHostToDevice, ctx().stream);
        if (e != hipSuccess) { rc = fail(ALP_EHIP, "H2D upload: %s", hipGetErrorString(e)); break; }
        const int grid = (int)((cnt + 255) / 256 < 4096 ? (cnt + 255) / 256 : 4096);
        hipLaunchKernelGGL((aos_to_planes_kernel<TIn, T, C>), dim3(grid), dim3(256), 0, ctx().stream,
                           stage, cnt, off, o[0], o[1], o[2], p0, p1, p2);
        e = hipStreamSynchronize(ctx().stream);   // staging buffer is reused
        if (e != hipSuccess) { rc = fail(ALP_EHIP, "upload kernel: %s", hipGetErrorString(e)); break; }
    }
    hipFree(stage);
    return rc;
}

template <int C>
int upload_any(const void *host, int in_dtype, int64_t n, const double o[3], int precision, void *p0,
               void *p1, void *p2) {
    if (in_dtype == ALP_F64 && precision == ALP_F64)
        return upload_columns<double, double, C>((const double *)host, n, o, (double *)p0, (double *)p1, (double *)p2);
    if (in_dtype == ALP_F64 && precision == ALP_F32)
        return upload_columns<double, float, C>((const double *)host, n, o, (float *)p0, (float *)p1, (float *)p2);
    if (in_dtype == ALP_F32 && precision == ALP_F64)
        return upload_columns<float, double, C>((const float *)host, n, o, (double *)p0, (double *)p1, (double *)p2);
    if (in_dtype == ALP_F32 && precision == ALP_F32)
        return upload_columns<float, float, C>((const float *)host, n, o, (float *)p0, (float *)p1, (float *)p2);
    return fail(ALP_EINVAL, "in_dtype must be ALP_F32 or ALP_F64");
}

int stream_grid(int64_t items) {
    // memory-bound streaming kernels: enough workgroups to fill 256 CUs x 8, grid-stride beyond
    const int64_t want = (items + 255) / 256;
    const int64_t cap = (int64_t)ctx().cu_count * 8;
    return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

template <typename T>
int launch_project(alp_points *p, const double params[ALP_NPARAM]) {
    PoseRec<T> pose;
    fold_pose_t<T>(params, p->origin, &pose);
    const int64_t nvec = (p->n + Num<T>::VEC - 1) / Num<T>::VEC;
    hipLaunchKernelGGL((project_kernel<T>), dim3(stream_grid(nvec)), dim3(256), 0, ctx().stream,
                       (const T *)p->x, (const T *)p->y, (const T *)p->z, (T *)p->u, (T *)p->v, nvec, pose);
    ALP_HIP(hipGetLastError());
    return ALP_OK;
}

int ensure_pop_scratch(alp_points *p, int64_t P, int nblk) {
    if (P > p->cand_cap) {
        const int64_t cap = round_up(P, 256);
        if (p->cand_dev) hipFree(p->cand_dev);
        if (p->cand_host) hipHostFree(p->cand_host);
        if (p->sums_dev) hipFree(p->sums_dev);
        if (p->sums_host) hipHostFree(p->sums_host);
        p->cand_dev = p->cand_host = nullptr;
        p->sums_dev = p->sums_host = nullptr;
        p->cand_cap = 0;
        const size_t rec = POSE_WORDS * p->esize();
        ALP_HIP(hipMalloc(&p->cand_dev, (size_t)cap * rec));
        ALP_HIP(hipHostMalloc(&p->cand_host, (size_t)cap * rec, hipHostMallocDefault));
        ALP_HIP(hipMalloc((void **)&p->sums_dev, (size_t)(cap + 1) * sizeof(double)));
        ALP_HIP(hipHostMalloc((void **)&p->sums_host, (size_t)(cap + 1) * sizeof(double), hipHostMallocDefault));
        p->cand_cap = cap;
    }
    const int64_t need = (int64_t)nblk * P;
    if (need > p->partials_cap) {
        if (p->partials) hipFree(p->partials);
        p->partials = nullptr;
        p->partials_cap = 0;
        ALP_HIP(hipMalloc((void **)&p->partials, (size_t)need * sizeof(double)));
        p->partials_cap = need;
    }
    return ALP_OK;
}

template <typename T>
int enqueue_popeval(alp_points *p, const double *cand, int64_t P, int loss_kind, double f_scale) {
    // persistent grid: LDS (40 KB/workgroup) admits 4 workgroups per CU = 4 waves per SIMD
    int nblk = ctx().cu_count * 4;
    const int64_t rows = (p->n + 255) / 256;
    if (rows < nblk) nblk = (int)(rows > 0 ? rows : 1);
    if (int rc = ensure_pop_scratch(p, P, nblk)) return rc;
    PoseRec<T> *h = (PoseRec<T> *)p->cand_host;
    for (int64_t i = 0; i < P; ++i) fold_pose_t<T>(cand + i * ALP_NPARAM, p->origin, &h[i]);
    ALP_HIP(hipMemcpyAsync(p->cand_dev, h, (size_t)P * sizeof(PoseRec<T>), hipMemcpyHostToDevice, ctx().stream));
    if (loss_kind == ALP_LOSS_HUBER)
        hipLaunchKernelGGL((popeval_kernel<T, ALP_LOSS_HUBER>), dim3(nblk), dim3(256), 0, ctx().stream,
                           (const T *)p->x, (const T *)p->y, (const T *)p->z, (const T *)p->uo,
                           (const T *)p->vo, p->n, (const PoseRec<T> *)p->cand_dev, (int)P, (T)f_scale,
                           p->partials);
    else
        hipLaunchKernelGGL((popeval_kernel<T, ALP_LOSS_MEAN_DIST>), dim3(nblk), dim3(256), 0, ctx().stream,
                           (const T *)p->x, (const T *)p->y, (const T *)p->z, (const T *)p->uo,
                           (const T *)p->vo, p->n, (const PoseRec<T> *)p->cand_dev, (int)P, (T)f_scale,
                           p->partials);
    ALP_HIP(hipGetLastError());
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((P + 31) / 32)), dim3(256), 0, ctx().stream,
                       p->partials, nblk, (int)P, (double)p->n, p->sums_dev);
    ALP_HIP(hipGetLastError());
    if (int rc = comm_allreduce_sum_f64(p->sums_dev, P + 1)) return rc;
    ALP_HIP(hipMemcpyAsync(p->sums_host, p->sums_dev, (size_t)(P + 1) * sizeof(double),
                           hipMemcpyDeviceToHost, ctx().stream));
    p->pending_P = P;
    return ALP_OK;
}

}  // namespace

extern "C" {

int alp_points_create(const void *xyz, int in_dtype, int64_t n, const double origin[3], int precision,
                      alp_points_t **out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(out, "out is NULL");
    *out = nullptr;
    ALP_REQUIRE(n >= 0, "n is negative");
    ALP_REQUIRE(n == 0 || xyz, "xyz is NULL");
    ALP_REQUIRE(origin, "origin is NULL");
    ALP_REQUIRE(precision == ALP_F32 || precision == ALP_F64, "precision must be ALP_F32 or ALP_F64");
    ALP_REQUIRE(in_dtype == ALP_F32 || in_dtype == ALP_F64, "in_dtype must be ALP_F32 or ALP_F64");
    alp_points *p = new alp_points();
    p->n = n;
    p->n_pad = round_up(n > 0 ? n : 1, 1024);
    p->precision = precision;
    memcpy(p->origin, origin, sizeof(p->origin));
    int rc = ALP_OK;
    for (void **pl : {&p->x, &p->y, &p->z, &p->u, &p->v})
        if ((rc = alloc_plane(pl, p->n_pad, p->esize()))) break;
    if (!rc && n > 0) rc = upload_any<3>(xyz, in_dtype, n, origin, precision, p->x, p->y, p->z);
    if (!rc) {
        hipError_t e = hipStreamSynchronize(ctx().stream);
        if (e != hipSuccess) rc = fail(ALP_EHIP, "points upload: %s", hipGetErrorString(e));
    }
    if (rc) {
        alp_points_destroy(p);
        return rc;
    }
    *out = p;
    return ALP_OK;
}

int alp_points_destroy(alp_points_t *p) {
    if (!p) return ALP_OK;
    if (ctx().ready) hipStreamSynchronize(ctx().stream);
    for (void *q : {p->x, p->y, p->z, p->uo, p->vo, p->u, p->v, p->cand_dev, (void *)p->partials,
                    (void *)p->sums_dev})
        if (q) hipFree(q);
    if (p->cand_host) hipHostFree(p->cand_host);
    if (p->sums_host) hipHostFree(p->sums_host);
    delete p;
    return ALP_OK;
}

int alp_points_count(const alp_points_t *p, int64_t *n) {
    ALP_REQUIRE(p && n, "NULL argument");
    *n = p->n;
    return ALP_OK;
}

int alp_points_set_observed(alp_points_t *p, const void *uv, int in_dtype) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p, "points handle is NULL");
    ALP_REQUIRE(p->n == 0 || uv, "uv is NULL");
    ALP_REQUIRE(in_dtype == ALP_F32 || in_dtype == ALP_F64, "in_dtype must be ALP_F32 or ALP_F64");
    if (!p->uo) {
        if (int rc = alloc_plane(&p->uo, p->n_pad, p->esize())) return rc;
        if (int rc = alloc_plane(&p->vo, p->n_pad, p->esize())) return rc;
    }
    const double zero[3] = {0, 0, 0};
    if (p->n > 0)
        if (int rc = upload_any<2>(uv, in_dtype, p->n, zero, p->precision, p->uo, p->vo, nullptr)) return rc;
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    return ALP_OK;
}

int alp_project(alp_points_t *p, const double params[ALP_NPARAM]) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p && params, "NULL argument");
    p->projected = true;
    if (p->n == 0) return ALP_OK;
    return p->precision == ALP_F64 ? launch_project<double>(p, params) : launch_project<float>(p, params);
}

int alp_projected_fetch(alp_points_t *p, void *u_out, void *v_out, int out_dtype) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p, "points handle is NULL");
    ALP_REQUIRE(out_dtype == ALP_F32 || out_dtype == ALP_F64, "out_dtype must be ALP_F32 or ALP_F64");
    if (!p->projected) return fail(ALP_ESTATE, "alp_projected_fetch: nothing projected yet");
    if (p->n == 0) return ALP_OK;
    ALP_REQUIRE(u_out && v_out, "output is NULL");
    const size_t es = p->esize();
    if ((out_dtype == ALP_F64) == (p->precision == ALP_F64)) {
        ALP_HIP(hipMemcpyAsync(u_out, p->u, (size_t)p->n * es, hipMemcpyDeviceToHost, ctx().stream));
        ALP_HIP(hipMemcpyAsync(v_out, p->v, (size_t)p->n * es, hipMemcpyDeviceToHost, ctx().stream));
        ALP_HIP(hipStreamSynchronize(ctx().stream));
        return ALP_OK;
    }
    std::vector<char> tmp((size_t)p->n * es * 2);
    ALP_HIP(hipMemcpyAsync(tmp.data(), p->u, (size_t)p->n * es, hipMemcpyDeviceToHost, ctx().stream));
    ALP_HIP(hipMemcpyAsync(tmp.data() + (size_t)p->n * es, p->v, (size_t)p->n * es, hipMemcpyDeviceToHost,
                           ctx().stream));
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    if (out_dtype == ALP_F64) {
        const float *s = (const float *)tmp.data();
        for (int64_t i = 0; i < p->n; ++i) {
            ((double *)u_out)[i] = s[i];
            ((double *)v_out)[i] = s[p->n + i];
        }
    } else {
        const double *s = (const double *)tmp.data();
        for (int64_t i = 0; i < p->n; ++i) {
            ((float *)u_out)[i] = (float)s[i];
            ((float *)v_out)[i] = (float)s[p->n + i];
        }
    }
    return ALP_OK;
}

int alp_projected_fetch_strided(alp_points_t *p, int64_t first, int64_t stride, int64_t count,
                                double *u_out, double *v_out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p && u_out && v_out, "NULL argument");
    if (!p->projected) return fail(ALP_ESTATE, "alp_projected_fetch_strided: nothing projected yet");
    ALP_REQUIRE(count >= 0 && first >= 0 && stride >= 1, "bad range");
    if (count == 0) return ALP_OK;
    ALP_REQUIRE(first + (count - 1) * stride < p->n, "range exceeds the point count");
    double *tmp = nullptr;
    ALP_HIP(hipMalloc((void **)&tmp, (size_t)count * 2 * sizeof(double)));
    const unsigned grid = (unsigned)((count + 255) / 256);
    if (p->precision == ALP_F64)
        hipLaunchKernelGGL(gather_strided_kernel<double>, dim3(grid), dim3(256), 0, ctx().stream,
                           (const double *)p->u, (const double *)p->v, first, stride, count, tmp);
    else
        hipLaunchKernelGGL(gather_strided_kernel<float>, dim3(grid), dim3(256), 0, ctx().stream,
                           (const float *)p->u, (const float *)p->v, first, stride, count, tmp);
    hipError_t e = hipMemcpyAsync(u_out, tmp, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx().stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(v_out, tmp + count, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx().stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);
    hipFree(tmp);
    if (e != hipSuccess) return fail(ALP_EHIP, "strided fetch: %s", hipGetErrorString(e));
    return ALP_OK;
}

int alp_residuals(alp_points_t *p, const double params[ALP_NPARAM], double *out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p && params, "NULL argument");
    if (!p->uo) return fail(ALP_ESTATE, "alp_residuals: observed uv not set");
    if (p->n == 0) return ALP_OK;
    ALP_REQUIRE(out, "out is NULL");
    double2 *dev = nullptr;
    ALP_HIP(hipMalloc((void **)&dev, (size_t)p->n * sizeof(double2)));
    const int grid = stream_grid(p->n);
    if (p->precision == ALP_F64) {
        PoseRec<double> pose;
        fold_pose_t<double>(params, p->origin, &pose);
        hipLaunchKernelGGL(residual_kernel<double>, dim3(grid), dim3(256), 0, ctx().stream,
                           (const double *)p->x, (const double *)p->y, (const double *)p->z,
                           (const double *)p->uo, (const double *)p->vo, dev, p->n, pose);
    } else {
        PoseRec<float> pose;
        fold_pose_t<float>(params, p->origin, &pose);
        hipLaunchKernelGGL(residual_kernel<float>, dim3(grid), dim3(256), 0, ctx().stream,
                           (const float *)p->x, (const float *)p->y, (const float *)p->z,
                           (const float *)p->uo, (const float *)p->vo, dev, p->n, pose);
    }
    hipError_t e = hipMemcpyAsync(out, dev, (size_t)p->n * sizeof(double2), hipMemcpyDeviceToHost, ctx().stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);
    hipFree(dev);
    if (e != hipSuccess) return fail(ALP_EHIP, "residuals: %s", hipGetErrorString(e));
    return ALP_OK;
}

int alp_loss_uv(const double *observed, const double *projected, int64_t n, int loss_kind, double f_scale,
                double *loss_out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(loss_out, "loss_out is NULL");
    ALP_REQUIRE(n >= 0, "n is negative");
    ALP_REQUIRE(loss_kind == ALP_LOSS_MEAN_DIST || loss_kind == ALP_LOSS_HUBER, "unknown loss_kind");
    if (n == 0) {                      // np.mean of an empty array
        *loss_out = NAN;
        return ALP_OK;
    }
    ALP_REQUIRE(observed && projected, "NULL input");
    const int grid = stream_grid(n);
    char *dev = nullptr;
    const size_t bytes = (size_t)n * sizeof(double2);
    ALP_HIP(hipMalloc((void **)&dev, 2 * bytes + (size_t)(grid + 2) * sizeof(double)));
    double *partials = (double *)(dev + 2 * bytes);
    hipError_t e = hipMemcpyAsync(dev, observed, bytes, hipMemcpyHostToDevice, ctx().stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dev + bytes, projected, bytes, hipMemcpyHostToDevice, ctx().stream);
    if (e == hipSuccess) {
        if (loss_kind == ALP_LOSS_HUBER)
            hipLaunchKernelGGL(loss_uv_kernel<ALP_LOSS_HUBER>, dim3(grid), dim3(256), 0, ctx().stream,
                               (const double2 *)dev, (const double2 *)(dev + bytes), n, f_scale, partials);
        else
            hipLaunchKernelGGL(loss_uv_kernel<ALP_LOSS_MEAN_DIST>, dim3(grid), dim3(256), 0, ctx().stream,
                               (const double2 *)dev, (const double2 *)(dev + bytes), n, f_scale, partials);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, ctx().stream, partials, grid, 1,
                           (double)n, partials + grid);
        e = hipGetLastError();
    }
    double res[2] = {0, 0};
    if (e == hipSuccess)
        e = hipMemcpyAsync(res, partials + grid, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx().stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);
    hipFree(dev);
    if (e != hipSuccess) return fail(ALP_EHIP, "alp_loss_uv: %s", hipGetErrorString(e));
    *loss_out = res[0] / (double)n;
    return ALP_OK;
}

int alp_eval_population_enqueue(alp_points_t *p, const double *cand, int64_t P, int loss_kind,
                                double f_scale) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p && cand, "NULL argument");
    ALP_REQUIRE(P >= 1 && P <= (1 << 20), "P out of range");
    ALP_REQUIRE(loss_kind == ALP_LOSS_MEAN_DIST || loss_kind == ALP_LOSS_HUBER, "unknown loss_kind");
    if (!p->uo) return fail(ALP_ESTATE, "alp_eval_population: observed uv not set");
    if (p->precision == ALP_F64) return enqueue_popeval<double>(p, cand, P, loss_kind, f_scale);
    return enqueue_popeval<float>(p, cand, P, loss_kind, f_scale);
}

int alp_eval_population_wait(alp_points_t *p, double *loss_out, int64_t *argmin_out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p, "points handle is NULL");
    if (p->pending_P <= 0) return fail(ALP_ESTATE, "alp_eval_population_wait: nothing enqueued");
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    const int64_t P = p->pending_P;
    p->pending_P = 0;
    const double n_total = p->sums_host[P];
    int64_t best = -1;
    double best_v = 0;
    for (int64_t i = 0; i < P; ++i) {
        const double l = p->sums_host[i] / n_total;    // np.mean over all vertices
        if (loss_out) loss_out[i] = l;
        if (l == l && (best < 0 || l < best_v)) {       // NaN never wins; first index on ties
            best = i;
            best_v = l;
        }
    }
    if (argmin_out) *argmin_out = best < 0 ? 0 : best;
    return ALP_OK;
}

int alp_eval_population(alp_points_t *p, const double *cand, int64_t P, int loss_kind, double f_scale,
                        double *loss_out, int64_t *argmin_out) {
    if (int rc = alp_eval_population_enqueue(p, cand, P, loss_kind, f_scale)) return rc;
    return alp_eval_population_wait(p, loss_out, argmin_out);
}

}  // extern "C"
