// Device code of the point-set path (included by alp_points.hip and by the development
// micro-benchmarks under tools/): element-type helpers, the projection arithmetic, and the
// kernels K1 project_kernel, K2 popeval_kernel (+ reduce_partials_kernel), K3 residual_batch_kernel,
// the stand-alone loss kernel and the upload helpers.
//
// Reference arithmetic: src/alproj/optimize.py  project :122-155, _distort :98-120,
// rmse :157-178, huber_loss :181-212, compute_residuals :215-237, and the generation loop
// of CMAOptimizer.optimize :418-424.  The pose-dependent scalars are folded on the host in
// float64 (alp_core.hip: fold_pose); everything per point happens here.
#pragma once

#include "alp_internal.h"

namespace alp {

// ------------------------------------------------------------------ element-type helpers
template <typename T> struct Num;
template <> struct Num<float> {
    using vec = float4;
    static constexpr int VEC = 4;
    static __device__ __forceinline__ float rcp(float a) { return __builtin_amdgcn_rcpf(a); }
    static __device__ __forceinline__ float sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float rcp_pop(float a) { return rcp(a); }
    static __device__ __forceinline__ float sqrt_pop(float a) { return sqrt(a); }
    // streaming (non-temporal) 16-byte accesses: data touched once per pass
    typedef float native4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ float4 nt_load(const float4 *p) {
        const native4 t = __builtin_nontemporal_load(reinterpret_cast<const native4 *>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    }
    static __device__ __forceinline__ void nt_store(const float4 &a, float4 *p) {
        const native4 t = {a.x, a.y, a.z, a.w};
        __builtin_nontemporal_store(t, reinterpret_cast<native4 *>(p));
    }
};
template <> struct Num<double> {
    using vec = double2;
    static constexpr int VEC = 2;
    // v_rcp_f64 + two Newton steps in FMAs: relative error ~1e-16 (not the correctly rounded quotient; the
    // parity mode is held to 1e-9 and observes ~1e-13) at 5 instructions instead of the ~15 of the IEEE
    // division expansion -- the float64 population kernel spends three of these per evaluation
    // (a = 0 or inf: the residual 1 - a y is NaN and the raw +-inf / 0 of v_rcp_f64 is returned, as 1.0 / a gives)
    static __device__ __forceinline__ double rcp(double a) {
        const double y0 = __builtin_amdgcn_rcp(a);
        const double e0 = __builtin_fma(-a, y0, 1.0);
        const double y1 = __builtin_fma(y0, e0, y0);
        const double y2 = __builtin_fma(y1, __builtin_fma(-a, y1, 1.0), y1);
        return e0 == e0 ? y2 : y0;
    }
    // v_rsq_f64, one Goldschmidt step, one correction: 7 instructions instead of the ~16 of the IEEE expansion
    // (which adds range scaling and a second correction).  tools/sqrt_f64.hip: equal to the IEEE square root on
    // 3 x 2^30 random inputs with exponents -1000..1000 (0 differences); +-0 and +inf come back as they are
    // (v_cmp_class), NaN and negative inputs give NaN; no intermediate overflows below 2^1023; below 2^-1022
    // (denormal input, a residual of 1e-154) the result keeps the input's few bits.
    static __device__ __forceinline__ double sqrt(double a) {
        const double y = __builtin_amdgcn_rsq(a);
        double g = a * y, h = 0.5 * y;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        g = __builtin_fma(__builtin_fma(-g, g, a), h, g);
        return __builtin_amdgcn_class(a, 0x260) ? a : g;       // 0x260: -0, +0, +inf
    }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    // The population kernel's forms (K2 is bound by float64 instruction issue, K1 / K3 by HBM and PCIe: those keep the forms
    // above and their bit patterns).  rcp_pop: the special operands (0, inf, NaN) are put right by ONE v_div_fixup_f64 instead
    // of a compare and two selects; POP_F64_NEWTON = 1 drops the second Newton step (relative error 2^-52 + rounding instead of
    // ~2^-53: measured against the fixtures in profiles/r05_popeval_f64_isa_census.txt).  sqrt_pop: the refinement of h is
    // dropped -- the correction term (a - g^2) h is itself of relative size 2^-52, so h's 2^-26 error moves the result by 2^-78.
#ifndef POP_F64_NEWTON
#define POP_F64_NEWTON 2
#endif
    static __device__ __forceinline__ double rcp_pop(double a) {
        const double y0 = __builtin_amdgcn_rcp(a);
        double y = __builtin_fma(y0, __builtin_fma(-a, y0, 1.0), y0);
        if (POP_F64_NEWTON >= 2) y = __builtin_fma(y, __builtin_fma(-a, y, 1.0), y);
        return __builtin_amdgcn_div_fixup(y, a, 1.0);
    }
    static __device__ __forceinline__ double sqrt_pop(double a) {
        const double y = __builtin_amdgcn_rsq(a);
        double g = a * y;
        const double h = 0.5 * y;
        g = __builtin_fma(g, __builtin_fma(-h, g, 0.5), g);
        g = __builtin_fma(__builtin_fma(-g, g, a), h, g);
        return __builtin_amdgcn_class(a, 0x260) ? a : g;       // 0x260: -0, +0, +inf
    }
    typedef double native2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ double2 nt_load(const double2 *p) {
        const native2 t = __builtin_nontemporal_load(reinterpret_cast<const native2 *>(p));
        return make_double2(t.x, t.y);
    }
    static __device__ __forceinline__ void nt_store(const double2 &a, double2 *p) {
        const native2 t = {a.x, a.y};
        __builtin_nontemporal_store(t, reinterpret_cast<native2 *>(p));
    }
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
    // One 16-byte vector per lane and NO grid-stride loop; every byte is touched exactly once,
    // so loads and stores are non-temporal.  Measured on 100 M vertices (tools/project_lab.hip):
    //   grid-stride, 2048 workgroups, plain accesses   0.400 ms  5.0 TB/s
    //   one vector per lane, plain accesses            0.343 ms  5.8 TB/s  (= a bare 3-in/2-out copy)
    //   one vector per lane, nt loads + nt stores      0.319 ms  6.27 TB/s
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    Vt qx = Num<T>::nt_load(x4 + i), qy = Num<T>::nt_load(y4 + i), qz = Num<T>::nt_load(z4 + i);
    Vt ou, ov;
#pragma unroll
    for (int k = 0; k < Num<T>::VEC; ++k) {
        T xd, yd;
        project_norm<T>(pose.v, vget<T>(qx, k), vget<T>(qy, k), vget<T>(qz, k), xd, yd);
        to_pixels<T>(pose.v, xd, yd, vget<T>(ou, k), vget<T>(ov, k));
    }
    Num<T>::nt_store(ou, u4 + i);
    Num<T>::nt_store(ov, v4 + i);
}

// ------------------------------------------------------------------ K3: residual vectors
// out[b][2i] = uo - u_b, out[b][2i+1] = vo - v_b  (optimize.py:233-236), float64 output.
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
// float64: the same butterfly with the two halves of the value moved by a DPP mov each (gfx950 has no 64-bit DPP add);
// until round 4 six __shfl_xor steps = twelve ds_bpermute_b32 through the LDS crossbar per candidate and group
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
    v += dpp_d<0xB1, 0xf>(v);
    v += dpp_d<0x4E, 0xf>(v);
    v += dpp_d<0x141, 0xf>(v);
    v += dpp_d<0x140, 0xf>(v);
    v += dpp_d<0x142, 0xa>(v);
    v += dpp_d<0x143, 0xc>(v);
    return v;
}

// ------------------------------------------------------------------ K2: population evaluation
// grid.x persistent workgroups, each owning a contiguous stripe of the points.  For each
// tile of TC candidate poses: the workgroup stages the tile's folded records in LDS, then
// walks its stripe in groups of 256*V points held in registers; for every candidate the
// V losses of a lane are summed, the wave is reduced with DPP, and lane 63 accumulates
// into a per-wave float64 slot in LDS.  The tile ends with one float64 row of TC partial
// sums per workgroup (deterministic: no atomics).
//
// The V evaluations of a lane are independent; they are written stage by stage (all V
// transforms, all V reciprocals, ...) so that the in-order wave always has V independent
// instructions between a result and its first use: v_rcp_f32 / v_sqrt_f32 issue at a
// quarter of the v_fma_f32 rate on gfx950 (measured: 8.2 vs 2.2 cycles per wave64
// instruction, tools/valu_rate.hip) and have the longest latency.
template <typename T, int V_, int TC_, int MINW_ = 1>
struct PopCfgT {
    static constexpr int V = V_;
    static constexpr int TC = TC_;
    static constexpr int MINW = MINW_;
};
template <typename T> struct PopCfg;
// Measured on MI355X (tools/popeval_lab.hip, 10 M points x 256 candidates, Huber):
//   V=1 TC=256 4 wg/CU 512 Gevals/s | V=4 TC=256 4 wg/CU 699 | V=8 TC=256 4 wg/CU 693
//   V=4 TC=128 8 wg/CU 736          | V=8 TC=128 6 wg/CU 744 | V=16 TC=256 2 wg/CU 549
// After the reciprocal / Horner trims (45 + 3 instructions) on the library kernel, 100 M x 2048:
//   V=3 251 ms | V=4 238.5 | V=5 234.1 | V=6 231.6 (230.0 with 4 waves/SIMD asked for) | V=7 -- (123 VGPRs) | V=8 241.2
//   candidate tile at V=6: TC=64 230.2 | TC=128 229.3 | TC=256 246.9
// The bare arithmetic of group_losses (no LDS, no reduction; tools/eval_rate.hip) runs at
// 780-880 Gevals/s: ~50 VALU instructions of which 4 are quarter-rate v_rcp/v_sqrt.
// Reading the records with scalar loads straight from global memory (s_load_dwordx16, SGPR
// operands, no LDS tile) measured 709-718 against 732 Gevals/s for the LDS tile, V=8 on top of
// it 679.
#ifndef POP_V
#define POP_V 6
#endif
#ifndef POP_MINW
#define POP_MINW 4
#endif
#ifndef POP_TC
#define POP_TC 128
#endif
template <> struct PopCfg<float> : PopCfgT<float, POP_V, POP_TC, POP_MINW> {};
// float64, 100 M x 2048 Huber after round 5's instruction trims (tools/sweep_popeval_f64.sh, ms): V=3 564 | V=4 546 | V=5 541
// (156 VGPRs, still 3 waves per SIMD) | V=4 with TC=64 564; stripes per CU at V=5: 3: 548 | 4: 541 | 6: 534 | 12: 531 | 24: 527 | 48: 525
#ifndef POP_VD
#define POP_VD 5
#endif
#ifndef POP_TCD
#define POP_TCD 128
#endif
// three waves per SIMD asked for (<= 168 VGPRs): what the compiler chose by itself for the main variants (154-158); round 6's
// second walk made it give the shared-pose Huber variant 192 without the hint
#ifndef POP_MINWD
#define POP_MINWD 3
#endif
template <> struct PopCfg<double> : PopCfgT<double, POP_VD, POP_TCD, POP_MINWD> {};
// lens-free variant (group_loss_sum_lens_free: 16 + 2 instructions per evaluation, five registers per point): more points per lane
// amortise the record reads and the cross-lane sum over more evaluations
#ifndef POP_V_LF
#define POP_V_LF 8
#endif
#ifndef POP_VD_LF
#define POP_VD_LF 6
#endif
template <typename T> struct PopCfgLF;
template <> struct PopCfgLF<float> : PopCfgT<float, POP_V_LF, POP_TC, POP_MINW> {};
template <> struct PopCfgLF<double> : PopCfgT<double, POP_VD_LF, POP_TCD, POP_MINWD> {};
// Q2 in the float64 evaluation (profiles/r05_popeval_f64_isa_census.txt: 0 saves 11 instructions of 83 and moves one pole-adjacent
// fixture value by 1.3e-12 -- kept at 1, the reference's form)
#ifndef POP_F64_Q2
#define POP_F64_Q2 1
#endif

// Sum of the losses of V points against one pose record r (wave-uniform).  uoc/voc are the
// observed pixels minus the image centre (c0, c1 are the same for every candidate of a call:
// checked on the host), so the residual is one fma: du = uoc - c0*a.
//
// Instruction count per point-candidate (float, Huber): 9 fma (folded transform) + 2 mul
// (perspective) + 3 (x^2, y^2, r2) + 4 fma (radial numerator/denominator polynomials) + 4 fma
// (+1, +1+a1, +1, +1+a2) + 13 (tangential/prism terms with shared 2p1xy and 2p2r2, ratios) +
// 4 (residuals, squared distance) + 3 (Huber as c (2r - c), c = min(r, f), halved once per group of V, fused into the
// accumulation) = 43 full-rate + 4 quarter-rate (1/Z, two denominators, sqrt) in float64; the
// float32 form shares ONE reciprocal between the two denominators (3 multiplies instead of a
// quarter-rate v_rcp_f32) and folds 2 p2 r2 into the thin-prism Horner form: 45 + 3.
// normalised, centred image coordinates of V points for the pose in rows 0..11 of r
template <typename T, int V>
struct NormCoords {
    T x1[V], y1[V], xx[V], yy[V], r2[V];
};

template <typename T, int V>
__device__ __forceinline__ void norm_coords(const T *r, const T (&qx)[V], const T (&qy)[V], const T (&qz)[V],
                                            NormCoords<T, V> &o) {
    using N = Num<T>;
    T zc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {     // rigid transform + K + mirror + centring, folded (rows 0..11)
        zc[j] = N::fma(r[8], qx[j], N::fma(r[9], qy[j], N::fma(r[10], qz[j], r[11])));
        o.x1[j] = N::fma(r[0], qx[j], N::fma(r[1], qy[j], N::fma(r[2], qz[j], r[3])));
        o.y1[j] = N::fma(r[4], qx[j], N::fma(r[5], qy[j], N::fma(r[6], qz[j], r[7])));
    }
#pragma unroll
    for (int j = 0; j < V; ++j) zc[j] = N::rcp_pop(zc[j]);
#pragma unroll
    for (int j = 0; j < V; ++j) {
        o.x1[j] *= zc[j];
        o.y1[j] *= zc[j];
        o.xx[j] = o.x1[j] * o.x1[j];
        o.yy[j] = o.y1[j] * o.y1[j];
        o.r2[j] = o.xx[j] + o.yy[j];
        if constexpr (sizeof(T) == 8 && POP_F64_Q2) {            // Q2: the reference squares sqrt(x^2+y^2)
            const T rr = N::sqrt_pop(o.r2[j]);
            o.r2[j] = rr * rr;
        }
    }
}

// normalised DISTORTED coordinates (x1_d, y1_d of optimize.py:112-116) of V points whose normalised coordinates are in nc;
// the stages are written one after the other over all V points (see above)
// EXACT_POLES: one reciprocal per denominator, as the reference divides (optimize.py:112-116).  The shared reciprocal below
// is wrong exactly where ONE denominator is zero: 1/dx = dy . inf is still +-inf, but 1/dy = dx . inf = 0 . inf = NaN where the
// reference's y1_d is finite -- a NaN loss instead of an infinite one.  popeval_kernel therefore walks a wave's share of a
// stripe again with EXACT_POLES when one of its candidate sums came out infinite or NaN (rare: an out-of-frame vertex exactly
// on a pole of the rational lens model, a vertex at the camera, a float32 overflow next to the camera plane).
template <typename T, int V, bool EXACT_POLES = false>
__device__ __forceinline__ void distort_group(const T *r, const NormCoords<T, V> &nc, T (&a)[V], T (&b)[V]) {
    using N = Num<T>;
    const T (&x1)[V] = nc.x1;
    const T (&y1)[V] = nc.y1;
    const T (&xx)[V] = nc.xx;
    const T (&yy)[V] = nc.yy;
    const T (&r2)[V] = nc.r2;
    T nx[V], ny[V], dx[V], dy[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const T tn = N::fma(N::fma(r[14], r2[j], r[13]), r2[j], r[12]);   // k1 + k2 r2 + k3 r4
        const T td = N::fma(N::fma(r[17], r2[j], r[16]), r2[j], r[15]);   // k4 + k5 r2 + k6 r4
        nx[j] = N::fma(tn, r2[j], (T)1);
        ny[j] = N::fma(tn, r2[j], r[18]);
        dx[j] = N::fma(td, r2[j], (T)1);
        dy[j] = N::fma(td, r2[j], r[19]);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        // one reciprocal for both denominators: 1/dx = dy / (dx dy), 1/dy = dx / (dx dy)  (float32: a quarter-rate
        // v_rcp_f32 saved; float64, since round 5: 3 multiplies instead of a second v_rcp_f64 + Newton steps + fix-up)
        if constexpr (EXACT_POLES) {
            dx[j] = N::rcp_pop(dx[j]);
            dy[j] = N::rcp_pop(dy[j]);
        } else {
            const T inv = N::rcp_pop(dx[j] * dy[j]);
            const T idx = dy[j] * inv;
            dy[j] = dx[j] * inv;
            dx[j] = idx;
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        // x1_d = x1 num/den + 2 p1 x y + 2 p2 r2 x^2 + r2 (s1 + s2 r2)   (optimize.py:112-116, Q1)
        const T t1 = r[20] * (x1[j] * y1[j]);             // 2 p1 x y       (shared by x and y)
        // r2 (2 p2 x^2 + s1 + s2 r2) + t1: one multiply fewer per coordinate pair than 2 p2 r2 formed on its own
        a[j] = N::fma(r2[j], N::fma(r[21], xx[j], N::fma(r[23], r2[j], r[22])), t1);
        b[j] = N::fma(r2[j], N::fma(r[21], yy[j], N::fma(r[25], r2[j], r[24])), t1);
        a[j] = N::fma(x1[j], nx[j] * dx[j], a[j]);
        b[j] = N::fma(y1[j], ny[j] * dy[j], b[j]);
    }
}

// SHARED_POSE: every candidate of the call has the same rows 0..11 (only distortion
// coefficients are optimised, the reference's second phase, example.py:75-78): the
// normalised coordinates `pre` were computed once per point outside the candidate loop.
template <typename T, int LOSS, int V, bool MASKED, bool SHARED_POSE, bool EXACT_POLES = false>
__device__ __forceinline__ T group_loss_sum(const T *r, const T (&qx)[V], const T (&qy)[V], const T (&qz)[V],
                                            const NormCoords<T, V> &pre, const T (&uoc)[V], const T (&voc)[V],
                                            const bool (&ok)[V], T f_scale) {
    using N = Num<T>;
    NormCoords<T, V> own;
    if constexpr (!SHARED_POSE) norm_coords<T, V>(r, qx, qy, qz, own);
    const NormCoords<T, V> &nc = SHARED_POSE ? pre : own;
    T a[V], b[V], d2[V], dist[V];
    distort_group<T, V, EXACT_POLES>(r, nc, a, b);
#pragma unroll
    for (int j = 0; j < V; ++j) {
        // pixels u = a c0 + c0 (optimize.py:117-118): residual uo - u = (uo - c0) - c0 a
        const T du = N::fma(a[j], r[28], uoc[j]);
        const T dv = N::fma(b[j], r[29], voc[j]);
        d2[j] = N::fma(dv, dv, du * du);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) dist[j] = N::sqrt_pop(d2[j]);
    T acc = 0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        if constexpr (LOSS == ALP_LOSS_MEAN_DIST) {
            acc += (MASKED && !ok[j]) ? (T)0 : dist[j];                                  // optimize.py:176
        } else {
            // Huber (optimize.py:207-211) without a branch: with c = min(r, f),
            // 0.5 c (2r - c) = 0.5 r^2 for r <= f and f (r - 0.5 f) beyond; a NaN r gives c = f and
            // a NaN term, an infinite r an infinite term, like np.where
            // (float64, since round 5: min and two fma instead of two multiplies, an fma, a compare, two selects and an add)
            const T c = sizeof(T) == 4 ? (T)__builtin_fminf((float)dist[j], (float)f_scale) : (T)__builtin_fmin((double)dist[j], (double)f_scale);
            const T t = N::fma((T)2, dist[j], -c);
            if (MASKED && !ok[j]) continue;
            acc = N::fma(c, t, acc);                      // twice the loss: halved once below (exact: a power of two)
        }
    }
    if constexpr (LOSS != ALP_LOSS_MEAN_DIST) acc *= (T)0.5;      // one multiply per V evaluations instead of one each
    return acc;
}

// LENS-FREE populations (every candidate has k1..k6 = p1 = p2 = s1..s4 = 0: the reference's first optimisation phase,
// example.py:51-54, BASELINE config 3).  optimize.py:112-118 then reduces to u = c0 x1 + c0, v = c1 y1 (1 + a1) / (1 + a2) + c1,
// and the host folds -c0 and -c1 (1 + a1) / (1 + a2) into the X' and Y' rows in float64 (host/alp_host.cpp:
// fold_pose_lens_free): per evaluation 9 FMA (the three rows), one reciprocal, 2 FMA (the residuals, perspective divide
// included), 2 for the squared distance, one square root and the loss -- 16 full-rate + 2 quarter-rate vector instructions where
// the general form needs 45 + 3.  What the general form would make of a non-finite value (0 . inf = NaN where a coefficient is
// zero) is restored by the second walk of popeval_kernel, which runs the GENERAL arithmetic on the general records.
template <typename T, int LOSS, int V, bool MASKED>
__device__ __forceinline__ T group_loss_sum_lens_free(const T *r, const T (&qx)[V], const T (&qy)[V], const T (&qz)[V],
                                                      const T (&uoc)[V], const T (&voc)[V], const bool (&ok)[V], T f_scale) {
    using N = Num<T>;
    T zc[V], xn[V], yn[V], d2[V], dist[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        zc[j] = N::fma(r[8], qx[j], N::fma(r[9], qy[j], N::fma(r[10], qz[j], r[11])));
        xn[j] = N::fma(r[0], qx[j], N::fma(r[1], qy[j], N::fma(r[2], qz[j], r[3])));
        yn[j] = N::fma(r[4], qx[j], N::fma(r[5], qy[j], N::fma(r[6], qz[j], r[7])));
    }
#pragma unroll
    for (int j = 0; j < V; ++j) zc[j] = N::rcp_pop(zc[j]);
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const T du = N::fma(xn[j], zc[j], uoc[j]);        // (uo - c0) - c0 x1
        const T dv = N::fma(yn[j], zc[j], voc[j]);
        d2[j] = N::fma(dv, dv, du * du);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) dist[j] = N::sqrt_pop(d2[j]);
    T acc = 0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        if constexpr (LOSS == ALP_LOSS_MEAN_DIST) {
            acc += (MASKED && !ok[j]) ? (T)0 : dist[j];
        } else {                                          // Huber as in group_loss_sum
            const T c = sizeof(T) == 4 ? (T)__builtin_fminf((float)dist[j], (float)f_scale) : (T)__builtin_fmin((double)dist[j], (double)f_scale);
            const T t = N::fma((T)2, dist[j], -c);
            if (MASKED && !ok[j]) continue;
            acc = N::fma(c, t, acc);
        }
    }
    if constexpr (LOSS != ALP_LOSS_MEAN_DIST) acc *= (T)0.5;
    return acc;
}

// ------------------------------------------------------------------ K3: residual vectors of B poses
// Residual vectors of B poses at once (finite-difference Jacobian of the least-squares path:
// scipy's 2-point scheme needs D+1 evaluations per iteration, optimize.py:510-528).  Each point
// is loaded once; the pose records are read with wave-uniform (scalar) loads.
// out[b][i] = (uo - u_b, vo - v_b), b-major.
// RES_V points per lane: independent chains between a transcendental and its use.  float32 runs the population kernel's stages
// (norm_coords, distort_group with a reciprocal per denominator), float64 K1's project_norm (round 6: bit-equal to project()).
// 22 poses x 10 M points, float64 through the population stages (round 5), kernel ms: RES_V = 1: 0.825 | 2: 0.834 | 3: 0.771 | 4: 0.782
#ifndef RES_V
#define RES_V 3
#endif
template <typename T>
__global__ __launch_bounds__(256) void residual_batch_kernel(const T *__restrict__ x, const T *__restrict__ y,
                                                             const T *__restrict__ z, const T *__restrict__ uo,
                                                             const T *__restrict__ vo, double2 *__restrict__ out,
                                                             int64_t n, const PoseRec<T> *__restrict__ poses, int B) {
    constexpr int V = RES_V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * V;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x * V + threadIdx.x; base < n; base += stride) {
        T qx[V], qy[V], qz[V], ou[V], ov[V];
        int64_t idx[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int64_t i = base + (int64_t)j * blockDim.x;
            idx[j] = i < n ? i : -1;
            const int64_t k = i < n ? i : base;            // a lane's first point always exists
            qx[j] = x[k]; qy[j] = y[k]; qz[j] = z[k]; ou[j] = uo[k]; ov[j] = vo[k];
        }
        for (int b = 0; b < B; ++b) {
            const T *r = poses[b].v;
            T xd[V], yd[V];
            if constexpr (sizeof(T) == 8) {
                // float64 (the parity mode): K1's own arithmetic point by point, so that the residuals ARE observed - project()
                // bit for bit, as in the reference (optimize.py:233-236).  Round 5 ran the population kernel's stages here
                // (0.77 against 0.83 ms of kernel at 22 poses x 10 M points, inside a call of 11 ms of PCIe time).
#pragma unroll
                for (int j = 0; j < V; ++j) project_norm<T>(r, qx[j], qy[j], qz[j], xd[j], yd[j]);
            } else {
                NormCoords<T, V> nc;
                norm_coords<T, V>(r, qx, qy, qz, nc);
                distort_group<T, V, true>(r, nc, xd, yd);           // a reciprocal per denominator: +-inf at a pole, like the reference
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                T u, v;
                to_pixels<T>(r, xd[j], yd[j], u, v);
                if (idx[j] >= 0)      // written once, read by the copy engine
                    Num<double>::nt_store(make_double2((double)(ou[j] - u), (double)(ov[j] - v)), out + ((int64_t)b * n + idx[j]));
            }
        }
    }
}

// TS = element type of the planes in HBM, T = arithmetic type (TS = float with T = double is the
// float64 re-evaluation of a float32 point set: alp_eval_population's argmin confirmation)
template <typename T, int LOSS, int V, bool MASKED, bool SHARED_POSE, typename TS = T, bool EXACT_POLES = false, bool LENS_FREE = false>
__device__ __forceinline__ void pop_group(const TS *__restrict__ x, const TS *__restrict__ y,
                                          const TS *__restrict__ z, const TS *__restrict__ uo,
                                          const TS *__restrict__ vo, int64_t base, int64_t end,
                                          const PoseRec<T> *s_c, double *s_sum_wave, int tc, T f_scale,
                                          unsigned long long redo_lo = ~0ull, unsigned long long redo_hi = ~0ull) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    T qx[V], qy[V], qz[V], uoc[V], voc[V];
    bool ok[V];
    const T c0 = s_c[0].v[26], c1 = s_c[0].v[27];     // identical in every record of a call
#pragma unroll
    for (int j = 0; j < V; ++j) {
        int64_t i = base + (int64_t)j * 256 + tid;
        ok[j] = MASKED ? (i < end) : true;
        if (MASKED && !ok[j]) i = base;       // any valid point; its loss is discarded
        qx[j] = (T)x[i]; qy[j] = (T)y[i]; qz[j] = (T)z[i];
        uoc[j] = (T)uo[i] - c0;
        voc[j] = (T)vo[i] - c1;
    }
    NormCoords<T, V> pre;
    if constexpr (SHARED_POSE && !LENS_FREE) norm_coords<T, V>(s_c[0].v, qx, qy, qz, pre);
    constexpr int WORDS = LENS_FREE ? 12 : 32;        // a lens-free record is its three rows
    for (int c = 0; c < tc; ++c) {
        if constexpr (EXACT_POLES)         // the second walk: only the candidates whose sums were not finite (wave-uniform bits)
            if (!(((c < 64 ? redo_lo >> c : redo_hi >> (c - 64)) & 1ull))) continue;
        T r[WORDS];
        const typename Num<T>::vec *rv = reinterpret_cast<const typename Num<T>::vec *>(s_c[c].v);
#pragma unroll
        for (int k = 0; k < WORDS / Num<T>::VEC; ++k) {
            typename Num<T>::vec t = rv[k];
#pragma unroll
            for (int e = 0; e < Num<T>::VEC; ++e) r[k * Num<T>::VEC + e] = vget<T>(t, e);
        }
        T acc;
        if constexpr (LENS_FREE) acc = group_loss_sum_lens_free<T, LOSS, V, MASKED>(r, qx, qy, qz, uoc, voc, ok, f_scale);
        else acc = group_loss_sum<T, LOSS, V, MASKED, SHARED_POSE, EXACT_POLES>(r, qx, qy, qz, pre, uoc, voc, ok, f_scale);
        acc = wave_sum_to_lane63(acc);
        if (lane == 63) s_sum_wave[c] += (double)acc;
    }
}

// which kernels take the second walk (see popeval_kernel): float64 arithmetic, and lens-free tiles of either precision
#define POP_SECOND_WALK(T, LENS_FREE) (sizeof(T) == 8 || (LENS_FREE))

// one workgroup's stripe [beg, end) against the tc staged records: wide groups of V rows, the rows they leave over two at a
// time, the ragged last row masked
template <typename T, int LOSS, int V, bool SHARED_POSE, typename TS, bool EXACT_POLES, bool LENS_FREE = false>
__device__ __forceinline__ void pop_walk_stripe(const TS *__restrict__ x, const TS *__restrict__ y, const TS *__restrict__ z,
                                                const TS *__restrict__ uo, const TS *__restrict__ vo, int64_t beg, int64_t end,
                                                const PoseRec<T> *recs, double *s_sum_wave, int tc, T f_scale,
                                                unsigned long long redo_lo = ~0ull, unsigned long long redo_hi = ~0ull) {
    // (the second walk, EXACT_POLES, goes row by row: it is rare, and its two reciprocals per point over V points in flight
    // would set the kernel's register count -- 178 instead of 154 VGPRs in float64, two waves per SIMD instead of three)
    constexpr int VW = EXACT_POLES ? 1 : V;
    int64_t base = beg;
    for (; base + 256 * VW <= end; base += 256 * VW)
        pop_group<T, LOSS, VW, false, SHARED_POSE, TS, EXACT_POLES, LENS_FREE>(x, y, z, uo, vo, base, end, recs, s_sum_wave, tc, f_scale, redo_lo, redo_hi);
    if constexpr (VW > 2)      // the rows left over by the wide groups, two at a time
        for (; base + 512 <= end; base += 512)
            pop_group<T, LOSS, 2, false, SHARED_POSE, TS, EXACT_POLES, LENS_FREE>(x, y, z, uo, vo, base, end, recs, s_sum_wave, tc, f_scale, redo_lo, redo_hi);
    for (; base < end; base += 256)
        pop_group<T, LOSS, 1, true, SHARED_POSE, TS, EXACT_POLES, LENS_FREE>(x, y, z, uo, vo, base, end, recs, s_sum_wave, tc, f_scale, redo_lo, redo_hi);
}

// LENS_FREE: `cands` holds the lens-free records (fold_pose_lens_free) the first walk runs on, `cands_general` the general ones
// (fold_pose) of the same candidates for the second walk; otherwise both are the general records.
template <typename T, int LOSS, typename Cfg = PopCfg<T>, bool SHARED_POSE = false, typename TS = T, bool LENS_FREE = false>
__global__ __launch_bounds__(256, Cfg::MINW) void popeval_kernel(
    const TS *__restrict__ x, const TS *__restrict__ y, const TS *__restrict__ z, const TS *__restrict__ uo,
    const TS *__restrict__ vo, int64_t n, const PoseRec<T> *__restrict__ cands, int P, T f_scale,
    double *__restrict__ partials, const PoseRec<T> *__restrict__ cands_general) {
    constexpr int TC = Cfg::TC;
    constexpr int V = Cfg::V;
    static_assert(TC <= 128, "the second walk's candidate mask is two 64-bit words");
    __shared__ PoseRec<T> s_c[TC];
    __shared__ double s_sum[4][TC];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;

    // stripe of this workgroup: multiples of 256 points so that only the last stripe is ragged
    const int64_t rows = (n + 255) / 256;
    const int64_t rows_per = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t beg = (int64_t)blockIdx.x * rows_per * 256;
    const int64_t end = (beg + rows_per * 256 < n) ? beg + rows_per * 256 : n;

    // gridDim.y > 1 (few candidate tiles, see enqueue_popeval): this workgroup takes every gridDim.y-th tile only
    for (int c0 = (int)blockIdx.y * TC; c0 < P; c0 += (int)gridDim.y * TC) {
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
        pop_walk_stripe<T, LOSS, V, SHARED_POSE, TS, false, LENS_FREE>(x, y, z, uo, vo, beg, end, s_c, s_sum[wave], tc, f_scale);
        // A sum that is not finite stays so (inf and NaN are sticky under +): looked for ONCE per wave, tile and stripe -- nothing
        // in the loop above pays for it.  The wave's rows are its own (lane t owns points t, t + 256, ...), so are its sums:
        // it clears the sums of the candidates concerned and walks its share again for THOSE candidates, row by row, with the
        // reference's arithmetic: a reciprocal per denominator (distort_group), the general records for a lens-free tile.
        // No barrier needed.  Where it runs: float64 (the parity mode) and lens-free tiles.  NOT for the general float32
        // variants: a population of wild float32 candidates overflows (inf - inf = NaN in a third of the candidates of a first
        // CMA-ES generation at sigma = 1, none of them in float64: tools/probe_cma_nonfinite.py), the walk cannot mend that, and
        // it doubled those generations' kernel time; float32 mode therefore keeps NaN at an exact pole (include/alproj_hip.h).
        if constexpr (POP_SECOND_WALK(T, LENS_FREE)) {
            __builtin_amdgcn_wave_barrier();       // lane 63 wrote the sums, every lane reads them: same wave, LDS in order
            const int l = tid & 63;
            const bool bad_lo = l < tc && (__builtin_bit_cast(unsigned long long, s_sum[wave][l < tc ? l : 0]) & 0x7ff0000000000000ull) == 0x7ff0000000000000ull;
            const bool bad_hi = l + 64 < tc && (__builtin_bit_cast(unsigned long long, s_sum[wave][l + 64 < tc ? l + 64 : 0]) & 0x7ff0000000000000ull) == 0x7ff0000000000000ull;
            const unsigned long long redo_lo = __builtin_amdgcn_ballot_w64(bad_lo), redo_hi = __builtin_amdgcn_ballot_w64(bad_hi);
            if ((redo_lo | redo_hi) != 0) {
                if (bad_lo) s_sum[wave][l] = 0.0;
                if (bad_hi) s_sum[wave][l + 64] = 0.0;
                __builtin_amdgcn_wave_barrier();
                // (a lens-free tile: the LDS holds the folded rows only, so the general records are read where they lie in HBM --
                // wave-uniform loads, slow and rare; SHARED_POSE's hoisted coordinates are not used by either walk then)
                if constexpr (LENS_FREE)
                    pop_walk_stripe<T, LOSS, V, false, TS, true>(x, y, z, uo, vo, beg, end, cands_general + c0, s_sum[wave], tc, f_scale, redo_lo, redo_hi);
                else
                    pop_walk_stripe<T, LOSS, V, SHARED_POSE, TS, true>(x, y, z, uo, vo, beg, end, s_c, s_sum[wave], tc, f_scale, redo_lo, redo_hi);
            }
        }
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

// Stand-alone loss of two (n x 2) float64 arrays, each either interleaved (b == NULL: a holds u0 v0 u1 v1 ...) or as its two
// columns (a = u[n], b = v[n]): one float64 partial per workgroup.  The order of the additions does not depend on the layout.
template <int LOSS, bool OBS_COLUMNS, bool PRJ_COLUMNS>
__global__ __launch_bounds__(256) void loss_uv_kernel(const double *__restrict__ obs_a, const double *__restrict__ obs_b,
                                                      const double *__restrict__ prj_a, const double *__restrict__ prj_b, int64_t n,
                                                      double f_scale, double *__restrict__ partials) {
    __shared__ double s[4];
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double2 o, q;
        if constexpr (OBS_COLUMNS) o = make_double2(obs_a[i], obs_b[i]);
        else o = reinterpret_cast<const double2 *>(obs_a)[i];
        if constexpr (PRJ_COLUMNS) q = make_double2(prj_a[i], prj_b[i]);
        else q = reinterpret_cast<const double2 *>(prj_a)[i];
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
