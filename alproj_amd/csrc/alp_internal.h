// Internal declarations shared by the translation units of libalproj_hip.so.
// Not part of the ABI (that is include/alproj_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "alproj_hip.h"

namespace alp {

// ------------------------------------------------------------------ errors
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define ALP_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return ::alp::fail(ALP_EHIP, "%s failed: %s (%s:%d)", #expr,                  \
                               hipGetErrorString(e__), __FILE__, __LINE__);               \
    } while (0)

#define ALP_REQUIRE(cond, msg)                                                            \
    do {                                                                                  \
        if (!(cond)) return ::alp::fail(ALP_EINVAL, "%s: %s", __func__, msg);             \
    } while (0)

// ------------------------------------------------------------------ global context
struct Context {
    bool ready = false;
    int device = -1;
    int cu_count = 0;
    hipStream_t stream = nullptr;
    hipEvent_t events[64] = {};
    // grow-only device scratch shared by the entry points that stage data for ONE call and
    // synchronise before they return (alp_residuals*, alp_loss_uv, alp_render_gather, ...): no
    // hipMalloc / hipFree on the per-call path once it has reached its working size
    void *scratch = nullptr;
    size_t scratch_cap = 0;
    // RCCL
    void *comm = nullptr;   // ncclComm_t
    int rank = 0;
    int world = 1;
};
Context &ctx();
int require_init();
// *out = device scratch of at least `bytes` bytes (valid until the next scratch_reserve / alp_shutdown)
int scratch_reserve(size_t bytes, void **out);

// all-reduce (sum, double) of `count` doubles in place on the library stream; no-op
// when no communicator exists.
int comm_allreduce_sum_f64(double *dev_buf, int64_t count);

// ------------------------------------------------------------------ pose record
// One camera pose folded, in float64 on the host, into the 32 numbers the kernels use.
// Layout (index):
//   0..3   row X' : x1 = (X'.[q;1]) / (Z.[q;1])   normalised, centred image x (see fold_pose)
//   4..7   row Y' : y1 = (Y'.[q;1]) / (Z.[q;1])
//   8..11  row Z  : depth along the optical axis
//   12..17 k1..k6
//   18,19  1+a1, 1+a2
//   20,21  2*p1, 2*p2
//   22..25 s1..s4
//   26,27  c0, c1 : float32-rounded image centre (w-1)/2, (h-1)/2
//   28,29  -c0, -c1
//   30,31  unused (0)
constexpr int POSE_WORDS = 32;

template <typename T>
struct alignas(16) PoseRec {
    T v[POSE_WORDS];
};

// params: the 25 ABI parameters; origin: local origin of the point set (absolute coords).
void fold_pose(const double params[ALP_NPARAM], const double origin[3], double rec[POSE_WORDS]);

template <typename T>
inline void fold_pose_t(const double params[ALP_NPARAM], const double origin[3], PoseRec<T> *out) {
    double r[POSE_WORDS];
    fold_pose(params, origin, r);
    for (int i = 0; i < POSE_WORDS; ++i) out->v[i] = (T)r[i];
}

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

}  // namespace alp
