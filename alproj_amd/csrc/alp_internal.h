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

// ------------------------------------------------------------------ kernel-section timer
// Entry points that mix kernels with PCIe copies (alp_residuals_batch, alp_rasterize_points, alp_mesh_from_rasters,
// alp_render_gather, the reverse_proj compaction) bracket their KERNEL sections with these; while the timer is
// on (alp_kernel_timing(1): bench.py, profiling) each section is a pair of HIP events on the library stream and
// alp_kernel_time_ms() returns their sum.  Off (the default) they do nothing.
void ktime_begin();
void ktime_end();
struct KTimeScope {
    KTimeScope() { ktime_begin(); }
    ~KTimeScope() { ktime_end(); }
};

// Shared-scratch invariant.  `Context::scratch` is ONE grow-only device buffer handed out by scratch_reserve() to
// whichever entry point runs: alp_comm_bcast, alp_residuals*, alp_loss_uv, alp_render_gather / fetch_visibility /
// fetch_u8, the typed uploads of alp_mesh_create / alp_mesh_set_value, alp_distort_*.  It is safe ONLY because
// (a) a handle is used from one thread at a time and the library has one stream, and (b) every user either
// synchronises the stream before it returns or leaves nothing in flight that reads the scratch (the typed uploads
// end with hipStreamSynchronize).  An entry point that returns with scratch reads still queued (an *_enqueue
// variant) must own its buffer instead -- alp_eval_population_enqueue does (pinned staging + per-handle partials).
// scratch_reserve() itself synchronises before it frees a too-small buffer.

// Development switches compiled into the render translation unit ("" = none: a release build); alp_build_flags()
const char *raster_dev_flags();

// alp_shutdown: the pinned staging buffers and events of the converting fetch (alp_points.hip) belong to the device context
void points_release_staging();

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
