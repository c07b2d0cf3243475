// Internal declarations shared by the translation units of libalproj_hip.so.
// Not part of the ABI (that is include/alproj_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include "host/alp_host.h"      // errors (set_error / fail / ALP_REQUIRE), pose record + fold_pose, the HIP-free host helpers

namespace alp {

#define ALP_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return ::alp::fail(ALP_EHIP, "%s failed: %s (%s:%d)", #expr,                  \
                               hipGetErrorString(e__), __FILE__, __LINE__);               \
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

}  // namespace alp
