// libalproj_hip.so -- device-resident point sets and the three kernels that run on them:
//   project_kernel   forward projection of every point with ONE pose (HBM-bound stream)
//   popeval_kernel   P candidate poses x every point -> per-candidate loss sums
//                    (VALU-bound; candidates staged in LDS, wave64 DPP reductions)
//   residual_batch_kernel  observed - projected for B poses, interleaved (least-squares path)
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

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "alp_point_kernels.h"


using namespace alp;

// ------------------------------------------------------------------ the handle
struct alp_points {
    int64_t n = 0;
    int64_t n_pad = 0;
    int precision = ALP_F32;
    double origin[3] = {0, 0, 0};
    // the planes live in at most three allocations (a hipMalloc / hipFree pair of this size costs ~1 ms: seven of them were a
    // third of what compute_residuals spent at 10 M points): coordinates at creation, observed pixels at alp_points_set_observed*,
    // projected pixels at the first alp_project
    void *slab_xyz = nullptr, *slab_obs = nullptr, *slab_uv = nullptr;
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
    int64_t last_info[3] = {0, 0, 0};     // alp_eval_population_info: variant, stripes, tile columns of the last launch
    int64_t pending_P = 0;
    int pending_loss = 0;
    double pending_f_scale = 0;
    std::vector<double> cand_copy;   // the P x 25 parameter vectors of the pending call (argmin confirmation)
    // argmin confirmation (float32 sets): float64 records, partial sums and sums of up to CONFIRM_MAX candidates
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // last population evaluation: before the kernels, after them, after the all-reduce
    bool timed = false;
    void *conf_dev = nullptr;
    int conf_nblk = 0;
    double *conf_host = nullptr;   // pinned, CONFIRM_MAX + 1
    size_t esize() const { return precision == ALP_F64 ? 8 : 4; }
};

// argmin confirmation of float32 point sets: CONFIRM_MAX, CONFIRM_GAP and the selection itself live in host/alp_host.h
using alp::host::CONFIRM_MAX;

namespace {

// k planes of n_pad elements in ONE zeroed allocation (n_pad is a multiple of 1024 elements: every plane starts 4 KB-aligned)
int alloc_planes(void **slab, int k, int64_t n_pad, size_t es, void **planes[]) {
    ALP_HIP(hipMalloc(slab, (size_t)k * n_pad * es));
    ALP_HIP(hipMemsetAsync(*slab, 0, (size_t)k * n_pad * es, ctx().stream));
    for (int i = 0; i < k; ++i) *planes[i] = (char *)*slab + (size_t)i * n_pad * es;
    return ALP_OK;
}

template <typename TIn, typename T, int C>
int upload_columns(const TIn *host, int64_t n, const double o[3], T *p0, T *p1, T *p2) {
    // points per staging chunk: 16 M (192 MB of float32 triples).  Measured on the MI355X box (tools/h2d_rate.hip):
    // one pageable hipMemcpy sustains 56 GB/s at this size, 48 MB chunks with a host sync each 36 GB/s.
    const int64_t CH = 16 << 20;
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
        e = hipGetLastError();                    // the staging buffer is reused in stream order: no host sync per chunk
        if (e != hipSuccess) { rc = fail(ALP_EHIP, "upload kernel: %s", hipGetErrorString(e)); break; }
    }
    if (hipStreamSynchronize(ctx().stream) != hipSuccess && !rc) rc = fail(ALP_EHIP, "upload: stream failed");
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

// One host COLUMN (n contiguous values of TIn) -> one device plane of T, minus its origin component in float64: the columns of a
// table as they lie (a pandas block is columns x rows: a DataFrame's x, y, z are three contiguous runs), no host-side
// interleaving.  Same chunking as upload_columns.
template <typename TIn, typename T>
__global__ __launch_bounds__(256) void column_to_plane_kernel(const TIn *__restrict__ src, int64_t count, int64_t dst_off, double o,
                                                              T *__restrict__ plane) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
        plane[dst_off + i] = (T)((double)src[i] - o);
}

template <typename TIn, typename T>
int upload_planes_t(const void *const *cols, int ncols, int64_t n, const double *o, void *const *planes) {
    const int64_t CH = 48 << 20;                      // values per staging chunk (192 MB of float32)
    const int64_t ch = n < CH ? (n > 0 ? n : 1) : CH;
    TIn *stage = nullptr;
    ALP_HIP(hipMalloc((void **)&stage, (size_t)ch * sizeof(TIn)));
    int rc = ALP_OK;
    for (int c = 0; c < ncols && !rc; ++c)
        for (int64_t off = 0; off < n; off += ch) {
            const int64_t cnt = (n - off < ch) ? (n - off) : ch;
            hipError_t e = hipMemcpyAsync(stage, (const TIn *)cols[c] + off, (size_t)cnt * sizeof(TIn), hipMemcpyHostToDevice, ctx().stream);
            if (e != hipSuccess) { rc = fail(ALP_EHIP, "H2D upload: %s", hipGetErrorString(e)); break; }
            const int grid = (int)((cnt + 255) / 256 < 4096 ? (cnt + 255) / 256 : 4096);
            hipLaunchKernelGGL((column_to_plane_kernel<TIn, T>), dim3(grid), dim3(256), 0, ctx().stream, stage, cnt, off, o[c], (T *)planes[c]);
            e = hipGetLastError();                    // the staging buffer is reused in stream order
            if (e != hipSuccess) { rc = fail(ALP_EHIP, "upload kernel: %s", hipGetErrorString(e)); break; }
        }
    if (hipStreamSynchronize(ctx().stream) != hipSuccess && !rc) rc = fail(ALP_EHIP, "upload: stream failed");
    hipFree(stage);
    return rc;
}

int upload_planes(const void *const *cols, int ncols, int in_dtype, int64_t n, const double *o, int precision, void *const *planes) {
    if (in_dtype == ALP_F64 && precision == ALP_F64) return upload_planes_t<double, double>(cols, ncols, n, o, planes);
    if (in_dtype == ALP_F64 && precision == ALP_F32) return upload_planes_t<double, float>(cols, ncols, n, o, planes);
    if (in_dtype == ALP_F32 && precision == ALP_F64) return upload_planes_t<float, double>(cols, ncols, n, o, planes);
    if (in_dtype == ALP_F32 && precision == ALP_F32) return upload_planes_t<float, float>(cols, ncols, n, o, planes);
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
    const unsigned grid = (unsigned)((nvec + 255) / 256);        // one 16-byte vector per lane
    hipLaunchKernelGGL((project_kernel<T>), dim3(grid), dim3(256), 0, ctx().stream,
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
        // two records per candidate: the general one and, behind all of those, the lens-free one (enqueue_popeval)
        ALP_HIP(hipMalloc(&p->cand_dev, (size_t)cap * rec * 2));
        ALP_HIP(hipHostMalloc(&p->cand_host, (size_t)cap * rec * 2, hipHostMallocDefault));
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
    // one pinned staging buffer per handle: a second enqueue would rewrite it under the first one's
    // asynchronous copy (and lose its result)
    if (p->pending_P > 0)
        return fail(ALP_ESTATE, "alp_eval_population_enqueue: the previous enqueue has not been waited for");
    if (int rc = ensure_pop_scratch(p, P, 0)) return rc;
    PoseRec<T> *h = (PoseRec<T> *)p->cand_host;
    // the kernel centres the observations once per point: every candidate of a call must share
    // the image size (the reference never optimises w, h: optimize.py:240-247)
    for (int64_t i = 1; i < P; ++i)
        if (cand[i * ALP_NPARAM + 21] != cand[21] || cand[i * ALP_NPARAM + 22] != cand[22])
            return fail(ALP_EINVAL, "alp_eval_population: candidates %lld and 0 differ in w or h", (long long)i);
    // lens-free populations (no candidate has a lens coefficient other than a1, a2: the reference's first phase, example.py:51-54;
    // BASELINE config 3) take the kernel variant that runs on rows with the lens folded in: its records follow the general ones
    bool lens_free = !getenv("ALP_POP_NO_LENS_FREE");
    for (int64_t i = 0; i < P && lens_free; ++i) lens_free = pose_is_lens_free(cand + i * ALP_NPARAM);
    for (int64_t i = 0; i < P; ++i) {
        double g[POSE_WORDS], lf[POSE_WORDS];
        fold_pose(cand + i * ALP_NPARAM, p->origin, g);
        for (int k = 0; k < POSE_WORDS; ++k) h[i].v[k] = (T)g[k];
        if (lens_free) {
            lens_free_from_general(g, lf);
            for (int k = 0; k < POSE_WORDS; ++k) h[p->cand_cap + i].v[k] = (T)lf[k];
        }
    }
    ALP_HIP(hipMemcpyAsync(p->cand_dev, h, (size_t)P * sizeof(PoseRec<T>), hipMemcpyHostToDevice, ctx().stream));
    if (lens_free)
        ALP_HIP(hipMemcpyAsync((PoseRec<T> *)p->cand_dev + p->cand_cap, h + p->cand_cap, (size_t)P * sizeof(PoseRec<T>), hipMemcpyHostToDevice,
                               ctx().stream));
    // distortion-only populations (the reference's second phase, example.py:75-78) share the
    // folded 3x4 matrix: its 12 words are identical in every record, and the kernel then
    // computes the normalised coordinates once per point instead of once per candidate
    bool shared_pose = P > 1 && !lens_free;
    for (int64_t i = 1; i < P && shared_pose; ++i)
        shared_pose = memcmp(h[i].v, h[0].v, 12 * sizeof(T)) == 0;
    using Kernel = void (*)(const T *, const T *, const T *, const T *, const T *, int64_t, const PoseRec<T> *, int, T,
                            double *, const PoseRec<T> *);
    const int which = (loss_kind == ALP_LOSS_HUBER ? 3 : 0) + (lens_free ? 2 : (shared_pose ? 1 : 0));
    const Kernel kernels[6] = {popeval_kernel<T, ALP_LOSS_MEAN_DIST, PopCfg<T>, false>,
                               popeval_kernel<T, ALP_LOSS_MEAN_DIST, PopCfg<T>, true>,
                               popeval_kernel<T, ALP_LOSS_MEAN_DIST, PopCfgLF<T>, false, T, true>,
                               popeval_kernel<T, ALP_LOSS_HUBER, PopCfg<T>, false>,
                               popeval_kernel<T, ALP_LOSS_HUBER, PopCfg<T>, true>,
                               popeval_kernel<T, ALP_LOSS_HUBER, PopCfgLF<T>, false, T, true>};
    // one workgroup per stripe of ~24 rows of 256 points (four groups of V = 6), between 4 and 64
    // workgroups per CU: a stripe is re-read once per tile of 128 candidates and a short one stays
    // in cache between those passes.  Measured, 100 M x 2048 float32: 4 workgroups per CU 244 ms,
    // 8: 229, 16: 224, 32: 221, 64: 219, 128: 219; 10 M x 256: 8 per CU (stripes of 19 rows) 3.31
    // ms, 16: 3.46, 32: 3.73.  float64 (three workgroups resident per CU): 24 per CU (round 5: 527 ms against 541 with 4).
    int nblk = ctx().cu_count * (sizeof(T) == 8 ? 24 : 4);
    int ytiles = 1;
    const int64_t rows = (p->n + 255) / 256;
    const int VV = lens_free ? PopCfgLF<T>::V : PopCfg<T>::V;       // rows of a full group
    if (sizeof(T) == 4) {
        const int64_t want = (rows + 4 * VV - 1) / (4 * VV);            // ~four full groups per stripe
        const int64_t lo = (int64_t)ctx().cu_count * 4, hi = (int64_t)ctx().cu_count * 64;
        // whole rounds of the 4 workgroups a CU holds at once while the grid is only a few rounds deep
        const int64_t rounded = (want + lo - 1) / lo * lo;
        nblk = (int)(want < lo ? lo : (want > hi ? hi : (want < 4 * lo ? rounded : want)));
        // Two candidate tiles or more: the grid is stripes x tiles -- a workgroup runs ONE tile of 128 candidates over a stripe of
        // whole groups of V rows.  Round 3 introduced it for populations of a few tiles whose one-column grid was only a few
        // rounds deep (10 M x 256: 1954 stripes of 20 rows = 1.9 rounds of the 1024 resident workgroups, 2 rows of every 20 in
        // the narrow groups; tools/sweep_popeval_grid.py, ms for P = 256 / 384 / 512 at 10 M points: one column of 2048 stripes
        // 3.15 / 4.64 / 6.14; stripes of 18 rows x tiles 2.90 / 4.33 / 5.61).  Round 6 measured it at every other shape as well
        // (profiles/r06_popeval_grid_sweep.txt, one column -> stripes x tiles, general | lens-free variant): 10 M x 1024 12.2 ->
        // 11.0 | 5.81 -> 4.91 ms; 10 M x 2048 24.6 -> 21.7 | 11.5 -> 9.64; 30 M x 1024 34.9 -> 32.5 | 15.8 -> 14.4; 100 M x 2048
        // 219.5 -> 215.2 | 95.7 -> 93.4 -- never slower, so it is the rule.  Stripes of k groups, k grown with the point count
        // (about two stripes per resident slot and tile column for small sets, up to 16 groups = ~25 000 points for large ones:
        // the timings are flat from 4 to 16 groups and the partial-sum buffer shrinks with the stripe count).
        // The sums depend on the shape in the last bits only (1e-9 relative between shapes: other group boundaries).
        const int tiles = (int)((P + POP_TC - 1) / POP_TC);
        if (tiles >= 2) {
            int64_t k = (int64_t)((double)rows / ((double)VV * 2.12 * (double)lo) + 0.5);      // groups of V rows per stripe
            if (k < 1) k = 1;
            if (k > 16) k = 16;
            const int64_t stripes = (rows + VV * k - 1) / (VV * k);
            if (stripes * tiles >= 4 * lo) {
                nblk = (int)stripes;
                ytiles = tiles;
            }
        }
    }
    if (sizeof(T) == 8) {
        // the per-stripe partial sums (nblk x P doubles, read once per generation by reduce_partials_kernel) stay below 128 MB:
        // 24 stripes per CU at P = 2048 are 100 MB (kept: 527 ms against 533 with 8 per CU); a population of 8192 gets 2048 stripes
        const int64_t cap = ((int64_t)128 << 20) / (8 * P);
        const int64_t lo = (int64_t)ctx().cu_count * 3;                  // one round of the three resident workgroups per CU
        if (nblk > cap) nblk = (int)(cap > lo ? cap : lo);
    }
    // tuning hook: "stripes,ytiles".  float32: any value gives the same losses (float64 additions of float32 group sums of this
    // magnitude are exact).  float64: the stripe count sets the ORDER of the float64 additions, so the last bits of the losses move
    // with it -- a development switch, not a setting
    if (const char *e = getenv("ALP_POP_GRID")) {
        const int tiles = (int)((P + PopCfg<T>::TC - 1) / PopCfg<T>::TC);
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && b >= 1 && b <= tiles) { nblk = a; ytiles = b; }
    }
    if (rows < nblk) nblk = (int)(rows > 0 ? rows : 1);
    if (int rc = ensure_pop_scratch(p, P, nblk)) return rc;
    if (!p->ev[0]) {            // all three or none: a partial failure must not leave ev[1] / ev[2] NULL for good
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        for (auto &e : ev)
            if (hipEventCreate(&e) != hipSuccess) {
                for (auto &d : ev)
                    if (d) hipEventDestroy(d);
                return fail(ALP_EHIP, "hipEventCreate failed");
            }
        for (int k = 0; k < 3; ++k) p->ev[k] = ev[k];
    }
    p->last_info[0] = lens_free ? ALP_POP_LENS_FREE : (shared_pose ? ALP_POP_SHARED_POSE : ALP_POP_GENERAL);
    p->last_info[1] = nblk;
    p->last_info[2] = ytiles;
    ALP_HIP(hipEventRecord(p->ev[0], ctx().stream));
    const PoseRec<T> *recs_general = (const PoseRec<T> *)p->cand_dev;
    hipLaunchKernelGGL(kernels[which], dim3(nblk, ytiles), dim3(256), 0, ctx().stream, (const T *)p->x, (const T *)p->y,
                       (const T *)p->z, (const T *)p->uo, (const T *)p->vo, p->n, lens_free ? recs_general + p->cand_cap : recs_general,
                       (int)P, (T)f_scale, p->partials, recs_general);
    ALP_HIP(hipGetLastError());
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((P + 31) / 32)), dim3(256), 0, ctx().stream,
                       p->partials, nblk, (int)P, (double)p->n, p->sums_dev);
    ALP_HIP(hipGetLastError());
    ALP_HIP(hipEventRecord(p->ev[1], ctx().stream));
    if (int rc = comm_allreduce_sum_f64(p->sums_dev, P + 1)) return rc;
    ALP_HIP(hipEventRecord(p->ev[2], ctx().stream));
    p->timed = true;
    ALP_HIP(hipMemcpyAsync(p->sums_host, p->sums_dev, (size_t)(P + 1) * sizeof(double),
                           hipMemcpyDeviceToHost, ctx().stream));
    p->pending_P = P;
    p->pending_loss = loss_kind;
    p->pending_f_scale = f_scale;
    if (sizeof(T) == 4) p->cand_copy.assign(cand, cand + P * ALP_NPARAM);
    return ALP_OK;
}

// float64 re-evaluation of K <= CONFIRM_MAX candidates of a float32 point set (same stored
// coordinates, double arithmetic, same fixed-order reduction and the same all-reduce):
// sums_out[0..K) = loss sums, sums_out[K] = vertex count.  Synchronous.
int confirm_losses(alp_points *p, const double *cand, const int64_t *which, int K, int loss_kind, double f_scale,
                   double *sums_out) {
    const int nblk_want = ctx().cu_count * 4;
    const int64_t rows = (p->n + 255) / 256;
    const int nblk = (int)(rows < nblk_want ? (rows > 0 ? rows : 1) : nblk_want);
    const size_t rec_bytes = (size_t)CONFIRM_MAX * sizeof(PoseRec<double>);
    if (!p->conf_dev || p->conf_nblk < nblk) {
        if (p->conf_dev) hipFree(p->conf_dev);
        p->conf_dev = nullptr;
        ALP_HIP(hipMalloc(&p->conf_dev, rec_bytes + (size_t)(nblk + 2) * CONFIRM_MAX * sizeof(double)));
        p->conf_nblk = nblk;
    }
    if (!p->conf_host) ALP_HIP(hipHostMalloc((void **)&p->conf_host, (CONFIRM_MAX + 1) * sizeof(double), hipHostMallocDefault));
    PoseRec<double> recs[CONFIRM_MAX];
    for (int k = 0; k < K; ++k) fold_pose_t<double>(cand + which[k] * ALP_NPARAM, p->origin, &recs[k]);
    PoseRec<double> *recs_dev = (PoseRec<double> *)p->conf_dev;
    double *partials = (double *)((char *)p->conf_dev + rec_bytes);
    double *sums_dev = partials + (size_t)nblk * CONFIRM_MAX;
    ALP_HIP(hipMemcpyAsync(recs_dev, recs, (size_t)K * sizeof(PoseRec<double>), hipMemcpyHostToDevice, ctx().stream));
    const float *x = (const float *)p->x, *y = (const float *)p->y, *z = (const float *)p->z;
    const float *uo = (const float *)p->uo, *vo = (const float *)p->vo;
    if (loss_kind == ALP_LOSS_HUBER)
        hipLaunchKernelGGL((popeval_kernel<double, ALP_LOSS_HUBER, PopCfg<double>, false, float>), dim3(nblk), dim3(256), 0,
                           ctx().stream, x, y, z, uo, vo, p->n, recs_dev, K, f_scale, partials, recs_dev);
    else
        hipLaunchKernelGGL((popeval_kernel<double, ALP_LOSS_MEAN_DIST, PopCfg<double>, false, float>), dim3(nblk), dim3(256),
                           0, ctx().stream, x, y, z, uo, vo, p->n, recs_dev, K, f_scale, partials, recs_dev);
    ALP_HIP(hipGetLastError());
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, ctx().stream, partials, nblk, K, (double)p->n, sums_dev);
    ALP_HIP(hipGetLastError());
    if (int rc = comm_allreduce_sum_f64(sums_dev, K + 1)) return rc;
    ALP_HIP(hipMemcpyAsync(p->conf_host, sums_dev, (size_t)(K + 1) * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
    ALP_HIP(hipStreamSynchronize(ctx().stream));      // recs (stack) must outlive the H2D copy too
    memcpy(sums_out, p->conf_host, (size_t)(K + 1) * sizeof(double));
    return ALP_OK;
}

// device staging of the residual entry points: at most RES_CHUNK_BYTES of output per launch
constexpr size_t RES_CHUNK_BYTES = (size_t)256 << 20;

template <typename T>
int residuals_impl(alp_points *p, const double *cand, int64_t B, double *out) {
    // B pose records (kernel argument for B == 1 would save the copy; one path keeps it simple)
    const size_t rec_bytes = round_up((int64_t)(B * sizeof(PoseRec<T>)), 256);
    int64_t chunk = (int64_t)(RES_CHUNK_BYTES / ((size_t)B * sizeof(double2)));
    chunk = chunk / 1024 * 1024;
    if (chunk < 1024) chunk = 1024;
    if (chunk > p->n) chunk = p->n;
    char *dev = nullptr;
    if (int rc = scratch_reserve(rec_bytes + (size_t)B * chunk * sizeof(double2), (void **)&dev)) return rc;
    PoseRec<T> *poses_dev = (PoseRec<T> *)dev;
    double2 *res_dev = (double2 *)(dev + rec_bytes);
    std::vector<PoseRec<T>> poses((size_t)B);
    for (int64_t b = 0; b < B; ++b) fold_pose_t<T>(cand + b * ALP_NPARAM, p->origin, &poses[b]);
    hipStream_t st = ctx().stream;
    ALP_HIP(hipMemcpyAsync(poses_dev, poses.data(), (size_t)B * sizeof(PoseRec<T>), hipMemcpyHostToDevice, st));
    for (int64_t off = 0; off < p->n; off += chunk) {
        const int64_t cnt = p->n - off < chunk ? p->n - off : chunk;
        ktime_begin();
        hipLaunchKernelGGL(residual_batch_kernel<T>, dim3(stream_grid(cnt)), dim3(256), 0, st, (const T *)p->x + off,
                           (const T *)p->y + off, (const T *)p->z + off, (const T *)p->uo + off, (const T *)p->vo + off,
                           res_dev, cnt, poses_dev, (int)B);
        ktime_end();
        ALP_HIP(hipGetLastError());
        // row b of the chunk -> out[b][off .. off + cnt); a chunk that holds whole rows is one contiguous run (a pitched copy of
        // the same bytes took 2-3 x as long)
        if (cnt == p->n)
            ALP_HIP(hipMemcpyAsync(out, res_dev, (size_t)B * cnt * sizeof(double2), hipMemcpyDeviceToHost, st));
        else
            ALP_HIP(hipMemcpy2DAsync(out + 2 * off, (size_t)p->n * sizeof(double2), res_dev, (size_t)cnt * sizeof(double2),
                                     (size_t)cnt * sizeof(double2), (size_t)B, hipMemcpyDeviceToHost, st));
        ALP_HIP(hipStreamSynchronize(st));       // the staging buffer is reused; poses must outlive their copy
    }
    return ALP_OK;
}

}  // namespace

// ---- fetch with a change of element type (float32 set -> float64 arrays, the reference's type; or the reverse)
// The planes are converted ON THE HOST while they arrive: a chunk crosses PCIe in its stored type (a float32 set moves 4 bytes
// per value, not 8) into one of two pinned staging buffers, and host threads widen / narrow the previous chunk into the
// caller's array meanwhile -- the copy engine and the host cores overlap, nothing of the size of the result is allocated.
// The reverse (a float64 set fetched as float32) is cast on the device into the scratch area first, so that again the narrow
// type crosses PCIe.  ALP_FETCH_CONVERT=host|device forces either way (tests, tools/probe_fetch.py).
namespace {
constexpr int64_t FETCH_CHUNK = (int64_t)8 << 20;         // values per chunk: 32 MB of float32, 64 MB of float64
void *g_fetch_stage[2] = {nullptr, nullptr};               // pinned, FETCH_CHUNK * 8 bytes each; lives until the process ends
hipEvent_t g_fetch_ev[2] = {nullptr, nullptr};

using alp::host::convert_threads;      // the widening / narrowing workers: host/alp_host.h (HIP-free, run under the sanitizers)
using alp::host::fetch_threads;

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_plane_kernel(const S *__restrict__ src, D *__restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (D)src[i];
}

template <typename S, typename D>
int fetch_converted_t(alp_points *p, D *u_out, D *v_out) {
    hipStream_t st = ctx().stream;
    const S *planes[2] = {(const S *)p->u, (const S *)p->v};
    D *outs[2] = {u_out, v_out};
    // the NARROWER type crosses PCIe: widening happens on the host (measured, 100 M points: 17 ms against 30 ms through the device
    // cast and 14 ms for the plain float32 fetch), narrowing on the device (15 ms against 33 ms; the plain float64 fetch: 28 ms)
    const char *mode = getenv("ALP_FETCH_CONVERT");
    const bool on_device = mode ? !strcmp(mode, "device") : sizeof(D) < sizeof(S);
    if (on_device) {
        D *tmp = nullptr;
        if (int rc = scratch_reserve((size_t)FETCH_CHUNK * sizeof(D), (void **)&tmp)) return rc;
        for (int pl = 0; pl < 2; ++pl)
            for (int64_t off = 0; off < p->n; off += FETCH_CHUNK) {
                const int64_t cnt = std::min(FETCH_CHUNK, p->n - off);
                hipLaunchKernelGGL((cast_plane_kernel<S, D>), dim3(stream_grid(cnt)), dim3(256), 0, st, planes[pl] + off, tmp, (long long)cnt);
                ALP_HIP(hipMemcpyAsync(outs[pl] + off, tmp, (size_t)cnt * sizeof(D), hipMemcpyDeviceToHost, st));   // in stream order: the next cast waits
            }
        ALP_HIP(hipStreamSynchronize(st));
        return ALP_OK;
    }
    for (auto &b : g_fetch_stage)
        if (!b) ALP_HIP(hipHostMalloc(&b, (size_t)FETCH_CHUNK * 8, hipHostMallocDefault));
    const int T = fetch_threads();
    for (auto &e : g_fetch_ev)
        if (!e) ALP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t *ev = g_fetch_ev;
    struct Job { int pl; int64_t off, cnt; };
    std::vector<Job> jobs;
    for (int pl = 0; pl < 2; ++pl)
        for (int64_t off = 0; off < p->n; off += FETCH_CHUNK) jobs.push_back({pl, off, std::min(FETCH_CHUNK, p->n - off)});
    for (size_t k = 0; k <= jobs.size(); ++k) {
        if (k < jobs.size()) {       // chunk k is on its way into stage[k & 1] (chunk k - 2, its last user, was converted in round k - 1)
            const Job &j = jobs[k];
            ALP_HIP(hipMemcpyAsync(g_fetch_stage[k & 1], planes[j.pl] + j.off, (size_t)j.cnt * sizeof(S), hipMemcpyDeviceToHost, st));
            ALP_HIP(hipEventRecord(ev[k & 1], st));
        }
        if (k > 0) {                 // ... while the host cores convert chunk k - 1
            const Job &j = jobs[k - 1];
            ALP_HIP(hipEventSynchronize(ev[(k - 1) & 1]));
            convert_threads((const S *)g_fetch_stage[(k - 1) & 1], outs[j.pl] + j.off, j.cnt, T);
        }
    }
    return ALP_OK;
}

}  // namespace
namespace alp {
void points_release_staging() {
    for (auto &b : g_fetch_stage) {
        if (b) hipHostFree(b);
        b = nullptr;
    }
    for (auto &e : g_fetch_ev) {
        if (e) hipEventDestroy(e);
        e = nullptr;
    }
}
}  // namespace alp
namespace {
int fetch_converted(alp_points *p, void *u_out, void *v_out, int out_dtype) {
    return out_dtype == ALP_F64 ? fetch_converted_t<float, double>(p, (double *)u_out, (double *)v_out)
                                : fetch_converted_t<double, float>(p, (float *)u_out, (float *)v_out);
}
}  // namespace

extern "C" {

// xyz: n x 3 row-major (cols == NULL), or cols[0..2]: the three columns as they lie
static int points_create(const void *xyz, const void *const *cols, int in_dtype, int64_t n, const double origin[3], int precision,
                         alp_points_t **out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(out, "out is NULL");
    *out = nullptr;
    ALP_REQUIRE(n >= 0, "n is negative");
    ALP_REQUIRE(n == 0 || xyz || (cols && cols[0] && cols[1] && cols[2]), "coordinates are NULL");
    ALP_REQUIRE(origin, "origin is NULL");
    ALP_REQUIRE(precision == ALP_F32 || precision == ALP_F64, "precision must be ALP_F32 or ALP_F64");
    ALP_REQUIRE(in_dtype == ALP_F32 || in_dtype == ALP_F64, "in_dtype must be ALP_F32 or ALP_F64");
    alp_points *p = new alp_points();
    p->n = n;
    p->n_pad = round_up(n > 0 ? n : 1, 1024);
    p->precision = precision;
    memcpy(p->origin, origin, sizeof(p->origin));
    void **xyz_planes[3] = {&p->x, &p->y, &p->z};
    int rc = alloc_planes(&p->slab_xyz, 3, p->n_pad, p->esize(), xyz_planes);
    if (!rc && n > 0) {
        void *const planes[3] = {p->x, p->y, p->z};
        rc = cols ? upload_planes(cols, 3, in_dtype, n, origin, precision, planes)
                  : upload_any<3>(xyz, in_dtype, n, origin, precision, p->x, p->y, p->z);
    }
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

int alp_points_create(const void *xyz, int in_dtype, int64_t n, const double origin[3], int precision,
                      alp_points_t **out) {
    return points_create(xyz, nullptr, in_dtype, n, origin, precision, out);
}

int alp_points_create_columns(const void *x, const void *y, const void *z, int in_dtype, int64_t n, const double origin[3],
                              int precision, alp_points_t **out) {
    const void *const cols[3] = {x, y, z};
    return points_create(nullptr, cols, in_dtype, n, origin, precision, out);
}

int alp_points_destroy(alp_points_t *p) {
    if (!p) return ALP_OK;
    if (ctx().ready) hipStreamSynchronize(ctx().stream);
    for (void *q : {p->slab_xyz, p->slab_obs, p->slab_uv, p->cand_dev, (void *)p->partials, (void *)p->sums_dev})
        if (q) hipFree(q);
    if (p->cand_host) hipHostFree(p->cand_host);
    if (p->sums_host) hipHostFree(p->sums_host);
    if (p->conf_dev) hipFree(p->conf_dev);
    if (p->conf_host) hipHostFree(p->conf_host);
    for (auto &e : p->ev)
        if (e) hipEventDestroy(e);
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
        void **obs_planes[2] = {&p->uo, &p->vo};
        if (int rc = alloc_planes(&p->slab_obs, 2, p->n_pad, p->esize(), obs_planes)) return rc;
    }
    const double zero[3] = {0, 0, 0};
    if (p->n > 0)
        if (int rc = upload_any<2>(uv, in_dtype, p->n, zero, p->precision, p->uo, p->vo, nullptr)) return rc;
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    return ALP_OK;
}

int alp_points_set_observed_columns(alp_points_t *p, const void *u, const void *v, int in_dtype) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p, "points handle is NULL");
    ALP_REQUIRE(p->n == 0 || (u && v), "u or v is NULL");
    ALP_REQUIRE(in_dtype == ALP_F32 || in_dtype == ALP_F64, "in_dtype must be ALP_F32 or ALP_F64");
    if (!p->uo) {
        void **obs_planes[2] = {&p->uo, &p->vo};
        if (int rc = alloc_planes(&p->slab_obs, 2, p->n_pad, p->esize(), obs_planes)) return rc;
    }
    const double zero[2] = {0, 0};
    const void *const cols[2] = {u, v};
    void *const planes[2] = {p->uo, p->vo};
    if (p->n > 0)
        if (int rc = upload_planes(cols, 2, in_dtype, p->n, zero, p->precision, planes)) return rc;
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    return ALP_OK;
}

int alp_project(alp_points_t *p, const double params[ALP_NPARAM]) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p && params, "NULL argument");
    if (!p->u) {
        void **uv_planes[2] = {&p->u, &p->v};
        if (int rc = alloc_planes(&p->slab_uv, 2, p->n_pad, p->esize(), uv_planes)) return rc;
    }
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
    return fetch_converted(p, u_out, v_out, out_dtype);
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
    if (int rc = scratch_reserve((size_t)count * 2 * sizeof(double), (void **)&tmp)) return rc;
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
    if (e != hipSuccess) return fail(ALP_EHIP, "strided fetch: %s", hipGetErrorString(e));
    return ALP_OK;
}

int alp_residuals(alp_points_t *p, const double params[ALP_NPARAM], double *out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p && params, "NULL argument");
    if (!p->uo) return fail(ALP_ESTATE, "alp_residuals: observed uv not set");
    if (p->n == 0) return ALP_OK;
    ALP_REQUIRE(out, "out is NULL");
    return p->precision == ALP_F64 ? residuals_impl<double>(p, params, 1, out) : residuals_impl<float>(p, params, 1, out);
}

int alp_residuals_batch(alp_points_t *p, const double *cand, int64_t B, double *out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p && cand, "NULL argument");
    ALP_REQUIRE(B >= 1 && B <= 4096, "B out of range");
    if (!p->uo) return fail(ALP_ESTATE, "alp_residuals_batch: observed uv not set");
    if (p->n == 0) return ALP_OK;
    ALP_REQUIRE(out, "out is NULL");
    return p->precision == ALP_F64 ? residuals_impl<double>(p, cand, B, out) : residuals_impl<float>(p, cand, B, out);
}

// obs_b / prj_b NULL: that array is interleaved (n x 2 row-major); else a = the u column, b = the v column
static int loss_uv_any(const double *obs_a, const double *obs_b, const double *prj_a, const double *prj_b, int64_t n, int loss_kind,
                       double f_scale, double *loss_out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(loss_out, "loss_out is NULL");
    ALP_REQUIRE(n >= 0, "n is negative");
    ALP_REQUIRE(loss_kind == ALP_LOSS_MEAN_DIST || loss_kind == ALP_LOSS_HUBER, "unknown loss_kind");
    if (n == 0) {                      // np.mean of an empty array
        *loss_out = NAN;
        return ALP_OK;
    }
    ALP_REQUIRE(obs_a && prj_a, "NULL input");
    const int grid = stream_grid(n);
    char *dev = nullptr;
    const size_t col = (size_t)n * sizeof(double);
    if (int rc = scratch_reserve(4 * col + (size_t)(grid + 2) * sizeof(double), (void **)&dev)) return rc;
    double *d_obs = (double *)dev, *d_prj = (double *)(dev + 2 * col);
    double *partials = (double *)(dev + 4 * col);
    hipStream_t st = ctx().stream;
    hipError_t e = hipMemcpyAsync(d_obs, obs_a, obs_b ? col : 2 * col, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && obs_b) e = hipMemcpyAsync(d_obs + n, obs_b, col, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_prj, prj_a, prj_b ? col : 2 * col, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && prj_b) e = hipMemcpyAsync(d_prj + n, prj_b, col, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        using Kernel = void (*)(const double *, const double *, const double *, const double *, int64_t, double, double *);
        static const Kernel kernels[2][2][2] = {
            {{loss_uv_kernel<ALP_LOSS_MEAN_DIST, false, false>, loss_uv_kernel<ALP_LOSS_MEAN_DIST, false, true>},
             {loss_uv_kernel<ALP_LOSS_MEAN_DIST, true, false>, loss_uv_kernel<ALP_LOSS_MEAN_DIST, true, true>}},
            {{loss_uv_kernel<ALP_LOSS_HUBER, false, false>, loss_uv_kernel<ALP_LOSS_HUBER, false, true>},
             {loss_uv_kernel<ALP_LOSS_HUBER, true, false>, loss_uv_kernel<ALP_LOSS_HUBER, true, true>}}};
        hipLaunchKernelGGL(kernels[loss_kind == ALP_LOSS_HUBER][obs_b != nullptr][prj_b != nullptr], dim3(grid), dim3(256), 0, st,
                           d_obs, d_obs + n, d_prj, d_prj + n, n, f_scale, partials);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, partials, grid, 1, (double)n, partials + grid);
        e = hipGetLastError();
    }
    double res[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(res, partials + grid, 2 * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(ALP_EHIP, "alp_loss_uv: %s", hipGetErrorString(e));
    *loss_out = res[0] / (double)n;
    return ALP_OK;
}

int alp_loss_uv(const double *observed, const double *projected, int64_t n, int loss_kind, double f_scale,
                double *loss_out) {
    return loss_uv_any(observed, nullptr, projected, nullptr, n, loss_kind, f_scale, loss_out);
}

int alp_loss_uv_columns(const double *obs_u, const double *obs_v, const double *prj_u, const double *prj_v, int64_t n,
                        int loss_kind, double f_scale, double *loss_out) {
    return loss_uv_any(obs_u, obs_v, prj_u, prj_v, n, loss_kind, f_scale, loss_out);
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
    const int64_t P = p->pending_P;
    p->pending_P = 0;                  // before the sync: a failed wait must not lock the handle (every later enqueue refused)
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    const double n_total = p->sums_host[P];
    std::vector<double> local;
    double *loss = loss_out;
    if (!loss) {
        local.resize((size_t)P);
        loss = local.data();
    }
    // (the selection below is host/alp_host.h: NaN never wins, first index on ties, the CONFIRM_MAX smallest of a band)
    double best_v = 0;
    int64_t best = host::losses_and_argmin(p->sums_host, P, n_total, loss, &best_v);
    // (a caller that passes no argmin_out wants the losses only: no confirmation -- CMAOptimizer needs the argmin of its LAST
    // generation alone, optimize.py:427, and the confirmation costs a float64 pass over every point)
    if (argmin_out && best >= 0 && p->precision == ALP_F32 && P > 1 && best_v < INFINITY) {
        // candidates whose float32 loss lies within CONFIRM_GAP of the smallest one: if there is
        // more than one, float32 cannot order them -- evaluate (up to CONFIRM_MAX of) them again in
        // float64 arithmetic and take the argmin of those; identical on every rank (the sums are
        // all-reduced, so every rank sees the same band and joins the same second all-reduce)
        int64_t which[CONFIRM_MAX];
        int K = 0;
        if (host::confirm_band(loss, P, best_v, which, &K) > 1) {
            double sums[CONFIRM_MAX + 1];
            if (int rc = confirm_losses(p, p->cand_copy.data(), which, K, p->pending_loss, p->pending_f_scale, sums)) return rc;
            best = host::merge_confirmed(loss, which, K, sums);
        }
    }
    if (argmin_out) *argmin_out = best < 0 ? 0 : best;
    return ALP_OK;
}

int alp_eval_population_timing(alp_points_t *p, float *kernel_ms, float *allreduce_ms) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(p, "points handle is NULL");
    if (!p->timed) return fail(ALP_ESTATE, "alp_eval_population_timing: no population evaluation yet");
    if (p->pending_P > 0) return fail(ALP_ESTATE, "alp_eval_population_timing: wait for the pending evaluation first");
    float a = 0, b = 0;
    ALP_HIP(hipEventElapsedTime(&a, p->ev[0], p->ev[1]));
    ALP_HIP(hipEventElapsedTime(&b, p->ev[1], p->ev[2]));
    if (kernel_ms) *kernel_ms = a;
    if (allreduce_ms) *allreduce_ms = b;
    return ALP_OK;
}

int alp_eval_population_info(alp_points_t *p, int64_t info[3]) {
    ALP_REQUIRE(p && info, "NULL argument");
    if (!p->timed) return fail(ALP_ESTATE, "alp_eval_population_info: no population evaluation yet");
    for (int k = 0; k < 3; ++k) info[k] = p->last_info[k];
    return ALP_OK;
}

int alp_eval_population(alp_points_t *p, const double *cand, int64_t P, int loss_kind, double f_scale,
                        double *loss_out, int64_t *argmin_out) {
    if (int rc = alp_eval_population_enqueue(p, cand, P, loss_kind, f_scale)) return rc;
    return alp_eval_population_wait(p, loss_out, argmin_out);
}

}  // extern "C"
