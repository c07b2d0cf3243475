/*
 * alproj_hip.h -- C ABI of libalproj_hip.so: the MI355X (gfx950) implementation of the
 * camera-projection hot path of 0kam/alproj (reference v1.1.1).
 *
 * The reference is pure Python and has no FFI layer; its boundary is its public Python
 * signatures.  Each entry point below names the reference function it replaces (paths
 * relative to the reference checkout).  The Python side in alproj_amd/ binds these with
 * ctypes and keeps the reference's signatures (INTEGRATION.md shows the stub a maintainer
 * of the reference would add).
 *
 * Conventions
 *   - every function returns 0 on success, a negative ALP_E* code on failure; the message
 *     for the calling thread's last failure is alp_last_error().  Numerical problems
 *     (points behind / at the camera) are NOT errors: they surface as inf/NaN exactly like
 *     the reference (src/alproj/optimize.py:146-149).
 *   - camera parameters travel as `const double[25]` in the fixed order ALP_PARAM_ORDER
 *     (keys documented at src/alproj/project.py:157-189).
 *   - host buffers belong to the caller and are only touched during the call; device
 *     memory is owned by opaque handles released by the matching *_destroy.
 *   - one handle is used from one thread at a time (the reference is single-threaded and
 *     non-reentrant: src/alproj/optimize.py:341-351).
 *   - there is NO CPU fallback: without a usable HIP device alp_init fails with
 *     ALP_ENODEVICE and every other call fails with ALP_ENOTINIT.
 */
#ifndef ALPROJ_HIP_H
#define ALPROJ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALP_ABI_VERSION 7

/* x,y,z,fov,pan,tilt,roll,a1,a2,k1..k6,p1,p2,s1..s4,w,h,cx,cy */
#define ALP_NPARAM 25
#define ALP_PARAM_ORDER "x,y,z,fov,pan,tilt,roll,a1,a2,k1,k2,k3,k4,k5,k6,p1,p2,s1,s2,s3,s4,w,h,cx,cy"

enum alp_error {
    ALP_OK = 0,
    ALP_EINVAL = -1,     /* bad argument (NULL pointer, negative size, unknown enum) */
    ALP_ENOTINIT = -2,   /* alp_init has not succeeded in this process */
    ALP_ENODEVICE = -3,  /* no HIP device / device index out of range */
    ALP_EHIP = -4,       /* a HIP runtime call failed (message has the hipError string) */
    ALP_ERCCL = -5,      /* an RCCL call failed */
    ALP_ESTATE = -6      /* handle is missing something the call needs (e.g. observed uv) */
};

enum alp_dtype {
    ALP_F32 = 0,
    ALP_F64 = 1,
    ALP_I32 = 2,
    ALP_I64 = 3,
    ALP_U8 = 4,
    ALP_U16 = 5
};

/* Loss kinds of the population evaluation. */
enum alp_loss {
    ALP_LOSS_MEAN_DIST = 0, /* "rmse" of the reference = mean Euclidean distance,
                               src/alproj/optimize.py:176-177 */
    ALP_LOSS_HUBER = 1      /* src/alproj/optimize.py:205-212 */
};

/* ---------------------------------------------------------------- library / device --- */

int alp_abi_version(void);
const char *alp_last_error(void);

/* Select HIP device `device` for this process, create the library stream and scratch
 * buffers.  Idempotent for the same device. */
int alp_init(int device);
int alp_shutdown(void);
int alp_device_count(int *count);
/* name[len] receives the gcnArchName ("gfx950:sramecc+:xnack-"); cu_count the CU number. */
int alp_device_info(char *name, int len, int *cu_count, int64_t *hbm_bytes);
/* PCI bus id ("0000:75:00.0") of the device this process was initialised on: what tells the ranks of a
 * multi-GPU job apart in a benchmark record.  len >= 16. */
int alp_device_pci_bus_id(char *id, int len);
/* 64-bit content digest of a HOST array, computed by `threads` host threads (0 = all cores); needs no device.
 * A change of any single 8-byte word always changes the digest.  The render wrappers use it to make sure a mesh
 * kept on the device still equals the caller's arrays (the reference uploads on every call,
 * src/alproj/project.py:213-215, so an in-place edit between two calls must be seen). */
int alp_host_hash64(const void *buf, int64_t bytes, int threads, uint64_t *digest);
/* (ABI 5) out[0] = min, out[1] = max of n float64 values (n >= 1) on host threads (0: up to 8), NaN for both if any value
 * is NaN -- numpy's `a.min()`, `a.max()` in one pass: `x.min(), x.max(), y.min(), y.max()` of to_geotiff (project.py:420-423)
 * took 14-30 ms of numpy on one core for the 11.7 M rows of the 100 M-vertex frame's table, a multiple of the device's work.
 * No device needed. */
int alp_host_minmax(const double *values, int64_t n, int threads, double out[2]);
/* (ABI 5) Bring the pages of a fresh host buffer into existence before the device -> host copy that fills it:
 * madvise(MADV_HUGEPAGE) + madvise(MADV_POPULATE_WRITE) from `threads` host threads (0: 4).  A copy into pages nobody has
 * touched runs at 8-13 GB/s on the bench host (one fault per 4 KB page inside the copy), into existing pages at 56 GB/s, and
 * populating 237 MB this way takes 2.7 ms.  Advice only: where the kernel refuses, nothing changes.  No device needed. */
int alp_host_prefault(void *buf, int64_t bytes, int threads);
/* Block until everything queued on the library stream is done. */
int alp_synchronize(void);

/* HIP-event timer slots on the library stream (what bench.py brackets kernels with).
 * slot in [0, 64).  (The package's render wrappers do not use them: a frame's device time comes from
 * alp_mesh_frame_ms.) */
int alp_event_record(int slot);
int alp_event_elapsed_ms(int slot_start, int slot_stop, float *ms); /* synchronises on stop */

/* Kernel-section timer.  The entry points that mix kernels with PCIe copies (alp_residuals_batch,
 * alp_rasterize_points, alp_mesh_from_rasters, alp_render_gather, alp_render_valid_count /
 * _fetch_valid) bracket their kernel sections with HIP events on the library stream while the
 * timer is on; alp_kernel_time_ms returns (and clears) the sum over the sections since the last
 * call and their number.  For bench.py's per-kernel roofline figures; off by default. */
int alp_kernel_timing(int enable);
int alp_kernel_time_ms(float *ms, int *sections);

/* Development switches this library was compiled with, comma separated; "" for a release build
 * (see the top of csrc/alp_raster.hip: several of them render wrong images by design). */
const char *alp_build_flags(void);

/* ---------------------------------------------------------------- multi-GPU (RCCL) ---- */
/* One process per GPU.  Rank 0 calls alp_comm_unique_id and ships the 128 bytes to the
 * other ranks by any means (the Python side: a localhost socket served by its own launcher, alproj_amd/launch.py, a
 * shared file, or any callable that broadcasts 128 bytes -- alproj_amd/dist.py); every rank then
 * calls alp_comm_init.  With a communicator present, alp_eval_population sums the
 * per-candidate partial losses and the vertex counts of all ranks with ONE
 * ncclAllReduce(sum, double, P+1) per call, on the library stream.
 * A failed ncclCommInitRank returns ALP_ERCCL and alp_last_error() then holds RCCL's error string, ncclGetLastError's
 * text, this rank's world position, HIP device and PCI bus id, and the environment RCCL / ROCr read. */
#define ALP_UNIQUE_ID_BYTES 128
int alp_comm_unique_id(char id[ALP_UNIQUE_ID_BYTES]);
int alp_comm_init(const char id[ALP_UNIQUE_ID_BYTES], int rank, int world_size);
int alp_comm_destroy(void);
int alp_comm_info(int *rank, int *world_size); /* 0,1 when no communicator */
/* Broadcast `bytes` bytes of the host buffer `buf` from rank `root` to every rank (ncclBroadcast on
 * the library stream); no-op without a communicator.  The optimiser shim uses it so that every
 * rank draws the SAME population: the reference constructs its sampler without a seed
 * (src/alproj/optimize.py:410-416), which is fine for one process and silently wrong for several
 * that must all-reduce the sums of identical candidates. */
int alp_comm_bcast(void *buf, int64_t bytes, int root);
/* All-gather of host buffers of different sizes: alp_comm_allgather_counts tells every rank the byte count of every
 * rank (counts[world_size]); alp_comm_allgatherv then fills `recv` (sum of the counts) with rank 0's `send`, rank
 * 1's, ... (one ncclBroadcast per rank in one group, on the library stream).  Without a communicator: a copy.
 * LsqOptimizer (src/alproj/optimize.py:442-539) solves ONE least-squares problem over all points: with the points
 * sharded over ranks, every rank gathers the residual vector and the Jacobian rows of all shards and runs the
 * identical scipy solve on them. */
int alp_comm_allgather_counts(int64_t bytes, int64_t *counts);
int alp_comm_allgatherv(const void *send, void *recv, const int64_t *counts);

/* ---------------------------------------------------------------- point sets ---------- */
/* Device-resident set of 3-D points (GCPs or DSM vertices) -- the `obj_points` DataFrame
 * of src/alproj/optimize.py:122-141 -- plus, optionally, observed image points
 * (`img_points`, src/alproj/optimize.py:173-174).
 *
 * xyz     n x 3 row-major host array (x, y, z = easting, northing, elevation), in_dtype
 *         ALP_F32 or ALP_F64, ABSOLUTE coordinates.
 * origin  local origin subtracted (in float64) before the coordinates are stored; choose
 *         it near the camera so that float32 storage does not lose the near field.
 * precision ALP_F32: coordinates, arithmetic and outputs float32 (20 B/vertex streamed);
 *           ALP_F64: everything float64 (parity mode, 40 B/vertex).
 */
typedef struct alp_points alp_points_t;

int alp_points_create(const void *xyz, int in_dtype, int64_t n, const double origin[3],
                      int precision, alp_points_t **out);
/* The same from the three COLUMNS as they lie (x[n], y[n], z[n] contiguous): a pandas DataFrame keeps its columns as rows of a
 * block, so `obj_points[["x", "y", "z"]]` of the reference's signature (optimize.py:139-141) reaches the device without the
 * host-side interleaving `DataFrame.to_numpy()` would do (96 ms for 10 M rows on one core, measured). */
int alp_points_create_columns(const void *x, const void *y, const void *z, int in_dtype, int64_t n,
                              const double origin[3], int precision, alp_points_t **out);
int alp_points_destroy(alp_points_t *pts);
int alp_points_count(const alp_points_t *pts, int64_t *n);
/* uv: n x 2 row-major host array of observed pixel coordinates (u, v). */
int alp_points_set_observed(alp_points_t *pts, const void *uv, int in_dtype);
/* ... or the two columns u[n], v[n] as they lie (img_points[["u", "v"]], optimize.py:173-174). */
int alp_points_set_observed_columns(alp_points_t *pts, const void *u, const void *v, int in_dtype);

/* Forward projection: replaces project(), src/alproj/optimize.py:122-155 (intrinsic_mat
 * :8-44, extrinsic_mat :46-96 and _distort :98-120 fused into one kernel).
 * Results stay on the device (planar u[], v[] in the set's precision) ... */
int alp_project(alp_points_t *pts, const double params[ALP_NPARAM]);
/* ... until fetched: u_out/v_out are host arrays of n elements of out_dtype (F32/F64).  out_dtype may differ from the set's
 * precision (the reference's float64 from a float32 set): the narrower type crosses PCIe -- widened by host threads while the
 * next chunk arrives, narrowed on the device -- and no temporary of the result's size is made. */
int alp_projected_fetch(alp_points_t *pts, void *u_out, void *v_out, int out_dtype);
/* Fetch a strided sample (indices first, first+stride, ...; count elements). */
int alp_projected_fetch_strided(alp_points_t *pts, int64_t first, int64_t stride,
                                int64_t count, double *u_out, double *v_out);

/* Residual vector (observed - projected), interleaved du0,dv0,du1,dv1,... (2n doubles):
 * replaces compute_residuals(), src/alproj/optimize.py:215-237.  Needs observed uv.
 * Float64 point sets: bit for bit `observed - alp_project()`, the identity the reference has by construction
 * (:233-236) -- the kernel runs alp_project's own arithmetic.  Float32 sets: same formula, one reciprocal per
 * denominator (+-inf at a pole of the lens model, like the reference), not bit-equal to alp_project's float32. */
int alp_residuals(alp_points_t *pts, const double params[ALP_NPARAM], double *out);
/* The same for B parameter vectors in one launch (cand: B x 25 row-major; out: B x 2n doubles,
 * row b = residual vector of pose b): the D+1 evaluations of a 2-point finite-difference
 * Jacobian for LsqOptimizer, src/alproj/optimize.py:461-463, :510-528. */
int alp_residuals_batch(alp_points_t *pts, const double *cand, int64_t B, double *out);

/* Population-wide reprojection error: replaces the inner loop of CMAOptimizer.optimize,
 * src/alproj/optimize.py:420-423, i.e. P calls of _proj_error (:347-356) = project +
 * rmse (:157-178) or huber_loss (:181-212).
 *
 * cand      P x 25 row-major candidate parameter vectors (already de-normalised).
 * loss_out  P doubles: mean over ALL vertices (of all ranks when a communicator exists).
 * argmin_out index of the smallest loss, first index on ties, NaN never wins unless all
 *           are NaN (then 0) -- the contract of solutions[0] after CMA.tell's stable sort,
 *           src/alproj/optimize.py:424-427.
 * Float32 point sets: when other losses lie within 5e-5 (relative) of the smallest, up to 16
 * of those candidates are evaluated again in float64 arithmetic on the stored points before
 * the index is returned (the north star's "argmin bit-exact"); loss_out then holds the
 * float64 re-evaluations for THOSE candidates and the float32-path losses for all others.
 * Poles of the rational lens model (src/alproj/optimize.py:112-116: a vertex for which EXACTLY one of
 * 1 + k4 r2 + k5 r4 + k6 r6 and 1 + a2 + k4 r2 + ... is zero): the reference's coordinate on that axis is +-inf, the
 * other finite, the candidate's loss +inf.  The kernel shares one reciprocal between the two denominators, which
 * turns the finite coordinate into NaN.  FLOAT64 point sets (the parity mode) mend it: a wave whose sum for a candidate
 * comes out infinite or NaN walks its share of the points again for that candidate with a reciprocal per denominator,
 * so the loss is +inf as in the reference (NaN only where the reference is NaN too).  FLOAT32 point sets keep the NaN:
 * the loss of such a candidate is NaN where the reference has +inf -- it ranks last either way (argmin_out skips NaN,
 * CMA.tell sorts NaN as +inf); a wild float32 population has overflow NaNs in a third of its candidates anyway, which no
 * second walk can mend.  tests/test_gpu_points.py constructs the case in both modes.
 * Lens-free populations (ALP_POP_LENS_FREE, below) take the second walk in either precision: it restores the NaN the
 * reference's k * inf produces at a vertex at the camera.
 * argmin_out == NULL: losses only, no confirmation pass (the reference uses the argmin of the LAST
 * generation only, src/alproj/optimize.py:427; every earlier generation needs the losses for
 * CMA.tell and nothing else).
 */
int alp_eval_population(alp_points_t *pts, const double *cand, int64_t P, int loss_kind,
                        double f_scale, double *loss_out, int64_t *argmin_out);
/* Same, but only enqueues the work on the library stream (H2D of the candidate records,
 * kernels, all-reduce, D2H into an internal pinned buffer).  alp_eval_population_wait
 * synchronises and delivers the results of the last enqueue. */
int alp_eval_population_enqueue(alp_points_t *pts, const double *cand, int64_t P,
                                int loss_kind, double f_scale);
int alp_eval_population_wait(alp_points_t *pts, double *loss_out, int64_t *argmin_out);
/* Device time of the last completed population evaluation of this handle, from HIP events on the
 * library stream: kernel_ms = the evaluation and reduction kernels, allreduce_ms = the
 * ncclAllReduce(sum, double, P+1) behind them (about 0 without a communicator).  What bench.py
 * reports as the collective's share of a CMA-ES generation (src/alproj/optimize.py:418-424 has no
 * counterpart: the reference is one process). */
int alp_eval_population_timing(alp_points_t *pts, float *kernel_ms, float *allreduce_ms);
/* (ABI 7) Which kernel variant the last population evaluation of this handle was given, and its launch shape:
 * info[0] = ALP_POP_GENERAL, ALP_POP_SHARED_POSE (every candidate has the same position / angles / fov: the reference's second
 * phase, example.py:75-78 -- the transform is hoisted out of the candidate loop) or ALP_POP_LENS_FREE (no candidate has a lens
 * coefficient other than a1, a2: the reference's first phase, example.py:51-54 -- the lens is folded into the pose rows on
 * the host and optimize.py:112-118 costs no arithmetic per point); info[1] = stripes of points, info[2] = columns of
 * candidate tiles of the launch grid.  The losses of one candidate agree between the variants to the tolerances of the
 * precision mode (float64: 1e-12 relative; float32: 2e-6), not bit for bit. */
enum alp_pop_variant { ALP_POP_GENERAL = 0, ALP_POP_SHARED_POSE = 1, ALP_POP_LENS_FREE = 2 };
int alp_eval_population_info(alp_points_t *pts, int64_t info[3]);

/* The candidate sampler of the CMA-ES loop on the device: replaces the `population_size` calls of
 * `optimizer.ask()` per generation, src/alproj/optimize.py:420-421 (third-party cmaes==0.12.0,
 * requirements.txt:14; bounds handling documented at optimize.py:381-384).  Candidate c of generation g:
 * x = mean + sigma * BD z with z ~ N(0, I_D) (BD = B diag(D) of the covariance C = B D^2 B^T, D x D
 * row-major); a draw outside [lower, upper] is re-drawn, the first feasible one of n_max_resampling tries
 * wins, otherwise draw number n_max_resampling is clipped to the box.  lower = upper = NULL: no box.
 * Counter-based (Philox4x32-10 keyed by seed; counter = try, candidate, generation): the same numbers on
 * every rank.  x_out: P x D row-major host array; tries_out (optional, P ints): index of the draw used
 * (n_max_resampling = clipped).  D <= 32. */
int alp_cma_sample(const double *mean, double sigma, const double *BD, const double *lower, const double *upper,
                   int D, int64_t P, int n_max_resampling, uint64_t seed, uint64_t generation, double *x_out,
                   int32_t *tries_out);

/* Loss of two host arrays of pixel coordinates (n x 2 row-major doubles each): replaces the
 * stand-alone rmse(), src/alproj/optimize.py:157-178 (loss_kind ALP_LOSS_MEAN_DIST) and
 * huber_loss(), :181-212 (ALP_LOSS_HUBER).  Float64 arithmetic on the device. */
int alp_loss_uv(const double *observed, const double *projected, int64_t n, int loss_kind,
                double f_scale, double *loss_out);
/* The same with either array given as its two COLUMNS as they lie in a table (obs_u[n], obs_v[n] / prj_u[n], prj_v[n]); an
 * array whose second pointer is NULL is taken as interleaved n x 2 like alp_loss_uv's.  `projected` is what project()
 * returned -- a DataFrame whose u, v are two separate runs -- so the reference's call rmse(img_points, projected) needs no
 * host-side interleaving (90 -> ~8 ms at 10 M rows).  Same sum order as alp_loss_uv: the same bits. */
int alp_loss_uv_columns(const double *obs_u, const double *obs_v, const double *prj_u, const double *prj_v, int64_t n,
                        int loss_kind, double f_scale, double *loss_out);

/* ---------------------------------------------------------------- mesh render --------- */
/* Device-resident triangle mesh: the vbo/cbo/ibo of src/alproj/project.py:213-215.
 *
 * vert   n_vert x 3, X, Z(up), Y order, relative to `offsets` (src/alproj/surface.py:189-190,
 *        :211); vert_dtype ALP_F32, or ALP_F64 -- what get_colored_surface returns: the cast of
 *        src/alproj/project.py:213 (astype("f4"), round to nearest even) then happens on the
 *        device during the chunked upload, no float32 copy is made on the host.
 * value  n_vert x 3 per-vertex values (colours, or the vertices themselves for reverse_proj,
 *        src/alproj/project.py:360), value_dtype ALP_F32 or ALP_F64 (cast of :214 likewise);
 *        NULL means value == vert.
 * ind    n_tri x 3 indices (ind_dtype ALP_I32 or ALP_I64), or NULL for the implicit
 *        regular grid of src/alproj/surface.py:194-201 with grid_h x grid_w vertices
 *        (n_vert == grid_h * grid_w; vertex id = row * grid_w + col).
 *        An index array that IS that grid, or that grid with the triangles of nodata
 *        vertices removed (src/alproj/surface.py:203-205, order kept), is recognised and
 *        rendered by the grid kernels (same result, no 12 B/triangle index reads); the full
 *        grid is recognised by host threads while the vertices are uploaded (or, without
 *        threads to spare, while it streams through the staging buffer) and is never stored;
 *        triangle ids reported by alp_render_fetch_visibility stay positions in `ind`.
 *        Indices outside [0, n_vert) are rejected (ALP_EINVAL; checked on the device).
 * (ABI 3: vert_dtype / value_dtype added; ABI 2 took float32 pointers only.)
 */
typedef struct alp_mesh alp_mesh_t;

int alp_mesh_create(const void *vert, int vert_dtype, const void *value, int value_dtype,
                    int64_t n_vert, const void *ind, int ind_dtype, int64_t n_tri,
                    int64_t grid_h, int64_t grid_w, alp_mesh_t **out);
int alp_mesh_destroy(alp_mesh_t *mesh);
/* How the mesh is held: info[0] = 1 when it is rendered as the implicit regular grid (given as
 * one, or an index array recognised as one -- possibly with a derived vertex mask), 0 when
 * through its index array; info[1], info[2] = grid rows, columns (0 for index meshes);
 * info[3] = number of triangles the kernels enumerate. */
int alp_mesh_info(alp_mesh_t *mesh, int64_t info[4]);

/* Replace (or, with NULL, drop) the stored per-vertex values of a resident mesh: sim_image's
 * colours for a mesh that reverse_proj created without any, src/alproj/project.py:214.  The
 * vertices, the index array and the visibility cache (below) are untouched. */
int alp_mesh_set_value(alp_mesh_t *mesh, const void *value, int value_dtype);

/* What the following renders interpolate: the stored per-vertex values (sim_image,
 * src/alproj/project.py:321) or the vertices themselves (reverse_proj, project.py:360) -- one
 * resident mesh serves both calls of the reference's pipeline (example.py:33-36). */
enum alp_value_source { ALP_VALUE_STORED = 0, ALP_VALUE_VERTICES = 1 };
int alp_mesh_set_value_source(alp_mesh_t *mesh, int source);

/* Per-vertex nodata mask (n_vert bytes, 0 = nodata; NULL removes the mask): triangles touching
 * a masked vertex are not drawn -- what get_colored_surface does by filtering its index array,
 * src/alproj/surface.py:203-205.  Triangle ids stay those of the unfiltered mesh. */
int alp_mesh_set_valid(alp_mesh_t *mesh, const uint8_t *valid);

/* Mesh construction of get_colored_surface() after its raster I/O, src/alproj/surface.py:173-212,
 * on the device: implicit-grid mesh with colours and nodata mask from
 *   dsm      rows x cols elevations (ALP_F32 or ALP_F64), nodata already filled (:171);
 *            clamped to [0, z_max] (:175-176, z_max = the `dsm_max_height` of :169)
 *   transform  the six affine coefficients a, b, c, d, e, f of the merged rasters:
 *            x = col * a + c, y = row * e + f (:179-180)
 *   aerial   3 x rows x cols band-planar colours (ALP_U8, ALP_U16 or ALP_F32), divided by
 *            color_div (0 = leave as they are) and clipped to [0, 1] (_normalize_aerial, :26-66;
 *            the caller picks the divisor from the source dtype / color_max)
 *   nodata   rows x cols bytes, nonzero = DSM nodata (:110-117), or NULL
 * Vertices are float32 (x, z, y) minus offsets_out = their float64 minimum (:211-212), vertex id
 * = row * cols + col, triangles (a, a+cols, a+cols+1), (a, a+cols+1, a+1) (:194-201). */
int alp_mesh_from_rasters(const void *dsm, int dsm_dtype, int64_t rows, int64_t cols,
                          const double transform[6], double z_max, const void *aerial,
                          int aerial_dtype, double color_div, const uint8_t *nodata,
                          double offsets_out[3], alp_mesh_t **out);

/* Copy the resident mesh back (any pointer may be NULL): vert, value n_vert x 3 float32, valid
 * n_vert bytes.  For inspection and parity tests. */
int alp_mesh_fetch(alp_mesh_t *mesh, float *vert, float *value, uint8_t *valid);

/* Depth-buffered render + lens-distortion remap: replaces persp_proj(),
 * src/alproj/project.py:145-294 (OpenGL draw :210-290, distort :111-143).
 *
 * params        camera parameters, ABSOLUTE camera position.
 * offsets       the 3 offsets (X,Z,Y order) subtracted from the camera position
 *               (src/alproj/project.py:204-207), or NULL.
 * min_distance  <= 0 disables the near-field mask (src/alproj/project.py:247, :264).
 * out           h x w x 3 float32 host image, row 0 = top (after the flipud of :281).
 */
int alp_render(alp_mesh_t *mesh, const double params[ALP_NPARAM], const double *offsets,
               double min_distance, float *out);
/* Same but leaves the image on the device (for timing); fetch with alp_render_fetch.
 *
 * Visibility cache: the reference renders the same mesh twice at one pose -- sim_image, then
 * reverse_proj (example.py:28,31; :57,59; :97,103), each a full GL draw
 * (src/alproj/project.py:276).  Here a render whose view (camera position and angles, fov,
 * w, h, offsets) equals the previous frame's on the same mesh and mask reuses that frame's
 * visibility buffer and runs only the resolve stage; value source, lens coefficients and
 * min_distance may differ.  Bit-identical to a full frame (the raster passes are
 * deterministic).  alp_mesh_frame_counts reports how many frames took each way:
 * counts[0] full, counts[1] resolve only. */
int alp_render_enqueue(alp_mesh_t *mesh, const double params[ALP_NPARAM],
                       const double *offsets, double min_distance);
int alp_render_fetch(alp_mesh_t *mesh, float *out);
int alp_mesh_frame_counts(alp_mesh_t *mesh, int64_t counts[2]);
/* Device time of the launches of the last alp_render_enqueue on this mesh (HIP events owned by the mesh, on the
 * library stream; waits for the frame). */
int alp_mesh_frame_ms(alp_mesh_t *mesh, float *ms);
/* Release the work areas the mesh keeps between calls of alp_render_rasterize_plan / alp_render_rasterize (the
 * compacted points and the 5 GB-class accumulator area of a 100 M-vertex frame); the mesh and its frame stay. */
int alp_mesh_trim(alp_mesh_t *mesh);
/* The last frame as h x w x 3 uint8: (image * scale) cast like numpy's astype(uint8)
 * (truncation toward zero, wrap), channels reversed when reverse_channels != 0 -- the tail of
 * sim_image(), src/alproj/project.py:322-324 (scale 255, RGB -> BGR), on the device, so that
 * 1 B instead of 4 B per channel cross PCIe. */
int alp_render_fetch_u8(alp_mesh_t *mesh, float scale, int reverse_channels, uint8_t *out);
/* The 64-bit visibility buffer of the last render, h x w, OpenGL window orientation (row 0 =
 * bottom), before the distortion remap: 0 = background, else (float32 bits of 1/depth) << 32 |
 * (0xFFFFFFFF - index of the winning triangle).  For inspection and parity tests. */
int alp_render_fetch_visibility(alp_mesh_t *mesh, uint64_t *out);

/* Post-processing of reverse_proj(), src/alproj/project.py:361-373, for a render whose
 * values are the vertices themselves (value == NULL): the pixels whose first channel (offset-
 * relative x) is > 0 (:369), in row-major order.  alp_render_valid_count returns their number
 * M; alp_render_fetch_valid then writes idx_out[M] (linear pixel index v * w + u) and
 * xyz_out[M][3] = (x, y, z) = channels (0, 2, 1) (:361) plus offsets[0], offsets[2],
 * offsets[1] (:370-373; NULL = no offsets), float64. */
/* Install a coordinate image produced elsewhere (h x w x 3 float32, row 0 = top: what persp_proj(vert,
 * vert, ...) returns, src/alproj/project.py:360) as the mesh's current frame, so that the
 * post-processing below runs on it as it would after alp_render_enqueue.  Its visibility buffer
 * is empty. */
int alp_render_load(alp_mesh_t *mesh, const float *image, int64_t h, int64_t w);
int alp_render_valid_count(alp_mesh_t *mesh, int64_t *count);
int alp_render_fetch_valid(alp_mesh_t *mesh, const double *offsets, uint32_t *idx_out, double *xyz_out);
/* The same survivors as three contiguous columns x_out[M], y_out[M], z_out[M] -- the layout a DataFrame keeps its
 * float64 columns in, so that the table of reverse_proj() is assembled without a transposing copy. */
int alp_render_fetch_valid_planes(alp_mesh_t *mesh, const double *offsets, uint32_t *idx_out, double *x_out,
                                  double *y_out, double *z_out);

/* The whole table of reverse_proj(), src/alproj/project.py:361-373, for the survivors of alp_render_valid_count (M rows, in
 * pixel order): index_out[M] = the table's labels (linear pixel index v * w + u: the reference filters a RangeIndex-ed frame),
 * u_out / v_out[M] = pixel column / row as int16 (:366), block_out = (3 + channels) contiguous float64 rows of M: x, y, z as
 * alp_render_fetch_valid_planes defines them, then the caller's image array[h][w][channels] (ALP_U8 / _U16 / _F32 / _F64, :364)
 * at each surviving pixel, cast to float64 -- the DataFrame's float64 block as pandas lays it out.  The array goes up
 * once (63 MB for a 5616 x 3744 photograph); the host no longer gathers, casts and divides 11.7 M rows on one core. */
int alp_render_fetch_valid_table(alp_mesh_t *mesh, const double *offsets, const void *array, int array_dtype,
                                 int64_t channels, int64_t *index_out, int16_t *u_out, int16_t *v_out, double *block_out);

/* set_gcp(), src/alproj/gcp.py:644-648, without the reverse_proj table: for n pixels (u[i], v[i])
 * of the last render (values = the vertices themselves) write xyz_out[i] = (x, y, z) as
 * alp_render_fetch_valid defines them, or NaN where the pixel lies outside the image or does
 * not see the surface (the rows the reference's left join + dropna removes). */
int alp_render_gather(alp_mesh_t *mesh, const int32_t *u, const int32_t *v, int64_t n,
                      const double *offsets, double *xyz_out);

/* filter_gcp_distance(), src/alproj/gcp.py:711-724: keep[i] = 1 for the rows of xyz[n][3] (x, y, z as alp_render_gather
 * writes them) without a NaN coordinate whose distance from camera[3] (params x, y, z) is >= min_distance and
 * <= max_distance, 0 otherwise.  A NaN bound is an absent one (the reference's None); the distance is formed as numpy
 * forms it (sqrt(dx**2 + dy**2 + dz**2), float64, no contraction), so the mask is the reference's.  Validation as
 * gcp.py:699-703 (negative minimum, maximum below minimum: ALP_EINVAL). */
int alp_distance_mask(const double *xyz, int64_t n, const double camera[3], double min_distance,
                      double max_distance, uint8_t *keep);

/* Compute part of to_geotiff(), src/alproj/project.py:376-503 (the GeoTIFF file itself is
 * written by the caller): n points (x, y, values[n][nb]) -> uint8 raster out[nb][height][width].
 * Pixel of a point: col = int((x - x_min) / resolution), row = int((y_max - y) / resolution),
 * both clipped (:435-436); per band and pixel the aggregate `agg` of the points that fall in it,
 * NaN values skipped (:450-459); then `sweeps` passes in which every empty pixel takes the
 * NaN-aware aggregate of its 3x3 neighbourhood (:462-479; sweeps = ceil(max_dist / resolution),
 * 0 = no interpolation); empty pixels -> nodata, others clipped to [0, 255] and truncated
 * (:483-485). */
enum alp_agg { ALP_AGG_MEAN = 0, ALP_AGG_MAX = 1, ALP_AGG_MIN = 2, ALP_AGG_MEDIAN = 3 };
/* The same computation fed from the RESIDENT coordinate image of the last render (values = the
 * vertices themselves), i.e. reverse_proj() -> to_geotiff() back to back, example.py:103-106,
 * without the multi-million-row table in between:
 *   alp_render_rasterize_plan  selects the pixels that see the surface (x > 0,
 *       src/alproj/project.py:369), keeps their x = channel 0 + offsets[0], y = channel 2 +
 *       offsets[2] (:361, :370-373) on the device and returns their number and bounds[4] =
 *       x_min, y_min, x_max, y_max (:420-421) -- from which the caller derives width / height
 *       exactly as :422-425 does;
 *   alp_render_rasterize  takes the caller's image `array` (h x w x channels of the frame's size;
 *       ALP_U8, ALP_U16, ALP_F32 or ALP_F64 -- the table's float64 channel columns, :364-367),
 *       the channel index of each of the nb output bands, and rasterises like
 *       alp_rasterize_points.  The plan belongs to the frame it was made for. */
int alp_render_rasterize_plan(alp_mesh_t *mesh, const double *offsets, int64_t *n_valid, double bounds[4]);
int alp_render_rasterize(alp_mesh_t *mesh, const void *array, int array_dtype, int64_t channels,
                         const int32_t *band_channel, int64_t nb, double x_min, double y_max,
                         double resolution, int64_t width, int64_t height, int agg, int sweeps,
                         int nodata, uint8_t *out);
int alp_rasterize_points(const double *x, const double *y, const double *values, int64_t n, int64_t nb,
                         double x_min, double y_max, double resolution, int64_t width, int64_t height,
                         int agg, int sweeps, int nodata, uint8_t *out);
/* The float32 raster (nb, height, width) itself, after the sweeps and before the byte conversion (the reference's
 * `raster_data`, src/alproj/project.py:448-479; NaN = empty): what the parity tests compare with pandas bit for bit. */
int alp_rasterize_points_f32(const double *x, const double *y, const double *values, int64_t n, int64_t nb,
                             double x_min, double y_max, double resolution, int64_t width, int64_t height,
                             int agg, int sweeps, float *out);
/* The same with the band values as nb separate columns (columns[b][i]: each n contiguous doubles -- how a DataFrame keeps
 * them, so that to_geotiff(df) hands its columns over without the transposed copy df[bands].to_numpy() makes, ~100 ms for
 * 16 M rows x 3 bands); byte-valued columns (a photograph's bands, at most four) are packed into the sort's payload as they lie,
 * any others interleaved on the device. */
int alp_rasterize_columns(const double *x, const double *y, const double *const *columns, int64_t n, int64_t nb,
                          double x_min, double y_max, double resolution, int64_t width, int64_t height,
                          int agg, int sweeps, int nodata, uint8_t *out);

/* Image-space distortion remap alone: replaces distort(), src/alproj/project.py:111-143.
 * img/out: h x w x c float32 host images; coeffs: a1,a2,k1..k6,p1,p2,s1..s4. */
int alp_distort_image(const float *img, int64_t h, int64_t w, int64_t c,
                      const double coeffs[14], float *out);
/* The float32 source map of that remap alone: map_x, map_y (h x w each) = what distort() hands to
 * cv2.remap, src/alproj/project.py:128-140 (_distort of the pixel grid with 1/a1, 1/a2 and every other
 * coefficient negated, cast to float32). */
int alp_distort_map(int64_t h, int64_t w, const double coeffs[14], float *map_x, float *map_y);

#ifdef __cplusplus
}
#endif
#endif /* ALPROJ_HIP_H */
