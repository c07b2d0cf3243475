// libalproj_hip.so -- depth-buffered mesh render + lens-distortion remap: the OpenGL
// replacement for persp_proj(), src/alproj/project.py:145-294 (call stack in SURVEY.md 3.2).
//
// What the reference makes OpenGL do, and where it is restated here:
//   project.py:203-207  camera position minus offsets (X,Z,Y order)        -> make_view()
//   project.py:13-54    projection_mat, used WITHOUT cx,cy (:257) and untransposed (:262):
//                       clip = (fx vx, fy vy, -1, vz) => near plane at view depth 1, no far
//                       plane, principal point ignored (quirks Q10, Q11)      -> to_window()
//   project.py:56-109   modelview_mat R = Rz(roll) Rx(tilt) Ry(360-pan)      -> make_view()
//   project.py:211-212  depth test GL_LESS + back-face culling (CCW front)   -> raster_tri()
//   project.py:217-253  varyings value, |view_pos|; min_dist mask            -> shade()
//   project.py:269-281  clear to 0, one indexed TRIANGLES draw, readback, flipud
//   project.py:111-143  distort(): inverted-coefficient source map, nearest gather, 0 border
//                                                                         -> remap_source()
//
// Pipeline (all on the library stream, no host round trip; DESIGN.md section 5 has the launch table):
//   1. clear the 64-bit visibility buffer (one word per pixel: float32 1/vz << 32 | ~triangle id) and the
//      frame's queue / list counters behind it
//   2. coverage, one of
//      regular-grid meshes (no index array, or an index array recognised at mesh creation as the grid or
//      as the grid minus the triangles of masked vertices):
//        tile_plan_kernel        tiles of 64x16 cells: frustum culling, NEAR / FAR lists
//        raster_grid_kernel      first round, NEAR tiles: one workgroup per tile, vertices transformed /
//                                projected / snapped once into LDS, one lane per cell = two triangles;
//                                fragments of small cells through an LDS depth patch where the tile's
//                                footprint fits one, larger cells and triangles parked in device queues
//        raster_parked_kernel    the parked cells (a wave per cell) and triangles (a wave per triangle)
//        hiz_build / hiz_top / tile_occlusion_kernel   depth pyramid, occlusion test of the FAR tiles
//        raster_grid_kernel, raster_parked_kernel      second round: the surviving FAR tiles
//      any other index array:
//        raster_kernel           one thread per triangle, three gathered vertices
//      all of them use emit_small(): 32-bit bounding-box rejection and back-face test, then an inline walk of
//      the bounding box with exact integer edge functions (64-bit atomicMax per covered pixel centre) for
//      triangles under 64 px; near-plane crossings and larger triangles go to raster_general_kernel, which
//      splits them into 64x64-pixel work items
//   3. raster_large_kernel: one wave per work item, one lane per pixel column
//   4. resolve_kernel: one thread per OUTPUT pixel: distortion source map (float64), fetch the
//      winning triangle, perspective-correct interpolation by ray/triangle intersection in view
//      space, min_distance mask, write h x w x 3 float32 (row 0 = top)
// The arithmetic that decides coverage and visibility is specified step by step in DESIGN.md
// section 5 and compiled with -ffp-contract=off so that it is reproducible bit for bit.
//
// Pinned by a real OpenGL: the reference's own persp_proj, run unmodified on Mesa llvmpipe in the
// build container, shows the same triangle and value on every pixel where a conformant GL has no freedom
// (tests/golden/g15_gl_render.npz, tests/test_gpu_gl.py).  The rules OpenGL leaves to the implementation
// (sub-pixel snapping, tie-break on shared edges, depth-buffer precision) are fixed here watertight and
// deterministic; how often they make a pixel differ from llvmpipe's is measured there too (0-2 pixels per frame).
#include "alp_raster_internal.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <thread>
#include <type_traits>
#include <vector>

// ---- development switches ---------------------------------------------------------------------------------
// Timing / census / stage-skipping builds used while the kernels were tuned (tools/build_variant.sh; DESIGN.md
// section 5 quotes their results).  Several produce WRONG IMAGES by design, so none of them can get into a
// library by accident: each needs -DALP_DEV next to it, the library reports what it was built with through
// alp_build_flags(), and tests/test_abi_symbols.py requires the shipped one to report nothing.  The development
// environment overrides (ALP_NEAR_PX, ALP_GRID_LANES, ALP_PATCH_NEAR / _FAR) are read by ALP_DEV builds only.
// ALP_NO_GRID_DETECT, ALP_QUEUE_CAP, ALP_NO_VIS_CACHE, ALP_NO_TILE_CULL and ALP_NO_OCCLUSION stay: they select
// between paths that produce the same image and are how the tests reach the index kernels, the queue growth,
// the full-frame path and the exact path without its culling.
#if defined(ALP_WG_TIMING) || defined(ALP_RASTER_STATS) || defined(VIS_PLAIN_STORE) || defined(VIS_NEVER) || defined(PARK_NOATOMIC) || \
    defined(PARKED_SKIP_CELLS) || defined(PARKED_SKIP_COOP) || defined(PARKED_SKIP_COOP4) || defined(GRID_STOP_AFTER) ||               \
    defined(GRID_NO_XCD_SWIZZLE) || defined(PT_SKIP_CELLS) || defined(PT_SKIP_SMALL) || defined(PT_SKIP_LARGE) || defined(PARKED_TILES_LAB) || \
    defined(INDEX_LDS_LAB)
#define ALP_DEV_SWITCHES 1
#ifndef ALP_DEV
#error "development switch given without -DALP_DEV: this would build a library that renders wrong images"
#endif
#endif
#if defined(INLINE_LOG2) || defined(FAST_MAX) || defined(COOP_MIN_W) || defined(COOP_MIN_PIX) || defined(GT_W_LOG2) || defined(GT_H_LOG2) || \
    defined(HIZ_SPAN) || defined(GRID_WAVES_PER_EU) || defined(PATCH_MIN_FAST) || defined(PATCH_WORDS_NEAR) || defined(PATCH_WORDS_FAR) ||    \
    defined(RASTER_BLOCKS_PER_CU) || defined(RESOLVE_BLOCKS_PER_CU) || defined(PARKED_TILES_WGS_PER_CU) || defined(PARKED_BY_TILES_DEFAULT)
#define ALP_DEV_TUNABLES 1
#ifndef ALP_DEV
#error "tuning parameter overridden without -DALP_DEV"
#endif
#endif

namespace alp {

const char *raster_dev_flags() {
    return ""
#ifdef ALP_DEV
           "ALP_DEV,"
#endif
#ifdef ALP_DEV_TUNABLES
           "tunables-overridden,"
#endif
#ifdef ALP_WG_TIMING
           "ALP_WG_TIMING,"
#endif
#ifdef ALP_RASTER_STATS
           "ALP_RASTER_STATS,"
#endif
#ifdef VIS_PLAIN_STORE
           "VIS_PLAIN_STORE(wrong image),"
#endif
#ifdef VIS_NEVER
           "VIS_NEVER(wrong image),"
#endif
#ifdef PARK_NOATOMIC
           "PARK_NOATOMIC(wrong image),"
#endif
#if defined(PARKED_SKIP_CELLS) || defined(PARKED_SKIP_COOP) || defined(PARKED_SKIP_COOP4)
           "PARKED_SKIP_*(wrong image),"
#endif
#ifdef PARKED_TILES_LAB
           "PARKED_TILES_LAB,"
#endif
#ifdef INDEX_LDS_LAB
           "INDEX_LDS_LAB,"
#endif
#if defined(PT_SKIP_CELLS) || defined(PT_SKIP_SMALL) || defined(PT_SKIP_LARGE)
           "PT_SKIP_*(wrong image),"
#endif
#ifdef GRID_STOP_AFTER
           "GRID_STOP_AFTER(wrong image),"
#endif
#ifdef GRID_NO_XCD_SWIZZLE
           "GRID_NO_XCD_SWIZZLE,"
#endif
        ;
}

#ifdef ALP_DEV
static const char *dev_getenv(const char *name) { return getenv(name); }
#else
static const char *dev_getenv(const char *) { return nullptr; }
#endif

#ifndef PARKED_BY_TILES_DEFAULT
#define PARKED_BY_TILES_DEFAULT 0       // 1: raster_parked_tiles_kernel draws the first round's parked work (ALP_PARKED=tiles / waves overrides)
#endif
#ifndef PARKED_TILES_WGS_PER_CU
#define PARKED_TILES_WGS_PER_CU 4       // persistent grid: 32 KB of LDS per workgroup
#endif

// The stages, in dependency order (one translation unit: the kernels inline each other's device functions):
#include "raster_common.h"
#include "raster_parked.h"
#include "raster_index.h"
#include "raster_plan.h"
#include "raster_grid.h"
#include "raster_large.h"
#include "raster_resolve.h"
#include "raster_post.h"

}  // namespace alp

using namespace alp;

namespace alp {

int upload_chunked(void *dst, const void *src, size_t bytes) {
    const size_t CH = (size_t)256 << 20;
    for (size_t off = 0; off < bytes; off += CH) {
        const size_t n = bytes - off < CH ? bytes - off : CH;
        ALP_HIP(hipMemcpyAsync((char *)dst + off, (const char *)src + off, n, hipMemcpyHostToDevice, ctx().stream));
    }
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    return ALP_OK;
}

int ensure_queue(alp_mesh *m, unsigned cap) {
    if (m->queue && m->qcap >= cap) return ALP_OK;
    if (m->queue) hipFree(m->queue);
    m->queue = nullptr;
    ALP_HIP(hipMalloc((void **)&m->queue, (size_t)cap * sizeof(WorkItem)));
    m->qcap = cap;
    return ALP_OK;
}

// both device queues start at 2^20 entries and grow on demand (finish_frame); ALP_QUEUE_CAP
// lowers the start so that tests can exercise the growth path
unsigned initial_queue_cap() {
    if (const char *e = getenv("ALP_QUEUE_CAP")) {
        const long v = atol(e);
        if (v >= 1 && v < (1l << 30)) return (unsigned)v;
    }
    return 1u << 20;
}

// Queues of parked work: [first round | second round] per kind.  The second round (far tiles: hardly
// anything to park) gets an eighth of the first round's capacity; finish_frame grows either on overflow.
int ensure_park(alp_mesh *m, unsigned cap_small, unsigned cap_large, unsigned cap_cell) {
    if (m->park_small && m->park_cap[0] >= cap_small && m->park_cap[1] >= cap_large && m->park_cap[2] >= cap_cell) return ALP_OK;
    if (m->park_small) hipFree(m->park_small);
    if (m->park_cell) hipFree(m->park_cell);
    m->park_small = nullptr;
    m->park_large = nullptr;
    m->park_cell = nullptr;
    const unsigned caps[3] = {cap_small, cap_large, cap_cell};
    unsigned b[3];
    for (int k = 0; k < 3; ++k) b[k] = std::max(m->park_cap_b[k], caps[k] / 8 + 64);
    ALP_HIP(hipMalloc((void **)&m->park_small, ((size_t)cap_small + b[0] + cap_large + b[1]) * sizeof(Deferred)));
    ALP_HIP(hipMalloc((void **)&m->park_cell, ((size_t)cap_cell + b[2]) * sizeof(ParkedCell)));
    m->park_large = (Deferred *)m->park_small + cap_small + b[0];
    for (int k = 0; k < 3; ++k) {
        m->park_cap[k] = caps[k];
        m->park_cap_b[k] = b[k];
    }
    return ALP_OK;
}

int apply_derived_mask(alp_mesh *m, const unsigned char *user) {
    hipStream_t st = ctx().stream;
    unsigned char *user_dev = nullptr;
    if (user) {
        if (int rc = scratch_reserve((size_t)m->n_vert, (void **)&user_dev)) return rc;
        if (int rc = upload_chunked(user_dev, user, (size_t)m->n_vert)) return rc;
    }
    if (!m->valid) ALP_HIP(hipMalloc((void **)&m->valid, (size_t)m->n_vert));
    hipLaunchKernelGGL(mask_and_kernel, dim3((unsigned)((m->n_vert + 255) / 256)), dim3(256), 0, st, m->valid_derived, user_dev,
                       (long long)m->n_vert, m->valid);
    ALP_HIP(hipGetLastError());
    ALP_HIP(hipStreamSynchronize(st));
    return ALP_OK;
}

// alp_mesh_create, explicit index array that is not the full grid: is it the grid with triangles removed
// such that a vertex mask says which (see subgrid_mark_kernel)?  On success the mesh becomes an implicit
// grid with that mask; on any mismatch it stays what it was.  `first` = the array's first triangle.
int try_subgrid(alp_mesh *m, const long long first[3]) {
    const long long a = first[0], b = first[1], c = first[2];
    const long long gw = c == a + 1 ? b - a - 1 : b - a;
    if (a < 0 || gw < 2 || m->n_vert % gw) return ALP_OK;
    const long long gh = m->n_vert / gw;
    if (gh < 2) return ALP_OK;
    const long long full = 2 * (gh - 1) * (gw - 1);
    // a small part of a large grid is cheaper as the index array it is
    if (m->n_tri >= full || m->n_tri * 4 < full) return ALP_OK;
    hipStream_t st = ctx().stream;
    const long long words = (full + 31) / 32, blocks = (words + 255) / 256;
    unsigned *block_dev = nullptr;
    auto giveup = [&](int code) {
        // (m->valid: the mesh is being created, a mask can only be this function's own, half-made one)
        for (void *p : {(void *)m->valid_derived, (void *)m->tri_present, (void *)m->tri_rank, (void *)block_dev, (void *)m->valid})
            if (p) hipFree(p);
        m->valid_derived = m->valid = nullptr;
        m->tri_present = m->tri_rank = nullptr;
        return code;
    };
    if (hipMalloc((void **)&m->valid_derived, (size_t)m->n_vert) != hipSuccess ||
        hipMalloc((void **)&m->tri_present, (size_t)words * 4) != hipSuccess ||
        hipMalloc((void **)&m->tri_rank, (size_t)words * 4) != hipSuccess ||
        hipMalloc((void **)&block_dev, (size_t)blocks * 4) != hipSuccess)
        return giveup(fail(ALP_EHIP, "sub-grid check: hipMalloc"));
    hipError_t e = hipMemsetAsync(m->valid_derived, 0, (size_t)m->n_vert, st);
    if (e == hipSuccess) e = hipMemsetAsync(m->tri_present, 0, (size_t)words * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(m->qcount_dev, 0, sizeof(unsigned), st);
    if (e != hipSuccess) return giveup(fail(ALP_EHIP, "sub-grid check: memset"));
    hipLaunchKernelGGL(subgrid_mark_kernel, dim3(ctx().cu_count * 8), dim3(256), 0, st, m->ind, (long long)m->n_tri, gw, gh,
                       m->tri_present, m->valid_derived, m->qcount_dev);
    hipLaunchKernelGGL(subgrid_absent_kernel, dim3(ctx().cu_count * 8), dim3(256), 0, st, full, gw, m->tri_present,
                       m->valid_derived, m->qcount_dev);
    hipLaunchKernelGGL(subgrid_blocksum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, m->tri_present, words, block_dev);
    std::vector<unsigned> sums((size_t)blocks);
    e = hipMemcpyAsync(m->qcount_host, m->qcount_dev, sizeof(unsigned), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(sums.data(), block_dev, (size_t)blocks * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return giveup(fail(ALP_EHIP, "sub-grid check: %s", hipGetErrorString(e)));
    if (*m->qcount_host != 0) return giveup(ALP_OK);                       // not a filtered grid: keep the index array
    unsigned long long run = 0;
    for (auto &s : sums) {
        const unsigned here = s;
        s = (unsigned)run;
        run += here;
    }
    if ((long long)run != m->n_tri) return giveup(ALP_OK);                  // cannot happen after the order check; be safe
    e = hipMemcpyAsync(block_dev, sums.data(), (size_t)blocks * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(subgrid_rank_kernel, dim3((unsigned)blocks), dim3(256), 0, st, m->tri_present, words, block_dev,
                           m->tri_rank);
        e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) return giveup(fail(ALP_EHIP, "sub-grid ranks: %s", hipGetErrorString(e)));
    hipFree(block_dev);
    block_dev = nullptr;
    if (int rc = apply_derived_mask(m, nullptr)) return giveup(rc);
    hipFree(m->ind);
    m->ind = nullptr;
    m->implicit = true;
    m->grid_h = gh;
    m->grid_w = gw;
    m->n_tri = full;
    return ALP_OK;
}

int ensure_gqueue(alp_mesh *m, unsigned cap) {
    if (m->gqueue && m->gcap >= cap) return ALP_OK;
    if (m->gqueue) hipFree(m->gqueue);
    m->gqueue = nullptr;
    ALP_HIP(hipMalloc((void **)&m->gqueue, (size_t)cap * sizeof(unsigned)));
    m->gcap = cap;
    return ALP_OK;
}

int finish_frame_of(alp_mesh *m);      // = finish_frame below (anonymous namespace)

// The x > 0 selection of reverse_proj (project.py:369) on the resident frame, in two steps: count + exclusive scan
// per chunk of the image (frame_valid_count, also waits for the frame and checks its queues), then the order-
// preserving write of the survivors' pixel index and (x, y, z) = channels (0, 2, 1) + offsets as float64
// (frame_valid_write; device pointers).  Shared by alp_render_fetch_valid and alp_render_rasterize_*.
int frame_valid_count(alp_mesh *m, int64_t *count) {
    if (int e = finish_frame_of(m)) return e;
    const long long npix = (long long)m->w * m->h;
    const int chunks = (int)((npix + COMPACT_CHUNK - 1) / COMPACT_CHUNK);
    if (chunks > m->compact_cap) {
        if (m->compact_counts) hipFree(m->compact_counts);
        if (m->compact_offsets) hipFree(m->compact_offsets);
        m->compact_counts = nullptr;
        m->compact_offsets = nullptr;
        m->compact_cap = 0;
        // counts | the chunks' extents (four floats each);  offsets | the total | the frame's extent (four floats)
        ALP_HIP(hipMalloc((void **)&m->compact_counts, (size_t)chunks * (sizeof(unsigned) + 4 * sizeof(float)) + 16));
        ALP_HIP(hipMalloc((void **)&m->compact_offsets, (size_t)(chunks + 3) * sizeof(unsigned long long)));
        m->compact_cap = chunks;
    }
    hipStream_t st = ctx().stream;
    ktime_begin();
    float *span = (float *)(((uintptr_t)(m->compact_counts + chunks) + 15) & ~(uintptr_t)15);
    hipLaunchKernelGGL(valid_count_kernel, dim3(chunks), dim3(256), 0, st, m->image, npix, m->compact_counts, span);
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, st, m->compact_counts, chunks, m->compact_offsets, span);
    ktime_end();
    ALP_HIP(hipGetLastError());
    unsigned long long tail[3] = {0, 0, 0};      // the total, then the extent of channels 0 and 2 over the survivors
    ALP_HIP(hipMemcpyAsync(tail, m->compact_offsets + chunks, sizeof(tail), hipMemcpyDeviceToHost, st));
    ALP_HIP(hipStreamSynchronize(st));
    const unsigned long long total = tail[0];
    memcpy(m->valid_span, tail + 1, sizeof(m->valid_span));
    m->valid_total = (int64_t)total;
    *count = m->valid_total;
    return ALP_OK;
}

int frame_valid_write(alp_mesh *m, const double *offsets, unsigned *idx_dev, double *xyz_dev, bool planar) {
    const long long npix = (long long)m->w * m->h;
    const int chunks = (int)((npix + COMPACT_CHUNK - 1) / COMPACT_CHUNK);
    const double o0 = offsets ? offsets[0] : 0.0, o1 = offsets ? offsets[1] : 0.0, o2 = offsets ? offsets[2] : 0.0;
    ktime_begin();
    hipLaunchKernelGGL(valid_write_kernel, dim3(chunks), dim3(256), 0, ctx().stream, m->image, npix, m->compact_offsets, o0, o1,
                       o2, idx_dev, xyz_dev, planar ? 1ll : 3ll, planar ? (long long)m->valid_total_planes : 1ll);
    ktime_end();
    ALP_HIP(hipGetLastError());
    return ALP_OK;
}

}  // namespace alp

namespace {

int ensure_frame(alp_mesh *m, int w, int h) {
    if (m->w == w && m->h == h && m->vis) return ALP_OK;
    if (m->vis) hipFree(m->vis);
    if (m->image) hipFree(m->image);
    m->vis = nullptr;
    m->image = nullptr;
    m->vis_current = false;
    ALP_HIP(hipMalloc((void **)&m->vis, (size_t)w * h * sizeof(unsigned long long) + QC_TOTAL * sizeof(unsigned)));   // + the frame's counters
    ALP_HIP(hipMalloc((void **)&m->image, (size_t)w * h * 3 * sizeof(float)));
    if (m->hiz) hipFree(m->hiz);
    m->hiz = nullptr;
    ALP_HIP(hipMalloc((void **)&m->hiz, (size_t)hiz_total(w, h) * sizeof(unsigned)));
    m->w = w;
    m->h = h;
    return ALP_OK;
}

// development: ALP_PATCH_NEAR / ALP_PATCH_FAR override the patch sizes (words; 0 switches the patches off)
static int patch_words_env(const char *name, int dflt) {
    if (const char *e = dev_getenv(name)) {
        const long w = atol(e);
        if (w >= 0 && w <= 5632) return (int)w;
    }
    return dflt;
}

// Enqueue one whole frame on the library stream, no host round trip: clear, raster passes (the
// queue lengths stay on the device), resolve.  The two queue counters are copied to pinned host
// memory at the end; finish_frame() checks them before anything reads the frame.
// `resolve_only`: the visibility buffer (and the frame's counters behind it) already hold this view's finished
// raster passes -- only the resolve runs (the visibility cache, see alp_mesh::vis_current).
template <bool IMPLICIT>
int render_impl(alp_mesh *m, const View &v, const RemapCoef &rc, double min_distance, bool resolve_only = false) {
    hipStream_t st = ctx().stream;
    const int cu = ctx().cu_count;
    unsigned *const fcount = (unsigned *)(m->vis + (size_t)v.w * v.h);
    m->vis_current = false;          // until every launch below has been accepted
    // one fill clears the visibility buffer AND the frame's queue / list counters, which live right behind it
    if (!resolve_only)
        ALP_HIP(hipMemsetAsync(m->vis, 0, (size_t)v.w * v.h * sizeof(unsigned long long) + QC_TOTAL * sizeof(unsigned), st));
    if (m->n_tri > 0 && !resolve_only) {
        // queue counters, four per round: [0] work items, [1] general entries, [2] small parked, [3] large parked
        // the consumers of one round: (a) the rare cases (near-plane crossings, 64 px and more), (b) what
        // raster_grid_kernel parked; the second round's parked entries follow the first round's in the queues
        auto drain_rare = [&](int round) -> int {
            unsigned *items = fcount + QC_STRIDE * round, *general = items + 1;
            hipLaunchKernelGGL((raster_general_kernel<IMPLICIT>), dim3(cu * 2), dim3(256), 0, st, m->vert, m->ind,
                               (long long)m->grid_w, v, m->vis, m->gqueue, general, m->gcap, m->queue, items, m->qcap);
            ALP_HIP(hipGetLastError());
            hipLaunchKernelGGL((raster_large_kernel<IMPLICIT>), dim3(cu * 8), dim3(256), 0, st, m->vert, m->ind,
                               (long long)m->grid_w, v, m->vis, m->queue, items, m->qcap);
            ALP_HIP(hipGetLastError());
            return ALP_OK;
        };
        // how the first round's parked work is drawn: by queue entry (raster_parked_kernel) or by tile through LDS depth
        // patches (raster_parked_tiles_kernel); the same image either way
        bool by_tiles = false;
#ifdef PARKED_TILES_LAB
        if constexpr (IMPLICIT) {
            const char *e = dev_getenv("ALP_PARKED");
            by_tiles = e ? e[0] == 't' : PARKED_BY_TILES_DEFAULT != 0;
        }
#endif
        auto drain_parked = [&](int round) -> int {
            unsigned *items = fcount + QC_STRIDE * round;
            const unsigned *cap = round ? m->park_cap_b : m->park_cap;
            const int wgs = round ? cu * 2 : cu * 8;
#ifdef PARKED_TILES_LAB
            if (round == 0 && by_tiles) {
                hipLaunchKernelGGL(raster_parked_tiles_kernel, dim3(cu * PARKED_TILES_WGS_PER_CU), dim3(256), 0, st, v, m->vis, m->park_small,
                                   m->park_large, m->park_cell, items + 2, cap[0], cap[1], cap[2], m->park_tiles, m->park_units,
                                   m->park_units_cap);
                ALP_HIP(hipGetLastError());
                return ALP_OK;
            }
#endif
            hipLaunchKernelGGL(raster_parked_kernel, dim3(wgs), dim3(256), 0, st, v, m->vis,
                               m->park_small + (round ? m->park_cap[0] : 0), m->park_large + (round ? m->park_cap[1] : 0),
                               m->park_cell + (round ? m->park_cap[2] : 0), items + 2, cap[0], cap[1], cap[2]);
            ALP_HIP(hipGetLastError());
            return ALP_OK;
        };
        if constexpr (IMPLICIT) {
            if (!m->park_small) {
                // ~0.7 M parked cells and a few 100 k parked triangles per 5616 x 3744 frame of the 100 M-vertex DSM
                const unsigned cap = initial_queue_cap();
                const bool dflt = cap == (1u << 20);
                if (int e = ensure_park(m, cap, cap, dflt ? 2u << 20 : cap)) return e;
            }
            const int tiles_x = (int)((m->grid_w - 1 + GT_W - 1) / GT_W);
            const long long tiles = (long long)tiles_x * ((m->grid_h - 1 + GT_H - 1) / GT_H);
            if (by_tiles && !m->park_tiles) {
                // one record per tile at most; units: a frame of the 100 M-vertex DSM makes ~45 k (grown on overflow, finish_frame)
                const unsigned ucap = std::max(m->park_units_cap, (unsigned)std::min<long long>(std::max<long long>(tiles, 1 << 16), 1 << 22));
                ParkedTile *pt = nullptr;
                ParkedUnit *pu = nullptr;
                if (hipMalloc((void **)&pt, (size_t)tiles * sizeof(ParkedTile)) != hipSuccess ||
                    hipMalloc((void **)&pu, (size_t)ucap * sizeof(ParkedUnit)) != hipSuccess) {
                    if (pt) hipFree(pt);
                    return fail(ALP_EHIP, "allocation of the parked-tile records failed");
                }
                m->park_tiles = pt;
                m->park_units = pu;
                m->park_tiles_cap = (unsigned)tiles;
                m->park_units_cap = ucap;
            }
            if (!m->tile_bounds) {      // once per mesh: the vertices never change
                // published only when both allocations and the launch succeeded: a half-made plan must not
                // make the next frame skip this block and read uninitialised boxes
                float *tb = nullptr;
                unsigned *tl = nullptr;
                hipError_t e = hipMalloc((void **)&tb, (size_t)tiles * 6 * sizeof(float));
                // three tile lists (near, far, far survivors) + their three counters
                if (e == hipSuccess) e = hipMalloc((void **)&tl, (size_t)(3 * tiles) * sizeof(unsigned));
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(tile_bounds_kernel, dim3((unsigned)tiles), dim3(256), 0, st, m->vert, (int)m->grid_h,
                                       (int)m->grid_w, tiles_x, tb);
                    e = hipGetLastError();
                }
                if (e != hipSuccess) {
                    if (tb) hipFree(tb);
                    if (tl) hipFree(tl);
                    return fail(ALP_EHIP, "frame plan of the mesh: %s", hipGetErrorString(e));
                }
                m->tile_bounds = tb;
                m->tile_lists = tl;
            }
            unsigned *near_list = m->tile_lists, *far_list = near_list + tiles, *second_list = far_list + tiles,
                     *counts = fcount + 2 * QC_STRIDE;   // [0] near, [1] far, [2] far survivors (cleared with the queue counters)
            TileCull cull;
            make_tile_cull(v, &cull);
            if (getenv("ALP_NO_TILE_CULL")) cull.enabled = 0;     // development: measure / cross-check the exact path alone
            if (getenv("ALP_NO_OCCLUSION")) cull.occlusion = 0;   // development: frustum culling only, one round
            // vertices are X, Z, Y: columns step X (R[0][0] on screen x), rows step Y (R[0][2])
            int along_rows = std::fabs(v.R[0][2]) > std::fabs(v.R[0][0]);
            if (const char *e = dev_getenv("ALP_GRID_LANES")) along_rows = e[0] == 'r';   // development override
            const unsigned plan_grid = (unsigned)((tiles + 255) / 256);
            const unsigned grid_wgs = (unsigned)((tiles + 7) / 8 * 8);     // whole turns of the 8 XCDs (see the kernel's phase 0)
            hipLaunchKernelGGL(tile_plan_kernel, dim3(plan_grid), dim3(256), 0, st, m->tile_bounds, (unsigned)tiles, cull,
                               near_list, far_list, counts, counts + 4);
            ALP_HIP(hipGetLastError());
            // first round: the near tiles (the occluders).  One workgroup per possible list entry; the
            // ones beyond the list's length leave at once.
            // LDS depth patches (words of 8 bytes; 0 = none).  Static LDS of the kernel is 19 KB: 64 KB per workgroup in all.
            static const int patch_near = patch_words_env("ALP_PATCH_NEAR", PATCH_WORDS_NEAR),
                             patch_far = patch_words_env("ALP_PATCH_FAR", PATCH_WORDS_FAR);
            hipLaunchKernelGGL(raster_grid_kernel, dim3(grid_wgs), dim3(256), (size_t)patch_near * 8, st, m->vert, m->valid,
                               (int)m->grid_h, (int)m->grid_w, v, m->vis, m->gqueue, fcount + 1, m->gcap,
                               along_rows, near_list, counts + 0, m->park_small, m->park_large, m->park_cell,
                               fcount + 2, m->park_cap[0], m->park_cap[1], m->park_cap[2], patch_near,
                               by_tiles ? m->park_tiles : nullptr, m->park_units, m->park_units_cap);
            ALP_HIP(hipGetLastError());
#ifdef ALP_WG_TIMING
            {   // duration of every workgroup of the first round
                ALP_HIP(hipStreamSynchronize(st));
                unsigned hc[4];
                ALP_HIP(hipMemcpy(hc, counts, sizeof(hc), hipMemcpyDeviceToHost));
                std::vector<unsigned long long> tt(8 * (size_t)hc[0]);
                ALP_HIP(hipMemcpyFromSymbol(tt.data(), HIP_SYMBOL(g_wgtime), tt.size() * 8));
                unsigned long long t0 = ~0ull, t1 = 0;
                std::vector<double> dur;
                double phase[4] = {0, 0, 0, 0};
                for (unsigned i = 0; i < hc[0] && i < 131072; ++i) {
                    t0 = std::min(t0, tt[8 * i]);
                    t1 = std::max(t1, tt[8 * i + 4]);
                    dur.push_back((tt[8 * i + 4] - tt[8 * i]) / 100.0);
                    for (int k = 0; k < 4; ++k) phase[k] += (tt[8 * i + k + 1] - tt[8 * i + k]) / 100.0;
                }
                std::vector<double> sorted = dur;
                std::sort(sorted.begin(), sorted.end());
                double sum = 0;
                for (double d : dur) sum += d;
                fprintf(stderr, "[wg timing] first round: %u workgroups, span %.1f us, sum of durations %.0f us (vertices %.0f, classify %.0f, fast %.0f, slow %.0f), "
                                "median %.1f, p90 %.1f, p99 %.1f, max %.1f us\n", hc[0], (t1 - t0) / 100.0, sum, phase[0], phase[1], phase[2], phase[3],
                        sorted[sorted.size() / 2], sorted[sorted.size() * 9 / 10], sorted[sorted.size() * 99 / 100], sorted.back());
                std::vector<unsigned> idx(dur.size());
                for (unsigned i = 0; i < idx.size(); ++i) idx[i] = i;
                std::partial_sort(idx.begin(), idx.begin() + std::min<size_t>(8, idx.size()), idx.end(), [&](unsigned a, unsigned b) { return dur[a] > dur[b]; });
                for (size_t k = 0; k < std::min<size_t>(8, idx.size()); ++k) {
                    const unsigned i = idx[k];
                    fprintf(stderr, "   wg %u: start +%.1f us, duration %.1f us = vertices %.1f + classify %.1f + fast %.1f + slow %.1f\n", i,
                            (tt[8 * i] - t0) / 100.0, dur[i], (tt[8 * i + 1] - tt[8 * i]) / 100.0, (tt[8 * i + 2] - tt[8 * i + 1]) / 100.0,
                            (tt[8 * i + 3] - tt[8 * i + 2]) / 100.0, (tt[8 * i + 4] - tt[8 * i + 3]) / 100.0);
                }
            }
#endif
            // The rare cases (near-plane crossings, triangles of 64 px and more) of BOTH rounds are drawn once, after
            // the second round's grid kernel: its general entries follow the first round's in the same queue.  The
            // pyramid then lacks those few triangles as occluders -- it stays conservative -- and a frame has two
            // launches fewer.
            const bool two_rounds = cull.enabled && cull.occlusion;
            if (!two_rounds)
                if (int e = drain_rare(0)) return e;
            if (int e = drain_parked(0)) return e;
            if (two_rounds) {
                // depth pyramid of everything the first round drew, occlusion test of the far tiles, second round.
                // (Measured and not kept: building the pyramid BEFORE the first round's parked cells / triangles
                // are drawn and running the second round on a second stream next to them -- the parked geometry
                // is the main occluder, three times as many far tiles survive, 1.23 instead of 1.06 ms.)
                const HizDims dm = hiz_dims(v.w, v.h);
                hipLaunchKernelGGL(hiz_build_kernel, dim3((unsigned)dm.w[3], (unsigned)dm.h[3]), dim3(256), 0, st, m->vis, v.w,
                                   v.h, dm, m->hiz, counts + 4);
                ALP_HIP(hipGetLastError());
                hipLaunchKernelGGL(tile_occlusion_kernel, dim3(plan_grid), dim3(256), 0, st, m->tile_bounds, cull, far_list,
                                   counts, dm, m->hiz, second_list, counts + 2);
                ALP_HIP(hipGetLastError());
                hipLaunchKernelGGL(raster_grid_kernel, dim3(grid_wgs), dim3(256), (size_t)patch_far * 8, st, m->vert, m->valid,
                                   (int)m->grid_h, (int)m->grid_w, v, m->vis, m->gqueue, fcount + 1, m->gcap,
                                   along_rows, second_list, counts + 2, m->park_small + m->park_cap[0],
                                   m->park_large + m->park_cap[1], m->park_cell + m->park_cap[2], fcount + QC_STRIDE + 2,
                                   m->park_cap_b[0], m->park_cap_b[1], m->park_cap_b[2], patch_far, (ParkedTile *)nullptr,
                                   (ParkedUnit *)nullptr, 0u);
                ALP_HIP(hipGetLastError());
                if (int e = drain_rare(0)) return e;
                if (int e = drain_parked(1)) return e;
            }
#ifdef ALP_RASTER_STATS
            if (by_tiles) {      // census of the parked-tile records: how many bin passes, how full
                unsigned pc[8];
                ALP_HIP(hipStreamSynchronize(st));
                ALP_HIP(hipMemcpy(pc, fcount, sizeof(pc), hipMemcpyDeviceToHost));
                std::vector<ParkedTile> rr(pc[7]);
                if (pc[7]) ALP_HIP(hipMemcpy(rr.data(), m->park_tiles, rr.size() * sizeof(ParkedTile), hipMemcpyDeviceToHost));
                long long bins = 0, area = 0, ent[3] = {0, 0, 0}, hist[8] = {0, 0, 0, 0, 0, 0, 0, 0}, scan = 0;
                for (const ParkedTile &r : rr) {
                    const int nb = (((r.i1 - (r.i0 & ~7)) >> 6) + 1) * (((r.j1 - r.j0) >> 6) + 1);
                    bins += nb;
                    area += (long long)(r.i1 - r.i0 + 1) * (r.j1 - r.j0 + 1);
                    for (int k = 0; k < 3; ++k) ent[k] += r.n[k];
                    scan += (long long)nb * (r.n[0] + r.n[1] + r.n[2]);
                    ++hist[nb <= 1 ? 0 : nb <= 2 ? 1 : nb <= 4 ? 2 : nb <= 8 ? 3 : nb <= 16 ? 4 : nb <= 32 ? 5 : nb <= 128 ? 6 : 7];
                }
                fprintf(stderr, "[parked tiles] records %u units %u | bins %lld (box area %lld px) | entries small %lld large %lld cells %lld | "
                                "entry scans over all passes %lld | records by bins <=1 %lld, 2 %lld, <=4 %lld, <=8 %lld, <=16 %lld, <=32 %lld, <=128 %lld, more %lld\n",
                        pc[7], pc[6], bins, area, ent[0], ent[1], ent[2], scan, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
            }
            {
                unsigned hc[4];
                ALP_HIP(hipMemcpyAsync(hc, counts, sizeof(hc), hipMemcpyDeviceToHost, st));
                ALP_HIP(hipStreamSynchronize(st));
                fprintf(stderr, "[frame plan] tiles %lld: near %u, far %u of which %u survive the occlusion test\n", tiles, hc[0],
                        hc[1], hc[2]);
                if (two_rounds) {
                    // how many NEAR tiles would an occlusion test against the FINISHED frame drop (an upper bound for
                    // what more rounds could gain)?  Full-frame pyramid, the NEAR list through tile_occlusion_kernel.
                    const HizDims dm = hiz_dims(v.w, v.h);
                    const unsigned full[4] = {65535u, (unsigned)v.w, 65535u, (unsigned)v.h}, zero = 0;
                    ALP_HIP(hipMemcpy(counts + 4, full, sizeof(full), hipMemcpyHostToDevice));
                    ALP_HIP(hipMemcpy(counts + 3, &zero, sizeof(zero), hipMemcpyHostToDevice));
                    hipLaunchKernelGGL(hiz_build_kernel, dim3((unsigned)dm.w[3], (unsigned)dm.h[3]), dim3(256), 0, st, m->vis, v.w, v.h, dm,
                                       m->hiz, counts + 4);
                    hipLaunchKernelGGL(tile_occlusion_kernel, dim3(plan_grid), dim3(256), 0, st, m->tile_bounds, cull, near_list,
                                       counts - 1, dm, m->hiz, second_list, counts + 3);      // counts[-1 + 1] = the NEAR count
                    unsigned left = 0;
                    ALP_HIP(hipStreamSynchronize(st));
                    ALP_HIP(hipMemcpy(&left, counts + 3, sizeof(left), hipMemcpyDeviceToHost));
                    fprintf(stderr, "[frame plan] of the %u NEAR tiles %u survive a test against the finished frame\n", hc[0], left);
                }
            }
#endif
        } else {
            const long long want = (m->n_tri + 255) / 256;
#ifndef RASTER_BLOCKS_PER_CU
#define RASTER_BLOCKS_PER_CU 64        // 16: 2.12 ms, 64: 2.01 (explicit int32 indices, 100 M vertices)
#endif
            const int grid = (int)(want < (long long)cu * RASTER_BLOCKS_PER_CU ? want : (long long)cu * RASTER_BLOCKS_PER_CU);
            bool through_lds = false;
#ifdef INDEX_LDS_LAB        // round 5: a block's vertices transformed once through an LDS hash set -- bit-exact, slower, not kept (raster_index.h)
            if constexpr (!IMPLICIT) {
                if (m->ind_sharing < 0.0f) {       // once per mesh: how many distinct vertices do 256 consecutive triangles name?
                    unsigned long long *sums = nullptr, host[2] = {0, 0};
                    if (int e = scratch_reserve(sizeof(host), (void **)&sums)) return e;
                    ALP_HIP(hipMemsetAsync(sums, 0, sizeof(host), st));
                    const long long blocks = want, step = blocks > 4096 ? blocks / 4096 : 1;
                    hipLaunchKernelGGL(index_sharing_kernel, dim3((unsigned)std::min<long long>((blocks + step - 1) / step, 4096)), dim3(256), 0, st,
                                       m->ind, (long long)m->n_tri, step, sums);
                    ALP_HIP(hipMemcpyAsync(host, sums, sizeof(host), hipMemcpyDeviceToHost, st));
                    ALP_HIP(hipStreamSynchronize(st));
                    m->ind_sharing = host[1] ? (float)((double)host[0] / (double)host[1]) : 1.0f;
                }
                const char *force = getenv("ALP_INDEX_LDS");     // "0" / "1": either kernel whatever the array
                through_lds = force ? atoi(force) != 0 : m->ind_sharing <= INDEX_SHARING_MAX;
                if (through_lds)
                    hipLaunchKernelGGL(raster_index_lds_kernel, dim3(grid), dim3(256), 0, st, m->vert, m->ind, m->valid, (long long)m->n_tri, v,
                                       m->vis, m->gqueue, fcount + 1, m->gcap);
            }
#endif
            if (!through_lds)
                hipLaunchKernelGGL((raster_kernel<IMPLICIT>), dim3(grid), dim3(256), 0, st, m->vert, m->ind, m->valid,
                                   (long long)m->n_tri, (long long)m->grid_w, v, m->vis, m->gqueue, fcount + 1,
                                   m->gcap);
            ALP_HIP(hipGetLastError());
            if (int e = drain_rare(0)) return e;
        }
    }
    const long long npix = (long long)v.w * v.h;
    const long long want = (npix + 255) / 256;
#ifndef RESOLVE_BLOCKS_PER_CU
#define RESOLVE_BLOCKS_PER_CU 64       // 16: 0.200 ms, 64: 0.176, one block per 256 pixels: 0.176 (100 M-vertex frame)
#endif
    const int grid = (int)(want < (long long)cu * RESOLVE_BLOCKS_PER_CU ? want : (long long)cu * RESOLVE_BLOCKS_PER_CU);
    const int identity = rc.a1 == 1 && rc.a2 == 1 && rc.k1 == 0 && rc.k2 == 0 && rc.k3 == 0 && rc.k4 == 0 && rc.k5 == 0 &&
                         rc.k6 == 0 && rc.p1 == 0 && rc.p2 == 0 && rc.s1 == 0 && rc.s2 == 0 && rc.s3 == 0 && rc.s4 == 0 &&
                         rc.c0 > 0 && rc.c1 > 0;
    hipLaunchKernelGGL((resolve_kernel<IMPLICIT>), dim3(grid), dim3(256), 0, st, m->vert,
                       m->coords_as_value ? nullptr : m->value, m->ind,
                       (long long)m->grid_w, v, rc, identity, min_distance, m->vis, m->image, fcount,
                       m->n_tri > 0 ? m->qcount_host : nullptr);
    ALP_HIP(hipGetLastError());
    m->last_v = v;
    m->last_rc = rc;
    m->last_min_distance = min_distance;
    m->unchecked = m->n_tri > 0;
    m->vis_current = true;
    m->rz_n = -1;                    // a rasterisation plan belongs to the frame it was made for
    ++(resolve_only ? m->frames_resolve_only : m->frames_full);
#ifdef ALP_RASTER_STATS
    {
        unsigned long long hs[24 + 64], zero[24 + 64] = {0};
        ALP_HIP(hipStreamSynchronize(st));
        ALP_HIP(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_rstat), sizeof(hs)));
        ALP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rstat), zero, sizeof(zero)));
        fprintf(stderr, "[raster stats] inline tris %llu | inline fragments by bbox width: 1px %llu, 2-3 %llu, 4-7 %llu, "
                        ">=8 %llu | coop tris %llu fragments %llu\n", hs[2], hs[3], hs[4], hs[5], hs[6], hs[8], hs[7]);
        static const char *bn[8] = {"<=512", "<=1K", "<=2K", "<=4K", "<=8K", "<=16K", "<=64K", ">64K"};
        for (int b = 0; b < 8; ++b)
            fprintf(stderr, "[footprint %6s px] tiles %7llu  area %10llu  cells FAST %9llu SLOW %9llu PARKED %8llu | box centres FAST %10llu "
                            "SLOW %10llu PARKED %10llu\n", bn[b], hs[24 + 8 * b], hs[25 + 8 * b], hs[26 + 8 * b], hs[27 + 8 * b], hs[28 + 8 * b],
                    hs[29 + 8 * b], hs[30 + 8 * b], hs[31 + 8 * b]);
        fprintf(stderr, "[parked cells] %llu: box height <= 2: %llu, <= 4: %llu; width <= 4: %llu; centres in boxes %llu\n", hs[23], hs[19], hs[20],
                hs[21], hs[22]);
        fprintf(stderr, "[grid stats] (unused %llu) tiles drawn %llu | FAST cells %llu (wave rounds %llu) SLOW cells %llu (wave "
                        "rounds %llu)\n", hs[9], hs[10], hs[11], hs[13], hs[12], hs[14]);
    }
#endif
    m->rendered = true;
    return ALP_OK;
}

// everything of a View the raster passes read (the resolve's float64 members follow from the same parameters)
static bool same_view(const View &a, const View &b) {
    return a.w == b.w && a.h == b.h && a.fx == b.fx && a.fy == b.fy && a.sx == b.sx && a.sy == b.sy && a.fxd == b.fxd && a.fyd == b.fyd &&
           !memcmp(a.R, b.R, sizeof(a.R)) && !memcmp(a.camf, b.camf, sizeof(a.camf)) && !memcmp(a.caml, b.caml, sizeof(a.caml)) &&
           !memcmp(a.Rd, b.Rd, sizeof(a.Rd)) && !memcmp(a.camd, b.camd, sizeof(a.camd));
}

// Before anything reads the last frame: wait for it and make sure neither queue overflowed.  A
// queue that was too small is grown and the frame rendered again (max is idempotent, but the
// dropped entries were never drawn).
int finish_frame(alp_mesh *m) {
    while (m->unchecked) {
        ALP_HIP(hipStreamSynchronize(ctx().stream));
        m->unchecked = false;
        const unsigned *h = m->qcount_host;
        const unsigned items = std::max(h[0], h[QC_STRIDE]), general = std::max(h[1], h[QC_STRIDE + 1]);
        bool park_ok = true;
        unsigned want_a[3], want_b[3];
        for (int k = 0; k < 3; ++k) {
            want_a[k] = m->park_cap[k];
            want_b[k] = m->park_cap_b[k];
            if (m->park_small && h[2 + k] > m->park_cap[k]) { park_ok = false; want_a[k] = h[2 + k] + h[2 + k] / 4 + 1024; }
            if (m->park_small && h[QC_STRIDE + 2 + k] > m->park_cap_b[k]) { park_ok = false; want_b[k] = h[QC_STRIDE + 2 + k] + h[QC_STRIDE + 2 + k] / 4 + 1024; }
        }
        const bool units_ok = !m->park_units || h[6] <= m->park_units_cap;      // [6]: units the first round's tiles asked for
        if (items <= m->qcap && general <= m->gcap && park_ok && units_ok) break;
        if (!units_ok) {
            hipFree(m->park_tiles);
            hipFree(m->park_units);
            m->park_tiles = nullptr;
            m->park_units = nullptr;
            m->park_units_cap = h[6] + h[6] / 4 + 1024;      // reallocated by the frame below
        }
        if (!park_ok) {
            for (int k = 0; k < 3; ++k) m->park_cap_b[k] = want_b[k];
            m->park_cap[0] = 0;           // force the reallocation
            if (int e = ensure_park(m, want_a[0], want_a[1], want_a[2])) return e;
        }
        if (items > m->qcap)
            if (int e = ensure_queue(m, items + items / 4 + 1024)) return e;
        if (general > m->gcap)
            if (int e = ensure_gqueue(m, general + general / 4 + 1024)) return e;
        if (int e = m->implicit ? render_impl<true>(m, m->last_v, m->last_rc, m->last_min_distance)
                                : render_impl<false>(m, m->last_v, m->last_rc, m->last_min_distance))
            return e;
    }
    return ALP_OK;
}

}  // namespace

int alp::finish_frame_of(alp_mesh *m) { return finish_frame(m); }

// n x 3 float32 or float64 host array -> n x 3 float32 on the device; float64 is staged through the library
// scratch in chunks and cast there (no host pass over the array, no float32 copy on the host)
int alp::upload_f32(float *dst, const void *src, int dtype, int64_t n_vert) {
    const size_t count = (size_t)n_vert * 3;
    if (dtype == ALP_F32) return upload_chunked(dst, src, count * 4);
    const size_t CH = (size_t)24 << 20;                 // doubles per chunk: 192 MB (tools/h2d_rate.hip: large chunks, no sync in between)
    const size_t ch = count < CH ? count : CH;
    double *stage = nullptr;
    if (int rc = scratch_reserve(ch * 8, (void **)&stage)) return rc;
    hipStream_t st = ctx().stream;
    for (size_t off = 0; off < count; off += ch) {
        const size_t cnt = count - off < ch ? count - off : ch;
        ALP_HIP(hipMemcpyAsync(stage, (const double *)src + off, cnt * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(cast_f64_f32_kernel, dim3(4096), dim3(256), 0, st, stage, (long long)cnt, (long long)off, dst);
        ALP_HIP(hipGetLastError());
    }
    ALP_HIP(hipStreamSynchronize(st));
    return ALP_OK;
}

// Regular-grid recognition in a host index array (HostGridCheck, host_check_threads, grid_candidate): host/alp_host.h --
// HIP-free, so that its threads run under the sanitizers on the CPU build.
using alp::host::HostGridCheck;
using alp::host::host_check_threads;

extern "C" {

int alp_mesh_create(const void *vert, int vert_dtype, const void *value, int value_dtype, int64_t n_vert, const void *ind,
                    int ind_dtype, int64_t n_tri, int64_t grid_h, int64_t grid_w, alp_mesh_t **out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(out, "out is NULL");
    *out = nullptr;
    ALP_REQUIRE(vert && n_vert > 0, "vert is NULL or empty");
    ALP_REQUIRE(vert_dtype == ALP_F32 || vert_dtype == ALP_F64, "vert_dtype must be ALP_F32 or ALP_F64");
    ALP_REQUIRE(!value || value_dtype == ALP_F32 || value_dtype == ALP_F64, "value_dtype must be ALP_F32 or ALP_F64");
    ALP_REQUIRE(n_vert < ((int64_t)1 << 31), "more than 2^31 vertices");
    const bool implicit = ind == nullptr;
    if (implicit) {
        ALP_REQUIRE(grid_h >= 2 && grid_w >= 2 && grid_h * grid_w == n_vert, "implicit grid: grid_h*grid_w != n_vert");
        n_tri = 2 * (grid_h - 1) * (grid_w - 1);
    } else {
        ALP_REQUIRE(ind_dtype == ALP_I32 || ind_dtype == ALP_I64, "ind_dtype must be ALP_I32 or ALP_I64");
        ALP_REQUIRE(n_tri >= 0, "n_tri is negative");
    }
    ALP_REQUIRE(n_tri < ((int64_t)1 << 32) - 1, "more than 2^32-2 triangles");
    alp_mesh *m = new alp_mesh();
    m->n_vert = n_vert;
    m->n_tri = n_tri;
    m->grid_h = grid_h;
    m->grid_w = grid_w;
    m->implicit = implicit;
    int rc = ALP_OK;
    auto bail = [&](int code) { alp_mesh_destroy(m); return code; };
    // The index array the reference builds (surface.py:194-201) is the full regular grid unless nodata triangles were
    // filtered out: an array with the grid's first triangle and exactly its triangle count is a candidate
    const bool detect = !getenv("ALP_NO_GRID_DETECT");       // env: keep the index path (tests, benchmarks)
    long long cand_gh = 0, cand_gw = 0;
    if (!implicit && detect) host::grid_candidate(ind, ind_dtype, n_tri, n_vert, &cand_gh, &cand_gw);
    HostGridCheck host_check;                 // its destructor joins on every way out of this function
    if (cand_gw) {
        const int T = host_check_threads(n_tri);
        if (T > 0) host_check.start(ind, ind_dtype, cand_gh, cand_gw, T);
    }
    if (hipMalloc((void **)&m->vert, (size_t)n_vert * 12) != hipSuccess) return bail(fail(ALP_EHIP, "hipMalloc vert"));
    if ((rc = upload_f32(m->vert, vert, vert_dtype, n_vert))) return bail(rc);
    if (value) {
        if (hipMalloc((void **)&m->value, (size_t)n_vert * 12) != hipSuccess) return bail(fail(ALP_EHIP, "hipMalloc value"));
        if ((rc = upload_f32(m->value, value, value_dtype, n_vert))) return bail(rc);
    }
    if (hipMalloc((void **)&m->qcount_dev, QC_TOTAL * sizeof(unsigned)) != hipSuccess ||
        hipHostMalloc((void **)&m->qcount_host, QC_TOTAL * sizeof(unsigned), hipHostMallocDefault) != hipSuccess)
        return bail(fail(ALP_EHIP, "hipMalloc queue counter"));
    // ... checked by the host threads started above while the vertices were uploaded -- then the array never crosses
    // PCIe -- or, where there are no threads to spare, WHILE IT STREAMS through the staging buffer; either way a full
    // grid is never stored: no 12 B/triangle buffer is allocated, nothing is narrowed or written, and the mesh is
    // rendered by the LDS-tiled grid kernels (same triangle ids, same result, no index traffic).  An array that only
    // starts like the grid takes the general path below.
    bool streamed_grid = false;
    if (cand_gw) {
        const long long gh = cand_gh, gw = cand_gw;
        if (host_check.started) {
            streamed_grid = host_check.is_grid();
        } else {
            hipStream_t st = ctx().stream;
            const size_t esize = ind_dtype == ALP_I32 ? 4 : 8;
            const int64_t total = n_tri * 3;
            const int64_t CH = (int64_t)(((size_t)192 << 20) / esize) / 3 * 3;      // whole triangles per chunk
            const int64_t ch = total < CH ? total : CH;
            void *stage = nullptr;
            if ((rc = scratch_reserve((size_t)ch * esize, &stage))) return bail(rc);
            hipError_t e = hipMemsetAsync(m->qcount_dev, 0, sizeof(unsigned), st);
            for (int64_t off = 0; off < total && e == hipSuccess; off += ch) {
                const int64_t cnt = total - off < ch ? total - off : ch;
                e = hipMemcpyAsync(stage, (const char *)ind + (size_t)off * esize, (size_t)cnt * esize, hipMemcpyHostToDevice, st);
                if (e != hipSuccess) break;
                if (ind_dtype == ALP_I32)
                    hipLaunchKernelGGL(check_grid_chunk_kernel<int>, dim3(4096), dim3(256), 0, st, (const int *)stage, cnt / 3, off / 3, gw, m->qcount_dev);
                else
                    hipLaunchKernelGGL(check_grid_chunk_kernel<long long>, dim3(4096), dim3(256), 0, st, (const long long *)stage, cnt / 3, off / 3, gw,
                                       m->qcount_dev);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(m->qcount_host, m->qcount_dev, sizeof(unsigned), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return bail(fail(ALP_EHIP, "grid check: %s", hipGetErrorString(e)));
            streamed_grid = *m->qcount_host == 0;
        }
        if (streamed_grid) {
            m->implicit = true;
            m->grid_h = gh;
            m->grid_w = gw;
        }
    }
    const bool want_ind = !implicit && n_tri > 0 && !streamed_grid;
    if (want_ind) {
        hipStream_t st = ctx().stream;
        if (hipMalloc((void **)&m->ind, (size_t)n_tri * 12) != hipSuccess) return bail(fail(ALP_EHIP, "hipMalloc ind"));
        if (hipMemsetAsync(m->qcount_dev, 0, sizeof(unsigned), st) != hipSuccess) return bail(fail(ALP_EHIP, "index check: memset"));
        const int64_t total = n_tri * 3;
        if (ind_dtype == ALP_I32) {
            if ((rc = upload_chunked(m->ind, ind, (size_t)total * 4))) return bail(rc);
            hipLaunchKernelGGL(check_index_range_kernel, dim3(4096), dim3(256), 0, st, m->ind, (long long)total, (long long)n_vert,
                               m->qcount_dev);
        } else {
            // int64 (what numpy builds, surface.py:194-201; project.py:215 casts with astype("i4")): narrowed on the
            // device, staged through the library scratch in chunks of 192 MB
            const int64_t CH = 24 << 20;
            const int64_t ch = total < CH ? total : CH;
            long long *stage = nullptr;
            if ((rc = scratch_reserve((size_t)ch * 8, (void **)&stage))) return bail(rc);
            for (int64_t off = 0; off < total; off += ch) {
                const int64_t cnt = total - off < ch ? total - off : ch;
                if (hipMemcpyAsync(stage, (const long long *)ind + off, (size_t)cnt * 8, hipMemcpyHostToDevice, st) != hipSuccess)
                    return bail(fail(ALP_EHIP, "index upload"));
                hipLaunchKernelGGL(narrow_indices_kernel, dim3(4096), dim3(256), 0, st, stage, (long long)cnt, (long long)off, m->ind,
                                   (long long)n_vert, m->qcount_dev);
            }
        }
        // range check (an out-of-range index would fault in the kernels): counted by the kernels above
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(m->qcount_host, m->qcount_dev, sizeof(unsigned), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return bail(fail(ALP_EHIP, "index upload: %s", hipGetErrorString(e)));
        if (*m->qcount_host != 0) {
            // name the first offender like the host check did (cold path: a scan of the caller's array)
            for (int64_t i = 0; i < total; ++i) {
                const long long v = ind_dtype == ALP_I32 ? (long long)((const int *)ind)[i] : ((const long long *)ind)[i];
                if (v < 0 || v >= n_vert) return bail(fail(ALP_EINVAL, "index %lld out of range at %lld", v, (long long)i));
            }
            return bail(fail(ALP_EINVAL, "%u indices out of range", *m->qcount_host));
        }
    }
    if ((rc = ensure_queue(m, initial_queue_cap()))) return bail(rc);
    if ((rc = ensure_gqueue(m, initial_queue_cap()))) return bail(rc);
    // ... and when they were (surface.py:203-205), the grid with a vertex mask
    if (!implicit && !m->implicit && n_tri >= 1 && detect) {
        long long first[3];
        for (int k = 0; k < 3; ++k)
            first[k] = ind_dtype == ALP_I32 ? (long long)((const int *)ind)[k] : ((const long long *)ind)[k];
        if ((rc = try_subgrid(m, first))) return bail(rc);
    }
    *out = m;
    return ALP_OK;
}

int alp_mesh_info(alp_mesh_t *m, int64_t info[4]) {
    ALP_REQUIRE(m && info, "NULL argument");
    info[0] = m->implicit ? 1 : 0;
    info[1] = m->grid_h;
    info[2] = m->grid_w;
    info[3] = m->n_tri;
    return ALP_OK;
}

int alp_mesh_destroy(alp_mesh_t *m) {
    if (!m) return ALP_OK;
    if (ctx().ready) hipStreamSynchronize(ctx().stream);
    for (void *p : {(void *)m->vert, (void *)m->value, (void *)m->ind, (void *)m->valid, (void *)m->valid_derived,
                    (void *)m->tri_present, (void *)m->tri_rank, (void *)m->vis, (void *)m->image,
                    (void *)m->queue, (void *)m->gqueue, (void *)m->qcount_dev, (void *)m->compact_counts, (void *)m->compact_offsets,
                    (void *)m->tile_bounds, (void *)m->tile_lists, (void *)m->hiz, (void *)m->park_small, (void *)m->park_cell,
                    (void *)m->rz_points, (void *)m->rz_work, (void *)m->park_tiles, (void *)m->park_units})
        if (p) hipFree(p);
    if (m->qcount_host) hipHostFree(m->qcount_host);
    for (auto &e : m->ev_frame)
        if (e) hipEventDestroy(e);
    delete m;
    return ALP_OK;
}

int alp_render_enqueue(alp_mesh_t *m, const double params[ALP_NPARAM], const double *offsets, double min_distance) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m && params, "NULL argument");
    ALP_REQUIRE(params[21] >= 1 && params[22] >= 1 && params[21] <= 32768 && params[22] <= 32768,
                "image size w,h must be in [1, 32768]");
    View v;
    RemapCoef rc;
    make_view(params, offsets, &v, &rc);
    if (int e = ensure_frame(m, v.w, v.h)) return e;
    // same view as the frame whose visibility buffer is still there: the raster passes would rebuild it bit for bit
    const bool no_cache = getenv("ALP_NO_VIS_CACHE") != nullptr;      // tests, benchmarks: force the full frame
    const bool cached = m->vis_current && !no_cache && same_view(v, m->last_v);
    if (!m->ev_frame[0]) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            if (a) hipEventDestroy(a);
            return fail(ALP_EHIP, "hipEventCreate failed");
        }
        m->ev_frame[0] = a;
        m->ev_frame[1] = b;
    }
    ALP_HIP(hipEventRecord(m->ev_frame[0], ctx().stream));
    const int e = m->implicit ? render_impl<true>(m, v, rc, min_distance, cached) : render_impl<false>(m, v, rc, min_distance, cached);
    if (e) return e;
    ALP_HIP(hipEventRecord(m->ev_frame[1], ctx().stream));
    return ALP_OK;
}

int alp_mesh_frame_ms(alp_mesh_t *m, float *ms) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m && ms, "NULL argument");
    if (!m->rendered || !m->ev_frame[1]) return fail(ALP_ESTATE, "alp_mesh_frame_ms: nothing rendered yet");
    ALP_HIP(hipEventSynchronize(m->ev_frame[1]));
    ALP_HIP(hipEventElapsedTime(ms, m->ev_frame[0], m->ev_frame[1]));
    return ALP_OK;
}

int alp_mesh_trim(alp_mesh_t *m) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m, "mesh handle is NULL");
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    if (m->rz_work) hipFree(m->rz_work);
    m->rz_work = nullptr;
    m->rz_work_cap = 0;
    if (m->rz_points) hipFree(m->rz_points);
    m->rz_points = nullptr;
    m->rz_cap = 0;
    m->rz_n = -1;
    return ALP_OK;
}

int alp_render_fetch(alp_mesh_t *m, float *out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m && out, "NULL argument");
    if (!m->rendered) return fail(ALP_ESTATE, "alp_render_fetch: nothing rendered yet");
    if (int e = finish_frame(m)) return e;
    ALP_HIP(hipMemcpyAsync(out, m->image, (size_t)m->w * m->h * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx().stream));
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    return ALP_OK;
}

int alp_render_fetch_u8(alp_mesh_t *m, float scale, int reverse_channels, uint8_t *out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m && out, "NULL argument");
    if (!m->rendered) return fail(ALP_ESTATE, "alp_render_fetch_u8: nothing rendered yet");
    if (int e = finish_frame(m)) return e;
    const long long npix = (long long)m->w * m->h;
    unsigned char *dev = nullptr;
    if (int rc = scratch_reserve((size_t)npix * 3, (void **)&dev)) return rc;
    hipLaunchKernelGGL(image_u8_kernel, dim3(ctx().cu_count * 16), dim3(256), 0, ctx().stream, m->image, npix, scale,
                       reverse_channels, dev);
    ALP_HIP(hipGetLastError());
    ALP_HIP(hipMemcpyAsync(out, dev, (size_t)npix * 3, hipMemcpyDeviceToHost, ctx().stream));
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    return ALP_OK;
}

int alp_render_fetch_visibility(alp_mesh_t *m, uint64_t *out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m && out, "NULL argument");
    if (!m->rendered) return fail(ALP_ESTATE, "alp_render_fetch_visibility: nothing rendered yet");
    if (int e = finish_frame(m)) return e;
    const unsigned long long *src = m->vis;
    if (m->tri_present) {                    // filtered grid: the caller's triangle numbering
        const long long npix = (long long)m->w * m->h;
        unsigned long long *tr = nullptr;
        if (int rc = scratch_reserve((size_t)npix * sizeof(uint64_t), (void **)&tr)) return rc;
        hipLaunchKernelGGL(vis_translate_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx().stream, m->vis, npix,
                           m->tri_present, m->tri_rank, tr);
        ALP_HIP(hipGetLastError());
        src = tr;
    }
    ALP_HIP(hipMemcpyAsync(out, src, (size_t)m->w * m->h * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx().stream));
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    return ALP_OK;
}

int alp_render_gather(alp_mesh_t *m, const int32_t *u, const int32_t *v, int64_t n, const double *offsets,
                      double *xyz_out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m, "mesh handle is NULL");
    ALP_REQUIRE(n >= 0, "n is negative");
    if (!m->rendered) return fail(ALP_ESTATE, "alp_render_gather: nothing rendered yet");
    if (int e = finish_frame(m)) return e;
    if (n == 0) return ALP_OK;
    ALP_REQUIRE(u && v && xyz_out, "NULL argument");
    char *dev = nullptr;
    const size_t uv_bytes = (size_t)n * sizeof(int32_t), xyz_bytes = (size_t)n * 3 * sizeof(double);
    if (int rc = scratch_reserve(xyz_bytes + 2 * uv_bytes, (void **)&dev)) return rc;
    double *xyz_dev = (double *)dev;
    int32_t *u_dev = (int32_t *)(dev + xyz_bytes), *v_dev = u_dev + n;
    hipStream_t st = ctx().stream;
    const double o0 = offsets ? offsets[0] : 0.0, o1 = offsets ? offsets[1] : 0.0, o2 = offsets ? offsets[2] : 0.0;
    hipError_t e = hipMemcpyAsync(u_dev, u, uv_bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(v_dev, v, uv_bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        KTimeScope kt;
        hipLaunchKernelGGL(gather_pixels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m->image, m->w,
                           m->h, u_dev, v_dev, n, o0, o1, o2, xyz_dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(xyz_out, xyz_dev, xyz_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(ALP_EHIP, "alp_render_gather: %s", hipGetErrorString(e));
    return ALP_OK;
}

int alp_distance_mask(const double *xyz, int64_t n, const double camera[3], double min_distance, double max_distance,
                      uint8_t *keep) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(n >= 0, "n is negative");
    if (n == 0) return ALP_OK;
    ALP_REQUIRE(xyz && camera && keep, "NULL argument");
    ALP_REQUIRE(!(min_distance < 0), "min_distance must be non-negative");
    ALP_REQUIRE(!(max_distance < min_distance), "max_distance must be >= min_distance");
    char *dev = nullptr;
    const size_t xyz_bytes = (size_t)n * 3 * sizeof(double);
    if (int rc = scratch_reserve(xyz_bytes + (size_t)n, (void **)&dev)) return rc;
    unsigned char *keep_dev = (unsigned char *)(dev + xyz_bytes);
    hipStream_t st = ctx().stream;
    hipError_t e = hipMemcpyAsync(dev, xyz, xyz_bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        KTimeScope kt;
        hipLaunchKernelGGL(distance_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double *)dev,
                           (long long)n, camera[0], camera[1], camera[2], min_distance, max_distance, keep_dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(keep, keep_dev, (size_t)n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(ALP_EHIP, "alp_distance_mask: %s", hipGetErrorString(e));
    return ALP_OK;
}

int alp_render_valid_count(alp_mesh_t *m, int64_t *count) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m && count, "NULL argument");
    if (!m->rendered) return fail(ALP_ESTATE, "alp_render_valid_count: nothing rendered yet");
    return frame_valid_count(m, count);
}

int alp_render_fetch_valid(alp_mesh_t *m, const double *offsets, uint32_t *idx_out, double *xyz_out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m, "mesh handle is NULL");
    if (m->valid_total < 0) return fail(ALP_ESTATE, "alp_render_fetch_valid: call alp_render_valid_count first");
    const int64_t M = m->valid_total;
    m->valid_total = -1;
    if (M == 0) return ALP_OK;
    ALP_REQUIRE(idx_out && xyz_out, "output is NULL");
    char *dev = nullptr;
    const size_t xyz_bytes = (size_t)M * 3 * sizeof(double), idx_bytes = (size_t)M * sizeof(unsigned);
    if (int rc = scratch_reserve(xyz_bytes + idx_bytes, (void **)&dev)) return rc;
    double *xyz_dev = (double *)dev;
    unsigned *idx_dev = (unsigned *)(dev + xyz_bytes);
    hipStream_t st = ctx().stream;
    if (int rc = frame_valid_write(m, offsets, idx_dev, xyz_dev, false)) return rc;
    hipError_t e = hipMemcpyAsync(xyz_out, xyz_dev, xyz_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(idx_out, idx_dev, idx_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(ALP_EHIP, "alp_render_fetch_valid: %s", hipGetErrorString(e));
    return ALP_OK;
}

int alp_render_fetch_valid_planes(alp_mesh_t *m, const double *offsets, uint32_t *idx_out, double *x_out, double *y_out, double *z_out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m, "mesh handle is NULL");
    if (m->valid_total < 0) return fail(ALP_ESTATE, "alp_render_fetch_valid_planes: call alp_render_valid_count first");
    const int64_t M = m->valid_total;
    m->valid_total = -1;
    if (M == 0) return ALP_OK;
    ALP_REQUIRE(idx_out && x_out && y_out && z_out, "output is NULL");
    char *dev = nullptr;
    const size_t plane = (size_t)M * sizeof(double), idx_bytes = (size_t)M * sizeof(unsigned);
    if (int rc = scratch_reserve(3 * plane + idx_bytes, (void **)&dev)) return rc;
    double *xyz_dev = (double *)dev;
    unsigned *idx_dev = (unsigned *)(dev + 3 * plane);
    hipStream_t st = ctx().stream;
    m->valid_total_planes = M;
    if (int rc = frame_valid_write(m, offsets, idx_dev, xyz_dev, true)) return rc;
    hipError_t e = hipMemcpyAsync(x_out, xyz_dev, plane, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(y_out, xyz_dev + M, plane, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(z_out, xyz_dev + 2 * M, plane, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(idx_out, idx_dev, idx_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(ALP_EHIP, "alp_render_fetch_valid_planes: %s", hipGetErrorString(e));
    return ALP_OK;
}

int alp_render_fetch_valid_table(alp_mesh_t *m, const double *offsets, const void *array, int array_dtype, int64_t channels,
                                 int64_t *index_out, int16_t *u_out, int16_t *v_out, double *block_out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m, "mesh handle is NULL");
    if (m->valid_total < 0) return fail(ALP_ESTATE, "alp_render_fetch_valid_table: call alp_render_valid_count first");
    const int64_t M = m->valid_total;
    m->valid_total = -1;
    if (M == 0) return ALP_OK;
    ALP_REQUIRE(index_out && u_out && v_out && block_out, "output is NULL");
    ALP_REQUIRE(channels >= 0 && channels <= 64, "channel count out of range");
    ALP_REQUIRE(channels == 0 || array, "array is NULL");
    ALP_REQUIRE(array_dtype == ALP_U8 || array_dtype == ALP_U16 || array_dtype == ALP_F32 || array_dtype == ALP_F64,
                "array_dtype must be ALP_U8, ALP_U16, ALP_F32 or ALP_F64");
    const size_t esize = array_dtype == ALP_U8 ? 1 : array_dtype == ALP_U16 ? 2 : array_dtype == ALP_F32 ? 4 : 8;
    const size_t npix = (size_t)m->w * m->h, arr_bytes = npix * (size_t)channels * esize;
    // block (x | y | z | channels: float64 rows of M) | labels (int64) | pixel index (u32) | u | v (int16) | the caller's array
    const size_t plane = (size_t)M * sizeof(double);
    const size_t block_bytes = (3 + (size_t)channels) * plane;
    const size_t small = (size_t)M * (8 + 4 + 2 + 2);
    char *dev = nullptr;
    if (int rc = scratch_reserve(block_bytes + small + 256 + arr_bytes + 256, (void **)&dev)) return rc;
    double *block_dev = (double *)dev;
    long long *index_dev = (long long *)(dev + block_bytes);
    unsigned *idx_dev = (unsigned *)(index_dev + M);
    short *u_dev = (short *)(idx_dev + M), *v_dev = u_dev + M;
    char *arr_dev = (char *)(((uintptr_t)(v_dev + M) + 255) & ~(uintptr_t)255);
    hipStream_t st = ctx().stream;
    if (arr_bytes)
        if (int rc = upload_chunked(arr_dev, array, arr_bytes)) return rc;
    m->valid_total_planes = M;
    if (int rc = frame_valid_write(m, offsets, idx_dev, block_dev, true)) return rc;
    {
        KTimeScope kt;
        const dim3 grid((unsigned)((M + 255) / 256));
#define ALP_COLUMNS(A) hipLaunchKernelGGL(table_columns_kernel<A>, grid, dim3(256), 0, st, idx_dev, (long long)M, m->w, (const A *)arr_dev, \
                                          (int)channels, index_dev, u_dev, v_dev, block_dev + 3 * M)
        if (array_dtype == ALP_U8) ALP_COLUMNS(unsigned char);
        else if (array_dtype == ALP_U16) ALP_COLUMNS(unsigned short);
        else if (array_dtype == ALP_F32) ALP_COLUMNS(float);
        else ALP_COLUMNS(double);
#undef ALP_COLUMNS
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(block_out, block_dev, block_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(index_out, index_dev, (size_t)M * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(u_out, u_dev, (size_t)M * 2, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(v_out, v_dev, (size_t)M * 2, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(ALP_EHIP, "alp_render_fetch_valid_table: %s", hipGetErrorString(e));
    return ALP_OK;
}

int alp_render(alp_mesh_t *m, const double params[ALP_NPARAM], const double *offsets, double min_distance, float *out) {
    if (int rc = alp_render_enqueue(m, params, offsets, min_distance)) return rc;
    return alp_render_fetch(m, out);
}

int alp_distort_image(const float *img, int64_t h, int64_t w, int64_t c, const double coeffs[14], float *out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(img && out && coeffs, "NULL argument");
    ALP_REQUIRE(h >= 1 && w >= 1 && c >= 1 && h <= 32768 && w <= 32768, "bad image shape");
    double p[ALP_NPARAM] = {0};
    for (int i = 0; i < 14; ++i) p[7 + i] = coeffs[i];
    p[3] = 60; p[21] = (double)w; p[22] = (double)h;
    View v;
    RemapCoef rc;
    make_view(p, nullptr, &v, &rc);
    const size_t bytes = (size_t)h * w * c * sizeof(float);
    float *dev = nullptr;
    if (int e2 = scratch_reserve(2 * bytes, (void **)&dev)) return e2;
    hipError_t e = hipMemcpyAsync(dev, img, bytes, hipMemcpyHostToDevice, ctx().stream);
    if (e == hipSuccess) {
        const long long want = ((long long)h * w + 255) / 256;
        const int grid = (int)(want < 4096 ? want : 4096);
        hipLaunchKernelGGL(distort_image_kernel, dim3(grid), dim3(256), 0, ctx().stream, dev, (int)w, (int)h, (int)c, rc,
                           (float *)((char *)dev + bytes));
        e = hipMemcpyAsync(out, (char *)dev + bytes, bytes, hipMemcpyDeviceToHost, ctx().stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);
    if (e != hipSuccess) return fail(ALP_EHIP, "alp_distort_image: %s", hipGetErrorString(e));
    return ALP_OK;
}

int alp_distort_map(int64_t h, int64_t w, const double coeffs[14], float *map_x, float *map_y) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(coeffs && map_x && map_y, "NULL argument");
    ALP_REQUIRE(h >= 1 && w >= 1 && h <= 32768 && w <= 32768, "bad image shape");
    double p[ALP_NPARAM] = {0};
    for (int i = 0; i < 14; ++i) p[7 + i] = coeffs[i];
    p[3] = 60; p[21] = (double)w; p[22] = (double)h;
    View v;
    RemapCoef rc;
    make_view(p, nullptr, &v, &rc);
    const size_t bytes = (size_t)h * w * sizeof(float);
    float *dev = nullptr;
    if (int e2 = scratch_reserve(2 * bytes, (void **)&dev)) return e2;
    const long long want = ((long long)h * w + 255) / 256;
    const int grid = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(distort_map_kernel, dim3(grid), dim3(256), 0, ctx().stream, (int)w, (int)h, rc, dev, dev + (size_t)h * w);
    ALP_HIP(hipGetLastError());
    ALP_HIP(hipMemcpyAsync(map_x, dev, bytes, hipMemcpyDeviceToHost, ctx().stream));
    ALP_HIP(hipMemcpyAsync(map_y, dev + (size_t)h * w, bytes, hipMemcpyDeviceToHost, ctx().stream));
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    return ALP_OK;
}

int alp_render_load(alp_mesh_t *m, const float *image, int64_t h, int64_t w) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m && image, "NULL argument");
    ALP_REQUIRE(h >= 1 && w >= 1 && h <= 32768 && w <= 32768, "image size w,h must be in [1, 32768]");
    if (m->unchecked) ALP_HIP(hipStreamSynchronize(ctx().stream));
    m->unchecked = false;
    if (int e = ensure_frame(m, (int)w, (int)h)) return e;
    if (int e = upload_chunked(m->image, image, (size_t)h * w * 3 * sizeof(float))) return e;
    ALP_HIP(hipMemsetAsync(m->vis, 0, (size_t)h * w * sizeof(unsigned long long), ctx().stream));   // no visibility belongs to it
    m->vis_current = false;
    m->rz_n = -1;
    m->rendered = true;
    m->valid_total = -1;
    return ALP_OK;
}

}  // extern "C"
