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
// Rasterisation rules OpenGL leaves to the implementation (sub-pixel snapping, tie-break on
// shared edges, depth-buffer precision) cannot be pinned against the reference's GL driver:
// see DESIGN.md "parity unpinned" -- the choices made are watertight and deterministic.
#include "alp_raster_internal.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
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
    defined(GRID_NO_XCD_SWIZZLE)
#define ALP_DEV_SWITCHES 1
#ifndef ALP_DEV
#error "development switch given without -DALP_DEV: this would build a library that renders wrong images"
#endif
#endif
#if defined(INLINE_LOG2) || defined(FAST_MAX) || defined(COOP_MIN_W) || defined(COOP_MIN_PIX) || defined(GT_W_LOG2) || defined(GT_H_LOG2) || \
    defined(HIZ_SPAN) || defined(GRID_WAVES_PER_EU) || defined(PATCH_MIN_FAST) || defined(PATCH_WORDS_NEAR) || defined(PATCH_WORDS_FAR) ||    \
    defined(RASTER_BLOCKS_PER_CU) || defined(RESOLVE_BLOCKS_PER_CU)
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

constexpr int SUB = 256;                     // sub-pixel units per pixel
constexpr float COORD_LIMIT = 4194304.0f;    // 2^22 px
constexpr int SMALL_PIXELS = 32;             // bbox pixel count finished inside raster_kernel
constexpr int TILE = 64;                     // work-item edge for large triangles

static void make_view(const double *p, const double *offsets, View *v, RemapCoef *rc) {
    double x = p[0], y = p[1], z = p[2];
    if (offsets) { x -= offsets[0]; y -= offsets[2]; z -= offsets[1]; }
    const double pi = M_PI;
    const double pan = (360 - p[4]) * pi / 180, tilt = p[5] * pi / 180, roll = p[6] * pi / 180;
    const double rx[3][3] = {{1, 0, 0}, {0, std::cos(tilt), -std::sin(tilt)}, {0, std::sin(tilt), std::cos(tilt)}};
    const double ry[3][3] = {{std::cos(pan), 0, std::sin(pan)}, {0, 1, 0}, {-std::sin(pan), 0, std::cos(pan)}};
    const double rz[3][3] = {{std::cos(roll), -std::sin(roll), 0}, {std::sin(roll), std::cos(roll), 0}, {0, 0, 1}};
    double t[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += rz[i][k] * rx[k][j];
            t[i][j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += t[i][k] * ry[k][j];
            v->R[i][j] = (float)s;
            v->Rd[i][j] = s;
        }
    const double cam[3] = {x, z, y};      // vertices are stored X, Z(up), Y
    for (int i = 0; i < 3; ++i) {
        v->camd[i] = cam[i];
        v->camf[i] = (float)cam[i];
        v->caml[i] = (float)(cam[i] - (double)v->camf[i]);
    }
    const double w = p[21], h = p[22];
    const double fov_x = p[3] * pi / 180, fov_y = fov_x * h / w;
    v->fxd = 1 / std::tan(fov_x / 2);
    v->fyd = 1 / std::tan(fov_y / 2);
    v->fx = (float)v->fxd;
    v->fy = (float)v->fyd;
    v->w = (int)w;
    v->h = (int)h;
    v->sx = 0.5f * (float)v->w;
    v->sy = 0.5f * (float)v->h;
    v->kx = 1.0 / (double)v->sx;
    v->ky = 1.0 / (double)v->sy;
    v->ifx = 1.0 / v->fxd;
    v->ify = 1.0 / v->fyd;
    if (rc) {
        rc->a1 = 1 / p[7]; rc->a2 = 1 / p[8];
        rc->k1 = -p[9]; rc->k2 = -p[10]; rc->k3 = -p[11]; rc->k4 = -p[12]; rc->k5 = -p[13]; rc->k6 = -p[14];
        rc->p1 = -p[15]; rc->p2 = -p[16]; rc->s1 = -p[17]; rc->s2 = -p[18]; rc->s3 = -p[19]; rc->s4 = -p[20];
        rc->c0 = (double)(float)((w - 1) / 2.0);
        rc->c1 = (double)(float)((h - 1) / 2.0);
    }
}

// ------------------------------------------------------------------ device helpers
#ifdef ALP_WG_TIMING        // development build: start / end time of every workgroup of raster_grid_kernel (100 MHz)
__device__ unsigned long long g_wgtime[8 * 131072];
#define WGT(k) do { if (threadIdx.x == 0 && blockIdx.x < 131072) g_wgtime[8 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WGT(k) ((void)0)
#endif
#ifdef ALP_RASTER_STATS     // development build: fragment / request census printed after every frame
__device__ unsigned long long g_rstat[24 + 8 * 8];

#define RSTAT(k, n) atomicAdd(&g_rstat[k], (unsigned long long)(n))
#else
#define RSTAT(k, n) ((void)0)
#endif
__device__ __forceinline__ void to_view(const View &v, float px, float py, float pz, float out[3]) {
    const float dx = (px - v.camf[0]) - v.caml[0];
    const float dy = (py - v.camf[1]) - v.caml[1];
    const float dz = (pz - v.camf[2]) - v.caml[2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        out[i] = __builtin_fmaf(v.R[i][0], dx, __builtin_fmaf(v.R[i][1], dy, v.R[i][2] * dz));
}

// The correctly rounded float32 reciprocal 1.0f / x in three instructions: v_rcp_f32 and one Newton step
// with fused multiply-adds.  On gfx950 this equals the IEEE division for EVERY mantissa (exhaustive check:
// tools/rcp_exact.hip, all 2^23 mantissas for exponents 0 .. 60; scaling by a power of two is exact in
// that range), so the specification's "IEEE division"  is met bit for bit at a
// third of the ~11 instructions of the generic expansion.  Outside [1, 2^60) the generic division runs.
__device__ __forceinline__ float exact_rcp_unchecked(float x) {       // x in [1, 2^60) -- or the result is not used
#if defined(__gfx950__)
    const float y = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(y, __builtin_fmaf(-x, y, 1.0f), y);
#else       // the exhaustive check covers this chip's v_rcp_f32 table only: anywhere else, the division itself
    return 1.0f / x;
#endif
}
__device__ __forceinline__ float exact_rcp(float x) {
    if (__builtin_expect(!(x >= 1.0f && x < 1.0e18f), 0)) return 1.0f / x;
    return exact_rcp_unchecked(x);
}

__device__ __forceinline__ void to_window(const View &v, const float q[3], float &xw, float &yw, float &iw) {
    const float i = exact_rcp(q[2]);
    iw = i;
    xw = __builtin_fmaf((v.fx * q[0]) * i, v.sx, v.sx);
    yw = __builtin_fmaf((v.fy * q[1]) * i, v.sy, v.sy);
}

__device__ __forceinline__ long long floor_div(long long a, long long b) {
    long long q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

struct Idx3 { long long a, b, c; };

template <bool IMPLICIT>
__device__ __forceinline__ Idx3 tri_vertices(const int *__restrict__ ind, long long gw, long long t) {
    Idx3 r;
    if constexpr (IMPLICIT) {
        // regular grid of src/alproj/surface.py:194-201: (a, a+gw, a+gw+1), (a, a+gw+1, a+1)
        // 32-bit arithmetic: fewer than 2^32 triangles, 2^31 vertices (a 64-bit division is ~5x the work)
        const unsigned cell = (unsigned)t >> 1, gc = (unsigned)gw - 1u, row = cell / gc, col = cell - row * gc;
        const long long a = (long long)(row * (unsigned)gw + col);
        r.a = a;
        r.b = (t & 1) ? a + gw + 1 : a + gw;
        r.c = (t & 1) ? a + 1 : a + gw + 1;
    } else {
        r.a = ind[t * 3 + 0];
        r.b = ind[t * 3 + 1];
        r.c = ind[t * 3 + 2];
    }
    return r;
}

// integer set-up of one window-space triangle
struct TriSetup {
    long long X[3], Y[3];
    long long area2;
    int i0, i1, j0, j1;      // pixel bbox (inclusive), already clamped to the viewport
    float iw[3];
    bool valid;
};

__device__ __forceinline__ int snap(float w) { return (int)__builtin_rintf(w * (float)SUB); }

// set-up from already snapped window coordinates
__device__ __forceinline__ TriSetup setup_snapped(const View &v, const int X[3], const int Y[3],
                                                  const float iw[3]) {
    TriSetup s;
    s.valid = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s.X[k] = X[k];
        s.Y[k] = Y[k];
        s.iw[k] = iw[k];
    }
    s.area2 = (s.X[1] - s.X[0]) * (s.Y[2] - s.Y[0]) - (s.X[2] - s.X[0]) * (s.Y[1] - s.Y[0]);
    if (s.area2 <= 0) return s;                                   // back face / degenerate
    long long minx = s.X[0], maxx = s.X[0], miny = s.Y[0], maxy = s.Y[0];
#pragma unroll
    for (int k = 1; k < 3; ++k) {
        minx = s.X[k] < minx ? s.X[k] : minx;
        maxx = s.X[k] > maxx ? s.X[k] : maxx;
        miny = s.Y[k] < miny ? s.Y[k] : miny;
        maxy = s.Y[k] > maxy ? s.Y[k] : maxy;
    }
    long long i0 = -floor_div(-(minx - SUB / 2), SUB), i1 = floor_div(maxx - SUB / 2, SUB);
    long long j0 = -floor_div(-(miny - SUB / 2), SUB), j1 = floor_div(maxy - SUB / 2, SUB);
    if (i0 < 0) i0 = 0;
    if (j0 < 0) j0 = 0;
    if (i1 > v.w - 1) i1 = v.w - 1;
    if (j1 > v.h - 1) j1 = v.h - 1;
    if (i0 > i1 || j0 > j1) return s;
    s.i0 = (int)i0; s.i1 = (int)i1; s.j0 = (int)j0; s.j1 = (int)j1;
    s.valid = true;
    return s;
}

__device__ __forceinline__ TriSetup setup_tri(const View &v, const float xw[3], const float yw[3],
                                              const float iw[3]) {
    const int X[3] = {snap(xw[0]), snap(xw[1]), snap(xw[2])};
    const int Y[3] = {snap(yw[0]), snap(yw[1]), snap(yw[2])};
    return setup_snapped(v, X, Y, iw);
}

// coverage + depth of pixel (i, j); returns 0 when the centre is not covered
__device__ __forceinline__ unsigned long long pixel_key(const TriSetup &s, int i, int j, unsigned tri) {
    const long long px = (long long)i * SUB + SUB / 2, py = (long long)j * SUB + SUB / 2;
    long long e[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3, b = (k + 2) % 3;
        const long long dx = s.X[b] - s.X[a], dy = s.Y[b] - s.Y[a];
        e[k] = dx * (py - s.Y[a]) - dy * (px - s.X[a]);
        if (e[k] < 0 || (e[k] == 0 && !(dy < 0 || (dy == 0 && dx > 0)))) return 0ull;
    }
    const float q = __builtin_fmaf((float)e[2], s.iw[2], __builtin_fmaf((float)e[1], s.iw[1], (float)e[0] * s.iw[0])) *
                    (1.0f / (float)s.area2);
    return ((unsigned long long)__float_as_uint(q) << 32) | (unsigned long long)(0xFFFFFFFFu - tri);
}

__device__ __forceinline__ void vis_max(unsigned long long *vis, const View &v, int i, int j, unsigned long long key) {
    unsigned long long *dst = vis + (unsigned)(__umul24((unsigned)j, (unsigned)v.w) + (unsigned)i);   // j, w <= 2^15
    // unconditional: a plain-load pre-test ("only if larger") measured SLOWER (3.35 vs 3.02 ms per
    // 100 M-vertex frame) -- the load serialises behind the atomic it was meant to save
#ifdef VIS_PLAIN_STORE          // development: the same address arithmetic without the atomic (wrong image)
    __builtin_nontemporal_store(key, dst);
#elif defined(VIS_NEVER)        // development: the arithmetic stays, the memory operation (almost) never happens
    if (key == 0x123456789ull) atomicMax(dst, key);
#else
    atomicMax(dst, key);
#endif
}

// every pixel centre of the (small) bounding box: the three edge functions are stepped
// incrementally in exact integer arithmetic (same values as pixel_key)
__device__ __forceinline__ void raster_bbox(const TriSetup &s, unsigned tri, unsigned long long *vis, const View &v) {
    long long dx[3], dy[3], row[3];
    bool tl[3];
    const long long px0 = (long long)s.i0 * SUB + SUB / 2, py0 = (long long)s.j0 * SUB + SUB / 2;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3, b = (k + 2) % 3;
        dx[k] = s.X[b] - s.X[a];
        dy[k] = s.Y[b] - s.Y[a];
        row[k] = dx[k] * (py0 - s.Y[a]) - dy[k] * (px0 - s.X[a]);
        tl[k] = dy[k] < 0 || (dy[k] == 0 && dx[k] > 0);
    }
    const float inv_area = 1.0f / (float)s.area2;
    for (int j = s.j0; j <= s.j1; ++j) {
        long long e0 = row[0], e1 = row[1], e2 = row[2];
        for (int i = s.i0; i <= s.i1; ++i) {
            const bool in0 = e0 > 0 || (e0 == 0 && tl[0]);
            const bool in1 = e1 > 0 || (e1 == 0 && tl[1]);
            const bool in2 = e2 > 0 || (e2 == 0 && tl[2]);
            if (in0 && in1 && in2) {
                const float q = __builtin_fmaf((float)e2, s.iw[2], __builtin_fmaf((float)e1, s.iw[1], (float)e0 * s.iw[0])) *
                                inv_area;
                vis_max(vis, v, i, j, ((unsigned long long)__float_as_uint(q) << 32) | (unsigned long long)(0xFFFFFFFFu - tri));
            }
            e0 -= dy[0] * SUB;
            e1 -= dy[1] * SUB;
            e2 -= dy[2] * SUB;
        }
        row[0] += dx[0] * SUB;
        row[1] += dx[1] * SUB;
        row[2] += dx[2] * SUB;
    }
}

// float64 homogeneous fallback for triangles beyond the fixed-point range
// (executed by a whole wave: lane l takes pixels l, l+64, ...)
__device__ void raster_big(const View &v, const float q[3][3], unsigned tri, unsigned long long *vis, int lane) {
    double xh[3], yh[3], wh[3];
    for (int k = 0; k < 3; ++k) {
        wh[k] = q[k][2];
        xh[k] = ((double)v.fx * q[k][0] + wh[k]) * v.sx;
        yh[k] = ((double)v.fy * q[k][1] + wh[k]) * v.sy;
    }
    const double det = xh[0] * (yh[1] * wh[2] - yh[2] * wh[1]) - yh[0] * (xh[1] * wh[2] - xh[2] * wh[1]) +
                       wh[0] * (xh[1] * yh[2] - xh[2] * yh[1]);
    if (!(det > 0)) return;
    const long long npix = (long long)v.w * v.h;
    for (long long p = lane; p < npix; p += 64) {
        {
            const int j = (int)(p / v.w), i = (int)(p - (long long)j * v.w);
            const double px = i + 0.5, py = j + 0.5;
            double e[3];
            bool inside = true;
            for (int k = 0; k < 3; ++k) {
                const int a = (k + 1) % 3, b = (k + 2) % 3;
                e[k] = px * (yh[a] * wh[b] - yh[b] * wh[a]) - py * (xh[a] * wh[b] - xh[b] * wh[a]) +
                       (xh[a] * yh[b] - xh[b] * yh[a]);
                if (!(e[k] > 0)) inside = false;
            }
            if (!inside) continue;
            const float qq = (float)((e[0] + e[1] + e[2]) / det);
            if (!(qq <= 1.0f)) continue;
            vis_max(vis, v, i, j, ((unsigned long long)__float_as_uint(qq) << 32) | (unsigned long long)(0xFFFFFFFFu - tri));
        }
    }
}


// One triangle -> up to two window-space triangles (near-plane clip).  Returns the count and
// fills xw/yw/iw[0..3] (fan around vertex 0); `big` when the fixed-point range is exceeded.
__device__ __forceinline__ int clip_project(const View &v, const float q[3][3], float xw[4], float yw[4],
                                            float iw[4], bool &big) {
    const bool in0 = q[0][2] >= 1.0f, in1 = q[1][2] >= 1.0f, in2 = q[2][2] >= 1.0f;
    const int nin = (int)in0 + (int)in1 + (int)in2;
    big = false;
    if (nin == 0) return 0;
    {   // all three vertices beyond one side plane of the frustum: nothing can reach the viewport
        bool r = true, l = true, t = true, b = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float cx = v.fx * q[k][0], cy = v.fy * q[k][1], cw = q[k][2];
            r = r && cx > cw;
            l = l && cx < -cw;
            t = t && cy > cw;
            b = b && cy < -cw;
        }
        if (r || l || t || b) return 0;
    }
    int np = 0;
    if (nin == 3) {
#pragma unroll
        for (int k = 0; k < 3; ++k) to_window(v, q[k], xw[k], yw[k], iw[k]);
        np = 3;
    } else {
        float poly[4][3];
        const bool in[3] = {in0, in1, in2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int n = (k + 1) % 3;
            if (in[k]) { poly[np][0] = q[k][0]; poly[np][1] = q[k][1]; poly[np][2] = q[k][2]; ++np; }
            if (in[k] != in[n]) {
                const float *pi_ = in[k] ? q[k] : q[n];
                const float *po = in[k] ? q[n] : q[k];
                const float t = (1.0f - pi_[2]) / (po[2] - pi_[2]);
                poly[np][0] = __builtin_fmaf(t, po[0] - pi_[0], pi_[0]);
                poly[np][1] = __builtin_fmaf(t, po[1] - pi_[1], pi_[1]);
                poly[np][2] = 1.0f;
                ++np;
            }
        }
        for (int k = 0; k < np; ++k) to_window(v, poly[k], xw[k], yw[k], iw[k]);
    }
    for (int k = 0; k < np; ++k)
        if (!(fabsf(xw[k]) < COORD_LIMIT) || !(fabsf(yw[k]) < COORD_LIMIT)) big = true;
    return np - 2;
}

template <bool IMPLICIT>
__device__ __forceinline__ void load_view_tri(const View &v, const float *__restrict__ vert,
                                              const int *__restrict__ ind, long long gw, long long t,
                                              float q[3][3]) {
    const Idx3 id = tri_vertices<IMPLICIT>(ind, gw, t);
    const long long ids[3] = {id.a, id.b, id.c};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float *p = vert + 3 * ids[k];
        to_view(v, p[0], p[1], p[2], q[k]);
    }
}

// One snapped window-space triangle (all vertices in front of the near plane, inside the
// fixed-point range): cheap bounding-box rejection, then either the 32-bit inline walk
// (triangles under 2^INLINE_LOG2/256 px), the 64-bit inline walk (<= SMALL_PIXELS centres) or
// 64x64-pixel work items for raster_large_kernel.  `sub` = index in the clip fan.
#ifndef INLINE_LOG2
#define INLINE_LOG2 14      // triangles below 2^INLINE_LOG2 / 256 px are finished inside the thread
#endif
// 24-bit multiply (full rate; v_mul_lo_u32 issues at a quarter of it): every product of the
// 32-bit set-up has factors below 2^15 (triangle extent < 2^14 sub-pixels, pixel centres inside
// its bounding box)
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

#ifndef FAST_MAX
#define FAST_MAX 4           // cells whose box holds at most FAST_MAX x FAST_MAX pixel centres take the cell fast path
#endif
#ifndef COOP_MIN_W
#define COOP_MIN_W 3        // bounding boxes at least this many pixel columns wide go to coop_raster
#endif
#ifndef COOP_MIN_PIX
#define COOP_MIN_PIX 9      // ... if they also hold at least this many pixel centres
#endif


// The inline walk of emit_snapped done by all 64 lanes of the wave on ONE triangle (arguments
// wave-uniform): lane = one pixel of an 8x8 block (8 consecutive pixels of a row = one 64-byte
// line of the visibility buffer), the blocks tile the bounding box.  A lane-per-triangle walk
// sends every fragment as its own memory-side request; here the fragments of a row segment
// leave in one.  Same integers and the same float32 depth expression as the inline walk.
__device__ __forceinline__ void coop_raster(const View &v, const int X[3], const int Y[3], const float iw3[3],
                                            unsigned t, unsigned long long *__restrict__ vis, int lane) {
    const int minx = min(X[0], min(X[1], X[2])), maxx = max(X[0], max(X[1], X[2]));
    const int miny = min(Y[0], min(Y[1], Y[2])), maxy = max(Y[0], max(Y[1], Y[2]));
    const int ci0 = max((minx + SUB / 2 - 1) >> 8, 0), ci1 = min((maxx - SUB / 2) >> 8, v.w - 1);
    const int cj0 = max((miny + SUB / 2 - 1) >> 8, 0), cj1 = min((maxy - SUB / 2) >> 8, v.h - 1);
    const int area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (X[2] - X[0]) * (Y[1] - Y[0]);
    int dx[3], dy[3], bias[3], xa[3], ya[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3, b = (k + 2) % 3;
        dx[k] = X[b] - X[a];
        dy[k] = Y[b] - Y[a];
        xa[k] = X[a];
        ya[k] = Y[a];
        bias[k] = (dy[k] < 0 || (dy[k] == 0 && dx[k] > 0)) ? 0 : 1;
    }
    const float inv_area = exact_rcp_unchecked((float)area2);
    const unsigned long long lo = (unsigned long long)(0xFFFFFFFFu - t);
    const int lx = lane & 7, ly = lane >> 3;
    for (int by = cj0; by <= cj1; by += 8)
        for (int bx = ci0 & ~7; bx <= ci1; bx += 8) {
            const int i = bx + lx, j = by + ly;
            if (i < ci0 || i > ci1 || j > cj1) continue;
            const int px = i * SUB + SUB / 2, py = j * SUB + SUB / 2;
            const int w0 = mul24(dx[0], py - ya[0]) - mul24(dy[0], px - xa[0]) - bias[0];
            const int w1 = mul24(dx[1], py - ya[1]) - mul24(dy[1], px - xa[1]) - bias[1];
            const int w2 = mul24(dx[2], py - ya[2]) - mul24(dy[2], px - xa[2]) - bias[2];
            if ((w0 | w1 | w2) >= 0) {
                RSTAT(7, 1);
                const float q = __builtin_fmaf((float)(w2 + bias[2]), iw3[2],
                                               __builtin_fmaf((float)(w1 + bias[1]), iw3[1],
                                                              (float)(w0 + bias[0]) * iw3[0])) * inv_area;
                vis_max(vis, v, i, j, ((unsigned long long)__float_as_uint(q) << 32) | lo);
            }
        }
}

// Wave-converged: rasterise the parked triangles of all lanes, one after the other.
__device__ __forceinline__ void coop_drain(const View &v, bool parked, const Deferred &d,
                                           unsigned long long *__restrict__ vis) {
    unsigned long long mask = __ballot(parked);
    const int lane = (int)(threadIdx.x & 63);
    while (mask) {
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        int X[3], Y[3];
        float iw3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            X[k] = __builtin_amdgcn_readlane(d.X[k], src);
            Y[k] = __builtin_amdgcn_readlane(d.Y[k], src);
            iw3[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d.iw[k]), src));
        }
        const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)d.t, src);
        if (lane == 0) RSTAT(8, 1);
        coop_raster(v, X, Y, iw3, t, vis, lane);
    }
}

enum { EMIT_DONE = 0, EMIT_PARKED = 1, EMIT_GENERAL = 2, EMIT_PARKED_SMALL = 3 };   // _SMALL: box of at most 8 x 8 centres (if asked for)

// One snapped window-space triangle, 32-bit part: bounding-box rejection, then -- for triangles
// under 64 px -- back-face test and the inline walk (or parking for coop_raster if may_park).
// Returns EMIT_GENERAL, having done nothing, for a larger triangle.
__device__ __forceinline__ int emit_small(const View &v, const int X[3], const int Y[3], const float *iwsrc,
                                          int n0, int n1, int n2, unsigned t, unsigned long long *__restrict__ vis,
                                          Deferred *park, bool may_park, int coop_min_w = COOP_MIN_W,
                                          int coop_min_pix = COOP_MIN_PIX, bool tell_small = false) {
    // bounding box without a pixel centre, or entirely outside the viewport
    const int minx = min(X[0], min(X[1], X[2])), maxx = max(X[0], max(X[1], X[2]));
    const int miny = min(Y[0], min(Y[1], Y[2])), maxy = max(Y[0], max(Y[1], Y[2]));
    const int i0 = (minx + SUB / 2 - 1) >> 8, i1 = (maxx - SUB / 2) >> 8;       // SUB == 256
    const int j0 = (miny + SUB / 2 - 1) >> 8, j1 = (maxy - SUB / 2) >> 8;
    if (i0 > i1 || j0 > j1 || i1 < 0 || j1 < 0 || i0 > v.w - 1 || j0 > v.h - 1) return EMIT_DONE;
    if (!(maxx - minx < (1 << INLINE_LOG2) && maxy - miny < (1 << INLINE_LOG2))) return EMIT_GENERAL;
    // triangle smaller than 64 px: every product of the set-up fits 32 bits when taken
    // relative to the first pixel centre -- the same integers as the 64-bit path
    const int area2 = mul24(X[1] - X[0], Y[2] - Y[0]) - mul24(X[2] - X[0], Y[1] - Y[0]);
    if (area2 <= 0) return EMIT_DONE;
    const int ci0 = max(i0, 0), ci1 = min(i1, v.w - 1), cj0 = max(j0, 0), cj1 = min(j1, v.h - 1);
    const float iw3[3] = {iwsrc[n0], iwsrc[n1], iwsrc[n2]};     // only now: most triangles never get here
    if (may_park && ci1 - ci0 + 1 >= coop_min_w && mul24(ci1 - ci0 + 1, cj1 - cj0 + 1) >= coop_min_pix) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            park->X[k] = X[k];
            park->Y[k] = Y[k];
            park->iw[k] = iw3[k];
        }
        park->t = t;
        return (tell_small && ci1 - ci0 < 8 && cj1 - cj0 < 8) ? EMIT_PARKED_SMALL : EMIT_PARKED;
    }
    const int px0 = ci0 * SUB + SUB / 2, py0 = cj0 * SUB + SUB / 2;
    // the tie rule is folded into the stepped value: w = e - (edge owns its boundary ? 0 : 1),
    // so "inside" is simply w0, w1, w2 >= 0 = sign bit of (w0 | w1 | w2)
    int dx[3], dy[3], row[3], bias[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3, b = (k + 2) % 3;
        dx[k] = X[b] - X[a];
        dy[k] = Y[b] - Y[a];
        bias[k] = (dy[k] < 0 || (dy[k] == 0 && dx[k] > 0)) ? 0 : 1;
        row[k] = mul24(dx[k], py0 - Y[a]) - mul24(dy[k], px0 - X[a]) - bias[k];
    }
    const float inv_area = exact_rcp_unchecked((float)area2);
    const unsigned long long lo = (unsigned long long)(0xFFFFFFFFu - t);
    RSTAT(2, 1);
    for (int j = cj0; j <= cj1; ++j) {
        int w0 = row[0], w1 = row[1], w2 = row[2];
        for (int i = ci0; i <= ci1; ++i) {
            if ((w0 | w1 | w2) >= 0) {
                RSTAT((ci1 - ci0) == 0 ? 3 : (ci1 - ci0) < 3 ? 4 : (ci1 - ci0) < 7 ? 5 : 6, 1);
                const float q = __builtin_fmaf((float)(w2 + bias[2]), iw3[2],
                                               __builtin_fmaf((float)(w1 + bias[1]), iw3[1],
                                                              (float)(w0 + bias[0]) * iw3[0])) * inv_area;
                vis_max(vis, v, i, j, ((unsigned long long)__float_as_uint(q) << 32) | lo);
            }
            w0 -= dy[0] * SUB;
            w1 -= dy[1] * SUB;
            w2 -= dy[2] * SUB;
        }
        row[0] += dx[0] * SUB;
        row[1] += dx[1] * SUB;
        row[2] += dx[2] * SUB;
    }
    return EMIT_DONE;
}

// One snapped window-space triangle of any size (all vertices in front of the near plane, inside
// the fixed-point range): emit_small, else the 64-bit set-up and either the 64-bit inline walk
// (<= SMALL_PIXELS centres) or 64x64-pixel work items for raster_large_kernel.  `sub` = index in
// the clip fan.
__device__ __forceinline__ void emit_snapped(const View &v, const int X[3], const int Y[3], const float iw3[3],
                                             long long t, int sub, unsigned long long *__restrict__ vis,
                                             WorkItem *__restrict__ queue, unsigned *__restrict__ qcount,
                                             unsigned qcap) {
    if (emit_small(v, X, Y, iw3, 0, 1, 2, (unsigned)t, vis, nullptr, false) != EMIT_GENERAL) return;
    const TriSetup s = setup_snapped(v, X, Y, iw3);
    if (!s.valid) return;
    const int bw = s.i1 - s.i0 + 1, bh = s.j1 - s.j0 + 1;
    if ((long long)bw * bh <= SMALL_PIXELS) {
        raster_bbox(s, (unsigned)t, vis, v);
    } else {
        for (int ty = s.j0 / TILE; ty <= s.j1 / TILE; ++ty)
            for (int tx = s.i0 / TILE; tx <= s.i1 / TILE; ++tx) {
                const unsigned slot = atomicAdd(qcount, 1u);
                if (slot < qcap)
                    queue[slot] = WorkItem{(unsigned)t, (unsigned short)sub, (unsigned short)tx, (unsigned short)ty, 0};
            }
    }
}

// The general path for one triangle given by its view-space vertices: near-plane clip, then
// emit_snapped per fan triangle (or a whole-triangle work item beyond the fixed-point range).
__device__ __forceinline__ void emit_general(const View &v, const float q[3][3], long long t,
                                             unsigned long long *__restrict__ vis, WorkItem *__restrict__ queue,
                                             unsigned *__restrict__ qcount, unsigned qcap) {
    float xw[4], yw[4], iw[4];
    bool big;
    const int ntri = clip_project(v, q, xw, yw, iw, big);
    if (ntri <= 0) return;
    if (big) {                       // rare: hand the whole triangle to the large pass
        const unsigned slot = atomicAdd(qcount, 1u);
        if (slot < qcap) queue[slot] = WorkItem{(unsigned)t, 0xFFFF, 0, 0, 0};
        return;
    }
    for (int f = 0; f < ntri; ++f) {
        const int X[3] = {snap(xw[0]), snap(xw[f + 1]), snap(xw[f + 2])};
        const int Y[3] = {snap(yw[0]), snap(yw[f + 1]), snap(yw[f + 2])};
        const float i3[3] = {iw[0], iw[f + 1], iw[f + 2]};
        emit_snapped(v, X, Y, i3, t, f, vis, queue, qcount, qcap);
    }
}

// ------------------------------------------------------------------ parked triangles of the grid kernel
// raster_grid_kernel does not rasterise the triangles emit_small parks (boxes of at least 3 columns
// and 9 centres, under 64 px): a tile next to the camera holds a thousand of them, and walking them
// one after the other inside the workgroup made those few workgroups the critical path of the whole
// frame (0.8 ms for a handful of tiles while the rest of the chip idled).  They are appended to two
// device queues instead and rasterised by their own launches, spread over every CU:
//   raster_coop4_body    boxes of at most 8 x 8 centres: FOUR triangles per wave, 16 lanes = a 4 x 4
//                        pixel block each (a 6 x 3 box costs two steps of a quarter wave instead of two
//                        steps of a whole one);
//   raster_coop_body     larger boxes: one triangle per wave, 8 x 8 pixel blocks (coop_raster).
// Same integers and the same float32 depth expression as the inline walk.
__device__ __forceinline__ void park_append(bool take, const Deferred &d, Deferred *__restrict__ queue,
                                            unsigned *__restrict__ count, unsigned cap) {
    const unsigned long long m = __ballot(take);
    if (!m) return;
    const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
#ifdef PARK_NOATOMIC      // development: no global counter (wrong image), to time its contention
    base = (blockIdx.x * 64u) % (cap - 64u);
#else
    if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(m));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
#endif
    const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    if (take && slot < cap) queue[slot] = d;       // an overflow is noticed by finish_frame (queue grown, frame redone)
}

__device__ __forceinline__ void raster_coop_body(const View &v, unsigned long long *__restrict__ vis,
                                                 const Deferred *__restrict__ queue,
                                                 const unsigned *__restrict__ count, unsigned cap) {
    const unsigned n = min(*count, cap);
    const int lane = (int)(threadIdx.x & 63);
    // workgroups go to the 8 XCDs round-robin: XCD x takes the x-th contiguous eighth of the queue
    // (neighbouring entries are neighbouring triangles: their pixels meet in one L2)
    const unsigned chunk = (n + 7u) >> 3, xcd = blockIdx.x & 7u, lo = xcd * chunk, hi = min(lo + chunk, n);
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(((blockIdx.x >> 3) * blockDim.x + threadIdx.x) >> 6));
    const unsigned nwaves = ((gridDim.x >> 3) * blockDim.x) >> 6;
    if (lo + wave >= hi) return;               // (returns from this body only: it is inlined into raster_parked_kernel)
    Deferred nextd = queue[lo + wave];         // wave-uniform address
    for (unsigned it = lo + wave; it < hi; it += nwaves) {
        const Deferred d = nextd;
        if (it + nwaves < hi) nextd = queue[it + nwaves];        // requested before this one is rasterised
        int X[3], Y[3];
        float iw3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            X[k] = __builtin_amdgcn_readfirstlane(d.X[k]);
            Y[k] = __builtin_amdgcn_readfirstlane(d.Y[k]);
            iw3[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(d.iw[k])));
        }
        coop_raster(v, X, Y, iw3, (unsigned)__builtin_amdgcn_readfirstlane((int)d.t), vis, lane);
    }
}

__device__ __forceinline__ void raster_coop4_body(const View &v, unsigned long long *__restrict__ vis,
                                                  const Deferred *__restrict__ queue,
                                                  const unsigned *__restrict__ count, unsigned cap) {
    const unsigned n = min(*count, cap);
    const unsigned chunk = (n + 7u) >> 3, xcd = blockIdx.x & 7u, lo = xcd * chunk, hi = min(lo + chunk, n);   // as in raster_coop_kernel
    const unsigned group = ((blockIdx.x >> 3) * blockDim.x + threadIdx.x) >> 4, ngroups = ((gridDim.x >> 3) * blockDim.x) >> 4;
    const int lx = (int)(threadIdx.x & 3), ly = (int)((threadIdx.x >> 2) & 3);
    if (lo + group >= hi) return;
    Deferred nextd = queue[lo + group];        // the 16 lanes of a group read the same entry
    for (unsigned it = lo + group; it < hi; it += ngroups) {
        const Deferred d = nextd;
        if (it + ngroups < hi) nextd = queue[it + ngroups];      // requested before this one is rasterised
        const int minx = min(d.X[0], min(d.X[1], d.X[2])), maxx = max(d.X[0], max(d.X[1], d.X[2]));
        const int miny = min(d.Y[0], min(d.Y[1], d.Y[2])), maxy = max(d.Y[0], max(d.Y[1], d.Y[2]));
        const int ci0 = max((minx + SUB / 2 - 1) >> 8, 0), ci1 = min((maxx - SUB / 2) >> 8, v.w - 1);
        const int cj0 = max((miny + SUB / 2 - 1) >> 8, 0), cj1 = min((maxy - SUB / 2) >> 8, v.h - 1);
        const int area2 = mul24(d.X[1] - d.X[0], d.Y[2] - d.Y[0]) - mul24(d.X[2] - d.X[0], d.Y[1] - d.Y[0]);
        int dx[3], dy[3], bias[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int a = (k + 1) % 3, b = (k + 2) % 3;
            dx[k] = d.X[b] - d.X[a];
            dy[k] = d.Y[b] - d.Y[a];
            bias[k] = (dy[k] < 0 || (dy[k] == 0 && dx[k] > 0)) ? 0 : 1;
        }
        const float inv_area = exact_rcp_unchecked((float)area2);
        const unsigned long long lo = (unsigned long long)(0xFFFFFFFFu - d.t);
        for (int by = cj0; by <= cj1; by += 4)
            for (int bx = ci0 & ~3; bx <= ci1; bx += 4) {
                const int i = bx + lx, j = by + ly;
                if (i < ci0 || i > ci1 || j > cj1) continue;
                const int px = i * SUB + SUB / 2, py = j * SUB + SUB / 2;
                const int w0 = mul24(dx[0], py - d.Y[1]) - mul24(dy[0], px - d.X[1]) - bias[0];
                const int w1 = mul24(dx[1], py - d.Y[2]) - mul24(dy[1], px - d.X[2]) - bias[1];
                const int w2 = mul24(dx[2], py - d.Y[0]) - mul24(dy[2], px - d.X[0]) - bias[2];
                if ((w0 | w1 | w2) >= 0) {
                    const float q = __builtin_fmaf((float)(w2 + bias[2]), d.iw[2],
                                                   __builtin_fmaf((float)(w1 + bias[1]), d.iw[1],
                                                                  (float)(w0 + bias[0]) * d.iw[0])) * inv_area;
                    vis_max(vis, v, i, j, ((unsigned long long)__float_as_uint(q) << 32) | lo);
                }
            }
    }
}

// Parked CELLS (box of at most 8 x 8 pixel centres, at least 3 columns and 9 centres): one cell per wave,
// lane = one pixel of the 8 x 8 window anchored at the box's first centre, so the whole cell is decided in
// ONE step; the cell's data are wave-uniform (scalar registers, scalar set-up).  Both triangles are
// decided per pixel from five shared edge functions -- the arithmetic of the FAST path of
// raster_grid_kernel, evaluated directly at the pixel instead of stepped -- and a pixel sends ONE atomic
// with the larger of its (at most two) keys: the row segments of both triangles of a cell travel in the
// same 64-byte line-requests (the chip serves ~23 G atomic line-requests/s; 16-lane groups stepping
// 8 x 2 blocks measured 130 M vector instructions for this stage, a wave per cell needs half).
__device__ __forceinline__ void raster_cell_body(const View &v, unsigned long long *__restrict__ vis,
                                                 const ParkedCell *__restrict__ queue,
                                                 const unsigned *__restrict__ count, unsigned cap) {
    const unsigned n = min(*count, cap);
    const int lane = (int)(threadIdx.x & 63), lx = lane & 7, ly = lane >> 3;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned it = wave; it < n; it += nwaves) {
        const ParkedCell *e = queue + it;          // wave-uniform address: scalar loads
        const int ax = e->X[0], bx_ = e->X[1], cx = e->X[2], dx_ = e->X[3];
        const int ay = e->Y[0], by_ = e->Y[1], cy = e->Y[2], dy_ = e->Y[3];
        const float iwa = e->iw[0], iwb = e->iw[1], iwc = e->iw[2], iwd = e->iw[3];
        const unsigned cell = e->cell;
        const int minx = min(min(ax, bx_), min(cx, dx_)), maxx = max(max(ax, bx_), max(cx, dx_));
        const int miny = min(min(ay, by_), min(cy, dy_)), maxy = max(max(ay, by_), max(cy, dy_));
        const int ci0 = max((minx + SUB / 2 - 1) >> 8, 0), ci1 = min((maxx - SUB / 2) >> 8, v.w - 1);
        const int cj0 = max((miny + SUB / 2 - 1) >> 8, 0), cj1 = min((maxy - SUB / 2) >> 8, v.h - 1);
        // directed edges: 0 b->c, 1 c->a, 2 a->b (triangle 0); 3 c->d, 4 d->a, 5 a->c (triangle 1)
        const int ex0 = cx - bx_, ex1 = ax - cx, ex2 = bx_ - ax, ex3 = dx_ - cx, ex4 = ax - dx_, ex5 = -ex1;
        const int ey0 = cy - by_, ey1 = ay - cy, ey2 = by_ - ay, ey3 = dy_ - cy, ey4 = ay - dy_, ey5 = -ey1;
        // the edge owns its boundary iff dy < 0 or (dy == 0 and dx > 0) iff (dy << 12) - dx < 0 (|dx| < 2^12)
        const int bs0 = 1 + (((ey0 << 12) - ex0) >> 31), bs1 = 1 + (((ey1 << 12) - ex1) >> 31), bs2 = 1 + (((ey2 << 12) - ex2) >> 31);
        const int bs3 = 1 + (((ey3 << 12) - ex3) >> 31), bs4 = 1 + (((ey4 << 12) - ex4) >> 31), bs5 = 1 + (((ey5 << 12) - ex5) >> 31);
        // doubled areas = sum of a triangle's three edge functions at any point (here: at b, resp. at c, where
        // two of the three vanish); every product has factors below 2^12
        const int area0 = ex1 * (by_ - cy) - ey1 * (bx_ - cx) + ex2 * (by_ - ay) - ey2 * (bx_ - ax);
        const int area1 = ex4 * (cy - dy_) - ey4 * (cx - dx_);
        const float inv0 = exact_rcp_unchecked((float)area0), inv1 = exact_rcp_unchecked((float)area1);   // used only where area > 0
        const unsigned long long lo0 = 0xFFFFFFFFu - 2u * cell, lo1 = lo0 - 1u;
        const int i = ci0 + lx, j = cj0 + ly;
        if (i > ci1 || j > cj1) continue;
        const int px = i * SUB + SUB / 2, py = j * SUB + SUB / 2;
        const int r0 = mul24(ex0, py - by_) - mul24(ey0, px - bx_);       // unbiased edge values
        const int r1 = mul24(ex1, py - cy) - mul24(ey1, px - cx);
        const int r2 = mul24(ex2, py - ay) - mul24(ey2, px - ax);
        const int r3 = mul24(ex3, py - cy) - mul24(ey3, px - cx);
        const int r4 = mul24(ex4, py - dy_) - mul24(ey4, px - dx_);
        const int r5 = -r1;
        unsigned long long key = 0;
        if (((r0 - bs0) | (r1 - bs1) | (r2 - bs2)) >= 0) {      // weights: edge k is opposite vertex k of (a, b, c)
            const float q = __builtin_fmaf((float)r2, iwc, __builtin_fmaf((float)r1, iwb, (float)r0 * iwa)) * inv0;
            key = ((unsigned long long)__float_as_uint(q) << 32) | lo0;
        }
        if (((r3 - bs3) | (r4 - bs4) | (r5 - bs5)) >= 0) {      // (a, c, d)
            const float q = __builtin_fmaf((float)r5, iwd, __builtin_fmaf((float)r4, iwc, (float)r3 * iwa)) * inv1;
            const unsigned long long k1 = ((unsigned long long)__float_as_uint(q) << 32) | lo1;
            key = k1 > key ? k1 : key;
        }
        if (key) vis_max(vis, v, i, j, key);
    }
}

// The three consumers of the parked work in ONE launch (three launches per round cost ~15 us of gaps):
// every wave takes its share of the cells, then of the large triangles, then of the small ones.
__global__ __launch_bounds__(256) void raster_parked_kernel(View v, unsigned long long *__restrict__ vis,
                                                            const Deferred *__restrict__ small_q, const Deferred *__restrict__ large_q,
                                                            const ParkedCell *__restrict__ cell_q,
                                                            const unsigned *__restrict__ counts, unsigned cap_small,
                                                            unsigned cap_large, unsigned cap_cell) {
#ifndef PARKED_SKIP_CELLS       // development: the stages one by one (wrong image)
    raster_cell_body(v, vis, cell_q, counts + 2, cap_cell);
#endif
#ifndef PARKED_SKIP_COOP
    raster_coop_body(v, vis, large_q, counts + 1, cap_large);
#endif
#ifndef PARKED_SKIP_COOP4
    raster_coop4_body(v, vis, small_q, counts + 0, cap_small);
#endif
}

// ------------------------------------------------------------------ kernel 2: per-triangle raster
// One thread per triangle, three gathered vertices.  Like raster_grid_kernel it finishes only the
// common case itself (all vertices in front and in range, under 64 px) and sets the rest aside
// for raster_general_kernel.
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void raster_kernel(const float *__restrict__ vert, const int *__restrict__ ind,
                                                     const unsigned char *__restrict__ valid,
                                                     long long n_tri, long long gw, View v,
                                                     unsigned long long *__restrict__ vis,
                                                     unsigned *__restrict__ gqueue, unsigned *__restrict__ gcount,
                                                     unsigned gcap) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long rounds = (n_tri + stride - 1) / stride;       // every lane makes every round: coop_drain is wave-wide
    for (long long k = 0; k < rounds; ++k) {
        const long long t = k * stride + (long long)blockIdx.x * blockDim.x + threadIdx.x;
        Deferred park;
        int code = EMIT_DONE;
        bool draw = t < n_tri;
        if (draw && valid) {
            const Idx3 id = tri_vertices<IMPLICIT>(ind, gw, t);
            draw = valid[id.a] && valid[id.b] && valid[id.c];
        }
        if (draw) {
            float q[3][3];
            load_view_tri<IMPLICIT>(v, vert, ind, gw, t, q);
            const bool in0 = q[0][2] >= 1.0f, in1 = q[1][2] >= 1.0f, in2 = q[2][2] >= 1.0f;
            if (in0 && in1 && in2) {
                float xw[3], yw[3], iw[3];
                bool ok = true;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    to_window(v, q[c], xw[c], yw[c], iw[c]);
                    ok = ok && fabsf(xw[c]) < COORD_LIMIT && fabsf(yw[c]) < COORD_LIMIT;
                }
                if (ok) {
                    const int X[3] = {snap(xw[0]), snap(xw[1]), snap(xw[2])};
                    const int Y[3] = {snap(yw[0]), snap(yw[1]), snap(yw[2])};
                    // without the cell fast path in front of it, parking pays from 4 columns / 16 centres (measured)
                    code = emit_small(v, X, Y, iw, 0, 1, 2, (unsigned)t, vis, &park, true, 4, 16);
                } else {
                    code = EMIT_GENERAL;
                }
            } else if (in0 || in1 || in2) {
                code = EMIT_GENERAL;
            }
            if (code == EMIT_GENERAL) {
                const unsigned slot = atomicAdd(gcount, 1u);
                if (slot < gcap) gqueue[slot] = (unsigned)t;
            }
        }
        coop_drain(v, code == EMIT_PARKED, park, vis);
    }
}

// The triangles raster_grid_kernel set aside (near-plane crossings, 64 px and more): one thread
// per entry of the general queue, the same path as raster_kernel.  The entry count is read on
// the device, so no host round trip separates the passes.
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void raster_general_kernel(const float *__restrict__ vert,
                                                             const int *__restrict__ ind, long long gw, View v,
                                                             unsigned long long *__restrict__ vis,
                                                             const unsigned *__restrict__ gqueue,
                                                             const unsigned *__restrict__ gcount, unsigned gcap,
                                                             WorkItem *__restrict__ queue,
                                                             unsigned *__restrict__ qcount, unsigned qcap) {
    const unsigned n = min(*gcount, gcap);
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned it = blockIdx.x * blockDim.x + threadIdx.x; it < n; it += stride) {
        const long long t = gqueue[it];
        float q[3][3];
        load_view_tri<IMPLICIT>(v, vert, ind, gw, t, q);
        emit_general(v, q, t, vis, queue, qcount, qcap);
    }
}

// ------------------------------------------------------------------ kernel 2b: implicit grid, LDS-tiled
// One workgroup = a tile of GT_W x GT_H grid cells (64 x 16 = 1024 cells, four per thread).
//   Phase 0  the tile comes from a list made by the frame plan (tile_plan_kernel, one lane per
//            tile): tiles whose bounding box (precomputed once per mesh: tile_bounds_kernel) lies
//            entirely beyond a side plane of the frustum or behind the near plane are not listed at
//            all -- conservatively (margins far above float32 rounding): such a tile draws nothing
//            in the exact path either, every one of its triangles is dropped by step 3 of the
//            specification or has no pixel centre inside the viewport.  FAR tiles (cells under about
//            a pixel) are listed for a second launch, after tile_occlusion_kernel has dropped those
//            that the depth pyramid of the first round proves to be hidden.
//   Phase 1  transforms, projects and snaps the (GT_W+1) x (GT_H+1) vertices of the tile ONCE
//            into LDS (raster_kernel does it 6 times per vertex).
//   Phase 2  classifies the cells (one lane per cell, four rounds): no pixel centre / outside
//            the viewport -> nothing; box of at most FAST_MAX x FAST_MAX centres -> FAST queue;
//            anything else -> SLOW queue.  Both queues live in LDS (cell ids, 2 bytes).
//   Phase 3  the FAST queue, 64 entries per wave: both triangles of a cell decided at once from
//            five shared edge functions.  In the far field only 1-3 % of the cells hold a pixel
//            centre; compacting them means ONE wave of a workgroup runs this (the most expensive)
//            stage for the whole tile instead of sixteen waves running it for one or two lanes each.
//   Phase 4  the SLOW queue: per triangle emit_small (inline walk, or parking: appended to the
//            device queues of raster_coop4_kernel / raster_coop_kernel), rare cases recorded in the
//            global general queue.
// Same integers, same tie rule, same float32 depth expression as the per-triangle path.
#ifndef GT_W_LOG2
#define GT_W_LOG2 6
#endif
#ifndef GT_H_LOG2
#define GT_H_LOG2 4
#endif
constexpr int GT_W = 1 << GT_W_LOG2, GT_H = 1 << GT_H_LOG2, GT_VW = GT_W + 1, GT_VH = GT_H + 1, GT_NV = GT_VW * GT_VH,
              GT_NC = GT_W * GT_H;
static_assert(GT_NC % 256 == 0 && GT_NC <= 65536, "tile size");

// idx / GT_VW for idx < GT_NV as a 24-bit multiply and a shift
constexpr int GT_DIV_SHIFT = 18;
constexpr int GT_DIV_MAGIC = ((1 << GT_DIV_SHIFT) + GT_VW - 1) / GT_VW;
constexpr bool gt_div_ok() {
    for (int i = 0; i < GT_NV; ++i)
        if (((i * GT_DIV_MAGIC) >> GT_DIV_SHIFT) != i / GT_VW) return false;
    return (long long)GT_NV * GT_DIV_MAGIC < (1ll << 31) && GT_DIV_MAGIC < (1 << 23);
}
static_assert(gt_div_ok(), "magic division");

// per-tile bounding boxes of an implicit-grid mesh: centre and half extent per axis (6 floats)
__global__ __launch_bounds__(256) void tile_bounds_kernel(const float *__restrict__ vert, int gh, int gw, int tiles_x,
                                                          float *__restrict__ bounds) {
    __shared__ float s_min[4][3], s_max[4][3];
    const int tile_r = blockIdx.x / tiles_x, tile_c = blockIdx.x - tile_r * tiles_x;
    const int r0 = tile_r * GT_H, c0 = tile_c * GT_W;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int idx = threadIdx.x; idx < GT_NV; idx += 256) {
        const int lr = idx / GT_VW, lc = idx - lr * GT_VW;
        const int r = r0 + lr, c = c0 + lc;
        if (r < gh && c < gw) {
            const float *p = vert + 3ull * ((unsigned)r * (unsigned)gw + (unsigned)c);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(lo[k], p[k]);       // a NaN coordinate is ignored here; such a vertex fails every test later
                hi[k] = fmaxf(hi[k], p[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int m = 32; m >= 1; m >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], m, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], m, 64));
        }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) {
            s_min[threadIdx.x >> 6][k] = lo[k];
            s_max[threadIdx.x >> 6][k] = hi[k];
        }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        const float a = fminf(fminf(s_min[0][k], s_min[1][k]), fminf(s_min[2][k], s_min[3][k]));
        const float b = fmaxf(fmaxf(s_max[0][k], s_max[1][k]), fmaxf(s_max[2][k], s_max[3][k]));
        bounds[6 * blockIdx.x + k] = 0.5f * a + 0.5f * b;
        bounds[6 * blockIdx.x + 3 + k] = (0.5f * b - 0.5f * a) * 1.000001f + 1e-30f;
    }
}

// the five planes of the tile test, in the stored (X, Z, Y) frame relative to the camera:
//   0..3  right, left, top, bottom: a point with n . d > 0 is beyond the plane (fx vx > vz, ...)
//   4     near: a point with n . d < 1 is behind it
// an = |n|; am = the component-wise magnitude bound of the float32 expressions the exact path
// evaluates (fx |R0| + |R2|, ...), which scales the safety margin.
struct TileCull {
    float cam[3];
    float n[5][3], an[5][3], am[5][3];
    float R[3][3];            // view rotation (rows x, y, z)
    float fx, fy, sx, sy;     // as in View
    float near_limit;         // a tile is "near" (drawn in the first round) when vz_min < near_limit * cell size
    int w, h;
    int enabled, occlusion;
};

static void make_tile_cull(const View &v, TileCull *c) {
    for (int i = 0; i < 3; ++i) c->cam[i] = (float)((double)v.camf[i] + (double)v.caml[i]);
    const double f[2] = {(double)v.fx, (double)v.fy};
    for (int k = 0; k < 5; ++k)
        for (int i = 0; i < 3; ++i) {
            double n, am;
            if (k < 4) {
                const int axis = k >> 1;                          // 0: x (right / left), 1: y (top / bottom)
                const double sgn = (k & 1) ? -1.0 : 1.0;
                n = sgn * f[axis] * (double)v.R[axis][i] - (double)v.R[2][i];
                am = f[axis] * std::fabs((double)v.R[axis][i]) + std::fabs((double)v.R[2][i]);
            } else {
                n = (double)v.R[2][i];
                am = std::fabs(n);
            }
            c->n[k][i] = (float)n;
            c->an[k][i] = (float)std::fabs(n) * 1.000001f;
            c->am[k][i] = (float)am * 1.000001f;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c->R[i][j] = v.R[i][j];
    c->fx = v.fx; c->fy = v.fy; c->sx = v.sx; c->sy = v.sy;
    c->w = v.w; c->h = v.h;
    // cells that project to about a pixel or more: vz < focal length in pixels x cell size
    const double focal_px = std::fmax((double)v.fx * v.sx, (double)v.fy * v.sy);
    double near_px = 0.75;     // measured on the 100 M-vertex frame: 0.5 1.17 ms, 0.75 1.11, 1.0 1.20, 1.5 1.22, 2.5 1.36
    if (const char *e = dev_getenv("ALP_NEAR_PX")) near_px = atof(e);      // development: where the first round ends
    c->near_limit = (float)(focal_px * near_px);
    c->enabled = 1;
    c->occlusion = 1;
}

// ---- frame plan, one lane per tile: drop the tiles outside the frustum, split the rest into the NEAR
// list (drawn first: the occluders) and the FAR list (tested against the depth pyramid of the first
// round before they are drawn).  counts[0] = near, counts[1] = far.  Wave-aggregated appends keep the
// lists roughly in tile order.
// Workgroup-aggregated append (all 256 threads call it): ONE atomicAdd per workgroup and list -- a
// reservation per wave made the two list counters the cost of these tiny kernels (1500 same-address
// atomics: 23 us for tile_plan_kernel).  The order inside the list follows the thread order.
__device__ __forceinline__ void list_append(bool take, unsigned value, unsigned *__restrict__ list, unsigned *count) {
    __shared__ unsigned s_cnt[4], s_base;
    const unsigned long long m = __ballot(take);
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    if (lane == 0) s_cnt[wave] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_base = total ? atomicAdd(count, total) : 0u;
    }
    __syncthreads();
    unsigned before = 0;
    for (int w = 0; w < wave; ++w) before += s_cnt[w];
    if (take) list[s_base + before + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = value;
    __syncthreads();         // s_cnt / s_base are reused by the next call
}

// The screen rectangle of a FAR tile, for the occlusion test: the tile's bounding box is projected (its eight
// corners lie in front of the camera: vz_min >= 2) and the rectangle widened by two pixels (float32 rounding,
// 1/256-pixel snapping), clamped to the viewport.  false: no usable rectangle (the tile is kept untested).
// tile_plan_kernel and tile_occlusion_kernel must see the SAME rectangle: the pyramid is only built where
// the plan said rectangles lie.
__device__ __forceinline__ bool far_tile_rect(const float *__restrict__ tb, const TileCull &cull, int &px0, int &px1, int &py0,
                                              int &py1, float &zmin) {
    const float c[3] = {tb[0] - cull.cam[0], tb[1] - cull.cam[1], tb[2] - cull.cam[2]};
    const float e[3] = {tb[3], tb[4], tb[5]};
    float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    zmin = INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float d0 = c[0] + ((k & 1) ? e[0] : -e[0]), d1 = c[1] + ((k & 2) ? e[1] : -e[1]), d2 = c[2] + ((k & 4) ? e[2] : -e[2]);
        const float vx = cull.R[0][0] * d0 + cull.R[0][1] * d1 + cull.R[0][2] * d2;
        const float vy = cull.R[1][0] * d0 + cull.R[1][1] * d1 + cull.R[1][2] * d2;
        const float vz = cull.R[2][0] * d0 + cull.R[2][1] * d1 + cull.R[2][2] * d2;
        const float iz = 1.0f / vz;
        const float xw = (cull.fx * vx * iz + 1.0f) * cull.sx, yw = (cull.fy * vy * iz + 1.0f) * cull.sy;
        x0 = fminf(x0, xw); x1 = fmaxf(x1, xw);
        y0 = fminf(y0, yw); y1 = fmaxf(y1, yw);
        zmin = fminf(zmin, vz);
    }
    // zmin >= 2 by construction of the far list (up to rounding: re-checked, NaN gives no rectangle)
    if (!(zmin >= 1.5f && x1 - x0 < 2048.0f && y1 - y0 < 2048.0f)) return false;
    // pixels whose centres can be touched: [x0 - 2, x1 + 2] clamped to the viewport
    px0 = max((int)floorf(x0 - 2.0f), 0);
    px1 = min((int)floorf(x1 + 2.0f), cull.w - 1);
    py0 = max((int)floorf(y0 - 2.0f), 0);
    py1 = min((int)floorf(y1 + 2.0f), cull.h - 1);
    return true;
}

// max over the wave, result in lane 63 (DPP inside the rows of 16 lanes, then row broadcasts)
__device__ __forceinline__ unsigned wave_max_to_lane63(unsigned x) {
#define ALP_STEP(CTRL, ROWS) x = max(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROWS, 0xf, false));
    ALP_STEP(0xB1, 0xf)     // quad_perm [1,0,3,2]
    ALP_STEP(0x4E, 0xf)     // quad_perm [2,3,0,1]
    ALP_STEP(0x141, 0xf)    // row_half_mirror
    ALP_STEP(0x140, 0xf)    // row_mirror: every lane holds its row's maximum
    ALP_STEP(0x142, 0xa)    // row_bcast15 into rows 1 and 3
    ALP_STEP(0x143, 0xc)    // row_bcast31 into rows 2 and 3: lane 63 holds the wave's
#undef ALP_STEP
    return x;
}

// `region` (four words, zero when the frame starts): the union of the FAR tiles' rectangles as maxima --
// 65535 - first column, last column + 1, 65535 - first row, last row + 1 -- for hiz_build_kernel.
__global__ __launch_bounds__(256) void tile_plan_kernel(const float *__restrict__ tile_bounds, unsigned n_tiles, TileCull cull,
                                                        unsigned *__restrict__ near_list, unsigned *__restrict__ far_list,
                                                        unsigned *__restrict__ counts, unsigned *__restrict__ region) {
    __shared__ unsigned s_region[4];
    if (threadIdx.x < 4) s_region[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    int kind = 0;                       // 0 dropped, 1 near, 2 far
    unsigned reg[4] = {0u, 0u, 0u, 0u};
    if (t < n_tiles) {
        kind = 1;
        if (cull.enabled) {
            const float *tb = tile_bounds + 6ull * t;
            const float d0 = tb[0] - cull.cam[0], d1 = tb[1] - cull.cam[1], d2 = tb[2] - cull.cam[2];
            const float e0 = tb[3], e1 = tb[4], e2 = tb[5];
            const float a0 = fabsf(d0) + e0, a1 = fabsf(d1) + e1, a2 = fabsf(d2) + e2;
            // absolute part of the margin: d is a float32 difference of a float32 box centre and the float32-rounded
            // camera position, each off by up to half an ulp of its MAGNITUDE (0.03 m at coordinates of 1e6 without
            // offsets), which the margin relative to |d| does not see: 4e-7 (> 3 ulp) of |centre| + |camera|
            const float g0 = fabsf(tb[0]) + fabsf(cull.cam[0]), g1 = fabsf(tb[1]) + fabsf(cull.cam[1]), g2 = fabsf(tb[2]) + fabsf(cull.cam[2]);
            bool outside = false;
            float vz_min = 0.0f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float sd = cull.n[k][0] * d0 + cull.n[k][1] * d1 + cull.n[k][2] * d2;
                const float rr = cull.an[k][0] * e0 + cull.an[k][1] * e1 + cull.an[k][2] * e2;
                const float mg = 1e-5f * (cull.am[k][0] * a0 + cull.am[k][1] * a1 + cull.am[k][2] * a2) +
                                 4e-7f * (cull.am[k][0] * g0 + cull.am[k][1] * g1 + cull.am[k][2] * g2);
                if (k < 4) outside = outside || (sd - rr > mg);                 // every point beyond a side plane
                else {
                    outside = outside || (sd + rr < 1.0f - mg - 1e-5f);        // every point behind the near plane
                    vz_min = sd - rr - mg;                                      // lower bound of the view depth in the tile
                }
            }
            if (outside) kind = 0;
            else if (cull.occlusion) {
                const float cell = fmaxf(2.0f * e0 / (float)GT_W, 2.0f * e2 / (float)GT_H);
                kind = (vz_min >= 2.0f && vz_min >= cull.near_limit * cell) ? 2 : 1;
                int px0, px1, py0, py1;
                float zmin;
                if (kind == 2 && far_tile_rect(tb, cull, px0, px1, py0, py1, zmin) && px0 <= px1 && py0 <= py1) {
                    reg[0] = 65535u - (unsigned)px0;
                    reg[1] = (unsigned)px1 + 1u;
                    reg[2] = 65535u - (unsigned)py0;
                    reg[3] = (unsigned)py1 + 1u;
                }
            }
        }
    }
    if (__ballot(reg[1] != 0u)) {          // wave-uniform
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned m = wave_max_to_lane63(reg[k]);
            if ((threadIdx.x & 63) == 63) atomicMax(&s_region[k], m);
        }
    }
    list_append(kind == 1, t, near_list, counts + 0);
    list_append(kind == 2, t, far_list, counts + 1);     // (its barriers also order s_region)
    if (threadIdx.x < 4 && s_region[threadIdx.x]) atomicMax(region + threadIdx.x, s_region[threadIdx.x]);
}

// ---- depth pyramid of the visibility buffer after the first round.  Level L holds, per block of
// (8 << L) x (8 << L) pixels, the SMALLEST float32 1/vz among the block's pixels inside the viewport
// (0 where a pixel is still empty): whatever is drawn later with a strictly smaller 1/vz everywhere in
// the block cannot win a single pixel there (the visibility word only grows; equal depth is not
// "strictly smaller", so the lower-triangle-index tie rule is never pre-empted).
#ifndef HIZ_SPAN
#define HIZ_SPAN 8          // the occlusion test reads up to HIZ_SPAN x HIZ_SPAN texels of the finest level that covers the rectangle with them
#endif                      // (2: 14 330 of 58 934 FAR tiles survive, 8: 12 430; the second round 113 -> 104 us, the test 7 -> 12 us)
constexpr int HIZ_LEVELS = 4;       // blocks of 8, 16, 32, 64 pixels

struct HizDims { int w[HIZ_LEVELS], h[HIZ_LEVELS]; long long off[HIZ_LEVELS]; };

static HizDims hiz_dims(int w, int h) {
    HizDims d;
    long long off = 0;
    for (int l = 0; l < HIZ_LEVELS; ++l) {
        const int b = 8 << l;
        d.w[l] = (w + b - 1) / b;
        d.h[l] = (h + b - 1) / b;
        d.off[l] = off;
        off += (long long)d.w[l] * d.h[l];
    }
    return d;
}
static long long hiz_total(int w, int h) {
    const HizDims d = hiz_dims(w, h);
    return d.off[HIZ_LEVELS - 1] + (long long)d.w[HIZ_LEVELS - 1] * d.h[HIZ_LEVELS - 1];
}

// one workgroup per 64 x 64 pixels: levels 0..3
// -- only where FAR tiles can look: the union of their rectangles (tile_plan_kernel), rounded outwards to the
// 64-pixel blocks of the top level written here, so that every texel the occlusion test can read is complete;
// the far field is a band under the horizon, the rest of the 168 MB buffer is not read (35 -> 13 us per
// 100 M-vertex frame)
__global__ __launch_bounds__(256) void hiz_build_kernel(const unsigned long long *__restrict__ vis, int w, int h, HizDims dm,
                                                        unsigned *__restrict__ hiz, const unsigned *__restrict__ region) {
    __shared__ unsigned s_min[64 + 16 + 4 + 1];
    const int rx = blockIdx.x * 64, ry = blockIdx.y * 64;
    {
        const unsigned r0 = region[0], r1 = region[1], r2 = region[2], r3 = region[3];
        if (r1 == 0u || r3 == 0u) return;                              // no FAR tile has a rectangle
        const int X0 = (int)(65535u - r0) & ~63, X1 = (int)(r1 - 1u) | 63, Y0 = (int)(65535u - r2) & ~63, Y1 = (int)(r3 - 1u) | 63;
        if (rx + 63 < X0 || rx > X1 || ry + 63 < Y0 || ry > Y1) return;
    }
    if (threadIdx.x < 85) s_min[threadIdx.x] = 0x7F800000u;       // +inf: no pixel of the viewport in the block yet
    __syncthreads();
    const int col = threadIdx.x & 63;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int row = (threadIdx.x >> 6) + 4 * k;
        const int x = rx + col, y = ry + row;
        if (x < w && y < h) {
            const unsigned q = (unsigned)(vis[(size_t)y * w + x] >> 32);      // float32 bits of 1/vz (positive: ordered as integers)
            // 8 lanes share a block; one LDS atomic per lane is fine here (21 M pixels, ~30 us)
            atomicMin(&s_min[(row >> 3) * 8 + (col >> 3)], q);
        }
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int bx = threadIdx.x & 3, by = threadIdx.x >> 2;
        unsigned m = 0x7F800000u;
        for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 2; ++i) m = min(m, s_min[(2 * by + j) * 8 + 2 * bx + i]);
        s_min[64 + threadIdx.x] = m;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int bx = threadIdx.x & 1, by = threadIdx.x >> 1;
        unsigned m = 0x7F800000u;
        for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 2; ++i) m = min(m, s_min[64 + (2 * by + j) * 4 + 2 * bx + i]);
        s_min[80 + threadIdx.x] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_min[84] = min(min(s_min[80], s_min[81]), min(s_min[82], s_min[83]));
    __syncthreads();
    // write the texels of this region that exist in each level
    if (threadIdx.x < 64) {
        const int tx = blockIdx.x * 8 + (threadIdx.x & 7), ty = blockIdx.y * 8 + (threadIdx.x >> 3);
        if (tx < dm.w[0] && ty < dm.h[0]) hiz[dm.off[0] + (long long)ty * dm.w[0] + tx] = s_min[threadIdx.x];
    } else if (threadIdx.x < 80) {
        const int k = threadIdx.x - 64, tx = blockIdx.x * 4 + (k & 3), ty = blockIdx.y * 4 + (k >> 2);
        if (tx < dm.w[1] && ty < dm.h[1]) hiz[dm.off[1] + (long long)ty * dm.w[1] + tx] = s_min[threadIdx.x];
    } else if (threadIdx.x < 84) {
        const int k = threadIdx.x - 80, tx = blockIdx.x * 2 + (k & 1), ty = blockIdx.y * 2 + (k >> 1);
        if (tx < dm.w[2] && ty < dm.h[2]) hiz[dm.off[2] + (long long)ty * dm.w[2] + tx] = s_min[threadIdx.x];
    } else if (threadIdx.x == 84) {
        hiz[dm.off[3] + (long long)blockIdx.y * dm.w[3] + blockIdx.x] = s_min[84];
    }
}

// ---- occlusion test of the FAR tiles, one lane per tile: the tile's bounding box is projected
// (its eight corners lie in front of the camera: vz_min >= 2), the screen rectangle is widened by two
// pixels (float32 rounding, 1/256-pixel snapping), and the largest 1/vz anything in the tile can reach
// (1 / vz_min, with margin) is compared with the pyramid texels under the rectangle.  Survivors are
// appended to the list of the second round.
__global__ __launch_bounds__(256) void tile_occlusion_kernel(const float *__restrict__ tile_bounds, TileCull cull,
                                                             const unsigned *__restrict__ far_list,
                                                             const unsigned *__restrict__ counts, HizDims dm,
                                                             const unsigned *__restrict__ hiz, unsigned *__restrict__ out_list,
                                                             unsigned *__restrict__ out_count) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned n = counts[1];
    bool keep = false;
    unsigned t = 0;
    if (i < n) {
        t = far_list[i];
        keep = true;
        int px0, px1, py0, py1;
        float zmin;
        if (far_tile_rect(tile_bounds + 6ull * t, cull, px0, px1, py0, py1, zmin)) {
            if (px0 > px1 || py0 > py1) {
                keep = false;                       // nothing of it can reach the viewport
            } else {
                // >= every interpolated float32 1/vz of the tile: 2e-5 relative, and the box corners' own uncertainty
                // (float32 centre and camera, see tile_plan_kernel) taken off the depth first
                const float *tb = tile_bounds + 6ull * t;
                const float zabs = 4e-7f * (fabsf(cull.R[2][0]) * (fabsf(tb[0]) + fabsf(cull.cam[0])) + fabsf(cull.R[2][1]) * (fabsf(tb[1]) + fabsf(cull.cam[1])) +
                                            fabsf(cull.R[2][2]) * (fabsf(tb[2]) + fabsf(cull.cam[2])));
                const float qmax = (1.0f / (zmin - zabs)) * 1.00002f;
                // the finest level that covers the rectangle with at most HIZ_SPAN x HIZ_SPAN texels; hiz_build_kernel
                // writes levels 0..3 (8..64 pixels); a rectangle too large even for the top level (rare among FAR
                // tiles) is kept untested
                int L = 0;
                while (L < 3 && (((px1 >> (3 + L)) - (px0 >> (3 + L))) >= HIZ_SPAN || ((py1 >> (3 + L)) - (py0 >> (3 + L))) >= HIZ_SPAN)) ++L;
                const int tx0 = px0 >> (3 + L), tx1 = px1 >> (3 + L), ty0 = py0 >> (3 + L), ty1 = py1 >> (3 + L);
                if (tx1 - tx0 < 8 && ty1 - ty0 < 8) {
                    unsigned m = 0x7F800000u;
                    for (int ty = ty0; ty <= ty1; ++ty)
                        for (int tx = tx0; tx <= tx1; ++tx)
                            m = min(m, hiz[dm.off[L] + (long long)ty * dm.w[L] + tx]);
                    keep = !(qmax < __uint_as_float(m));
                }
            }
        }
    }
    list_append(keep, t, out_list, out_count);
}

#ifndef GRID_WAVES_PER_EU
#define GRID_WAVES_PER_EU 8
#endif
#ifndef PATCH_MIN_FAST
#define PATCH_MIN_FAST 64   // tiles with fewer FAST cells send their fragments straight to the visibility buffer
#endif
#ifndef PATCH_WORDS_NEAR
#define PATCH_WORDS_NEAR 4096       // LDS patch of the first round's workgroups (8 bytes per pixel)
#endif
#ifndef PATCH_WORDS_FAR
#define PATCH_WORDS_FAR 0           // ... and of the second round's: none.  Its fragments are sparse (0.3 per cell) and the round is not
                                    // request-bound; 512 / 1024 words cost 20 / 30 us of occupancy (100 M-vertex frame)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GRID_WAVES_PER_EU)))
void raster_grid_kernel(const float *__restrict__ vert, const unsigned char *__restrict__ valid, int gh, int gw, View v,
                        unsigned long long *__restrict__ vis, unsigned *__restrict__ gqueue,
                        unsigned *__restrict__ gcount, unsigned gcap, int lanes_along_rows,
                        const unsigned *__restrict__ tile_list, const unsigned *__restrict__ tile_count,
                        Deferred *__restrict__ park_small, Deferred *__restrict__ park_large,
                        ParkedCell *__restrict__ park_cell, unsigned *__restrict__ park_counts, unsigned park_cap_small,
                        unsigned park_cap_large, unsigned park_cap_cell, int patch_cap) {
    // s_xy[].x of a vertex without window coordinates: behind the near plane / outside the
    // fixed-point range / masked out (nodata: its triangles do not exist, surface.py:203-205)
    constexpr int BEHIND = INT_MIN, RANGE = INT_MIN + 1, NODATA = INT_MIN + 2;
    __shared__ int2 s_xy[GT_NV];          // snapped window coordinates
    __shared__ float s_iw[GT_NV];
    __shared__ unsigned short s_q[GT_NC]; // FAST cell ids from the front, SLOW cell ids from the back
    // parked work: cell ids [0, ncell), then triangles (2 * cell id + half) with small boxes upwards from
    // ncell and with large boxes downwards from the end (a cell is parked whole or contributes at most two
    // triangles, so 2 * GT_NC entries always suffice)
    __shared__ unsigned short s_park[2 * GT_NC];
    __shared__ unsigned s_nfast, s_nslow, s_npark[3], s_park_base[3];
    // the tile's depth patch (dynamic LDS, patch_cap words): see phase 3
    extern __shared__ unsigned long long s_patch[];
    __shared__ int s_wbb[4][4];
    // ---- phase 0: this workgroup's tile (the frame plan dropped, deferred or culled the others)
    // Workgroups are handed to the 8 XCDs round-robin; each XCD has its own L2.  List position =
    // (XCD) * chunk + (turn): one XCD walks a CONTIGUOUS eighth of the list, i.e. neighbouring tiles,
    // whose fragments fall on neighbouring pixels, meet in the same L2.
    const unsigned n_list = *tile_count;
#ifndef GRID_NO_XCD_SWIZZLE
    const unsigned chunk = (n_list + 7u) >> 3;
    const unsigned pos = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= chunk || pos >= n_list) return;
#else
    const unsigned pos = blockIdx.x;
    if (pos >= n_list) return;
#endif
    const unsigned tile = tile_list[pos];
    WGT(0);
    const int tiles_x = (gw - 1 + GT_W - 1) / GT_W;
    const int tile_r = (int)(tile / (unsigned)tiles_x), tile_c = (int)(tile - (unsigned)tile_r * (unsigned)tiles_x);
    const int r0 = tile_r * GT_H, c0 = tile_c * GT_W;
    if (threadIdx.x == 0) {
        s_nfast = 0;
        s_nslow = 0;
        s_npark[0] = 0;
        s_npark[1] = 0;
        s_npark[2] = 0;
    }
    // ---- phase 1: vertices.  Every load of the thread's (up to) GT_VPT vertices is issued before the
    // first one is used: ONE memory round trip per tile instead of one per vertex (the round trip is what
    // this phase costs: measured 50 us per workgroup with five dependent trips while the atomics of the
    // neighbouring workgroups keep the memory pipeline busy).
    constexpr int GT_VPT = (GT_NV + 255) / 256;
    float vx[GT_VPT], vy[GT_VPT], vz[GT_VPT];
    unsigned char vok[GT_VPT];
    int bb_x0 = INT_MAX, bb_x1 = INT_MIN, bb_y0 = INT_MAX, bb_y1 = INT_MIN;     // snapped vertices of this thread
#pragma unroll
    for (int k = 0; k < GT_VPT; ++k) {
        const int idx = (int)threadIdx.x + 256 * k;
        const int lr = mul24(idx, GT_DIV_MAGIC) >> GT_DIV_SHIFT, lc = idx - lr * GT_VW;
        const int r = r0 + lr, c = c0 + lc;
        const bool inside = idx < GT_NV && r < gh && c < gw;
        const unsigned vid = inside ? (unsigned)r * (unsigned)gw + (unsigned)c : 0u;   // < 2^31 vertices
        const float *p = vert + 3ull * vid;
        vx[k] = p[0];
        vy[k] = p[1];
        vz[k] = p[2];
        vok[k] = inside ? (valid ? (valid[vid] ? 1 : 2) : 1) : 0;       // 0 outside the grid, 1 vertex, 2 nodata
    }
#pragma unroll
    for (int k = 0; k < GT_VPT; ++k) {
        const int idx = (int)threadIdx.x + 256 * k;
        int2 xy = make_int2(BEHIND, 0);
        if (vok[k] == 2) {
            xy.x = NODATA;
        } else if (vok[k] == 1) {
            float q[3];
            to_view(v, vx[k], vy[k], vz[k], q);
            if (q[2] >= 1.0f) {
                float xw, yw, iw;
                to_window(v, q, xw, yw, iw);
                xy.x = RANGE;
                if (fabsf(xw) < COORD_LIMIT && fabsf(yw) < COORD_LIMIT) {
                    xy = make_int2(snap(xw), snap(yw));
                    s_iw[idx] = iw;
                    bb_x0 = min(bb_x0, xy.x);
                    bb_x1 = max(bb_x1, xy.x);
                    bb_y0 = min(bb_y0, xy.y);
                    bb_y1 = max(bb_y1, xy.y);
                }
            }
        }
        if (idx < GT_NV) s_xy[idx] = xy;
    }
    if (patch_cap) {                   // the tile's footprint: per wave here, combined after the barrier
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            bb_x0 = min(bb_x0, __shfl_xor(bb_x0, m, 64));
            bb_x1 = max(bb_x1, __shfl_xor(bb_x1, m, 64));
            bb_y0 = min(bb_y0, __shfl_xor(bb_y0, m, 64));
            bb_y1 = max(bb_y1, __shfl_xor(bb_y1, m, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            int *o = s_wbb[threadIdx.x >> 6];
            o[0] = bb_x0; o[1] = bb_x1; o[2] = bb_y0; o[3] = bb_y1;
        }
    }
    __syncthreads();
    WGT(1);
#if defined(GRID_STOP_AFTER) && GRID_STOP_AFTER == 1
    if (s_xy[threadIdx.x].x == 12345) vis[0] = 1;     // keep phase 1 alive
    return;
#endif
    // ---- phase 2: classify the cells.  Consecutive cell ids run along the grid axis that runs ACROSS
    // the view, so that the fragments of neighbouring queue entries fall on neighbouring pixels of one
    // row and the atomics of one instruction share 64-byte lines.
    const int lane = (int)(threadIdx.x & 63);
    auto cell_rc = [&](int id, int &lr, int &lc) {
        if (lanes_along_rows) { lr = id & (GT_H - 1); lc = id >> GT_H_LOG2; }
        else { lc = id & (GT_W - 1); lr = id >> GT_W_LOG2; }
    };
#pragma unroll 1
    for (int id = threadIdx.x; id < GT_NC; id += 256) {
        int lr, lc;
        cell_rc(id, lr, lc);
        const int ia = lr * GT_VW + lc;
        const int2 P0 = s_xy[ia], P1 = s_xy[ia + GT_VW], P2 = s_xy[ia + GT_VW + 1], P3 = s_xy[ia + 1];
        int kind = 0;                      // 0 nothing, 1 FAST, 2 SLOW
        if (r0 + lr < gh - 1 && c0 + lc < gw - 1) {
            if (P0.x > NODATA && P1.x > NODATA && P2.x > NODATA && P3.x > NODATA) {
                // the cell's bounding box holds no pixel centre of the viewport: neither can its triangles
                const int minx = min(min(P0.x, P1.x), min(P2.x, P3.x)), maxx = max(max(P0.x, P1.x), max(P2.x, P3.x));
                const int miny = min(min(P0.y, P1.y), min(P2.y, P3.y)), maxy = max(max(P0.y, P1.y), max(P2.y, P3.y));
                const int i0 = (minx + SUB / 2 - 1) >> 8, i1 = (maxx - SUB / 2) >> 8;
                const int j0 = (miny + SUB / 2 - 1) >> 8, j1 = (maxy - SUB / 2) >> 8;
                if (!(i0 > i1 || j0 > j1 || i1 < 0 || j1 < 0 || i0 > v.w - 1 || j0 > v.h - 1)) {
                    const int nx = min(i1, v.w - 1) - max(i0, 0), ny = min(j1, v.h - 1) - max(j0, 0);     // centres - 1
                    // FAST needs the UNCLAMPED box small too: a near-field cell that only pokes a corner
                    // into the viewport has edge vectors far beyond the 24-bit products and the 2^12 tie
                    // key; i1 - i0 < 8 bounds its extent by 10 px = 2560 sub-pixel units
                    kind = (nx < FAST_MAX && ny < FAST_MAX && ((i1 - i0) | (j1 - j0)) < 8) ? 1 : 2;
                    // a larger box of at most 8 x 8 centres (at least COOP_MIN_W columns and COOP_MIN_PIX
                    // centres; unclamped extent under 14 px = 3584 sub-pixel units for the same reasons):
                    // the whole cell goes to raster_cell_kernel
                    if (kind == 2 && nx < 8 && ny < 8 && nx + 1 >= COOP_MIN_W && mul24(nx + 1, ny + 1) >= COOP_MIN_PIX &&
                        i1 - i0 < 12 && j1 - j0 < 12)
                        kind = 3;
                }
            } else if (!(P0.x == BEHIND && P1.x == BEHIND && P2.x == BEHIND && P3.x == BEHIND)) {
                kind = 2;                  // sentinels among the corners: sorted out per triangle
            }
        }
        const unsigned long long mf = __ballot(kind == 1), ms = __ballot(kind == 2);
        if (mf) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_nfast, (unsigned)__popcll(mf));
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (kind == 1) s_q[base + __popcll(mf & ((1ull << lane) - 1ull))] = (unsigned short)id;
        }
        if (ms) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_nslow, (unsigned)__popcll(ms));
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (kind == 2) s_q[GT_NC - 1 - (base + __popcll(ms & ((1ull << lane) - 1ull)))] = (unsigned short)id;
        }
        const unsigned long long mc = __ballot(kind == 3);
        if (mc) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_npark[2], (unsigned)__popcll(mc));
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (kind == 3) s_park[base + __popcll(mc & ((1ull << lane) - 1ull))] = (unsigned short)id;
        }
    }
    __syncthreads();
    WGT(2);
    const int nfast = (int)s_nfast, nslow = (int)s_nslow;
    const unsigned ncell = s_npark[2];
#if defined(GRID_STOP_AFTER) && GRID_STOP_AFTER == 2
    if (nfast + nslow == 123456) vis[0] = s_q[threadIdx.x];
    return;
#endif
#ifdef ALP_RASTER_STATS
    {   // census by the tile's screen footprint: would an LDS patch of that size pay?
        __shared__ int s_bb[4];
        __shared__ unsigned s_boxpix[3];
        if (threadIdx.x == 0) { s_bb[0] = INT_MAX; s_bb[1] = INT_MIN; s_bb[2] = INT_MAX; s_bb[3] = INT_MIN; s_boxpix[0] = s_boxpix[1] = s_boxpix[2] = 0; }
        __syncthreads();
        bool sentinel = false;
        for (int i = threadIdx.x; i < GT_NV; i += 256) {
            const int2 P = s_xy[i];
            if (P.x > NODATA) { atomicMin(&s_bb[0], P.x); atomicMax(&s_bb[1], P.x); atomicMin(&s_bb[2], P.y); atomicMax(&s_bb[3], P.y); }
            else if (P.x != BEHIND || true) sentinel |= (P.x == RANGE);
        }
        auto boxpix = [&](int id) {
            int lr, lc; cell_rc(id, lr, lc);
            const int ia = lr * GT_VW + lc;
            const int2 P0 = s_xy[ia], P1 = s_xy[ia + GT_VW], P2 = s_xy[ia + GT_VW + 1], P3 = s_xy[ia + 1];
            if (!(P0.x > NODATA && P1.x > NODATA && P2.x > NODATA && P3.x > NODATA)) return 0;
            const int minx = min(min(P0.x, P1.x), min(P2.x, P3.x)), maxx = max(max(P0.x, P1.x), max(P2.x, P3.x));
            const int miny = min(min(P0.y, P1.y), min(P2.y, P3.y)), maxy = max(max(P0.y, P1.y), max(P2.y, P3.y));
            const int i0 = max((minx + SUB / 2 - 1) >> 8, 0), i1 = min((maxx - SUB / 2) >> 8, v.w - 1);
            const int j0 = max((miny + SUB / 2 - 1) >> 8, 0), j1 = min((maxy - SUB / 2) >> 8, v.h - 1);
            return (i1 >= i0 && j1 >= j0) ? (i1 - i0 + 1) * (j1 - j0 + 1) : 0;
        };
        for (int e = threadIdx.x; e < nfast; e += 256) atomicAdd(&s_boxpix[0], (unsigned)boxpix(s_q[e]));
        for (int e = threadIdx.x; e < nslow; e += 256) atomicAdd(&s_boxpix[1], (unsigned)boxpix(s_q[GT_NC - 1 - e]));
        for (int e = threadIdx.x; e < (int)ncell; e += 256) atomicAdd(&s_boxpix[2], (unsigned)boxpix(s_park[e]));
        __syncthreads();
        if (threadIdx.x == 0) {
            const int i0 = max((s_bb[0] + SUB / 2 - 1) >> 8, 0), i1 = min((s_bb[1] - SUB / 2) >> 8, v.w - 1);
            const int j0 = max((s_bb[2] + SUB / 2 - 1) >> 8, 0), j1 = min((s_bb[3] - SUB / 2) >> 8, v.h - 1);
            const long long area = (i1 >= i0 && j1 >= j0) ? (long long)(((i1 - i0 + 8) & ~7)) * (j1 - j0 + 1) : 0;
            const int b = area <= 512 ? 0 : area <= 1024 ? 1 : area <= 2048 ? 2 : area <= 4096 ? 3 : area <= 8192 ? 4 : area <= 16384 ? 5 : area <= 65536 ? 6 : 7;
            RSTAT(24 + 8 * b + 0, 1);
            RSTAT(24 + 8 * b + 1, area);
            RSTAT(24 + 8 * b + 2, nfast);
            RSTAT(24 + 8 * b + 3, nslow);
            RSTAT(24 + 8 * b + 4, ncell);
            RSTAT(24 + 8 * b + 5, s_boxpix[0]);
            RSTAT(24 + 8 * b + 6, s_boxpix[1]);
            RSTAT(24 + 8 * b + 7, s_boxpix[2]);
        }
        __syncthreads();
    }
#endif
    if (threadIdx.x == 0) {
        RSTAT(10, 1);
        RSTAT(11, nfast);
        RSTAT(12, nslow);
        RSTAT(13, (nfast + 63) / 64);
        RSTAT(14, (nslow + 63) / 64);
    }
    // ---- phase 3: FAST cells.  The cell's box holds at most FAST_MAX x FAST_MAX pixel centres.  Both
    // triangles (a, b, c), (a, c, d) are decided at those centres at once: five edge functions
    // e(P->Q)(p) = (Q - P) x (p - P) instead of two 3-edge set-ups (the diagonal is shared,
    // e(a->c) = -e(c->a) exactly), stepped by whole pixels -- the same integers, tie rule and depth
    // expression as emit_small.  A triangle with area <= 0 can never have all three biased values
    // >= 0, and a centre outside a triangle's own box is outside the triangle.
    //
    // Where the fragments go.  What bounds this stage is the chip's rate of atomic line-requests, and the
    // fragments of FAST cells arrive one or two per request.  A tile whose footprint (the pixel centres
    // inside the bounding box of its snapped vertices, rows of whole 8-pixel lines) fits the workgroup's
    // LDS patch therefore collects them there with ds_max_u64 -- the same keys, and max is associative --
    // and sends the patch to the visibility buffer afterwards: consecutive lanes = consecutive pixels,
    // 8 fragments per request, every pixel once per tile.
    int pI0 = 0, pJ0 = 0, pW = 0, pH = 0;
    bool use_patch = false;
    if (patch_cap && nfast >= PATCH_MIN_FAST) {
        const int x0 = min(min(s_wbb[0][0], s_wbb[1][0]), min(s_wbb[2][0], s_wbb[3][0]));
        const int x1 = max(max(s_wbb[0][1], s_wbb[1][1]), max(s_wbb[2][1], s_wbb[3][1]));
        const int y0 = min(min(s_wbb[0][2], s_wbb[1][2]), min(s_wbb[2][2], s_wbb[3][2]));
        const int y1 = max(max(s_wbb[0][3], s_wbb[1][3]), max(s_wbb[2][3], s_wbb[3][3]));
        // |snapped| < COORD_LIMIT * SUB: no overflow in the roundings below
        const int i0 = max((x0 + SUB / 2 - 1) >> 8, 0), i1 = min((x1 - SUB / 2) >> 8, v.w - 1);
        const int j0 = max((y0 + SUB / 2 - 1) >> 8, 0), j1 = min((y1 - SUB / 2) >> 8, v.h - 1);
        if (i1 >= i0 && j1 >= j0) {
            pI0 = i0 & ~7;
            pJ0 = j0;
            pW = (i1 - pI0 + 8) & ~7;
            pH = j1 - j0 + 1;
            use_patch = mul24(pW, pH) <= patch_cap;      // pW, pH <= 2^15
        }
    }
    const int patch_n = use_patch ? mul24(pW, pH) : 0;
    if (use_patch) {
        for (int k = threadIdx.x; k < patch_n; k += 256) s_patch[k] = 0ull;
        __syncthreads();
    }
    auto fast_cells = [&](auto to_patch) {
    constexpr bool PATCH = decltype(to_patch)::value;
    auto sink = [&](int i, int j, unsigned long long key) {
        if constexpr (PATCH)
            __hip_atomic_fetch_max(&s_patch[mul24(j - pJ0, pW) + (i - pI0)], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            vis_max(vis, v, i, j, key);
    };
#pragma unroll 1
    for (int e = threadIdx.x; e < nfast; e += 256) {
        const int id = s_q[e];
        int lr, lc;
        cell_rc(id, lr, lc);
        const int ia = lr * GT_VW + lc, ib = ia + GT_VW, ic = ib + 1, idd = ia + 1;
        const int2 a = s_xy[ia], b = s_xy[ib], cc = s_xy[ic], d = s_xy[idd];
        const unsigned cell = (unsigned)(r0 + lr) * (unsigned)(gw - 1) + (unsigned)(c0 + lc);      // < 2^31: 2 * cell + 1 fits
        const int minx = min(min(a.x, b.x), min(cc.x, d.x)), maxx = max(max(a.x, b.x), max(cc.x, d.x));
        const int miny = min(min(a.y, b.y), min(cc.y, d.y)), maxy = max(max(a.y, b.y), max(cc.y, d.y));
        const int ci0 = max((minx + SUB / 2 - 1) >> 8, 0), ci1 = min((maxx - SUB / 2) >> 8, v.w - 1);
        const int cj0 = max((miny + SUB / 2 - 1) >> 8, 0), cj1 = min((maxy - SUB / 2) >> 8, v.h - 1);
        const int px = ci0 * SUB + SUB / 2, py = cj0 * SUB + SUB / 2;
        const int pax = px - a.x, pay = py - a.y, pbx = px - b.x, pby = py - b.y;
        const int pcx = px - cc.x, pcy = py - cc.y, pdx = px - d.x, pdy = py - d.y;
        // directed edges: 0 b->c, 1 c->a, 2 a->b (triangle 0); 3 c->d, 4 d->a, 5 a->c (triangle 1)
        int ex[6] = {cc.x - b.x, a.x - cc.x, b.x - a.x, d.x - cc.x, a.x - d.x, 0};
        int ey[6] = {cc.y - b.y, a.y - cc.y, b.y - a.y, d.y - cc.y, a.y - d.y, 0};
        ex[5] = -ex[1];
        ey[5] = -ey[1];
        int bs[6], row[6];
        row[0] = mul24(ex[0], pby) - mul24(ey[0], pbx);
        row[1] = mul24(ex[1], pcy) - mul24(ey[1], pcx);
        row[2] = mul24(ex[2], pay) - mul24(ey[2], pax);
        row[3] = mul24(ex[3], pcy) - mul24(ey[3], pcx);
        row[4] = mul24(ex[4], pdy) - mul24(ey[4], pdx);
        row[5] = -row[1];
        const int area0 = row[0] + row[1] + row[2], area1 = row[3] + row[4] + row[5];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            // the edge owns its boundary iff dy < 0 or (dy == 0 and dx > 0) iff (dy << 12) - dx < 0
            // (|dx| < 2^12 here: the cell's unclamped box spans fewer than 10 pixels)
            bs[k] = 1 + (((ey[k] << 12) - ex[k]) >> 31);
            row[k] -= bs[k];
        }
        const float iwa = s_iw[ia], iwb = s_iw[ib], iwc = s_iw[ic], iwd = s_iw[idd];
        const float inv0 = exact_rcp_unchecked((float)area0), inv1 = exact_rcp_unchecked((float)area1);      // used only where area > 0
        const unsigned long long lo0 = 0xFFFFFFFFu - 2u * cell, lo1 = lo0 - 1u;
        for (int j = cj0; j <= cj1; ++j) {
            int u[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) u[k] = row[k];
            for (int i = ci0; i <= ci1; ++i) {
                if ((u[0] | u[1] | u[2]) >= 0) {      // weights: edge k is opposite vertex k of (a, b, c)
                    const float q = __builtin_fmaf((float)(u[2] + bs[2]), iwc,
                                                   __builtin_fmaf((float)(u[1] + bs[1]), iwb,
                                                                  (float)(u[0] + bs[0]) * iwa)) * inv0;
                    sink(i, j, ((unsigned long long)__float_as_uint(q) << 32) | lo0);
                }
                if ((u[3] | u[4] | u[5]) >= 0) {      // (a, c, d)
                    const float q = __builtin_fmaf((float)(u[5] + bs[5]), iwd,
                                                   __builtin_fmaf((float)(u[4] + bs[4]), iwc,
                                                                  (float)(u[3] + bs[3]) * iwa)) * inv1;
                    sink(i, j, ((unsigned long long)__float_as_uint(q) << 32) | lo1);
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) u[k] -= ey[k] * SUB;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) row[k] += ex[k] * SUB;
        }
    }
    };
    if (use_patch) {
        fast_cells(std::true_type{});
        __syncthreads();
        // the patch goes out row by row; a thread's next word is 256 further on
        const int step_rows = 256 / pW, step_cols = 256 - step_rows * pW;
        int row = (int)threadIdx.x / pW, col = (int)threadIdx.x - row * pW;
        for (int k = threadIdx.x; k < patch_n; k += 256) {
            const unsigned long long key = s_patch[k];
            if (key) vis_max(vis, v, pI0 + col, pJ0 + row, key);
            row += step_rows;
            col += step_cols;
            if (col >= pW) { col -= pW; ++row; }
        }
    } else {
        fast_cells(std::false_type{});
    }
#if defined(GRID_STOP_AFTER) && GRID_STOP_AFTER == 3
    return;
#endif
#ifdef ALP_WG_TIMING
    __syncthreads();
#endif
    WGT(3);
    // ---- phase 4: SLOW cells, triangle by triangle: (a, b, c) and (a, c, d) (surface.py:194-201).
    // Wave-converged (coop_drain is wave-wide): every lane of a wave makes the same number of rounds.
#pragma unroll 1
    for (int e0 = (int)(threadIdx.x & ~63u); e0 < nslow; e0 += 256) {
        const int e = e0 + lane;
        const bool work = e < nslow;
        const int id = work ? (int)s_q[GT_NC - 1 - e] : 0;
        int lr, lc;
        cell_rc(id, lr, lc);
        const int ia = lr * GT_VW + lc, ib = ia + GT_VW, ic = ib + 1, idd = ia + 1;
        const int2 P0 = s_xy[ia], P1 = s_xy[ib], P2 = s_xy[ic], P3 = s_xy[idd];
        const unsigned cell = (unsigned)(r0 + lr) * (unsigned)(gw - 1) + (unsigned)(c0 + lc);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k1 = half ? ic : ib, k2 = half ? idd : ic;
            const int2 A = P0, B = half ? P2 : P1, C = half ? P3 : P2;
            const unsigned t = 2u * cell + (unsigned)half;
            Deferred park;
            int code = EMIT_DONE;
            if (work) {
                if (A.x > NODATA && B.x > NODATA && C.x > NODATA) {
                    const int X[3] = {A.x, B.x, C.x}, Y[3] = {A.y, B.y, C.y};
                    code = emit_small(v, X, Y, s_iw, ia, k1, k2, t, vis, &park, true, COOP_MIN_W, COOP_MIN_PIX, true);
                } else if (A.x != NODATA && B.x != NODATA && C.x != NODATA &&
                           !(A.x == BEHIND && B.x == BEHIND && C.x == BEHIND)) {
                    code = EMIT_GENERAL;      // near-plane crossing or out of range (all three behind: nothing to draw)
                }
                if (code == EMIT_GENERAL) {   // rare: raster_general_kernel redoes this triangle from its vertices
                    const unsigned slot = atomicAdd(gcount, 1u);
                    if (slot < gcap) gqueue[slot] = t;
                }
            }
            // parked triangles are only noted here (every lane of the wave arrives here) ...
            const unsigned long long ms = __ballot(code == EMIT_PARKED_SMALL), ml = __ballot(code == EMIT_PARKED);
            if (ms) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&s_npark[0], (unsigned)__popcll(ms));
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                if (code == EMIT_PARKED_SMALL) s_park[ncell + base + __popcll(ms & ((1ull << lane) - 1ull))] = (unsigned short)(2 * id + half);
            }
            if (ml) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&s_npark[1], (unsigned)__popcll(ml));
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                if (code == EMIT_PARKED) s_park[2 * GT_NC - 1 - (base + __popcll(ml & ((1ull << lane) - 1ull)))] = (unsigned short)(2 * id + half);
            }
        }
    }
    // ---- phase 5: ... and leave the workgroup together: ONE reservation per queue and workgroup in the
    // device queues (a reservation per wave and round made the two global counters the bottleneck of the
    // near tiles: 0.5 ms of same-address atomics), then every thread writes whole entries.
    __syncthreads();
    const unsigned np_small = s_npark[0], np_large = s_npark[1];
    if (np_small + np_large + ncell == 0) {
        WGT(4);
        return;
    }
    if (threadIdx.x < 3) {
        const unsigned cnt = threadIdx.x == 0 ? np_small : (threadIdx.x == 1 ? np_large : ncell);
        s_park_base[threadIdx.x] = cnt ? atomicAdd(park_counts + threadIdx.x, cnt) : 0u;
    }
    __syncthreads();
    for (unsigned e = threadIdx.x; e < ncell; e += 256) {
        const int id = (int)s_park[e];
        int lr, lc;
        cell_rc(id, lr, lc);
        const int ia = lr * GT_VW + lc, ib = ia + GT_VW, ic = ib + 1, idd = ia + 1;
        const int2 A = s_xy[ia], B = s_xy[ib], C = s_xy[ic], D = s_xy[idd];
#ifdef ALP_RASTER_STATS
        {   // census of the parked cells' boxes
            const int minx = min(min(A.x, B.x), min(C.x, D.x)), maxx = max(max(A.x, B.x), max(C.x, D.x));
            const int miny = min(min(A.y, B.y), min(C.y, D.y)), maxy = max(max(A.y, B.y), max(C.y, D.y));
            const int bw = min((maxx - SUB / 2) >> 8, v.w - 1) - max((minx + SUB / 2 - 1) >> 8, 0) + 1;
            const int bh = min((maxy - SUB / 2) >> 8, v.h - 1) - max((miny + SUB / 2 - 1) >> 8, 0) + 1;
            RSTAT(19, bh <= 2 ? 1 : 0);
            RSTAT(20, bh <= 4 ? 1 : 0);
            RSTAT(21, bw <= 4 ? 1 : 0);
            RSTAT(22, bw * bh);
            RSTAT(23, 1);
        }
#endif
        ParkedCell pc;
        pc.X[0] = A.x; pc.X[1] = B.x; pc.X[2] = C.x; pc.X[3] = D.x;
        pc.Y[0] = A.y; pc.Y[1] = B.y; pc.Y[2] = C.y; pc.Y[3] = D.y;
        pc.iw[0] = s_iw[ia]; pc.iw[1] = s_iw[ib]; pc.iw[2] = s_iw[ic]; pc.iw[3] = s_iw[idd];
        pc.cell = (unsigned)(r0 + lr) * (unsigned)(gw - 1) + (unsigned)(c0 + lc);
        pc.pad[0] = pc.pad[1] = pc.pad[2] = 0;
        const unsigned slot = s_park_base[2] + e;
        if (slot < park_cap_cell) park_cell[slot] = pc;           // an overflow is noticed by finish_frame
    }
    for (unsigned e = threadIdx.x; e < np_small + np_large; e += 256) {
        const bool large = e >= np_small;
        const unsigned k = large ? e - np_small : e;
        const unsigned code = large ? s_park[2 * GT_NC - 1 - k] : s_park[ncell + k];
        const int id = (int)(code >> 1), half = (int)(code & 1u);
        int lr, lc;
        cell_rc(id, lr, lc);
        const int ia = lr * GT_VW + lc, ib = ia + GT_VW, ic = ib + 1, idd = ia + 1;
        const int k1 = half ? ic : ib, k2 = half ? idd : ic;
        const int2 A = s_xy[ia], B = s_xy[k1], C = s_xy[k2];
        Deferred d;
        d.X[0] = A.x; d.X[1] = B.x; d.X[2] = C.x;
        d.Y[0] = A.y; d.Y[1] = B.y; d.Y[2] = C.y;
        d.iw[0] = s_iw[ia]; d.iw[1] = s_iw[k1]; d.iw[2] = s_iw[k2];
        d.t = 2u * ((unsigned)(r0 + lr) * (unsigned)(gw - 1) + (unsigned)(c0 + lc)) + (unsigned)half;
        const unsigned slot = s_park_base[large ? 1 : 0] + k;
        Deferred *queue = large ? park_large : park_small;
        if (slot < (large ? park_cap_large : park_cap_small)) queue[slot] = d;    // an overflow is noticed by finish_frame
    }
#ifdef ALP_WG_TIMING
    __syncthreads();
#endif
    WGT(4);
}

// ------------------------------------------------------------------ kernel 3: large triangles
// one wave per (triangle, 64x64-pixel tile): lane = pixel column, loop over the rows
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void raster_large_kernel(const float *__restrict__ vert,
                                                           const int *__restrict__ ind, long long gw, View v,
                                                           unsigned long long *__restrict__ vis,
                                                           const WorkItem *__restrict__ queue,
                                                           const unsigned *__restrict__ qcount, unsigned qcap) {
    const unsigned count = min(*qcount, qcap);
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned it = wave; it < count; it += nwaves) {
        const WorkItem wi = queue[it];
        float q[3][3];
        load_view_tri<IMPLICIT>(v, vert, ind, gw, (long long)wi.tri, q);
        if (wi.sub == 0xFFFF) {
            raster_big(v, q, wi.tri, vis, lane);
            continue;
        }
        float xw[4], yw[4], iw[4];
        bool big;
        const int ntri = clip_project(v, q, xw, yw, iw, big);
        const int f = wi.sub;
        if (f >= ntri) continue;
        const float x3[3] = {xw[0], xw[f + 1], xw[f + 2]}, y3[3] = {yw[0], yw[f + 1], yw[f + 2]},
                    i3[3] = {iw[0], iw[f + 1], iw[f + 2]};
        const TriSetup s = setup_tri(v, x3, y3, i3);
        if (!s.valid) continue;
        const int i = wi.tx * TILE + lane;
        int ja = wi.ty * TILE, jb = ja + TILE - 1;
        ja = ja < s.j0 ? s.j0 : ja;
        jb = jb > s.j1 ? s.j1 : jb;
        if (i < s.i0 || i > s.i1) continue;
        for (int j = ja; j <= jb; ++j) {
            const unsigned long long key = pixel_key(s, i, j, wi.tri);
            if (key) vis_max(vis, v, i, j, key);
        }
    }
}

// ------------------------------------------------------------------ kernel 4: resolve + remap
__device__ __forceinline__ bool remap_source(const RemapCoef &c, int w, int h, int x, int y, int &sx, int &sy,
                                             float *mapx = nullptr, float *mapy = nullptr) {
    const double x1 = (x - c.c0) / c.c0, y1 = (y - c.c1) / c.c1;
    const double r = __builtin_sqrt(x1 * x1 + y1 * y1), r2 = r * r, r4 = r2 * r2, r6 = r4 * r2;
    const double xd = x1 * (1 + c.k1 * r2 + c.k2 * r4 + c.k3 * r6) / (1 + c.k4 * r2 + c.k5 * r4 + c.k6 * r6) +
                      2 * c.p1 * x1 * y1 + c.p2 * (r2 * 2 * x1 * x1) + c.s1 * r2 + c.s2 * r4;
    const double yd = y1 * (1 + c.a1 + c.k1 * r2 + c.k2 * r4 + c.k3 * r6) / (1 + c.a2 + c.k4 * r2 + c.k5 * r4 + c.k6 * r6) +
                      2 * c.p1 * x1 * y1 + c.p2 * (r2 * 2 * y1 * y1) + c.s3 * r2 + c.s4 * r4;
    const float mx = (float)(xd * c.c0 + c.c0), my = (float)(yd * c.c1 + c.c1);
    if (mapx) { *mapx = mx; *mapy = my; }
    const double rx = __builtin_rint((double)mx), ry = __builtin_rint((double)my);
    if (!(rx >= 0 && rx < w && ry >= 0 && ry < h)) return false;
    sx = (int)rx;
    sy = (int)ry;
    return true;
}

// the float32 source map itself (what distort() hands to cv2.remap, project.py:140-141)
__global__ __launch_bounds__(256) void distort_map_kernel(int w, int h, RemapCoef rc, float *__restrict__ map_x,
                                                          float *__restrict__ map_y) {
    const long long npix = (long long)w * h;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        const int y = (int)(p / w), x = (int)(p - (long long)y * w);
        int sx, sy;
        remap_source(rc, w, h, x, y, sx, sy, map_x + p, map_y + p);
    }
}

// (Measured and not kept, round 2: a two-stage software pipeline -- the loads of a thread's next pixel in
// flight during the float64 interpolation of the current one.  The kernel without its float64 arithmetic
// takes 115 us, with it 176 us; the pipelined form needs 111 VGPRs (4 waves per SIMD) and takes 167 us,
// 220 us at 5 waves and 500 us at 6 (spills).)
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void resolve_kernel(const float *__restrict__ vert, const float *__restrict__ value,
                                                      const int *__restrict__ ind, long long gw, View v,
                                                      RemapCoef rc, int identity_remap, double min_distance,
                                                      const unsigned long long *__restrict__ vis,
                                                      float *__restrict__ out, const unsigned *__restrict__ frame_counts,
                                                      unsigned *__restrict__ host_counts) {
    // the frame's queue counters go to pinned host memory for finish_frame (a copy node of its own costs 5 us)
    if (host_counts && blockIdx.x == 0 && threadIdx.x < 2 * QC_STRIDE) host_counts[threadIdx.x] = frame_counts[threadIdx.x];
    const long long npix = (long long)v.w * v.h;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        const int y = (int)((unsigned)p / (unsigned)v.w), x = (int)((unsigned)p - (unsigned)y * (unsigned)v.w);   // w * h <= 2^30
        float o[3] = {0.0f, 0.0f, 0.0f};
        int sx, sy;
        // no distortion at all (a1 = a2 = 1, everything else 0): the float64 map returns the
        // pixel itself for every image size (checked exhaustively up to 32768), skip it
        if (identity_remap ? (sx = x, sy = y, true) : remap_source(rc, v.w, v.h, x, y, sx, sy)) {
            const int j = v.h - 1 - sy;                               // flipud: image row -> GL row
            const unsigned long long key = vis[(size_t)j * v.w + sx];
            if (key) {
                const long long t = (long long)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
                const Idx3 id = tri_vertices<IMPLICIT>(ind, gw, t);
                const long long ids[3] = {id.a, id.b, id.c};
                // float64 view-space vertices: Rd (v - camd)
                double Q[3][3];
                float P[3][3];                     // the vertices themselves: they are the values when no value array is given
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float *pp = vert + 3 * ids[k];
                    P[k][0] = pp[0];
                    P[k][1] = pp[1];
                    P[k][2] = pp[2];
                    const double d0 = (double)P[k][0] - v.camd[0], d1 = (double)P[k][1] - v.camd[1], d2 = (double)P[k][2] - v.camd[2];
#pragma unroll
                    for (int c = 0; c < 3; ++c) Q[k][c] = (v.Rd[c][0] * d0 + v.Rd[c][1] * d1) + v.Rd[c][2] * d2;
                }
                const double *A = Q[0], *B = Q[1], *C = Q[2];
                const double r[3] = {(((double)sx + 0.5) * v.kx - 1.0) * v.ifx,
                                     (((double)j + 0.5) * v.ky - 1.0) * v.ify, 1.0};
                const double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
                const double pv[3] = {r[1] * e2[2] - r[2] * e2[1], r[2] * e2[0] - r[0] * e2[2], r[0] * e2[1] - r[1] * e2[0]};
                const double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
                const double tv[3] = {-A[0], -A[1], -A[2]};
                const double inv_det = 1.0 / det;      // the one division of the interpolation
                const double beta = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv_det;
                const double qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
                const double gamma = (r[0] * qv[0] + r[1] * qv[1] + r[2] * qv[2]) * inv_det;
                const double alpha = 1.0 - beta - gamma;
                bool masked = false;
                if (min_distance > 0) {
                    const double dA = __builtin_sqrt(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]);
                    const double dB = __builtin_sqrt(B[0] * B[0] + B[1] * B[1] + B[2] * B[2]);
                    const double dC = __builtin_sqrt(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
                    masked = alpha * dA + beta * dB + gamma * dC < min_distance;
                }
                if (!masked) {
                    if (value) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float *pv_ = value + 3 * ids[k];
                            P[k][0] = pv_[0];
                            P[k][1] = pv_[1];
                            P[k][2] = pv_[2];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) o[c] = (float)(alpha * P[0][c] + beta * P[1][c] + gamma * P[2][c]);
                }
            }
        }
        out[p * 3 + 0] = o[0];
        out[p * 3 + 1] = o[1];
        out[p * 3 + 2] = o[2];
    }
}

// stand-alone distort(): gather of an h x w x c image
__global__ __launch_bounds__(256) void distort_image_kernel(const float *__restrict__ img, int w, int h, int c,
                                                            RemapCoef rc, float *__restrict__ out) {
    const long long npix = (long long)w * h;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        const int y = (int)(p / w), x = (int)(p - (long long)y * w);
        int sx, sy;
        const bool ok = remap_source(rc, w, h, x, y, sx, sy);
        for (int k = 0; k < c; ++k) out[p * c + k] = ok ? img[((long long)sy * w + sx) * c + k] : 0.0f;
    }
}

// ------------------------------------------------------------------ reverse_proj post-processing
// src/alproj/project.py:361-373 on the device: keep the pixels whose first rendered channel
// (offset-relative x) is > 0 (quirk Q13), in row-major pixel order, and return their linear
// index and x, y, z = channels 0, 2, 1 (+ offsets, added in float64 like the reference does).
// Pass 1 counts per chunk of COMPACT_CHUNK pixels, a one-workgroup scan turns the counts into
// offsets, pass 2 writes (order-preserving stream compaction).
constexpr int COMPACT_CHUNK = 4096;

// set_gcp (src/alproj/gcp.py:644-648) against the resident coordinate image instead of a merge
// with the reverse_proj table: pixel (u[i], v[i]) -> x, y, z = channels (0, 2, 1) + offsets,
// NaN where the pixel is outside the image or does not see the surface (x <= 0, project.py:369)
// sim_image's tail (project.py:322-324): (raw * 255).astype(uint8), RGB -> BGR.  numpy's float32 -> uint8 cast is the
// x86 truncating conversion to int32 followed by a wrap to 8 bits (NaN and out-of-range give 0x80000000 -> 0).
__global__ __launch_bounds__(256) void image_u8_kernel(const float *__restrict__ img, long long npix, float scale, int reverse,
                                                       unsigned char *__restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        unsigned char b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = img[3 * i + c] * scale;
            const int q = (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : (int)0x80000000;   // false for NaN too
            b[c] = (unsigned char)(q & 0xFF);
        }
        out[3 * i + 0] = reverse ? b[2] : b[0];
        out[3 * i + 1] = b[1];
        out[3 * i + 2] = reverse ? b[0] : b[2];
    }
}

__global__ __launch_bounds__(256) void gather_pixels_kernel(const float *__restrict__ image, int w, int h,
                                                            const int *__restrict__ u, const int *__restrict__ v,
                                                            long long n, double o0, double o1, double o2,
                                                            double *__restrict__ xyz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double nan = __builtin_nan("");
    double x = nan, y = nan, z = nan;
    const int uu = u[i], vv = v[i];
    if (uu >= 0 && uu < w && vv >= 0 && vv < h) {
        const float *px = image + 3 * ((long long)vv * w + uu);
        if (px[0] > 0.0f) {
            x = (double)px[0] + o0;
            y = (double)px[2] + o2;
            z = (double)px[1] + o1;
        }
    }
    xyz[3 * i + 0] = x;
    xyz[3 * i + 1] = y;
    xyz[3 * i + 2] = z;
}

__global__ __launch_bounds__(256) void valid_count_kernel(const float *__restrict__ img, long long npix,
                                                          unsigned *__restrict__ counts) {
    __shared__ unsigned s[4];
    const long long base = (long long)blockIdx.x * COMPACT_CHUNK;
    unsigned c = 0;
    for (int k = threadIdx.x; k < COMPACT_CHUNK; k += 256) {
        const long long p = base + k;
        if (p < npix && img[p * 3] > 0.0f) ++c;
    }
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// exclusive scan of n counts (n up to a few ten thousand) by one workgroup; total -> offsets[n]
__global__ __launch_bounds__(1024) void scan_counts_kernel(const unsigned *__restrict__ counts, int n,
                                                           unsigned long long *__restrict__ offsets) {
    __shared__ unsigned long long s[1024];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    unsigned long long sum = 0;
    for (int i = lo; i < hi; ++i) sum += counts[i];
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                 // Hillis-Steele inclusive scan
        unsigned long long t = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned long long run = threadIdx.x ? s[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        offsets[i] = run;
        run += counts[i];
    }
    if (threadIdx.x == 1023) offsets[n] = s[1023];
}

__global__ __launch_bounds__(256) void valid_write_kernel(const float *__restrict__ img, long long npix,
                                                          const unsigned long long *__restrict__ offsets,
                                                          double o0, double o1, double o2,
                                                          unsigned *__restrict__ idx_out,
                                                          double *__restrict__ xyz_out) {
    __shared__ unsigned s_wave[4];
    const long long base = (long long)blockIdx.x * COMPACT_CHUNK;
    unsigned long long out = offsets[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k0 = 0; k0 < COMPACT_CHUNK; k0 += 256) {   // consecutive pixels per pass keep the order
        const long long p = base + k0 + threadIdx.x;
        float c0 = 0, c1 = 0, c2 = 0;
        bool valid = false;
        if (p < npix) {
            c0 = img[p * 3];
            valid = c0 > 0.0f;
            if (valid) { c1 = img[p * 3 + 1]; c2 = img[p * 3 + 2]; }
        }
        const unsigned long long m = __ballot(valid);
        const unsigned before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        unsigned wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_wave[w];
        const unsigned total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (valid) {
            const unsigned long long o = out + wbase + before;
            idx_out[o] = (unsigned)p;
            xyz_out[o * 3 + 0] = (double)c0 + o0;        // x  (channel 0 + offsets[0])
            xyz_out[o * 3 + 1] = (double)c2 + o2;        // y  (channel 2 + offsets[2])
            xyz_out[o * 3 + 2] = (double)c1 + o1;        // z  (channel 1 + offsets[1])
        }
        out += total;
        __syncthreads();
    }
}

// does an index array spell out exactly the regular grid of surface.py:194-201 with gw columns?
__global__ __launch_bounds__(256) void check_grid_kernel(const int *__restrict__ ind, long long n_tri, long long gw,
                                                         unsigned *__restrict__ mismatch) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    bool bad = false;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n_tri; t += stride) {
        const Idx3 e = tri_vertices<true>(nullptr, gw, t);
        bad |= ind[3 * t] != e.a || ind[3 * t + 1] != e.b || ind[3 * t + 2] != e.c;
    }
    if (bad) *mismatch = 1u;
}

// Is an index array a FILTERED regular grid (the triangles of surface.py:194-201 in their order, some
// removed -- what get_colored_surface returns for a DSM with nodata, surface.py:203-205)?  Pass 1, per
// triangle of the array: it must be a grid triangle, later in grid order than its predecessor; its bit is
// set in `present`, its vertices are marked.  Pass 2, per grid triangle NOT in the array: one of its
// vertices must be unmarked -- then "draw the triangles whose three vertices are marked" draws exactly the
// array, and the mesh is rendered by the implicit-grid kernels with that vertex mask.
__device__ __forceinline__ long long subgrid_id(const int *__restrict__ ind, long long t, long long gw, long long gh) {
    const long long a = ind[3 * t], b = ind[3 * t + 1], c = ind[3 * t + 2];
    int type;
    if (b == a + gw && c == a + gw + 1) type = 0;
    else if (b == a + gw + 1 && c == a + 1) type = 1;
    else return -1;
    const long long row = a / gw, col = a - row * gw;
    if (a < 0 || row >= gh - 1 || col >= gw - 1) return -1;
    return 2 * (row * (gw - 1) + col) + type;
}

__global__ __launch_bounds__(256) void subgrid_mark_kernel(const int *__restrict__ ind, long long n_tri, long long gw, long long gh,
                                                           unsigned *__restrict__ present, unsigned char *__restrict__ mark,
                                                           unsigned *__restrict__ mismatch) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    bool bad = false;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n_tri; t += stride) {
        const long long id = subgrid_id(ind, t, gw, gh);
        if (id < 0 || (t > 0 && subgrid_id(ind, t - 1, gw, gh) >= id)) { bad = true; continue; }
        atomicOr(&present[id >> 5], 1u << (id & 31));
        mark[ind[3 * t]] = 1;
        mark[ind[3 * t + 1]] = 1;
        mark[ind[3 * t + 2]] = 1;
    }
    if (bad) *mismatch = 1u;
}

__global__ __launch_bounds__(256) void subgrid_absent_kernel(long long n_grid_tri, long long gw, const unsigned *__restrict__ present,
                                                             const unsigned char *__restrict__ mark, unsigned *__restrict__ mismatch) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    bool bad = false;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n_grid_tri; id += stride) {
        if (present[id >> 5] >> (id & 31) & 1u) continue;
        const Idx3 e = tri_vertices<true>(nullptr, gw, id);
        bad |= mark[e.a] && mark[e.b] && mark[e.c];
    }
    if (bad) *mismatch = 1u;
}

// rank[w] = number of set bits in present[0 .. w): block sums, then (after the host scanned them) the words
__global__ __launch_bounds__(256) void subgrid_blocksum_kernel(const unsigned *__restrict__ present, long long n_words,
                                                               unsigned *__restrict__ block_sums) {
    __shared__ unsigned s[4];
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    unsigned c = w < n_words ? (unsigned)__popc(present[w]) : 0u;
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void subgrid_rank_kernel(const unsigned *__restrict__ present, long long n_words,
                                                           const unsigned *__restrict__ block_offsets, unsigned *__restrict__ rank) {
    __shared__ unsigned s[256];
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    s[threadIdx.x] = w < n_words ? (unsigned)__popc(present[w]) : 0u;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                  // Hillis-Steele inclusive scan
        const unsigned t = threadIdx.x >= (unsigned)d ? s[threadIdx.x - d] : 0u;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    if (w < n_words) rank[w] = block_offsets[blockIdx.x] + (threadIdx.x ? s[threadIdx.x - 1] : 0u);
}

// visibility words with the grid's triangle ids -> positions in the caller's (filtered) index array
__global__ __launch_bounds__(256) void vis_translate_kernel(const unsigned long long *__restrict__ vis, long long npix,
                                                            const unsigned *__restrict__ present, const unsigned *__restrict__ rank,
                                                            unsigned long long *__restrict__ out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    unsigned long long key = vis[p];
    if (key) {
        const unsigned id = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
        const unsigned pos = rank[id >> 5] + (unsigned)__popc(present[id >> 5] & ((1u << (id & 31)) - 1u));
        key = (key & 0xFFFFFFFF00000000ull) | (unsigned long long)(0xFFFFFFFFu - pos);
    }
    out[p] = key;
}

// valid = derived AND (user mask or all ones)
__global__ __launch_bounds__(256) void mask_and_kernel(const unsigned char *__restrict__ derived, const unsigned char *__restrict__ user,
                                                       long long n, unsigned char *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = derived[i] && (!user || user[i]);
}

// alp_mesh_create with float64 vertices / values (what get_colored_surface returns, surface.py:189-193): the cast of
// project.py:213-214 (``astype("f4")``: round to nearest even) on the device, chunk by chunk during the upload
__global__ __launch_bounds__(256) void cast_f64_f32_kernel(const double *__restrict__ src, long long count, long long dst_off,
                                                           float *__restrict__ dst) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[dst_off + i] = (float)src[i];
}

// an out-of-range index would fault in the raster kernels: counted on the device (the host loop over 6e8
// indices of a 100 M-vertex mesh took longer than their upload)
__global__ __launch_bounds__(256) void check_index_range_kernel(const int *__restrict__ ind, long long count, long long n_vert,
                                                                unsigned *__restrict__ bad) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned mine = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const int v = ind[i];
        mine += (v < 0 || v >= n_vert) ? 1u : 0u;
    }
    if (mine) atomicAdd(bad, mine);
}

__global__ __launch_bounds__(256) void narrow_indices_kernel(const long long *__restrict__ src, long long count,
                                                             long long dst_off, int *__restrict__ dst, long long n_vert,
                                                             unsigned *__restrict__ bad) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned mine = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const long long v = src[i];
        mine += (v < 0 || v >= n_vert) ? 1u : 0u;
        dst[dst_off + i] = (int)v;
    }
    if (mine) atomicAdd(bad, mine);
}

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
        ALP_HIP(hipMalloc((void **)&m->compact_counts, (size_t)chunks * sizeof(unsigned)));
        ALP_HIP(hipMalloc((void **)&m->compact_offsets, (size_t)(chunks + 1) * sizeof(unsigned long long)));
        m->compact_cap = chunks;
    }
    hipStream_t st = ctx().stream;
    ktime_begin();
    hipLaunchKernelGGL(valid_count_kernel, dim3(chunks), dim3(256), 0, st, m->image, npix, m->compact_counts);
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, st, m->compact_counts, chunks, m->compact_offsets);
    ktime_end();
    ALP_HIP(hipGetLastError());
    unsigned long long total = 0;
    ALP_HIP(hipMemcpyAsync(&total, m->compact_offsets + chunks, sizeof(total), hipMemcpyDeviceToHost, st));
    ALP_HIP(hipStreamSynchronize(st));
    m->valid_total = (int64_t)total;
    *count = m->valid_total;
    return ALP_OK;
}

int frame_valid_write(alp_mesh *m, const double *offsets, unsigned *idx_dev, double *xyz_dev) {
    const long long npix = (long long)m->w * m->h;
    const int chunks = (int)((npix + COMPACT_CHUNK - 1) / COMPACT_CHUNK);
    const double o0 = offsets ? offsets[0] : 0.0, o1 = offsets ? offsets[1] : 0.0, o2 = offsets ? offsets[2] : 0.0;
    ktime_begin();
    hipLaunchKernelGGL(valid_write_kernel, dim3(chunks), dim3(256), 0, ctx().stream, m->image, npix, m->compact_offsets, o0, o1,
                       o2, idx_dev, xyz_dev);
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
        auto drain_parked = [&](int round) -> int {
            unsigned *items = fcount + QC_STRIDE * round;
            const unsigned *cap = round ? m->park_cap_b : m->park_cap;
            const int wgs = round ? cu * 2 : cu * 8;
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
                               fcount + 2, m->park_cap[0], m->park_cap[1], m->park_cap[2], patch_near);
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
                                   m->park_cap_b[0], m->park_cap_b[1], m->park_cap_b[2], patch_far);
                ALP_HIP(hipGetLastError());
                if (int e = drain_rare(0)) return e;
                if (int e = drain_parked(1)) return e;
            }
#ifdef ALP_RASTER_STATS
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
        if (items <= m->qcap && general <= m->gcap && park_ok) break;
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
    if (hipMalloc((void **)&m->vert, (size_t)n_vert * 12) != hipSuccess) return bail(fail(ALP_EHIP, "hipMalloc vert"));
    if ((rc = upload_f32(m->vert, vert, vert_dtype, n_vert))) return bail(rc);
    if (value) {
        if (hipMalloc((void **)&m->value, (size_t)n_vert * 12) != hipSuccess) return bail(fail(ALP_EHIP, "hipMalloc value"));
        if ((rc = upload_f32(m->value, value, value_dtype, n_vert))) return bail(rc);
    }
    if (hipMalloc((void **)&m->qcount_dev, QC_TOTAL * sizeof(unsigned)) != hipSuccess ||
        hipHostMalloc((void **)&m->qcount_host, QC_TOTAL * sizeof(unsigned), hipHostMallocDefault) != hipSuccess)
        return bail(fail(ALP_EHIP, "hipMalloc queue counter"));
    if (!implicit && n_tri > 0) {
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
    // The index array the reference builds (surface.py:194-201) is the full regular grid unless
    // nodata triangles were filtered out: recognise it, drop the 12 B/triangle array and use the
    // LDS-tiled grid kernel (same triangle ids, same result, no index traffic).
    if (!implicit && n_tri >= 2 && (n_tri & 1) == 0 && !getenv("ALP_NO_GRID_DETECT")) {   // env: keep the index path (tests, benchmarks)
        long long first[3];
        for (int k = 0; k < 3; ++k)
            first[k] = ind_dtype == ALP_I32 ? (long long)((const int *)ind)[k] : ((const long long *)ind)[k];
        const long long gw = first[1] - first[0];
        if (first[0] == 0 && gw >= 2 && first[2] == gw + 1 && n_vert % gw == 0) {
            const long long gh = n_vert / gw;
            if (gh >= 2 && n_tri == 2 * (gh - 1) * (gw - 1)) {
                if (hipMemsetAsync(m->qcount_dev, 0, sizeof(unsigned), ctx().stream) != hipSuccess)
                    return bail(fail(ALP_EHIP, "grid check: memset"));
                hipLaunchKernelGGL(check_grid_kernel, dim3(ctx().cu_count * 8), dim3(256), 0, ctx().stream, m->ind,
                                   (long long)n_tri, gw, m->qcount_dev);
                hipError_t e = hipMemcpyAsync(m->qcount_host, m->qcount_dev, sizeof(unsigned), hipMemcpyDeviceToHost,
                                              ctx().stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);
                if (e != hipSuccess) return bail(fail(ALP_EHIP, "grid check: %s", hipGetErrorString(e)));
                if (*m->qcount_host == 0) {
                    hipFree(m->ind);
                    m->ind = nullptr;
                    m->implicit = true;
                    m->grid_h = gh;
                    m->grid_w = gw;
                }
            }
        }
    }
    // ... and when they were (surface.py:203-205), the grid with a vertex mask
    if (!implicit && !m->implicit && n_tri >= 1 && !getenv("ALP_NO_GRID_DETECT")) {
        long long first[3];
        for (int k = 0; k < 3; ++k)
            first[k] = ind_dtype == ALP_I32 ? (long long)((const int *)ind)[k] : ((const long long *)ind)[k];
        if ((rc = try_subgrid(m, first))) return bail(rc);
    }
    *out = m;
    return ALP_OK;
}

int alp_mesh_destroy(alp_mesh_t *m) {
    if (!m) return ALP_OK;
    if (ctx().ready) hipStreamSynchronize(ctx().stream);
    for (void *p : {(void *)m->vert, (void *)m->value, (void *)m->ind, (void *)m->valid, (void *)m->valid_derived,
                    (void *)m->tri_present, (void *)m->tri_rank, (void *)m->vis, (void *)m->image,
                    (void *)m->queue, (void *)m->gqueue, (void *)m->qcount_dev, (void *)m->compact_counts, (void *)m->compact_offsets,
                    (void *)m->tile_bounds, (void *)m->tile_lists, (void *)m->hiz, (void *)m->park_small, (void *)m->park_cell,
                    (void *)m->rz_points})
        if (p) hipFree(p);
    if (m->qcount_host) hipHostFree(m->qcount_host);
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
    return m->implicit ? render_impl<true>(m, v, rc, min_distance, cached) : render_impl<false>(m, v, rc, min_distance, cached);
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
    if (int rc = frame_valid_write(m, offsets, idx_dev, xyz_dev)) return rc;
    hipError_t e = hipMemcpyAsync(xyz_out, xyz_dev, xyz_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(idx_out, idx_dev, idx_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(ALP_EHIP, "alp_render_fetch_valid: %s", hipGetErrorString(e));
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
