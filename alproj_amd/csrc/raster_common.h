// Part of alp_raster.hip (one translation unit, included inside namespace alp in the order given there; not a
// stand-alone header): constants, the view, and the per-triangle device helpers every raster kernel shares: view transform, window
// coordinates, exact reciprocal, snapped set-up, pixel key, bounding-box walk, near-plane clipping, emit_small / emit_general.
#pragma once

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
