// Part of alp_raster.hip (one translation unit, included inside namespace alp in the order given there; not a
// stand-alone header): the frame plan of an implicit-grid mesh: tiles of 64 x 16 cells, their bounding boxes (once per mesh), frustum culling and
// the NEAR / FAR split, the depth pyramid of the first round and the occlusion test of the FAR tiles.
#pragma once

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
