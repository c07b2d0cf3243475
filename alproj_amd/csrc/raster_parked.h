// Part of alp_raster.hip (one translation unit, included inside namespace alp in the order given there; not a
// stand-alone header): the consumers of the work raster_grid_kernel parks in device queues: cells (a wave per cell), larger triangles (a wave per
// triangle), small triangles (16 lanes per triangle); one launch per round.
#pragma once

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

#ifdef PARKED_TILES_LAB
// LAB (round 4; needs -DALP_DEV like every development switch: tools/build_variant.sh parked_tiles -DPARKED_TILES_LAB, then
// ALP_PARKED=tiles).  Measured on the 100 M-vertex frame and NOT adopted -- 585 us against raster_parked_kernel's 343 us, bit
// for bit the same image; profiles/r04_parked_tiles_lab.txt and DESIGN.md section 5 say why.
// ------------------------------------------------------------------ the same work, tile by tile through LDS depth patches
// raster_parked_kernel above deals queue ENTRIES to waves: every fragment is a global atomic, and what bounds the frame's
// raster stages is the chip's rate of atomic line-requests (22-27 G/s) -- 7.7 M of the frame's 15 M
// requests are this kernel's.  Here the work is dealt by SCREEN AREA instead.  raster_grid_kernel leaves one record per
// tile that parked something (its contiguous ranges in the three queues, the pixel box they can touch); that box is
// covered with PT_BIN x PT_BIN-pixel bins, and a workgroup draws everything of the tile that touches a bin into a 32 KB
// LDS depth patch (ds_max_u64: the same keys, and max is associative), then sends the patch out with consecutive lanes
// = consecutive pixels: 8 pixels per line-request, every pixel once per (tile, bin), overdraw inside the tile resolved
// in LDS.  With fragments going to LDS a lane can walk its OWN cell or triangle (stepped edge functions, the arithmetic
// of raster_grid_kernel's FAST path) -- the set-up is per lane instead of per wave on the scalar pipe, and no lane waits
// for the 8 x 8 window of somebody else's cell; only triangles with larger boxes are still walked by a whole wave.
// Units of (record, slot) are dealt to a persistent grid; a tile whose box covers many bins (the nearest tiles: 100+) is
// spread over up to PT_MAX_UNITS units.  Same image bit for bit (tests: the frozen oracle, g15 / g16).
template <int NE>
__device__ __forceinline__ void patch_walk(const int (&ex)[NE], const int (&ey)[NE], int (&row)[NE], const int (&bs)[NE],
                                           int ci0, int ci1, int cj0, int cj1, int bi0, int bj0, float iwa, float iwb, float iwc,
                                           float iwd, float inv0, float inv1, unsigned long long lo0,
                                           unsigned long long *__restrict__ patch) {
    // NE == 6: a cell, edges 0..2 = triangle (a, b, c), 3..5 = (a, c, d); NE == 3: one triangle (a, b, c)
    for (int j = cj0; j <= cj1; ++j) {
        int u[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) u[k] = row[k];
        unsigned long long *prow = patch + ((j - bj0) << 6) - bi0;
        for (int i = ci0; i <= ci1; ++i) {
            unsigned long long key = 0;
            if ((u[0] | u[1] | u[2]) >= 0) {
                const float q = __builtin_fmaf((float)(u[2] + bs[2]), iwc,
                                               __builtin_fmaf((float)(u[1] + bs[1]), iwb, (float)(u[0] + bs[0]) * iwa)) * inv0;
                key = ((unsigned long long)__float_as_uint(q) << 32) | lo0;
            }
            if constexpr (NE == 6) {
                if ((u[3] | u[4] | u[5]) >= 0) {
                    const float q = __builtin_fmaf((float)(u[5] + bs[5]), iwd,
                                                   __builtin_fmaf((float)(u[4] + bs[4]), iwc, (float)(u[3] + bs[3]) * iwa)) * inv1;
                    const unsigned long long k1 = ((unsigned long long)__float_as_uint(q) << 32) | (lo0 - 1u);
                    key = k1 > key ? k1 : key;
                }
            }
            if (key) __hip_atomic_fetch_max(prow + i, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int k = 0; k < NE; ++k) u[k] -= ey[k] * SUB;
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) row[k] += ex[k] * SUB;
    }
}

__global__ __launch_bounds__(256) void raster_parked_tiles_kernel(View v, unsigned long long *__restrict__ vis,
                                                                  const Deferred *__restrict__ small_q, const Deferred *__restrict__ large_q,
                                                                  const ParkedCell *__restrict__ cell_q,
                                                                  const unsigned *__restrict__ counts, unsigned cap_small,
                                                                  unsigned cap_large, unsigned cap_cell,
                                                                  const ParkedTile *__restrict__ recs, const ParkedUnit *__restrict__ units,
                                                                  unsigned units_cap) {
    __shared__ unsigned long long s_patch[PT_BIN * PT_BIN];
    const unsigned n_units = min(counts[4], units_cap);
    const int lane = (int)(threadIdx.x & 63);
    for (unsigned u = blockIdx.x; u < n_units; u += gridDim.x) {
        const ParkedUnit un = units[u];
        const ParkedTile rc = recs[un.rec];
        const int I0 = rc.i0 & ~7;
        const int nbx = ((rc.i1 - I0) >> 6) + 1, nby = ((rc.j1 - rc.j0) >> 6) + 1, nbins = nbx * nby;
        const unsigned n_small = rc.base[0] < cap_small ? min(rc.n[0], cap_small - rc.base[0]) : 0u;      // what overflowed was never written
        const unsigned n_large = rc.base[1] < cap_large ? min(rc.n[1], cap_large - rc.base[1]) : 0u;
        const unsigned n_cell = rc.base[2] < cap_cell ? min(rc.n[2], cap_cell - rc.base[2]) : 0u;
        for (int b = un.slot; b < nbins; b += un.nslots) {
            const int by = b / nbx, bx = b - by * nbx;
            const int bi0 = I0 + (bx << 6), bj0 = rc.j0 + (by << 6);
            const int bi1 = min(bi0 + PT_BIN - 1, rc.i1), bj1 = min(bj0 + PT_BIN - 1, rc.j1);
            const int words = (bj1 - bj0 + 1) << 6;
            for (int k = threadIdx.x; k < words; k += 256) s_patch[k] = 0ull;
            __syncthreads();
            // ---- cells: a lane walks its own cell's box (both triangles from five shared edge functions)
#ifndef PT_SKIP_CELLS           // development: the stages one by one (wrong image)
            for (unsigned e = threadIdx.x; e < n_cell; e += 256) {
                const ParkedCell pc = cell_q[rc.base[2] + e];
                const int ax = pc.X[0], bx_ = pc.X[1], cx = pc.X[2], dx_ = pc.X[3];
                const int ay = pc.Y[0], by_ = pc.Y[1], cy = pc.Y[2], dy_ = pc.Y[3];
                const int minx = min(min(ax, bx_), min(cx, dx_)), maxx = max(max(ax, bx_), max(cx, dx_));
                const int miny = min(min(ay, by_), min(cy, dy_)), maxy = max(max(ay, by_), max(cy, dy_));
                const int ci0 = max(max((minx + SUB / 2 - 1) >> 8, 0), bi0), ci1 = min(min((maxx - SUB / 2) >> 8, v.w - 1), bi1);
                const int cj0 = max(max((miny + SUB / 2 - 1) >> 8, 0), bj0), cj1 = min(min((maxy - SUB / 2) >> 8, v.h - 1), bj1);
                if (ci0 > ci1 || cj0 > cj1) continue;
                const int px = ci0 * SUB + SUB / 2, py = cj0 * SUB + SUB / 2;
                // directed edges: 0 b->c, 1 c->a, 2 a->b (triangle 0); 3 c->d, 4 d->a, 5 a->c (triangle 1)
                int ex[6] = {cx - bx_, ax - cx, bx_ - ax, dx_ - cx, ax - dx_, 0};
                int ey[6] = {cy - by_, ay - cy, by_ - ay, dy_ - cy, ay - dy_, 0};
                ex[5] = -ex[1];
                ey[5] = -ey[1];
                int bs[6], row[6];
                row[0] = mul24(ex[0], py - by_) - mul24(ey[0], px - bx_);
                row[1] = mul24(ex[1], py - cy) - mul24(ey[1], px - cx);
                row[2] = mul24(ex[2], py - ay) - mul24(ey[2], px - ax);
                row[3] = mul24(ex[3], py - cy) - mul24(ey[3], px - cx);
                row[4] = mul24(ex[4], py - dy_) - mul24(ey[4], px - dx_);
                row[5] = -row[1];
                const int area0 = row[0] + row[1] + row[2], area1 = row[3] + row[4] + row[5];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    bs[k] = 1 + (((ey[k] << 12) - ex[k]) >> 31);      // 0 where the edge owns its boundary (|dx| < 2^12: unclamped extent under 14 px)
                    row[k] -= bs[k];
                }
                const float inv0 = exact_rcp_unchecked((float)area0), inv1 = exact_rcp_unchecked((float)area1);   // used only where area > 0
                patch_walk<6>(ex, ey, row, bs, ci0, ci1, cj0, cj1, bi0, bj0, pc.iw[0], pc.iw[1], pc.iw[2], pc.iw[3], inv0, inv1,
                              (unsigned long long)(0xFFFFFFFFu - 2u * pc.cell), s_patch);
            }
#endif
            // ---- triangles with boxes of at most 8 x 8 centres: a lane walks its own triangle
#ifndef PT_SKIP_SMALL
            for (unsigned e = threadIdx.x; e < n_small; e += 256) {
                const Deferred d = small_q[rc.base[0] + e];
                const int minx = min(d.X[0], min(d.X[1], d.X[2])), maxx = max(d.X[0], max(d.X[1], d.X[2]));
                const int miny = min(d.Y[0], min(d.Y[1], d.Y[2])), maxy = max(d.Y[0], max(d.Y[1], d.Y[2]));
                const int ci0 = max(max((minx + SUB / 2 - 1) >> 8, 0), bi0), ci1 = min(min((maxx - SUB / 2) >> 8, v.w - 1), bi1);
                const int cj0 = max(max((miny + SUB / 2 - 1) >> 8, 0), bj0), cj1 = min(min((maxy - SUB / 2) >> 8, v.h - 1), bj1);
                if (ci0 > ci1 || cj0 > cj1) continue;
                const int px = ci0 * SUB + SUB / 2, py = cj0 * SUB + SUB / 2;
                const int area2 = mul24(d.X[1] - d.X[0], d.Y[2] - d.Y[0]) - mul24(d.X[2] - d.X[0], d.Y[1] - d.Y[0]);
                int ex[3], ey[3], bs[3], row[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int a = (k + 1) % 3, bb = (k + 2) % 3;
                    ex[k] = d.X[bb] - d.X[a];
                    ey[k] = d.Y[bb] - d.Y[a];
                    bs[k] = (ey[k] < 0 || (ey[k] == 0 && ex[k] > 0)) ? 0 : 1;
                    row[k] = mul24(ex[k], py - d.Y[a]) - mul24(ey[k], px - d.X[a]) - bs[k];
                }
                patch_walk<3>(ex, ey, row, bs, ci0, ci1, cj0, cj1, bi0, bj0, d.iw[0], d.iw[1], d.iw[2], 0.0f,
                              exact_rcp_unchecked((float)area2), 0.0f, (unsigned long long)(0xFFFFFFFFu - d.t), s_patch);
            }
#endif
            // ---- triangles with larger boxes (under 64 px): found by a lane each, walked by the whole wave in 8 x 8 blocks
#ifndef PT_SKIP_LARGE
            for (unsigned e0 = (unsigned)(threadIdx.x & ~63u); e0 < n_large; e0 += 256) {
                const unsigned e = e0 + (unsigned)lane;
                Deferred d;
                bool hit = false;
                if (e < n_large) {
                    d = large_q[rc.base[1] + e];
                    const int minx = min(d.X[0], min(d.X[1], d.X[2])), maxx = max(d.X[0], max(d.X[1], d.X[2]));
                    const int miny = min(d.Y[0], min(d.Y[1], d.Y[2])), maxy = max(d.Y[0], max(d.Y[1], d.Y[2]));
                    hit = max((minx + SUB / 2 - 1) >> 8, bi0) <= min((maxx - SUB / 2) >> 8, bi1) &&
                          max((miny + SUB / 2 - 1) >> 8, bj0) <= min((maxy - SUB / 2) >> 8, bj1);
                }
                unsigned long long mask = __ballot(hit);
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
                    const int minx = min(X[0], min(X[1], X[2])), maxx = max(X[0], max(X[1], X[2]));
                    const int miny = min(Y[0], min(Y[1], Y[2])), maxy = max(Y[0], max(Y[1], Y[2]));
                    const int ci0 = max(max((minx + SUB / 2 - 1) >> 8, 0), bi0), ci1 = min(min((maxx - SUB / 2) >> 8, v.w - 1), bi1);
                    const int cj0 = max(max((miny + SUB / 2 - 1) >> 8, 0), bj0), cj1 = min(min((maxy - SUB / 2) >> 8, v.h - 1), bj1);
                    const int area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (X[2] - X[0]) * (Y[1] - Y[0]);
                    int dx[3], dy[3], bias[3], xa[3], ya[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int a = (k + 1) % 3, bb = (k + 2) % 3;
                        dx[k] = X[bb] - X[a];
                        dy[k] = Y[bb] - Y[a];
                        xa[k] = X[a];
                        ya[k] = Y[a];
                        bias[k] = (dy[k] < 0 || (dy[k] == 0 && dx[k] > 0)) ? 0 : 1;
                    }
                    const float inv_area = exact_rcp_unchecked((float)area2);
                    const unsigned long long lo = (unsigned long long)(0xFFFFFFFFu - t);
                    const int lx = lane & 7, ly = lane >> 3;
                    for (int yy = cj0; yy <= cj1; yy += 8)
                        for (int xx = ci0 & ~7; xx <= ci1; xx += 8) {
                            const int i = xx + lx, j = yy + ly;
                            if (i < ci0 || i > ci1 || j > cj1) continue;
                            const int px = i * SUB + SUB / 2, py = j * SUB + SUB / 2;
                            const int w0 = mul24(dx[0], py - ya[0]) - mul24(dy[0], px - xa[0]) - bias[0];
                            const int w1 = mul24(dx[1], py - ya[1]) - mul24(dy[1], px - xa[1]) - bias[1];
                            const int w2 = mul24(dx[2], py - ya[2]) - mul24(dy[2], px - xa[2]) - bias[2];
                            if ((w0 | w1 | w2) >= 0) {
                                const float q = __builtin_fmaf((float)(w2 + bias[2]), iw3[2],
                                                               __builtin_fmaf((float)(w1 + bias[1]), iw3[1],
                                                                              (float)(w0 + bias[0]) * iw3[0])) * inv_area;
                                __hip_atomic_fetch_max(&s_patch[((j - bj0) << 6) + (i - bi0)],
                                                       ((unsigned long long)__float_as_uint(q) << 32) | lo, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                }
            }
#endif
            __syncthreads();
            // ---- the patch goes out: consecutive lanes = consecutive pixels of a row
            for (int k = threadIdx.x; k < words; k += 256) {
                const unsigned long long key = s_patch[k];
                if (key) vis_max(vis, v, bi0 + (k & 63), bj0 + (k >> 6), key);
            }
            __syncthreads();
        }
    }
}
#endif   // PARKED_TILES_LAB
