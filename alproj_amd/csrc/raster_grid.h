// Part of alp_raster.hip (one translation unit, included inside namespace alp in the order given there; not a
// stand-alone header): raster_grid_kernel: one workgroup per listed tile of the implicit grid.
#pragma once

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
                        unsigned park_cap_large, unsigned park_cap_cell, int patch_cap,
                        ParkedTile *__restrict__ park_tiles, ParkedUnit *__restrict__ park_units, unsigned park_units_cap) {
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
#ifdef PARKED_TILES_LAB
    __shared__ int s_pbb[4];              // pixel box of everything parked (park_tiles)
    __shared__ unsigned s_unit0, s_rec, s_nunits;
#endif
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
#ifdef PARKED_TILES_LAB
        s_pbb[0] = INT_MAX; s_pbb[1] = INT_MIN; s_pbb[2] = INT_MAX; s_pbb[3] = INT_MIN;
#endif
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
#ifdef PARKED_TILES_LAB
    int pb_x0 = INT_MAX, pb_x1 = INT_MIN, pb_y0 = INT_MAX, pb_y1 = INT_MIN;      // snapped extent of this thread's parked entries
#endif
    for (unsigned e = threadIdx.x; e < ncell; e += 256) {
        const int id = (int)s_park[e];
        int lr, lc;
        cell_rc(id, lr, lc);
        const int ia = lr * GT_VW + lc, ib = ia + GT_VW, ic = ib + 1, idd = ia + 1;
        const int2 A = s_xy[ia], B = s_xy[ib], C = s_xy[ic], D = s_xy[idd];
#ifdef PARKED_TILES_LAB
        pb_x0 = min(pb_x0, min(min(A.x, B.x), min(C.x, D.x)));
        pb_x1 = max(pb_x1, max(max(A.x, B.x), max(C.x, D.x)));
        pb_y0 = min(pb_y0, min(min(A.y, B.y), min(C.y, D.y)));
        pb_y1 = max(pb_y1, max(max(A.y, B.y), max(C.y, D.y)));
#endif
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
#ifdef PARKED_TILES_LAB
        pb_x0 = min(pb_x0, min(A.x, min(B.x, C.x)));
        pb_x1 = max(pb_x1, max(A.x, max(B.x, C.x)));
        pb_y0 = min(pb_y0, min(A.y, min(B.y, C.y)));
        pb_y1 = max(pb_y1, max(A.y, max(B.y, C.y)));
#endif
        Deferred d;
        d.X[0] = A.x; d.X[1] = B.x; d.X[2] = C.x;
        d.Y[0] = A.y; d.Y[1] = B.y; d.Y[2] = C.y;
        d.iw[0] = s_iw[ia]; d.iw[1] = s_iw[k1]; d.iw[2] = s_iw[k2];
        d.t = 2u * ((unsigned)(r0 + lr) * (unsigned)(gw - 1) + (unsigned)(c0 + lc)) + (unsigned)half;
        const unsigned slot = s_park_base[large ? 1 : 0] + k;
        Deferred *queue = large ? park_large : park_small;
        if (slot < (large ? park_cap_large : park_cap_small)) queue[slot] = d;    // an overflow is noticed by finish_frame
    }
#ifdef PARKED_TILES_LAB
    if (park_tiles) {
        // the tile's record for raster_parked_tiles_kernel: where its entries are and which pixels they can touch;
        // record slot and first unit come from ONE 64-bit add (units in the low word), so both run in the same order
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            pb_x0 = min(pb_x0, __shfl_xor(pb_x0, m, 64));
            pb_x1 = max(pb_x1, __shfl_xor(pb_x1, m, 64));
            pb_y0 = min(pb_y0, __shfl_xor(pb_y0, m, 64));
            pb_y1 = max(pb_y1, __shfl_xor(pb_y1, m, 64));
        }
        if (lane == 0 && pb_x0 <= pb_x1) {
            atomicMin(&s_pbb[0], pb_x0); atomicMax(&s_pbb[1], pb_x1);
            atomicMin(&s_pbb[2], pb_y0); atomicMax(&s_pbb[3], pb_y1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            ParkedTile r;
            r.base[0] = s_park_base[0]; r.base[1] = s_park_base[1]; r.base[2] = s_park_base[2];
            r.n[0] = np_small; r.n[1] = np_large; r.n[2] = ncell;
            r.i0 = max((s_pbb[0] + SUB / 2 - 1) >> 8, 0); r.i1 = min((s_pbb[1] - SUB / 2) >> 8, v.w - 1);
            r.j0 = max((s_pbb[2] + SUB / 2 - 1) >> 8, 0); r.j1 = min((s_pbb[3] - SUB / 2) >> 8, v.h - 1);
            r.pad[0] = r.pad[1] = 0;
            unsigned nun = 0;
            if (r.i1 >= r.i0 && r.j1 >= r.j0) {
                const int nb = (((r.i1 - (r.i0 & ~7)) >> 6) + 1) * (((r.j1 - r.j0) >> 6) + 1);
                nun = (unsigned)min(nb, PT_MAX_UNITS);
            }
            s_nunits = nun;
            if (nun) {
                const unsigned long long old = atomicAdd((unsigned long long *)(park_counts + 4), (1ull << 32) | (unsigned long long)nun);
                s_unit0 = (unsigned)old;
                s_rec = (unsigned)(old >> 32);
                park_tiles[(unsigned)(old >> 32)] = r;         // at most one record per listed tile: the array holds them all
            }
        }
        __syncthreads();
        for (unsigned k = threadIdx.x; k < s_nunits; k += 256)
            if (s_unit0 + k < park_units_cap) {                // an overflow is noticed by finish_frame
                ParkedUnit u;
                u.rec = s_rec; u.slot = (unsigned short)k; u.nslots = (unsigned short)s_nunits;
                park_units[s_unit0 + k] = u;
            }
    }
#endif
#ifdef ALP_WG_TIMING
    __syncthreads();
#endif
    WGT(4);
}
