// Part of alp_rasterize.hip (one translation unit, included inside namespace alp in the order given there; not a
// stand-alone header): the cell of every point, and the aggregation of the runs of equal cells in the cell-sorted order -- the sequential
// Kahan walk (pandas' order), the order-free pieces + join, the checks that choose between them, the packing of byte-valued bands.
#pragma once

// cell (row * width + col, project.py:435-436) and index of every point
__global__ __launch_bounds__(256) void rz_cell_kernel(const double *__restrict__ x, const double *__restrict__ y, long long n,
                                                      double x_min, double y_max, double res, int width, int height,
                                                      unsigned *__restrict__ cell, unsigned *__restrict__ idx,
                                                      unsigned char *__restrict__ tile_used, int tiles_x) {      // idx NULL: the slot holds packed band values
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long col = (long long)((x[i] - x_min) / res);
        long long row = (long long)((y_max - y[i]) / res);
        col = col < 0 ? 0 : (col > width - 1 ? width - 1 : col);
        row = row < 0 ? 0 : (row > height - 1 ? height - 1 : row);
        cell[i] = (unsigned)(row * width + col);
        if (idx) idx[i] = (unsigned)i;
        if (tile_used) tile_used[(row >> 5) * tiles_x + (col >> 6)] = 1;      // RZ_TH = 32, RZ_TW = 64; every writer writes 1
    }
}

// Runs of equal cell in the (stably) cell-sorted order: the points of a run are the rows of one pandas group in their
// original order.  Each run is aggregated band by band, skipping NaN like pandas does:
//   mean   libgroupby.group_mean: Kahan summation  y = v - c; t = s + y; c = (t - s) - y; s = t  (c reset to 0 when it
//          turns NaN: an infinite value), then s / count -- checked against pandas 2.3 bit for bit (tests)
//   max / min   order-free
// and its float32 cast (project.py:459) goes into the NaN-filled raster.  One thread per run: the recurrence is sequential,
// the loads are not -- a run is walked eight points at a time, all their gathers in flight together (next to the camera
// thousands of camera pixels share a cell: with one gather per turn such a run alone took a millisecond, and a wave that
// ran the recurrence for 64 points with operands broadcast from lane to lane -- every lane computing the same -- took as
// long: measured 2.9 and 1.07 ms for the 11.7 M points of the 100 M-vertex frame).  A run's end is found by galloping and
// bisection, not by a load per point.
template <int AGG>
struct RzAcc {
    double s = 0.0, comp = 0.0, m = AGG == AGG_MAX ? -INFINITY : INFINITY;
    long long cnt = 0;
    __device__ __forceinline__ void take(double v) {
        if (v != v) return;
        ++cnt;
        if constexpr (AGG == AGG_MEAN) {
            const double yv = v - comp, t = s + yv;
            comp = (t - s) - yv;
            if (comp != comp) comp = 0.0;
            s = t;
        } else if constexpr (AGG == AGG_MAX) {
            m = v > m ? v : m;
        } else {
            m = v < m ? v : m;
        }
    }
    __device__ __forceinline__ float result() const { return AGG == AGG_MEAN ? (float)(s / (double)cnt) : (float)m; }
};

template <int AGG, int NB>
__device__ __forceinline__ void rz_walk_run(const unsigned *__restrict__ idx_s, const double *__restrict__ values, long long i,
                                            long long j, int nb, int b0, unsigned cell, long long hw, float *__restrict__ raster) {
    RzAcc<AGG> acc[NB];
    long long k = i;
    for (; k + 8 <= j; k += 8) {
        unsigned id[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) id[u] = idx_s[k + u];
        double v[8][NB];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int g = 0; g < NB; ++g) v[u][g] = values[(long long)id[u] * nb + b0 + g];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int g = 0; g < NB; ++g) acc[g].take(v[u][g]);
    }
    for (; k < j; ++k) {
        const double *row = values + (long long)idx_s[k] * nb + b0;
#pragma unroll
        for (int g = 0; g < NB; ++g) acc[g].take(row[g]);
    }
#pragma unroll
    for (int g = 0; g < NB; ++g)
        if (acc[g].cnt) raster[(long long)(b0 + g) * hw + cell] = acc[g].result();
}

template <int AGG>
__global__ __launch_bounds__(256) void rz_runs_kernel(const unsigned *__restrict__ cell_s, const unsigned *__restrict__ idx_s,
                                                      const double *__restrict__ values, long long n, int nb, long long hw,
                                                      float *__restrict__ raster) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned c = cell_s[i];
        if (i > 0 && cell_s[i - 1] == c) continue;             // not the head of a run
        long long lo = i, step = 1;                            // cell_s[lo] is in the run
        while (lo + step < n && cell_s[lo + step] == c) { lo += step; step <<= 1; }
        long long hi = lo + step < n ? lo + step : n;         // cell_s[hi] is not (or hi == n)
        while (hi - lo > 1) {
            const long long mid = lo + ((hi - lo) >> 1);
            if (cell_s[mid] == c) lo = mid; else hi = mid;
        }
        int b0 = 0;
        for (; b0 + 4 <= nb; b0 += 4) rz_walk_run<AGG, 4>(idx_s, values, i, hi, nb, b0, c, hw, raster);
        if (nb - b0 == 3) rz_walk_run<AGG, 3>(idx_s, values, i, hi, nb, b0, c, hw, raster);
        else if (nb - b0 == 2) rz_walk_run<AGG, 2>(idx_s, values, i, hi, nb, b0, c, hw, raster);
        else if (nb - b0 == 1) rz_walk_run<AGG, 1>(idx_s, values, i, hi, nb, b0, c, hw, raster);
    }
}

// ---- order-free aggregates in parallel pieces
// max / min never depend on the order, and neither does the mean of INTEGER-valued bands (image bytes in float64 columns:
// Kahan's compensation stays exactly 0 and every partial sum below 2^53 is exact: rz_integer_check_kernel bounds the values by
// min(2^31, 2^53 / n), n = the longest run there can be).  Then a run need not be walked by one
// thread -- next to the camera thousands of camera pixels share a cell, and the longest run alone set the kernel's time
// (1.1 ms of 2.2 for the 100 M-vertex frame).  rz_pieces_kernel: a thread per RZ_SEG consecutive sorted positions walks them,
// finishes the runs that lie inside and leaves (sum, count) of the at most two pieces that cross its borders;
// rz_join_kernel: the thread whose segment holds a crossing run's head adds the pieces of the segments after it.
constexpr int RZ_SEG = 16;
struct RzPiece {
    double v;            // sum, or max / min
    unsigned cnt;
    unsigned pad;
};

template <int AGG>
__device__ __forceinline__ void rz_piece_take(RzPiece &p, double v) {
    if (v != v) return;
    ++p.cnt;
    if constexpr (AGG == AGG_MEAN) p.v += v;
    else if constexpr (AGG == AGG_MAX) p.v = v > p.v ? v : p.v;
    else p.v = v < p.v ? v : p.v;
}
template <int AGG>
__device__ __forceinline__ void rz_piece_join(RzPiece &p, const RzPiece &q) {
    p.cnt += q.cnt;
    if constexpr (AGG == AGG_MEAN) p.v += q.v;
    else if constexpr (AGG == AGG_MAX) p.v = q.v > p.v ? q.v : p.v;
    else p.v = q.v < p.v ? q.v : p.v;
}
template <int AGG>
__device__ __forceinline__ float rz_piece_result(const RzPiece &p) {
    return AGG == AGG_MEAN ? (float)(p.v / (double)p.cnt) : (float)p.v;
}

// first[t * nb + b]: the piece that CONTINUES a run from segment t - 1 (it starts at the segment's first position);
// last[t * nb + b]: the piece that starts a run inside segment t (or at its first position) and continues into t + 1
// PACKED: `id` is not the point's index but its (at most four) byte-valued band values, one byte each -- the sort carried them
// along as its payload, nothing is gathered (image bytes: the reference's own use, project.py:364 on a uint8 photograph)
template <int AGG, int NB, bool PACKED>
__device__ __forceinline__ void rz_pieces_bands(const unsigned (&cs)[RZ_SEG], const unsigned (&id)[RZ_SEG], unsigned before, unsigned after,
                                                int count, const double *__restrict__ values, int nb, int b0, long long hw, long long t,
                                                float *__restrict__ raster, RzPiece *__restrict__ first, RzPiece *__restrict__ last) {
    const double ident = AGG == AGG_MEAN ? 0.0 : (AGG == AGG_MAX ? -INFINITY : INFINITY);
    RzPiece pc[NB];
#pragma unroll
    for (int g = 0; g < NB; ++g) pc[g] = {ident, 0u, 0u};
    bool from_head = before != cs[0];
#pragma unroll
    for (int u = 0; u < RZ_SEG; ++u) {
        if (u >= count) break;
        if constexpr (PACKED) {
#pragma unroll
            for (int g = 0; g < NB; ++g) rz_piece_take<AGG>(pc[g], (double)((id[u] >> (8 * (b0 + g))) & 0xFFu));
        } else {
            const double *row = values + (long long)id[u] * nb + b0;
#pragma unroll
            for (int g = 0; g < NB; ++g) rz_piece_take<AGG>(pc[g], row[g]);
        }
        const unsigned nextc = u + 1 < count ? cs[u + 1 < RZ_SEG ? u + 1 : 0] : after;
        if (nextc != cs[u]) {                          // the run ends here
#pragma unroll
            for (int g = 0; g < NB; ++g) {
                if (from_head) { if (pc[g].cnt) raster[(long long)(b0 + g) * hw + cs[u]] = rz_piece_result<AGG>(pc[g]); }
                else first[t * nb + b0 + g] = pc[g];
                pc[g].v = ident; pc[g].cnt = 0u;
            }
            from_head = true;
        } else if (u + 1 == count) {                   // ... or goes on in the next segment
#pragma unroll
            for (int g = 0; g < NB; ++g) {
                if (from_head) last[t * nb + b0 + g] = pc[g]; else first[t * nb + b0 + g] = pc[g];
            }
        }
    }
}

template <int AGG, bool PACKED = false>
__global__ __launch_bounds__(256) void rz_pieces_kernel(const unsigned *__restrict__ cell_s, const unsigned *__restrict__ idx_s,
                                                        const double *__restrict__ values, long long n, int nb, long long hw,
                                                        float *__restrict__ raster, RzPiece *__restrict__ first,
                                                        RzPiece *__restrict__ last) {
    const long long nseg = (n + RZ_SEG - 1) / RZ_SEG, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nseg; t += stride) {
        const long long p0 = t * RZ_SEG, p1 = p0 + RZ_SEG < n ? p0 + RZ_SEG : n;
        unsigned cs[RZ_SEG], id[RZ_SEG];
#pragma unroll
        for (int u = 0; u < RZ_SEG; ++u) {
            cs[u] = p0 + u < p1 ? cell_s[p0 + u] : 0xFFFFFFFFu;
            id[u] = p0 + u < p1 ? idx_s[p0 + u] : 0u;
        }
        const unsigned before = p0 > 0 ? cell_s[p0 - 1] : 0xFFFFFFFFu, after = p1 < n ? cell_s[p1] : 0xFFFFFFFFu;
        const int count = (int)(p1 - p0);
        int b0 = 0;
        for (; b0 + 4 <= nb; b0 += 4) rz_pieces_bands<AGG, 4, PACKED>(cs, id, before, after, count, values, nb, b0, hw, t, raster, first, last);
        if (nb - b0 == 3) rz_pieces_bands<AGG, 3, PACKED>(cs, id, before, after, count, values, nb, b0, hw, t, raster, first, last);
        else if (nb - b0 == 2) rz_pieces_bands<AGG, 2, PACKED>(cs, id, before, after, count, values, nb, b0, hw, t, raster, first, last);
        else if (nb - b0 == 1) rz_pieces_bands<AGG, 1, PACKED>(cs, id, before, after, count, values, nb, b0, hw, t, raster, first, last);
    }
}

template <int AGG>
__global__ __launch_bounds__(256) void rz_join_kernel(const unsigned *__restrict__ cell_s, long long n, int nb, long long hw,
                                                      float *__restrict__ raster, const RzPiece *__restrict__ first,
                                                      const RzPiece *__restrict__ last) {
    const long long nseg = (n + RZ_SEG - 1) / RZ_SEG, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nseg; t += stride) {
        const long long p0 = t * RZ_SEG, p1 = p0 + RZ_SEG;
        if (p1 >= n) continue;                                   // the last segment: nothing goes on behind it
        const unsigned c = cell_s[p1 - 1];
        if (cell_s[p1] != c) continue;                           // no run leaves this segment
        if (cell_s[p0] == c && p0 > 0 && cell_s[p0 - 1] == c) continue;      // the run's head is in an earlier segment
        // segments t + 1 ... e: the run fills t + 1 ... e - 1 and ends in e.  Gallop + bisect on "position still in the run".
        long long lo = p1, step = RZ_SEG;
        while (lo + step < n && cell_s[lo + step] == c) { lo += step; step <<= 1; }
        long long hi = lo + step < n ? lo + step : n;
        while (hi - lo > 1) {
            const long long mid = lo + ((hi - lo) >> 1);
            if (cell_s[mid] == c) lo = mid; else hi = mid;
        }
        const long long e = lo / RZ_SEG;                       // segment of the run's last position
        for (int b = 0; b < nb; ++b) {
            RzPiece acc = last[t * nb + b];
            long long u = t + 1;
            for (; u + 4 <= e + 1; u += 4) {
                const RzPiece q0 = first[u * nb + b], q1 = first[(u + 1) * nb + b], q2 = first[(u + 2) * nb + b], q3 = first[(u + 3) * nb + b];
                rz_piece_join<AGG>(acc, q0); rz_piece_join<AGG>(acc, q1); rz_piece_join<AGG>(acc, q2); rz_piece_join<AGG>(acc, q3);
            }
            for (; u <= e; ++u) rz_piece_join<AGG>(acc, first[u * nb + b]);
            if (acc.cnt) raster[(long long)b * hw + c] = rz_piece_result<AGG>(acc);
        }
    }
}

// what do the bands hold?  flag bit 0: some value is not an integer of magnitude below `limit`; bit 1: some value is not a byte
// (an integer in [0, 255]; NaN is not a byte either: a packed value has no way to say "skip me").  `limit` is what makes the
// order-free mean exact: a run is at most n points long, so with |v| < limit = min(2^31, 2^53 / n) every partial sum of
// every run stays below 2^53 (rz_integer_limit)
__host__ __device__ inline double rz_integer_limit(long long n) {
    const double by_run = 9007199254740992.0 / (double)(n < 1 ? 1 : n);
    return by_run < 2147483648.0 ? by_run : 2147483648.0;
}
__global__ __launch_bounds__(256) void rz_integer_check_kernel(const double *__restrict__ values, long long count, double limit,
                                                               unsigned *__restrict__ flag) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const double v = values[i];
        if (v != v) { bad |= 2u; continue; }
        if (!(fabs(v) < limit && v == (double)(long long)v)) bad |= 3u;
        else if (!(v >= 0.0 && v <= 255.0)) bad |= 2u;
    }
    for (int m = 32; m >= 1; m >>= 1) bad |= (unsigned)__shfl_xor((int)bad, m, 64);
    if (bad && (threadIdx.x & 63) == 0 && (*flag & bad) != bad) atomicOr(flag, bad);
}

// byte-valued bands (nb <= 4) in float64 -> one packed word per point: the sort's payload.  values[i * es + b * bs]:
// interleaved rows (es = nb, bs = 1) or a table's columns as they lie (es = 1, bs = n)
__global__ __launch_bounds__(256) void rz_pack_kernel(const double *__restrict__ values, long long n, int nb, long long es, long long bs,
                                                      unsigned *__restrict__ packed) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned w = 0;
        for (int b = 0; b < nb; ++b) w |= ((unsigned)values[i * es + b * bs] & 0xFFu) << (8 * b);
        packed[i] = w;
    }
}
