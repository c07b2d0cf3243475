// libalproj_hip.so -- compute part of to_geotiff(), src/alproj/project.py:376-503 (SURVEY.md
// 8(f) row f2): reverse-projected points -> regular raster.
//
//   project.py:435-436  col = int((x - x_min) / res) clipped to [0, width-1],
//                       row = int((y_max - y) / res) clipped to [0, height-1]
//   project.py:450-459  per band: groupby (row, col), aggregate (mean / max / min; NaN values are
//                       skipped like pandas does), store into a float32 raster, NaN elsewhere
//   project.py:462-479  ceil(max_dist / res) sweeps: every NaN cell takes the NaN-aware
//                       aggregate of its 3x3 neighbourhood of the PREVIOUS sweep (NaN outside
//                       the raster); the reference does it with scipy.ndimage.generic_filter
//                       and a Python lambda per pixel
//   project.py:483-485  NaN -> nodata, clip to [0, 255], truncate to uint8
// The GeoTIFF file itself (rasterio) stays on the host side of the ABI.
//
// Kernels (round 4: sort-based, no atomics, no accumulators): the cell of every point, ONE stable radix sort of (cell, point)
// shared by all bands, then a thread per run of equal cells walks its points IN THEIR ORIGINAL ORDER and forms pandas'
// aggregate -- for the mean the Kahan-compensated float64 sum of libgroupby's group_mean, so that the float64 value, its
// float32 cast and the truncated byte are the reference's for ANY band values, not only for byte-valued ones (the atomics of
// rounds 2-3 added in arrival order: exact for integers below 2^53, one ulp off for general floats); the median sorts one
// composite 64-bit key (cell : order-preserving float32 value) per band.  Then the fused tail (float32 raster -> focal
// sweeps in LDS -> bytes) or, for more than RZ_SMAX sweeps, separate sweep / conversion kernels.  The 3x3 mean adds its
// window in numpy's order (pairwise block of 8, then the ninth).
#include "alp_raster_internal.h"

#include <algorithm>
#include <cmath>

namespace alp {

enum { AGG_MEAN = 0, AGG_MAX = 1, AGG_MIN = 2 };

// ALP_RZ_SEPARATE_PASSES=1: sweeps and byte conversion as separate kernels over the whole float32 raster whatever the sweep
// count (the path for more than RZ_SMAX sweeps); the tests run both and compare bytes
static bool rz_separate_passes() {
    const char *e = getenv("ALP_RZ_SEPARATE_PASSES");
    return e && e[0] == '1';
}

// order-preserving map double -> uint64 (radix-sort keys of the two-sort median)
__device__ __forceinline__ unsigned long long d2ord(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

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
// Kahan's compensation stays exactly 0 and every partial sum below 2^53 is exact).  Then a run need not be walked by one
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

// what do the bands hold?  flag bit 0: some value is not an integer of magnitude below 2^31; bit 1: some value is not a byte
// (an integer in [0, 255]; NaN is not a byte either: a packed value has no way to say "skip me")
__global__ __launch_bounds__(256) void rz_integer_check_kernel(const double *__restrict__ values, long long count,
                                                               unsigned *__restrict__ flag) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const double v = values[i];
        if (v != v) { bad |= 2u; continue; }
        if (!(fabs(v) < 2147483648.0 && v == (double)(long long)v)) bad |= 3u;
        else if (!(v >= 0.0 && v <= 255.0)) bad |= 2u;
    }
    for (int m = 32; m >= 1; m >>= 1) bad |= (unsigned)__shfl_xor((int)bad, m, 64);
    if (bad && (threadIdx.x & 63) == 0 && (*flag & bad) != bad) atomicOr(flag, bad);
}

// byte-valued bands (nb <= 4), interleaved float64 -> one packed word per point: the sort's payload
__global__ __launch_bounds__(256) void rz_pack_kernel(const double *__restrict__ values, long long n, int nb, unsigned *__restrict__ packed) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned w = 0;
        for (int b = 0; b < nb; ++b) w |= ((unsigned)values[i * nb + b] & 0xFFu) << (8 * b);
        packed[i] = w;
    }
}

// one sweep of the NaN-only 3x3 focal fill
template <int AGG>
__global__ __launch_bounds__(256) void rz_focal_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                       int nb, int width, int height) {
    const long long hw = (long long)width * height;
    const long long total = hw * nb;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const float centre = src[i];
        if (centre == centre) { dst[i] = centre; continue; }
        const long long p = i % hw;
        const int row = (int)(p / width), col = (int)(p - (long long)row * width);
        const float *band = src + (i - p);
        double w[9];
        int k = 0, have = 0;
#pragma unroll
        for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc, ++k) {
                const int rr = row + dr, cc = col + dc;
                float val = __int_as_float(0x7fc00000);
                if (rr >= 0 && rr < height && cc >= 0 && cc < width) val = band[(long long)rr * width + cc];
                const bool ok = val == val;
                have += ok;
                if constexpr (AGG == AGG_MEAN) w[k] = ok ? (double)val : 0.0;           // nansum: NaN -> 0
                else if constexpr (AGG == AGG_MAX) w[k] = ok ? (double)val : -INFINITY;
                else w[k] = ok ? (double)val : INFINITY;
            }
        float out = __int_as_float(0x7fc00000);
        if (have) {
            if constexpr (AGG == AGG_MEAN) {
                // numpy's pairwise sum of 9 contiguous doubles: block of 8, then the rest
                const double s = (((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]))) + w[8];
                out = (float)(s / (double)have);
            } else {
                double m = w[0];
#pragma unroll
                for (int j = 1; j < 9; ++j) m = (AGG == AGG_MAX) ? fmax(m, w[j]) : fmin(m, w[j]);
                out = (float)m;
            }
        }
        dst[i] = out;
    }
}

__global__ __launch_bounds__(256) void rz_to_u8_kernel(const float *__restrict__ raster, long long total, int nodata,
                                                       unsigned char *__restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const float v = raster[i];
        unsigned char o;
        if (v != v) o = (unsigned char)nodata;
        else o = (unsigned char)(v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v));      // clip, then truncate
        out[i] = o;
    }
}

// ------------------------------------------------------------------ fused tail
// up to RZ_SMAX focal sweeps + uint8 in ONE pass over the raster: a workgroup owns a tile of RZ_TW x RZ_TH cells, loads the
// float32 raster of the tile and a halo of S cells into LDS (a cell S sweeps later depends on the cells within S of it,
// nothing else), sweeps there -- each sweep is valid on a region one cell smaller all round -- and writes bytes only.
// Every value is formed by the expressions of the separate kernels above (which stay as the path for more sweeps), so the
// bytes are the same; the raster does not cross HBM as float32 once per sweep and once more for the conversion, and a
// tile whose own and neighbouring tiles hold no point -- most of a georectified photograph's bounding box -- reads nothing.
constexpr int RZ_TW = 64, RZ_TH = 32, RZ_SMAX = 8;
enum { AGG_MEDIAN_FOCAL = 3 };

template <int AGG>
__device__ __forceinline__ float rz_window_value(const float *__restrict__ s, int lw, int at) {
    const float nan = __int_as_float(0x7fc00000);
    if constexpr (AGG == AGG_MEDIAN_FOCAL) {
        // the window's values in order, its NaN behind them as +inf: a 25-exchange network for nine (no loop whose length
        // differs from lane to lane: the insertion sort of rz_focal_median_kernel took 0.36 ms of the 100 M-vertex frame's tail,
        // this 0.1x).  Equal values (and +-0) may come out in another order than there: the same numbers, and the tail writes bytes
        float w[9];
        int have = 0, k = 0;
#pragma unroll
        for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc, ++k) {
                const float val = s[at + dr * lw + dc];
                const bool ok = val == val;
                have += ok;
                w[k] = ok ? val : INFINITY;
            }
        if (!have) return nan;
#define RZ_CE(i, j) { const float lo = w[i] < w[j] ? w[i] : w[j], hi = w[i] < w[j] ? w[j] : w[i]; w[i] = lo; w[j] = hi; }
        RZ_CE(0, 3) RZ_CE(1, 7) RZ_CE(2, 5) RZ_CE(4, 8)
        RZ_CE(0, 7) RZ_CE(2, 4) RZ_CE(3, 8) RZ_CE(5, 6)
        RZ_CE(0, 2) RZ_CE(1, 3) RZ_CE(4, 5) RZ_CE(7, 8)
        RZ_CE(1, 4) RZ_CE(3, 6) RZ_CE(5, 7)
        RZ_CE(0, 1) RZ_CE(2, 4) RZ_CE(3, 5) RZ_CE(6, 8)
        RZ_CE(2, 3) RZ_CE(4, 5) RZ_CE(6, 7)
        RZ_CE(1, 2) RZ_CE(3, 4) RZ_CE(5, 6)
#undef RZ_CE
        const int ka = (have - 1) >> 1, kb = have >> 1;
        float a = w[0], b = w[0];
#pragma unroll
        for (int u = 1; u < 9; ++u) { a = ka == u ? w[u] : a; b = kb == u ? w[u] : b; }
        return (have & 1) ? a : (float)(((double)a + (double)b) / 2);
    } else {
        double w[9];
        int k = 0, have = 0;
#pragma unroll
        for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc, ++k) {
                const float val = s[at + dr * lw + dc];
                const bool ok = val == val;
                have += ok;
                if constexpr (AGG == AGG_MEAN) w[k] = ok ? (double)val : 0.0;
                else if constexpr (AGG == AGG_MAX) w[k] = ok ? (double)val : -INFINITY;
                else w[k] = ok ? (double)val : INFINITY;
            }
        if (!have) return nan;
        if constexpr (AGG == AGG_MEAN) {
            const double sum = (((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]))) + w[8];     // numpy's order, as above
            return (float)(sum / (double)have);
        } else {
            double m = w[0];
#pragma unroll
            for (int j = 1; j < 9; ++j) m = (AGG == AGG_MAX) ? fmax(m, w[j]) : fmin(m, w[j]);
            return (float)m;
        }
    }
}

// the float32 raster the run kernels wrote (NaN = empty cell) -> S sweeps of the aggregate's own 3x3 window -> bytes
// NaN into the float32 raster of the tiles that hold a point (all bands); the tail never reads the others
// (the tiles: rz_tile_list_kernel's list; an entry = tile number | the 3 x 3 neighbourhood's "holds a point" bits << 20, bit 4 the
// tile itself)
constexpr int RZ_TILE_BITS = 20;
__global__ __launch_bounds__(256) void rz_fill_tiles_kernel(float *__restrict__ raster, const unsigned *__restrict__ list,
                                                            const unsigned *__restrict__ list_count, int nb, int width, int height,
                                                            int tiles_x) {
    const unsigned count = *list_count;
    const long long hw = (long long)width * height;
    const float nan = __int_as_float(0x7fc00000);
    for (unsigned e = blockIdx.x; e < count; e += gridDim.x) {
        const unsigned entry = list[e];
        if (!((entry >> (RZ_TILE_BITS + 4)) & 1u)) continue;
        const int t = (int)(entry & ((1u << RZ_TILE_BITS) - 1u));
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        for (int k = threadIdx.x; k < RZ_TW * RZ_TH * nb; k += 256) {
            const int b = k / (RZ_TW * RZ_TH), r = (k / RZ_TW) % RZ_TH, c = k % RZ_TW;
            const int gr = ty * RZ_TH + r, gc = tx * RZ_TW + c;
            if (gr < height && gc < width) raster[b * hw + (long long)gr * width + gc] = nan;
        }
    }
}

// the tiles a sweep can reach -- those with a point in their own or one of their eight neighbouring tiles -- as a compact list
// (any order): the fill and the tail walk it instead of launching a workgroup per tile of a mostly empty raster
__global__ __launch_bounds__(256) void rz_tile_list_kernel(const unsigned char *__restrict__ tile_used, int tiles_x, int tiles_y,
                                                           unsigned *__restrict__ list, unsigned *__restrict__ list_count) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= tiles_x * tiles_y) return;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    unsigned used9 = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = ty + dy, xx = tx + dx;
            if (yy >= 0 && yy < tiles_y && xx >= 0 && xx < tiles_x && tile_used[yy * tiles_x + xx]) used9 |= 1u << (3 * (dy + 1) + dx + 1);
        }
    if (used9) list[atomicAdd(list_count, 1u)] = (unsigned)t | (used9 << RZ_TILE_BITS);
}

// list: the tiles a sweep can reach (rz_tile_list_kernel) -- most of a georectified photograph's bounding box is empty; the
// other tiles keep the nodata launch_tail filled `out` with (one wide fill instead of byte stores tile by tile), and cells
// of neighbouring tiles without points are NaN without being read (they were never filled)
template <int AGG>
__global__ __launch_bounds__(256) void rz_tail_kernel(const float *__restrict__ raster, int width, int height, int S,
                                                      int nodata, int tiles_x, int nb, unsigned char *__restrict__ out,
                                                      const unsigned *__restrict__ list, const unsigned *__restrict__ list_count) {
    extern __shared__ float rz_tail_lds[];                   // two rasters of (RZ_TH + 2 S) x (RZ_TW + 2 S) floats: 18 KB at S = 1, 31 KB at S = 8
    __shared__ int s_any;
    const float nan = __int_as_float(0x7fc00000);
    const int tid = (int)threadIdx.x;
    const long long hw = (long long)width * height;
    const int lw = RZ_TW + 2 * S, lh = RZ_TH + 2 * S;
    const float inv_lw = 1.0f / (float)lw;                       // idx / lw through (idx + 0.5) * (1 / lw): idx < 3840, exact
    const unsigned work = list_count[0] * (unsigned)nb;          // (tile, band) pairs, the bands of a tile next to each other
    for (unsigned item = blockIdx.x; item < work; item += gridDim.x) {
        __syncthreads();                                             // the previous item's LDS is read no more
        if (tid == 0) s_any = 0;
        __syncthreads();
        const unsigned entry = list[item / (unsigned)nb];
        const int t = (int)(entry & ((1u << RZ_TILE_BITS) - 1u));
        const unsigned used9 = entry >> RZ_TILE_BITS;                // which of the 3 x 3 tiles around this one hold points (bit 3 * (dy + 1) + (dx + 1))
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const long long band_base = (long long)(item % (unsigned)nb) * hw;
        const int x0 = tx * RZ_TW - S, y0 = ty * RZ_TH - S;        // raster position of LDS cell (0, 0)
        float *buf_cur = rz_tail_lds, *buf_nxt = rz_tail_lds + lw * lh;
        bool any = false;
        for (int idx = tid; idx < lw * lh; idx += 256) {
            const int r = (int)(((float)idx + 0.5f) * inv_lw), c = idx - r * lw;
            const int gr = y0 + r, gc = x0 + c;
            float v = nan;                                           // outside the raster: NaN, in every sweep
            if (gr >= 0 && gr < height && gc >= 0 && gc < width) {
                const int dy = r < S ? 0 : (r >= S + RZ_TH ? 2 : 1), dx = c < S ? 0 : (c >= S + RZ_TW ? 2 : 1);
                if ((used9 >> (3 * dy + dx)) & 1u) v = raster[band_base + (long long)gr * width + gc];
            }
            buf_cur[idx] = v;
            any |= (v == v);
        }
        if (any) s_any = 1;
        __syncthreads();
        if (s_any) {
            for (int s = 0; s < S; ++s) {
                const int rw = lw - 2 * (s + 1), rh = lh - 2 * (s + 1);
                const float inv_rw = 1.0f / (float)rw;
                for (int idx = tid; idx < rw * rh; idx += 256) {
                    int r = (int)(((float)idx + 0.5f) * inv_rw), c = idx - r * rw;
                    r += s + 1;
                    c += s + 1;
                    const int at = r * lw + c;
                    float o = buf_cur[at];
                    if (o != o) {
                        const int gr = y0 + r, gc = x0 + c;
                        if (gr >= 0 && gr < height && gc >= 0 && gc < width) o = rz_window_value<AGG>(buf_cur, lw, at);
                    }
                    buf_nxt[at] = o;
                }
                __syncthreads();
                float *const done = buf_cur;
                buf_cur = buf_nxt;
                buf_nxt = done;
            }
        }
        for (int idx = tid; idx < RZ_TW * RZ_TH; idx += 256) {
            const int r = idx / RZ_TW, c = idx % RZ_TW;
            const int gr = y0 + S + r, gc = x0 + S + c;
            if (gr >= height || gc >= width) continue;
            const float v = buf_cur[(r + S) * lw + c + S];
            unsigned char o;
            if (v != v) o = (unsigned char)nodata;
            else o = (unsigned char)(v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v));      // clip, then truncate
            out[band_base + (long long)gr * width + gc] = o;
        }
    }
}

// ------------------------------------------------------------------ median
// groupby median needs the values of every cell in order.  Image bands are bytes or float32 in float64 columns: when every
// non-NaN value of the band IS a float32 (checked on the device), ONE radix sort of the composite key
// (cell : order-preserving float32 bits) per band puts every cell's values in order, and the middle key(s) of a run ARE the
// median's operands.  Any other band takes two stable sorts (by value, then by cell) as before.
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) { return __uint_as_float((o >> 31) ? (o & 0x7fffffffu) : ~o); }

// what kind of values do the bands hold?  flag[band] bit 0: some value is not a float32; bit 1: some value is not an integer in
// [0, 65535]; bit 2: not an integer in [0, 255] (image bytes and 16-bit samples: their composite key needs 8 / 16 value bits, three
// / two radix passes fewer); bit 3: some value is NaN (bytes without one can ride the sort as its payload: rz_median_packed_kernel).
// All bands in one launch: one wait of the host instead of one per band.
__global__ __launch_bounds__(256) void rz_median_check_kernel(const double *__restrict__ values, long long count, int nb,
                                                              unsigned *__restrict__ flag) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const double val = values[i];
        unsigned bad = 0;
        if (val != val) bad = 8u;
        else {
            if ((double)(float)val != val) bad |= 1u;
            const bool whole = val >= 0.0 && val <= 65535.0 && val == (double)(unsigned)val;
            if (!whole) bad |= 6u;
            else if (val > 255.0) bad |= 4u;
        }
        if (bad) {
            unsigned *f = flag + (int)(i % nb);
            if ((*f & bad) != bad) atomicOr(f, bad);           // a plain look first: the word settles after a few writers
        }
    }
}

// VBITS = 32: key = cell : order-preserving float32 bits; VBITS = 16 / 8: key = cell : the integer itself
template <int VBITS>
__global__ __launch_bounds__(256) void rz_median_key_kernel(const unsigned *__restrict__ cell, const double *__restrict__ values,
                                                            long long n, int nb, int band, unsigned long long *__restrict__ key) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double val = values[i * nb + band];
        const unsigned long long lo = VBITS == 32 ? (unsigned long long)f2ord((float)val) : (unsigned long long)(unsigned)val;
        key[i] = val != val ? ~0ull : (((unsigned long long)cell[i] << VBITS) | lo);      // NaN: behind every cell (cells are below 2^31)
    }
}

template <int VBITS>
__global__ __launch_bounds__(256) void rz_median_runs32_kernel(const unsigned long long *__restrict__ key_s, long long n,
                                                               float *__restrict__ raster_band) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long k0 = key_s[i];
        const unsigned c = (unsigned)(k0 >> VBITS);
        if (k0 == ~0ull || (i > 0 && (unsigned)(key_s[i - 1] >> VBITS) == c)) continue;      // NaN tail, or not the head of a run
        // the run's end: gallop, then bisect (a load per element would cost a run of thousands a millisecond)
        long long lo = i, step = 1;                            // key_s[lo] is in the run
        while (lo + step < n && key_s[lo + step] != ~0ull && (unsigned)(key_s[lo + step] >> VBITS) == c) { lo += step; step <<= 1; }
        long long hi = lo + step < n ? lo + step : n;         // key_s[hi] is not (or hi == n)
        while (hi - lo > 1) {
            const long long mid = lo + ((hi - lo) >> 1);
            if (key_s[mid] != ~0ull && (unsigned)(key_s[mid] >> VBITS) == c) lo = mid; else hi = mid;
        }
        const long long k = hi - i;
        const unsigned long long ka = key_s[i + (k - 1) / 2], kb = key_s[i + k / 2];
        const double a = VBITS == 32 ? (double)ord2f((unsigned)ka) : (double)(unsigned)(ka & ((1ull << VBITS) - 1ull));
        const double b = VBITS == 32 ? (double)ord2f((unsigned)kb) : (double)(unsigned)(kb & ((1ull << VBITS) - 1ull));
        raster_band[c] = (float)((k & 1) ? a : (a + b) / 2);
    }
}

// ---- byte-valued bands (at most four, no NaN): ONE sort by cell with the packed values as its payload (the mean's sort), then
// the middle value(s) of every run are SELECTED from its words -- the order inside a run does not matter to a median.  A run
// of up to 16 points (nearly all: the frame's cells hold 1.5 points on average) is sorted in the registers of the thread at
// its head (a bitonic network over its bytes, padded with 256); a longer one (the 100 M-vertex frame: 138 000 of 1.74 M runs,
// up to 671 points, holding 46 % of the points) is taken by the whole wave: its bytes are counted into a 256-bin histogram in
// LDS, four bins to a lane, and a prefix sum over the lanes finds the bin of the middle.  (A list of the long runs for a
// second kernel, appended to with one atomic per run: 1.1 ms -- the 138 000 atomics on one word.)
template <int N>
__device__ __forceinline__ void rz_sort_small(unsigned (&a)[N]) {
#pragma unroll
    for (int k = 2; k <= N; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const unsigned lo = a[i] < a[l] ? a[i] : a[l], hi = a[i] < a[l] ? a[l] : a[i];
                    a[i] = up ? lo : hi;
                    a[l] = up ? hi : lo;
                }
            }
}
template <int N>
__device__ __forceinline__ unsigned rz_pick(const unsigned (&a)[N], int k) {
    unsigned r = a[0];
#pragma unroll
    for (int u = 1; u < N; ++u) r = k == u ? a[u] : r;
    return r;
}
__device__ __forceinline__ float rz_middle(unsigned lo, unsigned hi, long long len) {       // pandas' median of a group: its middle
    return (float)((len & 1) ? (double)lo : ((double)lo + (double)hi) / 2);                 // value, or the mean of the two
}
template <int N>
__device__ __forceinline__ void rz_median_small(const unsigned *__restrict__ pay_s, long long i, int len, int nb, unsigned cell,
                                                long long hw, float *__restrict__ raster) {
    unsigned w[N];
#pragma unroll
    for (int u = 0; u < N; ++u) w[u] = u < len ? pay_s[i + u] : 0u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g >= nb) break;
        unsigned a[N];
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] = u < len ? ((w[u] >> (8 * g)) & 0xFFu) : 256u;
        rz_sort_small<N>(a);
        raster[(long long)g * hw + cell] = rz_middle(rz_pick<N>(a, (len - 1) >> 1), rz_pick<N>(a, len >> 1), len);
    }
}

// one wave (a workgroup of 64) per 64 consecutive sorted positions: the heads among them take their runs
__global__ __launch_bounds__(64) void rz_median_packed_kernel(const unsigned *__restrict__ cell_s, const unsigned *__restrict__ pay_s,
                                                              long long n, int nb, long long hw, float *__restrict__ raster) {
    __shared__ unsigned hist[4][256];
    const int lane = (int)threadIdx.x;
    const long long chunks = (n + 63) >> 6;
    for (long long ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
        const long long base = ch << 6, i = base + lane;
        const unsigned c = i < n ? cell_s[i] : 0xFFFFFFFFu;
        const bool head = i < n && (i == 0 || cell_s[i - 1] != c);
        const unsigned long long heads = __ballot(head);
        long long hi = i + 1;
        if (head) {
            const unsigned long long later = lane < 63 ? heads >> (lane + 1) : 0ull;
            if (later) hi = i + 1 + __builtin_ctzll(later);           // the next head among the 64
            else {                                                      // the run reaches the end of the 64: gallop, then bisect (rz_runs_kernel)
                long long lo = base + 63 < n ? base + 63 : n - 1, step = 1;
                while (lo + step < n && cell_s[lo + step] == c) { lo += step; step <<= 1; }
                hi = lo + step < n ? lo + step : n;
                while (hi - lo > 1) {
                    const long long mid = lo + ((hi - lo) >> 1);
                    if (cell_s[mid] == c) lo = mid; else hi = mid;
                }
            }
        }
        const unsigned len = head ? (unsigned)(hi - i) : 0u;
        if (len == 1) {
            const unsigned w = pay_s[i];
            for (int g = 0; g < nb; ++g) raster[(long long)g * hw + c] = (float)((w >> (8 * g)) & 0xFFu);
        } else if (len == 2) {
            const unsigned w0 = pay_s[i], w1 = pay_s[i + 1];
            for (int g = 0; g < nb; ++g) raster[(long long)g * hw + c] = rz_middle((w0 >> (8 * g)) & 0xFFu, (w1 >> (8 * g)) & 0xFFu, 2);
        } else if (len > 2 && len <= 4) rz_median_small<4>(pay_s, i, (int)len, nb, c, hw, raster);
        else if (len > 4 && len <= 8) rz_median_small<8>(pay_s, i, (int)len, nb, c, hw, raster);
        else if (len > 8 && len <= 16) rz_median_small<16>(pay_s, i, (int)len, nb, c, hw, raster);
        // the longer runs, one after the other, by the whole wave
        unsigned long long longs = __ballot(len > 16u);
        while (longs) {
            const int src = __builtin_ctzll(longs);
            longs &= longs - 1;
            const long long ri = base + src;
            const unsigned rlen = __shfl(len, src), rcell = __shfl(c, src);
            for (int q = lane; q < nb * 256; q += 64) (&hist[0][0])[q] = 0u;
            __syncthreads();
            for (unsigned j = (unsigned)lane; j < rlen; j += 64u) {
                const unsigned w = pay_s[ri + j];
                for (int g = 0; g < nb; ++g) atomicAdd(&hist[g][(w >> (8 * g)) & 0xFFu], 1u);
            }
            __syncthreads();
            for (int g = 0; g < nb; ++g) {
                const uint4 cnt = *(const uint4 *)&hist[g][4 * lane];      // this lane's four bins
                const unsigned sum = cnt.x + cnt.y + cnt.z + cnt.w;
                unsigned upto = sum;                                        // inclusive prefix over the lanes
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned o = __shfl_up(upto, d);
                    if (lane >= d) upto += o;
                }
                const unsigned before = upto - sum;
                unsigned mid[2];
                for (int q = 0; q < 2; ++q) {
                    const unsigned k = q == 0 ? (rlen - 1u) >> 1 : rlen >> 1;      // the k-th smallest (from 0)
                    unsigned v = 4u * (unsigned)lane, r = k - before;
                    if (r >= cnt.x) { r -= cnt.x; ++v; if (r >= cnt.y) { r -= cnt.y; ++v; if (r >= cnt.z) ++v; } }
                    mid[q] = __shfl(v, __builtin_ctzll(__ballot(before <= k && k < upto)));
                }
                if (lane == 0) raster[(long long)g * hw + rcell] = rz_middle(mid[0], mid[1], (long long)rlen);
            }
            __syncthreads();                                                // the histograms are read no more
        }
    }
}

__global__ __launch_bounds__(256) void rz_median_keys_kernel(const unsigned *__restrict__ cell_in, const double *__restrict__ values,
                                                             long long n, int nb, int band, unsigned long long *__restrict__ vkey,
                                                             unsigned *__restrict__ idx, unsigned *__restrict__ cell) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double val = values[i * nb + band];
        vkey[i] = d2ord(val);
        idx[i] = (unsigned)i;
        cell[i] = (val != val) ? 0xFFFFFFFFu : cell_in[i];      // NaN: sorts behind every pixel
    }
}

__global__ __launch_bounds__(256) void rz_gather_cell_kernel(const unsigned *__restrict__ idx_sorted,
                                                             const unsigned *__restrict__ cell, long long n,
                                                             unsigned *__restrict__ cell_sorted) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        cell_sorted[i] = cell[idx_sorted[i]];
}

// runs of equal pixel in the (pixel, value)-sorted order -> median into the float32 raster
__global__ __launch_bounds__(256) void rz_median_runs_kernel(const unsigned *__restrict__ cell_sorted,
                                                             const unsigned *__restrict__ idx_sorted,
                                                             const double *__restrict__ values, long long n, int nb,
                                                             int band, float *__restrict__ raster_band) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned c = cell_sorted[i];
        if (c == 0xFFFFFFFFu || (i > 0 && cell_sorted[i - 1] == c)) continue;      // not the head of a run
        long long lo = i, step = 1;                            // gallop, then bisect, as above
        while (lo + step < n && cell_sorted[lo + step] == c) { lo += step; step <<= 1; }
        long long hi = lo + step < n ? lo + step : n;
        while (hi - lo > 1) {
            const long long mid = lo + ((hi - lo) >> 1);
            if (cell_sorted[mid] == c) lo = mid; else hi = mid;
        }
        const long long k = hi - i;
        const double a = values[(long long)idx_sorted[i + (k - 1) / 2] * nb + band];
        const double b = values[(long long)idx_sorted[i + k / 2] * nb + band];
        raster_band[c] = (float)((k & 1) ? a : (a + b) / 2);
    }
}

__global__ __launch_bounds__(256) void rz_focal_median_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                              int nb, int width, int height) {
    const long long hw = (long long)width * height;
    const long long total = hw * nb;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const float centre = src[i];
        if (centre == centre) { dst[i] = centre; continue; }
        const long long p = i % hw;
        const int row = (int)(p / width), col = (int)(p - (long long)row * width);
        const float *band = src + (i - p);
        float w[9];
        int have = 0;
        for (int dr = -1; dr <= 1; ++dr)
            for (int dc = -1; dc <= 1; ++dc) {
                const int rr = row + dr, cc = col + dc;
                if (rr < 0 || rr >= height || cc < 0 || cc >= width) continue;
                const float val = band[(long long)rr * width + cc];
                if (val != val) continue;
                int k = have++;                                   // insertion sort of at most 9 values
                while (k > 0 && w[k - 1] > val) { w[k] = w[k - 1]; --k; }
                w[k] = val;
            }
        float out = __int_as_float(0x7fc00000);
        if (have) out = (have & 1) ? w[have / 2] : (float)(((double)w[have / 2 - 1] + (double)w[have / 2]) / 2);
        dst[i] = out;
    }
}

}  // namespace alp

#include <rocprim/rocprim.hpp>

namespace alp {

enum { AGG_MEDIAN = 3 };

// The (cell, payload) pair sort with NINE bits per onesweep pass (workgroups of 1024 x 8 items): a raster of up to 2^27 cells
// is sorted in three passes where the library's tuned gfx950 configuration (8 bits) takes four, and a pass costs no more --
// 11.7 M pairs, 27-bit keys: 0.273 ms against 0.432; 30 bits 0.348 against 0.431; 18 bits 0.199 against 0.329
// (tools/sort_rate.hip: 6, 7, 9, 10, 12 items, 512 threads, 10 and 7 bits all measured slower).  The 64-bit composite keys of a
// median whose bands are not bytes take it too: 59 bits 0.602 ms against 0.672.
using RzPairSort = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 9, rocprim::block_radix_rank_algorithm::match>>;

template <int AGG>
static void launch_tail(const float *raster, int nb, int width, int height, int sweeps, int nodata, unsigned char *out_dev,
                        const unsigned *list, const unsigned *list_count) {
    const int tiles_x = (width + RZ_TW - 1) / RZ_TW, tiles_y = (height + RZ_TH - 1) / RZ_TH;
    const size_t lds = 2 * sizeof(float) * (size_t)(RZ_TW + 2 * sweeps) * (size_t)(RZ_TH + 2 * sweeps);
    const long long items = (long long)tiles_x * tiles_y * nb;
    // 32 workgroups per CU, four times what is resident: a tile at the edge of the points costs many times one inside them
    // (every NaN cell next to a value takes the 3 x 3 window), so the hardware's dispatcher evens the shares out (measured on the
    // 100 M-vertex frame, mean / median tail: 139 / 566 us at 8 per CU, 103 / 358 at 32, 109 / 369 at 128; a work counter drawn
    // from with an atomic: 197 / 368 -- profiles/r04_f2_kernel_breakdown.txt)
    const unsigned grid = (unsigned)std::min<long long>(items, (long long)ctx().cu_count * 32);
    (void)hipMemsetAsync(out_dev, nodata & 0xFF, (size_t)width * height * nb, ctx().stream);       // an error: hipGetLastError below
    hipLaunchKernelGGL((rz_tail_kernel<AGG>), dim3(grid), dim3(256), lds, ctx().stream, raster, width, height, sweeps, nodata, tiles_x,
                       nb, out_dev, list, list_count);
}

// bytes of device scratch run_rasterize needs for n points (sort buffers + rocPRIM's temporary storage)
static size_t rz_sort_bytes(long long n, size_t *tmp_out) {
    size_t t1 = 0, t2 = 0, t3 = 0;
    const size_t count = (size_t)n;
    rocprim::radix_sort_pairs<RzPairSort>(nullptr, t1, (unsigned *)nullptr, (unsigned *)nullptr, (unsigned *)nullptr, (unsigned *)nullptr, count, 0u, 32u,
                              ctx().stream);
    rocprim::radix_sort_keys<RzPairSort>(nullptr, t2, (unsigned long long *)nullptr, (unsigned long long *)nullptr, count, 0u, 64u, ctx().stream);
    rocprim::radix_sort_pairs(nullptr, t3, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (unsigned *)nullptr,
                              (unsigned *)nullptr, count, 0u, 64u, ctx().stream);
    const size_t tmp = std::max(t1, std::max(t2, t3));
    if (tmp_out) *tmp_out = tmp;
    // cell, idx (x 2: in / out) | 64-bit keys x 2 | third cell array of the two-sort median | flag, tile count | tile bytes, tile
    // list (at most 2^RZ_TILE_BITS tiles: the fused tail's condition) | temporary storage
    return (size_t)n * (16 + 16 + 4) + 256 + ((size_t)5 << RZ_TILE_BITS) + 256 + tmp + 256;
}

// dx, dy, dv: the points on the device (values interleaved n x nb); ra, rb: float32 rasters (rb only for the separate-pass
// path); `sort_area`: rz_sort_bytes(n) bytes.  agg: AGG_MEAN / _MAX / _MIN / AGG_MEDIAN.
// do the order-free pieces fit where the median keeps its 64-bit keys?  (they do unless the table is tiny or very wide)
static bool rz_pieces_fit(int agg, long long n, int nb) {
    const long long nseg = (n + RZ_SEG - 1) / RZ_SEG;
    return agg != AGG_MEDIAN && (size_t)nseg * nb * 2 * sizeof(RzPiece) <= (size_t)n * 16 && !getenv("ALP_RZ_SEQUENTIAL");
}
// may byte-valued bands ride through the sort as its payload?
static bool rz_can_pack(int agg, long long n, int nb) {
    return nb <= 4 && (agg == AGG_MEDIAN || rz_pieces_fit(agg, n, nb)) && !getenv("ALP_RZ_NO_PACKED");
}
static unsigned *rz_payload_slot(char *sort_area, long long n) { return (unsigned *)sort_area + 2 * n; }

static int run_rasterize(int agg, const double *dx, const double *dy, const double *dv, long long n, int nb, double x_min,
                         double y_max, double res, int width, int height, int sweeps, int nodata, float *ra, float *rb,
                         unsigned char *out_dev, char *sort_area, float **f32_out = nullptr, bool packed_ready = false) {
    // packed_ready: the bands are bytes (nb <= 4) and the caller has already put their packed words into the sort's payload slot
    // (rz_payload_slot); dv is not read then
    hipStream_t st = ctx().stream;
    const long long hw = (long long)width * height, total = hw * nb;
    const int cu = ctx().cu_count;
    auto grid = [&](long long items) {
        const long long want = (items + 255) / 256;
        return (unsigned)(want < 1 ? 1 : (want < (long long)cu * 8 ? want : (long long)cu * 8));
    };
    size_t tmp = 0;
    rz_sort_bytes(n, &tmp);
    const size_t count = (size_t)n;
    unsigned *cell = (unsigned *)sort_area, *cell_s = cell + n, *idx = cell_s + n, *idx_s = idx + n;
    unsigned long long *key = (unsigned long long *)(idx_s + n), *key_s = key + n;
    unsigned *cell3 = (unsigned *)(key_s + n);
    unsigned *flag = (unsigned *)(((uintptr_t)(cell3 + n) + 63) & ~(uintptr_t)63);
    unsigned *tile_count = flag + 8;
    unsigned *tile_list = flag + 16;
    unsigned char *tile_used = (unsigned char *)(tile_list + ((size_t)1 << RZ_TILE_BITS));
    void *sort_tmp = (void *)(((uintptr_t)(tile_used + ((size_t)1 << RZ_TILE_BITS)) + 255) & ~(uintptr_t)255);
    const int tiles_x = (width + RZ_TW - 1) / RZ_TW, tiles_y = (height + RZ_TH - 1) / RZ_TH;
    // the fused tail reads the tiles that hold points only; the separate-pass path reads every cell (and takes the rasters of
    // more tiles than the list holds: a raster one cell wide and 2^31 tall has 2^26 of them)
    const bool fused = sweeps <= RZ_SMAX && !rz_separate_passes() && !f32_out && (long long)tiles_x * tiles_y <= (1ll << RZ_TILE_BITS);
    unsigned cell_bits = 1;
    while (cell_bits < 32 && (1ll << cell_bits) < hw) ++cell_bits;
    if (fused) {
        ALP_HIP(hipMemsetAsync(tile_used, 0, (size_t)tiles_x * tiles_y, st));
        ALP_HIP(hipMemsetAsync(tile_count, 0, sizeof(unsigned), st));
    } else {
        ALP_HIP(hipMemsetD32Async((hipDeviceptr_t)ra, 0x7fc00000, (size_t)total, st));      // NaN everywhere (the runtime's fill)
    }
    // the order-free cases go through parallel pieces (the pieces live where the median keeps its 64-bit keys); byte-valued bands
    // (at most four) travel through the sort as its payload
    const long long nseg = (n + RZ_SEG - 1) / RZ_SEG;
    bool pieces = rz_pieces_fit(agg, n, nb);
    bool packed = packed_ready;
    if (packed_ready && !rz_can_pack(agg, n, nb)) return fail(ALP_EINVAL, "rasterisation: packed band values on a path that does not take them");
    unsigned kinds[64];             // the median's look at its bands (rz_median_check_kernel)
    unsigned *const kinds_dev = (unsigned *)key_s;
    if (!packed_ready && agg == AGG_MEDIAN) {
        ALP_HIP(hipMemsetAsync(kinds_dev, 0, (size_t)nb * sizeof(unsigned), st));
        hipLaunchKernelGGL(rz_median_check_kernel, dim3(grid(n * nb)), dim3(256), 0, st, dv, n * nb, nb, kinds_dev);
        ALP_HIP(hipMemcpyAsync(kinds, kinds_dev, (size_t)nb * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        ALP_HIP(hipStreamSynchronize(st));
        packed = rz_can_pack(agg, n, nb);
        for (int b = 0; b < nb; ++b) packed = packed && !(kinds[b] & (4u | 8u));
        if (packed) hipLaunchKernelGGL(rz_pack_kernel, dim3(grid(n)), dim3(256), 0, st, dv, n, nb, idx);
    }
    if (!packed_ready && agg != AGG_MEDIAN && pieces && (agg == AGG_MEAN || nb <= 4)) {
        unsigned kind = 3;
        ALP_HIP(hipMemsetAsync(flag, 0, sizeof(unsigned), st));
        hipLaunchKernelGGL(rz_integer_check_kernel, dim3(grid(n * nb)), dim3(256), 0, st, dv, n * nb, flag);
        ALP_HIP(hipMemcpyAsync(&kind, flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        ALP_HIP(hipStreamSynchronize(st));
        packed = rz_can_pack(agg, n, nb) && !(kind & 2u);
        if (agg == AGG_MEAN && (kind & 1u)) pieces = false;        // a float-valued mean: pandas' order (rz_runs_kernel)
        if (packed) hipLaunchKernelGGL(rz_pack_kernel, dim3(grid(n)), dim3(256), 0, st, dv, n, nb, idx);
    }
    hipLaunchKernelGGL(rz_cell_kernel, dim3(grid(n)), dim3(256), 0, st, dx, dy, n, x_min, y_max, res, width, height, cell,
                       packed ? (unsigned *)nullptr : idx, fused ? tile_used : nullptr, tiles_x);
    if (fused) {
        const int tiles = tiles_x * tiles_y;
        hipLaunchKernelGGL(rz_tile_list_kernel, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, st, tile_used, tiles_x, tiles_y, tile_list, tile_count);
        hipLaunchKernelGGL(rz_fill_tiles_kernel, dim3((unsigned)std::min(tiles, cu * 8)), dim3(256), 0, st, ra, tile_list, tile_count, nb, width,
                           height, tiles_x);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && agg != AGG_MEDIAN) {
        size_t t = tmp;
        e = rocprim::radix_sort_pairs<RzPairSort>(sort_tmp, t, cell, cell_s, idx, idx_s, count, 0u, cell_bits, st);      // stable: a run keeps the rows' order
        if (e == hipSuccess) {
            RzPiece *first = (RzPiece *)key, *last = first + (size_t)nseg * nb;
            const unsigned gs = grid(nseg);
#define ALP_RZ_PIECES(A)                                                                                                                  \
    do {                                                                                                                                  \
        if (packed) hipLaunchKernelGGL((rz_pieces_kernel<A, true>), dim3(gs), dim3(256), 0, st, cell_s, idx_s, dv, n, nb, hw, ra, first, last); \
        else hipLaunchKernelGGL((rz_pieces_kernel<A, false>), dim3(gs), dim3(256), 0, st, cell_s, idx_s, dv, n, nb, hw, ra, first, last);       \
        hipLaunchKernelGGL((rz_join_kernel<A>), dim3(gs), dim3(256), 0, st, cell_s, n, nb, hw, ra, first, last);                          \
    } while (0)
            if (agg == AGG_MEAN && pieces) ALP_RZ_PIECES(AGG_MEAN);
            else if (agg == AGG_MEAN) hipLaunchKernelGGL((rz_runs_kernel<AGG_MEAN>), dim3(grid(n)), dim3(256), 0, st, cell_s, idx_s, dv, n, nb, hw, ra);
            else if (agg == AGG_MAX && pieces) ALP_RZ_PIECES(AGG_MAX);
            else if (agg == AGG_MAX) hipLaunchKernelGGL((rz_runs_kernel<AGG_MAX>), dim3(grid(n)), dim3(256), 0, st, cell_s, idx_s, dv, n, nb, hw, ra);
            else if (pieces) ALP_RZ_PIECES(AGG_MIN);
            else hipLaunchKernelGGL((rz_runs_kernel<AGG_MIN>), dim3(grid(n)), dim3(256), 0, st, cell_s, idx_s, dv, n, nb, hw, ra);
#undef ALP_RZ_PIECES
            e = hipGetLastError();
        }
    } else if (e == hipSuccess && packed) {
        size_t t = tmp;
        e = rocprim::radix_sort_pairs<RzPairSort>(sort_tmp, t, cell, cell_s, idx, idx_s, count, 0u, cell_bits, st);
        if (e == hipSuccess) {
            const unsigned chunks = (unsigned)std::min<long long>((n + 63) / 64, (long long)cu * 256);
            hipLaunchKernelGGL(rz_median_packed_kernel, dim3(chunks), dim3(64), 0, st, cell_s, idx_s, n, nb, hw, ra);
            e = hipGetLastError();
        }
    } else if (e == hipSuccess) {
        // (the flag words, one per band (nb <= 64), were parked at the head of the sorted-key buffer: the first sort writes that buffer
        // only now, after they have been read)
        for (int b = 0; b < nb && e == hipSuccess; ++b) {
            const unsigned kind = kinds[b];
            size_t t = tmp;
            if (!(kind & 4u)) {             // bytes: cell : value in cell_bits + 8 bits (the NaN keys, all ones, end up last)
                hipLaunchKernelGGL(rz_median_key_kernel<8>, dim3(grid(n)), dim3(256), 0, st, cell, dv, n, nb, b, key);
                e = rocprim::radix_sort_keys<RzPairSort>(sort_tmp, t, key, key_s, count, 0u, std::min(64u, cell_bits + 9u), st);
                if (e != hipSuccess) break;
                hipLaunchKernelGGL(rz_median_runs32_kernel<8>, dim3(grid(n)), dim3(256), 0, st, key_s, n, ra + b * hw);
            } else if (!(kind & 2u)) {      // integers below 2^16: cell : value in cell_bits + 16 bits
                hipLaunchKernelGGL(rz_median_key_kernel<16>, dim3(grid(n)), dim3(256), 0, st, cell, dv, n, nb, b, key);
                e = rocprim::radix_sort_keys<RzPairSort>(sort_tmp, t, key, key_s, count, 0u, std::min(64u, cell_bits + 17u), st);
                if (e != hipSuccess) break;
                hipLaunchKernelGGL(rz_median_runs32_kernel<16>, dim3(grid(n)), dim3(256), 0, st, key_s, n, ra + b * hw);
            } else if (!(kind & 1u)) {      // float32 values
                hipLaunchKernelGGL(rz_median_key_kernel<32>, dim3(grid(n)), dim3(256), 0, st, cell, dv, n, nb, b, key);
                e = rocprim::radix_sort_keys<RzPairSort>(sort_tmp, t, key, key_s, count, 0u, 64u, st);
                if (e != hipSuccess) break;
                hipLaunchKernelGGL(rz_median_runs32_kernel<32>, dim3(grid(n)), dim3(256), 0, st, key_s, n, ra + b * hw);
            } else {
                // values that are not float32: by value (64-bit keys), then stably by cell
                hipLaunchKernelGGL(rz_median_keys_kernel, dim3(grid(n)), dim3(256), 0, st, cell, dv, n, nb, b, key, idx, cell3);
                e = rocprim::radix_sort_pairs(sort_tmp, t, key, key_s, idx, idx_s, count, 0u, 64u, st);
                if (e != hipSuccess) break;
                hipLaunchKernelGGL(rz_gather_cell_kernel, dim3(grid(n)), dim3(256), 0, st, idx_s, cell3, n, cell_s);
                t = tmp;
                unsigned *cell_s2 = (unsigned *)key;           // the value keys are spent
                e = rocprim::radix_sort_pairs<RzPairSort>(sort_tmp, t, cell_s, cell_s2, idx_s, idx, count, 0u, 32u, st);
                if (e != hipSuccess) break;
                hipLaunchKernelGGL(rz_median_runs_kernel, dim3(grid(n)), dim3(256), 0, st, cell_s2, idx, dv, n, nb, b, ra + b * hw);
            }
            e = hipGetLastError();
        }
    }
    if (e != hipSuccess) return fail(ALP_EHIP, "rasterisation: %s", hipGetErrorString(e));
    if (fused) {
        if (agg == AGG_MEAN) launch_tail<AGG_MEAN>(ra, nb, width, height, sweeps, nodata, out_dev, tile_list, tile_count);
        else if (agg == AGG_MAX) launch_tail<AGG_MAX>(ra, nb, width, height, sweeps, nodata, out_dev, tile_list, tile_count);
        else if (agg == AGG_MIN) launch_tail<AGG_MIN>(ra, nb, width, height, sweeps, nodata, out_dev, tile_list, tile_count);
        else launch_tail<AGG_MEDIAN_FOCAL>(ra, nb, width, height, sweeps, nodata, out_dev, tile_list, tile_count);
        ALP_HIP(hipGetLastError());
        return ALP_OK;
    }
    float *cur = ra, *nxt = rb;
    for (int s = 0; s < sweeps; ++s) {
        if (agg == AGG_MEAN) hipLaunchKernelGGL((rz_focal_kernel<AGG_MEAN>), dim3(grid(total)), dim3(256), 0, st, cur, nxt, nb, width, height);
        else if (agg == AGG_MAX) hipLaunchKernelGGL((rz_focal_kernel<AGG_MAX>), dim3(grid(total)), dim3(256), 0, st, cur, nxt, nb, width, height);
        else if (agg == AGG_MIN) hipLaunchKernelGGL((rz_focal_kernel<AGG_MIN>), dim3(grid(total)), dim3(256), 0, st, cur, nxt, nb, width, height);
        else hipLaunchKernelGGL(rz_focal_median_kernel, dim3(grid(total)), dim3(256), 0, st, cur, nxt, nb, width, height);
        float *t = cur; cur = nxt; nxt = t;
    }
    if (f32_out) {                 // the float32 raster itself (project.py:479, before the byte conversion)
        *f32_out = cur;
        ALP_HIP(hipGetLastError());
        return ALP_OK;
    }
    hipLaunchKernelGGL(rz_to_u8_kernel, dim3(grid(total)), dim3(256), 0, st, cur, total, nodata, out_dev);
    ALP_HIP(hipGetLastError());
    return ALP_OK;
}

// ------------------------------------------------------------------ fed from the resident coordinate image
// reverse_proj + to_geotiff back to back (example.py:103-106) without the DataFrame in between: the points are the
// frame's pixels that see the surface, compacted in pixel order (the rows of the reference's table), their x / y the
// coordinate image's channels 0 / 2 plus the offsets (project.py:361, :370-373), their band values the caller's
// image array at the same pixel (project.py:364).

// values[i][b] = (double) array[pixel idx[i]][band_channel[b]]  (the float64 columns of the reference's table)
template <typename A>
__global__ __launch_bounds__(256) void rz_gather_bands_kernel(const A *__restrict__ array, const unsigned *__restrict__ idx, long long n,
                                                              int channels, int nb, const int *__restrict__ band_channel,
                                                              double *__restrict__ values) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const A *px = array + (long long)idx[i] * channels;
        for (int b = 0; b < nb; ++b) values[i * nb + b] = (double)px[band_channel[b]];
    }
}

// the same for a uint8 photograph and at most four bands: one packed word per point, straight into the sort's payload slot
__global__ __launch_bounds__(256) void rz_gather_packed_kernel(const unsigned char *__restrict__ array, const unsigned *__restrict__ idx,
                                                               long long n, int channels, int nb, const int *__restrict__ band_channel,
                                                               unsigned *__restrict__ packed) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    int ch[4];
    for (int b = 0; b < 4; ++b) ch[b] = b < nb ? band_channel[b] : 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned char *px = array + (long long)idx[i] * channels;
        unsigned w = 0;
        for (int b = 0; b < nb; ++b) w |= (unsigned)px[ch[b]] << (8 * b);
        packed[i] = w;
    }
}

}  // namespace alp

using namespace alp;

extern "C" int alp_render_rasterize_plan(alp_mesh_t *m, const double *offsets, int64_t *n_valid, double bounds[4]) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m && n_valid && bounds, "NULL argument");
    if (!m->rendered) return fail(ALP_ESTATE, "alp_render_rasterize_plan: nothing rendered yet");
    int64_t M = 0;
    if (int rc = frame_valid_count(m, &M)) return rc;
    m->valid_total = -1;                         // this count is not an alp_render_fetch_valid in waiting
    m->rz_n = -1;
    *n_valid = M;
    bounds[0] = bounds[1] = bounds[2] = bounds[3] = NAN;
    if (M == 0) { m->rz_n = 0; return ALP_OK; }
    const size_t need = (size_t)M * (8 + 8 + 8 + 4) + 64;      // x | y | z planes (the compaction writes all three), then the pixel index
    if (need > m->rz_cap) {
        if (m->rz_points) hipFree(m->rz_points);
        m->rz_points = nullptr;
        m->rz_cap = 0;
        ALP_HIP(hipMalloc((void **)&m->rz_points, need));
        m->rz_cap = need;
    }
    double *x = (double *)m->rz_points;
    unsigned *idx = (unsigned *)(x + 3 * M);
    // the compaction writes planes directly (x = channel 0 + offset, y = channel 2 + offset: project.py:361, :370-373): no
    // interleaved copy to split afterwards (0.08 ms for the 100 M-vertex frame's 11.7 M pixels)
    m->valid_total_planes = M;
    if (int rc = frame_valid_write(m, offsets, idx, x, true)) return rc;
    // x.min(), y.min(), x.max(), y.max() of the table (project.py:420-423): the count's pass over the image took the extent of
    // the channels along (frame_valid_count), and adding the offset keeps the order -- the minimum of the sums is the sum at
    // the minimum, bit for bit (a pass of its own over x and y: 0.08 ms and one more wait of the host)
    const double ox = offsets ? offsets[0] : 0.0, oy = offsets ? offsets[2] : 0.0;
    bounds[0] = (double)m->valid_span[0] + ox;
    bounds[1] = (double)m->valid_span[2] + oy;
    bounds[2] = (double)m->valid_span[1] + ox;
    bounds[3] = (double)m->valid_span[3] + oy;
    m->rz_n = M;
    return ALP_OK;
}

extern "C" int alp_render_rasterize(alp_mesh_t *m, const void *array, int array_dtype, int64_t channels,
                                    const int32_t *band_channel, int64_t nb, double x_min, double y_max, double resolution,
                                    int64_t width, int64_t height, int agg, int sweeps, int nodata, uint8_t *out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m && array && band_channel && out, "NULL argument");
    if (m->rz_n < 0) return fail(ALP_ESTATE, "alp_render_rasterize: call alp_render_rasterize_plan for this frame first");
    ALP_REQUIRE(m->rz_n >= 1, "no pixel of the frame sees the surface");
    ALP_REQUIRE(array_dtype == ALP_U8 || array_dtype == ALP_U16 || array_dtype == ALP_F32 || array_dtype == ALP_F64,
                "array_dtype must be ALP_U8, ALP_U16, ALP_F32 or ALP_F64");
    ALP_REQUIRE(channels >= 1 && channels <= 64 && nb >= 1 && nb <= 64, "channel or band count out of range");
    for (int64_t b = 0; b < nb; ++b) ALP_REQUIRE(band_channel[b] >= 0 && band_channel[b] < channels, "band_channel out of range");
    ALP_REQUIRE(width >= 1 && height >= 1 && width * height <= ((int64_t)1 << 31), "raster size out of range");
    ALP_REQUIRE(resolution > 0, "resolution must be positive");
    ALP_REQUIRE(agg == ALP_AGG_MEAN || agg == ALP_AGG_MAX || agg == ALP_AGG_MIN || agg == ALP_AGG_MEDIAN,
                "agg must be ALP_AGG_MEAN, _MAX, _MIN or _MEDIAN");
    ALP_REQUIRE(sweeps >= 0 && sweeps <= 4096, "sweeps out of range");
    const int64_t n = m->rz_n;
    const size_t esize = array_dtype == ALP_U8 ? 1 : array_dtype == ALP_U16 ? 2 : array_dtype == ALP_F32 ? 4 : 8;
    const size_t npix = (size_t)m->w * m->h, arr_bytes = npix * (size_t)channels * esize;
    const size_t total = (size_t)width * height * nb;
    // values | raster a | raster b | out (u8) | band table | the caller's array   (the sort buffers: the library scratch)
    const size_t bytes = (size_t)n * nb * 8 + total * (4 + 4 + 1) + 64 * 4 + 256 + arr_bytes + 64;
    if (bytes > m->rz_work_cap) {
        if (m->rz_work) hipFree(m->rz_work);
        m->rz_work = nullptr;
        m->rz_work_cap = 0;
        ALP_HIP(hipMalloc((void **)&m->rz_work, bytes));
        m->rz_work_cap = bytes;
    }
    char *dev = m->rz_work;
    double *dv = (double *)dev;
    float *ra = (float *)(dv + (size_t)n * nb), *rb = ra + total;
    unsigned char *out_dev = (unsigned char *)(rb + total);
    int *bands_dev = (int *)(((uintptr_t)(out_dev + total) + 15) & ~(uintptr_t)15);
    char *arr_dev = (char *)(((uintptr_t)(bands_dev + 64) + 255) & ~(uintptr_t)255);
    const double *dx = (const double *)m->rz_points, *dy = dx + n;
    const unsigned *idx = (const unsigned *)(dx + 3 * n);
    hipStream_t st = ctx().stream;
    char *sort_area = nullptr;
    if (int rc0 = scratch_reserve(rz_sort_bytes(n, nullptr), (void **)&sort_area)) return rc0;
    int rc = upload_chunked(arr_dev, array, arr_bytes);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(bands_dev, band_channel, (size_t)nb * sizeof(int), hipMemcpyHostToDevice, st);
    if (!rc && e == hipSuccess) {
        KTimeScope kt;
        const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, (long long)ctx().cu_count * 8);
#define ALP_GATHER(A) hipLaunchKernelGGL(rz_gather_bands_kernel<A>, dim3(grid), dim3(256), 0, st, (const A *)arr_dev, idx, (long long)n, \
                                         (int)channels, (int)nb, bands_dev, dv)
        const bool packed = array_dtype == ALP_U8 && rz_can_pack(agg, n, (int)nb);
        if (packed) hipLaunchKernelGGL(rz_gather_packed_kernel, dim3(grid), dim3(256), 0, st, (const unsigned char *)arr_dev, idx, (long long)n,
                                       (int)channels, (int)nb, bands_dev, rz_payload_slot(sort_area, n));
        else if (array_dtype == ALP_U8) ALP_GATHER(unsigned char);
        else if (array_dtype == ALP_U16) ALP_GATHER(unsigned short);
        else if (array_dtype == ALP_F32) ALP_GATHER(float);
        else ALP_GATHER(double);
#undef ALP_GATHER
        rc = run_rasterize(agg, dx, dy, dv, n, (int)nb, x_min, y_max, resolution, (int)width, (int)height, sweeps, nodata, ra, rb, out_dev,
                           sort_area, nullptr, packed);
    }
    if (!rc && e == hipSuccess) e = hipMemcpyAsync(out, out_dev, total, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    else hipStreamSynchronize(st);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ALP_EHIP, "alp_render_rasterize: %s", hipGetErrorString(e));
    return ALP_OK;
}

namespace alp {

// columns of a table (each contiguous, as a DataFrame keeps them) -> the interleaved values[i][b] the kernels read
__global__ __launch_bounds__(256) void rz_interleave_kernel(const double *__restrict__ planar, long long n, int nb,
                                                            double *__restrict__ values) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        for (int b = 0; b < nb; ++b) values[i * nb + b] = planar[(long long)b * n + i];
}

// values (n x nb, interleaved) or cols (nb pointers to n contiguous doubles): exactly one is given
static int rasterize_host_points(const char *who, const double *x, const double *y, const double *values,
                                 const double *const *cols, int64_t n, int64_t nb, double x_min, double y_max,
                                 double resolution, int64_t width, int64_t height, int agg, int sweeps, int nodata,
                                 uint8_t *out, float *out_f32 = nullptr) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(x && y && (values || cols) && (out || out_f32), "NULL argument");
    ALP_REQUIRE(n >= 1 && nb >= 1 && nb <= 64, "n or band count out of range");
    ALP_REQUIRE(width >= 1 && height >= 1 && width * height <= ((int64_t)1 << 31), "raster size out of range");
    ALP_REQUIRE(resolution > 0, "resolution must be positive");
    ALP_REQUIRE(agg == ALP_AGG_MEAN || agg == ALP_AGG_MAX || agg == ALP_AGG_MIN || agg == ALP_AGG_MEDIAN,
                "agg must be ALP_AGG_MEAN, _MAX, _MIN or _MEDIAN");
    ALP_REQUIRE(n < ((int64_t)1 << 31), "more than 2^31 points");
    ALP_REQUIRE(sweeps >= 0 && sweeps <= 4096, "sweeps out of range");
    if (cols)
        for (int64_t b = 0; b < nb; ++b) ALP_REQUIRE(cols[b], "a band column is NULL");
    const size_t total = (size_t)width * height * nb;
    const size_t pts_bytes = (size_t)n * sizeof(double);
    char *dev = nullptr;
    // x | y | values | raster a | raster b | out (u8) | sort buffers; the columns are staged in the sort buffers when they
    // fit (they are interleaved into `values` before the first sort), else behind them
    const size_t sort_bytes = rz_sort_bytes(n, nullptr);
    const bool stage_in_sort = cols && pts_bytes * nb <= sort_bytes;
    const size_t bytes = pts_bytes * (2 + nb) + total * (4 + 4 + 1) + 256 + sort_bytes + ((cols && !stage_in_sort) ? pts_bytes * nb + 64 : 0);
    ALP_HIP(hipMalloc((void **)&dev, bytes));
    double *dx = (double *)dev, *dy = dx + n, *dv = dy + n;
    float *ra = (float *)(dv + (size_t)n * nb), *rb = ra + total;
    unsigned char *out_dev = (unsigned char *)(rb + total);
    char *sort_area = (char *)(((uintptr_t)(out_dev + total) + 255) & ~(uintptr_t)255);
    double *planar = stage_in_sort ? (double *)sort_area : (double *)(((uintptr_t)(sort_area + sort_bytes) + 63) & ~(uintptr_t)63);
    hipStream_t st = ctx().stream;
    hipError_t e = hipMemcpyAsync(dx, x, pts_bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dy, y, pts_bytes, hipMemcpyHostToDevice, st);
    if (cols) {
        for (int64_t b = 0; b < nb && e == hipSuccess; ++b)
            e = hipMemcpyAsync(planar + (size_t)b * n, cols[b], pts_bytes, hipMemcpyHostToDevice, st);
    } else if (e == hipSuccess) {
        e = hipMemcpyAsync(dv, values, pts_bytes * nb, hipMemcpyHostToDevice, st);
    }
    int rc = ALP_OK;
    if (e == hipSuccess) {
        KTimeScope kt;
        if (cols) {
            const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, (long long)ctx().cu_count * 8);
            hipLaunchKernelGGL(rz_interleave_kernel, dim3(grid), dim3(256), 0, st, planar, (long long)n, (int)nb, dv);
        }
        float *f32_dev = nullptr;
        rc = run_rasterize(agg, dx, dy, dv, n, (int)nb, x_min, y_max, resolution, (int)width, (int)height, sweeps, nodata, ra, rb,
                           out_dev, sort_area, out_f32 ? &f32_dev : nullptr);
        if (rc == ALP_OK && out_f32) e = hipMemcpyAsync(out_f32, f32_dev, total * sizeof(float), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess && rc == ALP_OK && !out_f32) e = hipMemcpyAsync(out, out_dev, total, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(dev);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ALP_EHIP, "%s: %s", who, hipGetErrorString(e));
    return ALP_OK;
}

}  // namespace alp

extern "C" int alp_rasterize_points(const double *x, const double *y, const double *values, int64_t n, int64_t nb,
                                    double x_min, double y_max, double resolution, int64_t width, int64_t height,
                                    int agg, int sweeps, int nodata, uint8_t *out) {
    return alp::rasterize_host_points("alp_rasterize_points", x, y, values, nullptr, n, nb, x_min, y_max, resolution, width, height,
                                      agg, sweeps, nodata, out);
}

extern "C" int alp_rasterize_points_f32(const double *x, const double *y, const double *values, int64_t n, int64_t nb,
                                        double x_min, double y_max, double resolution, int64_t width, int64_t height,
                                        int agg, int sweeps, float *out) {
    return alp::rasterize_host_points("alp_rasterize_points_f32", x, y, values, nullptr, n, nb, x_min, y_max, resolution, width, height,
                                      agg, sweeps, 0, nullptr, out);
}

extern "C" int alp_rasterize_columns(const double *x, const double *y, const double *const *columns, int64_t n, int64_t nb,
                                     double x_min, double y_max, double resolution, int64_t width, int64_t height,
                                     int agg, int sweeps, int nodata, uint8_t *out) {
    return alp::rasterize_host_points("alp_rasterize_columns", x, y, nullptr, columns, n, nb, x_min, y_max, resolution, width, height,
                                      agg, sweeps, nodata, out);
}
