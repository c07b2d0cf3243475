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
// Kernels: scatter (float64 atomics per band: sum+count, or ordered-integer max/min),
// finalize (float32 raster), focal sweep (ping-pong), uint8 conversion.  The 3x3 mean adds its
// window in numpy's order (pairwise block of 8, then the ninth) so that the float64 sum -- and
// with it the float32 value and the truncated byte -- match the reference.
#include "alp_raster_internal.h"

#include <algorithm>
#include <cmath>

namespace alp {

enum { AGG_MEAN = 0, AGG_MAX = 1, AGG_MIN = 2 };

// ALP_RZ_SEPARATE_PASSES=1: finalize / sweep / conversion as separate kernels whatever the sweep count (the path for more than
// RZ_SMAX sweeps); the tests run both and compare bytes
static bool rz_separate_passes() {
    const char *e = getenv("ALP_RZ_SEPARATE_PASSES");
    return e && e[0] == '1';
}

// order-preserving map double -> uint64 (so that integer atomicMax/Min order like the doubles)
__device__ __forceinline__ unsigned long long d2ord(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord2d(unsigned long long o) {
    const unsigned long long u = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o;
    return __longlong_as_double((long long)u);
}

// Points arrive in the pixel order of the camera image, so neighbouring lanes of a wave mostly fall into the same
// raster cell (near the camera hundreds of pixels share a 1 m cell): every RUN of equal cells inside a wave is reduced
// with a segmented shuffle reduction first and only its first lane goes to memory -- one float64 atomic and one count
// atomic per run and band instead of per point (the per-point version spent 2.7 ... 8 ms on 11.7 M points x 3 bands,
// serialised on the hot cells).  Sums of a cell are formed in another order than point by point; the aggregates the
// reference forms are order-free for max / min and, for the byte-valued channels the path carries, exact for mean.
template <int AGG>
__global__ __launch_bounds__(256) void rz_scatter_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ values, long long n, int nb,
                                                         double x_min, double y_max, double res, int width,
                                                         int height, double *__restrict__ acc,
                                                         unsigned *__restrict__ cnt) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long hw = (long long)width * height;
    const int lane = (int)(threadIdx.x & 63);
    const long long rounds = (n + stride - 1) / stride;                 // every lane makes every round: the shuffles are wave-wide
    for (long long k = 0; k < rounds; ++k) {
        const long long i = k * stride + (long long)blockIdx.x * blockDim.x + threadIdx.x;
        const bool in = i < n;
        long long cell = -1 - lane;                                      // lanes past the end: runs of their own, never stored
        if (in) {
            long long col = (long long)((x[i] - x_min) / res);
            long long row = (long long)((y_max - y[i]) / res);
            col = col < 0 ? 0 : (col > width - 1 ? width - 1 : col);
            row = row < 0 ? 0 : (row > height - 1 ? height - 1 : row);
            cell = row * width + col;
        }
        const long long prev = __shfl_up(cell, 1);
        const bool head = lane == 0 || prev != cell;
        const unsigned long long heads = __ballot(head);
        const int run = __popcll(heads & (~0ull >> (63 - lane)));        // run number of this lane (1-based, monotone)
        int same[6];                                                     // does lane + 2^j belong to this lane's run?
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int r2 = __shfl_down(run, 1 << j);
            same[j] = (lane + (1 << j) < 64) && r2 == run;
        }
        for (int b = 0; b < nb; ++b) {
            const double val = in ? values[i * nb + b] : __longlong_as_double(0x7ff8000000000000ll);
            const bool ok = val == val;                                  // pandas skips NaN
            unsigned c = ok ? 1u : 0u;
            if constexpr (AGG == AGG_MEAN) {
                double sum = ok ? val : 0.0;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const double s2 = __shfl_down(sum, 1 << j);
                    const unsigned c2 = __shfl_down(c, 1 << j);
                    if (same[j]) { sum += s2; c += c2; }
                }
                if (head && c) {
                    atomicAdd(&acc[b * hw + cell], sum);
                    atomicAdd(&cnt[b * hw + cell], c);
                }
            } else {
                unsigned long long key = ok ? d2ord(val) : (AGG == AGG_MAX ? 0ull : ~0ull);      // the identities of max / min
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const unsigned long long k2 = __shfl_down(key, 1 << j);
                    const unsigned c2 = __shfl_down(c, 1 << j);
                    if (same[j]) {
                        key = (AGG == AGG_MAX) ? (k2 > key ? k2 : key) : (k2 < key ? k2 : key);
                        c += c2;
                    }
                }
                if (head && c) {
                    if constexpr (AGG == AGG_MAX) atomicMax(reinterpret_cast<unsigned long long *>(&acc[b * hw + cell]), key);
                    else atomicMin(reinterpret_cast<unsigned long long *>(&acc[b * hw + cell]), key);
                    atomicAdd(&cnt[b * hw + cell], c);
                }
            }
        }
    }
}

template <int AGG>
__global__ __launch_bounds__(256) void rz_finalize_kernel(const double *__restrict__ acc, const unsigned *__restrict__ cnt,
                                                          long long total, float *__restrict__ raster) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const unsigned c = cnt[i];
        float r = __int_as_float(0x7fc00000);                 // NaN
        if (c) {
            if constexpr (AGG == AGG_MEAN) r = (float)(acc[i] / (double)c);
            else r = (float)ord2d(reinterpret_cast<const unsigned long long *>(acc)[i]);
        }
        raster[i] = r;
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
// finalize + up to RZ_SMAX focal sweeps + uint8 in ONE pass over the raster: a workgroup owns a tile of RZ_TW x RZ_TH cells,
// forms the float32 raster of the tile and a halo of S cells in LDS (a cell S sweeps later depends on the cells within S of
// it, nothing else), sweeps there -- each sweep is valid on a region one cell smaller all round -- and writes bytes only.
// Every value is formed by the expressions of the separate kernels above (which stay as the path for more sweeps), so the
// bytes are the same; the raster no longer crosses HBM as float32 three times (finalize, sweep, conversion: 2.24 ms of the
// 3.79 ms of the 3 x 8088 x 9786 raster of bench.py's f2 leg), and a tile whose cells and halo are all empty -- most of a
// georectified photograph's bounding box -- skips its sweeps.
constexpr int RZ_TW = 64, RZ_TH = 32, RZ_SMAX = 8;
enum { AGG_MEDIAN_FOCAL = 3 };

template <int AGG>
__device__ __forceinline__ float rz_window_value(const float *__restrict__ s, int lw, int at) {
    const float nan = __int_as_float(0x7fc00000);
    if constexpr (AGG == AGG_MEDIAN_FOCAL) {
        float w[9];
        int have = 0;
#pragma unroll
        for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc) {
                const float val = s[at + dr * lw + dc];
                if (val != val) continue;
                int k = have++;                                   // insertion sort of at most 9 values
                while (k > 0 && w[k - 1] > val) { w[k] = w[k - 1]; --k; }
                w[k] = val;
            }
        if (!have) return nan;
        return (have & 1) ? w[have / 2] : (float)(((double)w[have / 2 - 1] + (double)w[have / 2]) / 2);
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

// AGG_MEAN / _MAX / _MIN read the scatter's accumulators, AGG_MEDIAN_FOCAL the float32 raster the median runs wrote
template <int AGG>
__global__ __launch_bounds__(256) void rz_tail_kernel(const double *__restrict__ acc, const unsigned *__restrict__ cnt,
                                                      const float *__restrict__ raster, int width, int height, int S,
                                                      int nodata, int tiles_x, int tiles_y, unsigned char *__restrict__ out) {
    extern __shared__ float rz_tail_lds[];                   // two rasters of (RZ_TH + 2 S) x (RZ_TW + 2 S) floats: 18 KB at S = 1, 31 KB at S = 8
    __shared__ int s_any;
    const float nan = __int_as_float(0x7fc00000);
    const int tid = (int)threadIdx.x;
    const long long hw = (long long)width * height;
    int t = (int)blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const long long band_base = (long long)(t / tiles_y) * hw;
    const int x0 = tx * RZ_TW - S, y0 = ty * RZ_TH - S;        // raster position of LDS cell (0, 0)
    const int lw = RZ_TW + 2 * S, lh = RZ_TH + 2 * S;
    float *buf_cur = rz_tail_lds, *buf_nxt = rz_tail_lds + lw * lh;
    const float inv_lw = 1.0f / (float)lw;                       // idx / lw through (idx + 0.5) * (1 / lw): idx < 3840, exact
    if (tid == 0) s_any = 0;
    __syncthreads();
    bool any = false;
    if constexpr (AGG == AGG_MEDIAN_FOCAL) {
        for (int idx = tid; idx < lw * lh; idx += 256) {
            const int r = (int)(((float)idx + 0.5f) * inv_lw), c = idx - r * lw;
            const int gr = y0 + r, gc = x0 + c;
            float v = nan;                                       // outside the raster: NaN, in every sweep
            if (gr >= 0 && gr < height && gc >= 0 && gc < width) v = raster[band_base + (long long)gr * width + gc];
            buf_cur[idx] = v;
            any |= (v == v);
        }
    } else {
        // two passes with the trip count fixed, so that all count loads of a lane are in flight together and the accumulator
        // loads (occupied cells only) after them, instead of one count -> accumulator dependency per turn of a rolled loop
        constexpr int TURNS = ((RZ_TW + 2 * RZ_SMAX) * (RZ_TH + 2 * RZ_SMAX) + 255) / 256;
        unsigned have[TURNS];
        long long at[TURNS];
        const int cells = lw * lh;
#pragma unroll
        for (int k = 0; k < TURNS; ++k) {
            const int idx = tid + k * 256;
            have[k] = 0;
            at[k] = 0;
            if (idx < cells) {
                const int r = (int)(((float)idx + 0.5f) * inv_lw), c = idx - r * lw;
                const int gr = y0 + r, gc = x0 + c;
                if (gr >= 0 && gr < height && gc >= 0 && gc < width) {       // outside the raster: NaN, in every sweep
                    at[k] = band_base + (long long)gr * width + gc;
                    have[k] = cnt[at[k]];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < TURNS; ++k) {
            const int idx = tid + k * 256;
            if (idx < cells) {
                float v = nan;
                if (have[k]) {
                    if constexpr (AGG == AGG_MEAN) v = (float)(acc[at[k]] / (double)have[k]);
                    else v = (float)ord2d(reinterpret_cast<const unsigned long long *>(acc)[at[k]]);
                    any = true;
                }
                buf_cur[idx] = v;
            }
        }
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

template <int AGG>
static void launch_tail(const double *acc, const unsigned *cnt, const float *raster, int nb, int width, int height, int sweeps,
                        int nodata, unsigned char *out_dev) {
    const int tiles_x = (width + RZ_TW - 1) / RZ_TW, tiles_y = (height + RZ_TH - 1) / RZ_TH;
    const size_t lds = 2 * sizeof(float) * (size_t)(RZ_TW + 2 * sweeps) * (size_t)(RZ_TH + 2 * sweeps);
    hipLaunchKernelGGL((rz_tail_kernel<AGG>), dim3((unsigned)((long long)tiles_x * tiles_y * nb)), dim3(256), lds, ctx().stream, acc,
                       cnt, raster, width, height, sweeps, nodata, tiles_x, tiles_y, out_dev);
}

template <int AGG>
static int run_rasterize(const double *dx, const double *dy, const double *dv, long long n, int nb, double x_min,
                         double y_max, double res, int width, int height, int sweeps, int nodata, double *acc,
                         unsigned *cnt, float *ra, float *rb, unsigned char *out_dev) {
    hipStream_t st = ctx().stream;
    const long long total = (long long)width * height * nb;
    const int cu = ctx().cu_count;
    auto grid = [&](long long items) {
        const long long want = (items + 255) / 256;
        return (unsigned)(want < 1 ? 1 : (want < (long long)cu * 8 ? want : (long long)cu * 8));
    };
    if constexpr (AGG == AGG_MEAN) {
        ALP_HIP(hipMemsetAsync(acc, 0, (size_t)total * sizeof(double), st));
    } else {
        // identity of max over the ordered keys is 0, of min all ones
        ALP_HIP(hipMemsetAsync(acc, AGG == AGG_MAX ? 0x00 : 0xff, (size_t)total * sizeof(double), st));
    }
    ALP_HIP(hipMemsetAsync(cnt, 0, (size_t)total * sizeof(unsigned), st));
    hipLaunchKernelGGL((rz_scatter_kernel<AGG>), dim3(grid(n)), dim3(256), 0, st, dx, dy, dv, n, nb, x_min, y_max, res,
                       width, height, acc, cnt);
    if (sweeps <= RZ_SMAX && !rz_separate_passes()) {
        launch_tail<AGG>(acc, cnt, nullptr, nb, width, height, sweeps, nodata, out_dev);
        ALP_HIP(hipGetLastError());
        return ALP_OK;
    }
    hipLaunchKernelGGL((rz_finalize_kernel<AGG>), dim3(grid(total)), dim3(256), 0, st, acc, cnt, total, ra);
    float *cur = ra, *nxt = rb;
    for (int s = 0; s < sweeps; ++s) {
        hipLaunchKernelGGL((rz_focal_kernel<AGG>), dim3(grid(total)), dim3(256), 0, st, cur, nxt, nb, width, height);
        float *t = cur; cur = nxt; nxt = t;
    }
    hipLaunchKernelGGL(rz_to_u8_kernel, dim3(grid(total)), dim3(256), 0, st, cur, total, nodata, out_dev);
    ALP_HIP(hipGetLastError());
    return ALP_OK;
}

// ------------------------------------------------------------------ median
// groupby median needs the values of every pixel in order: the points of one band are sorted
// by value, then stably by pixel (two rocPRIM radix sorts: a library sort, nothing to hand-tune),
// and the middle element(s) of each pixel's run are averaged like numpy/pandas do.
__global__ __launch_bounds__(256) void rz_median_keys_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                             const double *__restrict__ values, long long n, int nb,
                                                             int band, double x_min, double y_max, double res,
                                                             int width, int height,
                                                             unsigned long long *__restrict__ vkey,
                                                             unsigned *__restrict__ idx, unsigned *__restrict__ cell) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long col = (long long)((x[i] - x_min) / res);
        long long row = (long long)((y_max - y[i]) / res);
        col = col < 0 ? 0 : (col > width - 1 ? width - 1 : col);
        row = row < 0 ? 0 : (row > height - 1 ? height - 1 : row);
        const double val = values[i * nb + band];
        vkey[i] = d2ord(val);
        idx[i] = (unsigned)i;
        cell[i] = (val != val) ? 0xFFFFFFFFu : (unsigned)(row * width + col);      // NaN: sorts behind every pixel
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
        long long j = i + 1;
        while (j < n && cell_sorted[j] == c) ++j;
        const long long k = j - i;
        const double a = values[(long long)idx_sorted[i + (k - 1) / 2] * nb + band];
        const double b = values[(long long)idx_sorted[i + k / 2] * nb + band];
        raster_band[c] = (float)((k & 1) ? a : (a + b) / 2);
    }
}

__global__ __launch_bounds__(256) void rz_fill_nan_kernel(float *__restrict__ p, long long total) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
        p[i] = __int_as_float(0x7fc00000);
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

static int run_rasterize_median(const double *dx, const double *dy, const double *dv, long long n, int nb, double x_min,
                                double y_max, double res, int width, int height, int sweeps, int nodata, float *ra,
                                float *rb, unsigned char *out_dev) {
    hipStream_t st = ctx().stream;
    const long long hw = (long long)width * height, total = hw * nb;
    const int cu = ctx().cu_count;
    auto grid = [&](long long items) {
        const long long want = (items + 255) / 256;
        return (unsigned)(want < 1 ? 1 : (want < (long long)cu * 8 ? want : (long long)cu * 8));
    };
    // scratch: value keys (2 x u64), point ids (2 x u32), pixel ids (3 x u32), rocPRIM temporary storage
    size_t tmp1 = 0, tmp2 = 0;
    const size_t count = (size_t)n;
    rocprim::radix_sort_pairs(nullptr, tmp1, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                              (unsigned *)nullptr, (unsigned *)nullptr, count, 0u, 64u, st);
    rocprim::radix_sort_pairs(nullptr, tmp2, (unsigned *)nullptr, (unsigned *)nullptr, (unsigned *)nullptr,
                              (unsigned *)nullptr, count, 0u, 32u, st);
    const size_t tmp = tmp1 > tmp2 ? tmp1 : tmp2;
    char *scratch = nullptr;
    ALP_HIP(hipMalloc((void **)&scratch, (size_t)n * (16 + 8 + 12) + tmp + 256));
    unsigned long long *vkey = (unsigned long long *)scratch, *vkey2 = vkey + n;
    unsigned *idx = (unsigned *)(vkey2 + n), *idx2 = idx + n, *cell = idx2 + n, *cell_s = cell + n, *cell_s2 = cell_s + n;
    void *sort_tmp = (void *)(((uintptr_t)(cell_s2 + n) + 255) & ~(uintptr_t)255);
    hipLaunchKernelGGL(rz_fill_nan_kernel, dim3(grid(total)), dim3(256), 0, st, ra, total);
    hipError_t e = hipSuccess;
    for (int b = 0; b < nb && e == hipSuccess; ++b) {
        hipLaunchKernelGGL(rz_median_keys_kernel, dim3(grid(n)), dim3(256), 0, st, dx, dy, dv, n, nb, b, x_min, y_max, res,
                           width, height, vkey, idx, cell);
        size_t t = tmp;
        e = rocprim::radix_sort_pairs(sort_tmp, t, vkey, vkey2, idx, idx2, count, 0u, 64u, st);   // by value
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(rz_gather_cell_kernel, dim3(grid(n)), dim3(256), 0, st, idx2, cell, n, cell_s);
        t = tmp;
        e = rocprim::radix_sort_pairs(sort_tmp, t, cell_s, cell_s2, idx2, idx, count, 0u, 32u, st);   // stably by pixel
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(rz_median_runs_kernel, dim3(grid(n)), dim3(256), 0, st, cell_s2, idx, dv, n, nb, b, ra + b * hw);
    }
    if (e == hipSuccess && sweeps <= RZ_SMAX && !rz_separate_passes()) {
        launch_tail<AGG_MEDIAN_FOCAL>(nullptr, nullptr, ra, nb, width, height, sweeps, nodata, out_dev);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);          // scratch is freed below
        hipFree(scratch);
        if (e != hipSuccess) return fail(ALP_EHIP, "median rasterisation: %s", hipGetErrorString(e));
        return ALP_OK;
    }
    float *cur = ra, *nxt = rb;
    for (int s = 0; s < sweeps && e == hipSuccess; ++s) {
        hipLaunchKernelGGL(rz_focal_median_kernel, dim3(grid(total)), dim3(256), 0, st, cur, nxt, nb, width, height);
        float *t = cur; cur = nxt; nxt = t;
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rz_to_u8_kernel, dim3(grid(total)), dim3(256), 0, st, cur, total, nodata, out_dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);          // scratch is freed below
    hipFree(scratch);
    if (e != hipSuccess) return fail(ALP_EHIP, "median rasterisation: %s", hipGetErrorString(e));
    return ALP_OK;
}

// ------------------------------------------------------------------ fed from the resident coordinate image
// reverse_proj + to_geotiff back to back (example.py:103-106) without the DataFrame in between: the points are the
// frame's pixels that see the surface, compacted in pixel order (the rows of the reference's table), their x / y the
// coordinate image's channels 0 / 2 plus the offsets (project.py:361, :370-373), their band values the caller's
// image array at the same pixel (project.py:364).

// interleaved (x, y, z) of the compaction -> planar x[M], y[M]
__global__ __launch_bounds__(256) void rz_split_xy_kernel(const double *__restrict__ xyz, long long n, double *__restrict__ x,
                                                          double *__restrict__ y) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        x[i] = xyz[3 * i];
        y[i] = xyz[3 * i + 1];
    }
}

// min / max of x and y over the M points: ordered-integer atomics on four words (x_min, y_min, x_max, y_max)
__global__ __launch_bounds__(256) void rz_bounds_kernel(const double *__restrict__ x, const double *__restrict__ y, long long n,
                                                        unsigned long long *__restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned long long lo_x = ~0ull, lo_y = ~0ull, hi_x = 0ull, hi_y = 0ull;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long a = d2ord(x[i]), b = d2ord(y[i]);
        lo_x = a < lo_x ? a : lo_x; hi_x = a > hi_x ? a : hi_x;
        lo_y = b < lo_y ? b : lo_y; hi_y = b > hi_y ? b : hi_y;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long a = __shfl_xor(lo_x, off), b = __shfl_xor(lo_y, off), c = __shfl_xor(hi_x, off), d = __shfl_xor(hi_y, off);
        lo_x = a < lo_x ? a : lo_x; lo_y = b < lo_y ? b : lo_y;
        hi_x = c > hi_x ? c : hi_x; hi_y = d > hi_y ? d : hi_y;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(out + 0, lo_x); atomicMin(out + 1, lo_y);
        atomicMax(out + 2, hi_x); atomicMax(out + 3, hi_y);
    }
}

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
    const size_t need = (size_t)M * (8 + 8 + 4) + 64;
    if (need > m->rz_cap) {
        if (m->rz_points) hipFree(m->rz_points);
        m->rz_points = nullptr;
        m->rz_cap = 0;
        ALP_HIP(hipMalloc((void **)&m->rz_points, need));
        m->rz_cap = need;
    }
    double *x = (double *)m->rz_points, *y = x + M;
    unsigned *idx = (unsigned *)(y + M);
    char *dev = nullptr;
    if (int rc = scratch_reserve((size_t)M * 3 * sizeof(double) + 64, (void **)&dev)) return rc;
    double *xyz = (double *)dev;
    unsigned long long *mm = (unsigned long long *)(dev + (size_t)M * 3 * sizeof(double));
    hipStream_t st = ctx().stream;
    if (int rc = frame_valid_write(m, offsets, idx, xyz, false)) return rc;
    const unsigned long long init[4] = {~0ull, ~0ull, 0ull, 0ull};
    ALP_HIP(hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)std::min<long long>((M + 255) / 256, (long long)ctx().cu_count * 8);
    ktime_begin();
    hipLaunchKernelGGL(rz_split_xy_kernel, dim3(grid), dim3(256), 0, st, xyz, (long long)M, x, y);
    hipLaunchKernelGGL(rz_bounds_kernel, dim3(grid), dim3(256), 0, st, x, y, (long long)M, mm);
    ktime_end();
    ALP_HIP(hipGetLastError());
    unsigned long long h[4];
    ALP_HIP(hipMemcpyAsync(h, mm, sizeof(h), hipMemcpyDeviceToHost, st));
    ALP_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < 4; ++k) {
        const unsigned long long u = (h[k] >> 63) ? (h[k] & 0x7fffffffffffffffull) : ~h[k];     // ord2d on the host
        memcpy(&bounds[k], &u, 8);
    }
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
    // values | acc (f64) | cnt (u32) | raster a | raster b | out (u8) | band table | the caller's array
    const size_t bytes = (size_t)n * nb * 8 + total * (8 + 4 + 4 + 4 + 1) + 64 * 4 + 256 + arr_bytes + 64;
    if (bytes > m->rz_work_cap) {
        if (m->rz_work) hipFree(m->rz_work);
        m->rz_work = nullptr;
        m->rz_work_cap = 0;
        ALP_HIP(hipMalloc((void **)&m->rz_work, bytes));
        m->rz_work_cap = bytes;
    }
    char *dev = m->rz_work;
    double *dv = (double *)dev;
    double *acc = dv + (size_t)n * nb;
    unsigned *cnt = (unsigned *)(acc + total);
    float *ra = (float *)(cnt + total), *rb = ra + total;
    unsigned char *out_dev = (unsigned char *)(rb + total);
    int *bands_dev = (int *)(((uintptr_t)(out_dev + total) + 15) & ~(uintptr_t)15);
    char *arr_dev = (char *)(((uintptr_t)(bands_dev + 64) + 255) & ~(uintptr_t)255);
    const double *dx = (const double *)m->rz_points, *dy = dx + n;
    const unsigned *idx = (const unsigned *)(dy + n);
    hipStream_t st = ctx().stream;
    int rc = upload_chunked(arr_dev, array, arr_bytes);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(bands_dev, band_channel, (size_t)nb * sizeof(int), hipMemcpyHostToDevice, st);
    if (!rc && e == hipSuccess) {
        KTimeScope kt;
        const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, (long long)ctx().cu_count * 8);
#define ALP_GATHER(A) hipLaunchKernelGGL(rz_gather_bands_kernel<A>, dim3(grid), dim3(256), 0, st, (const A *)arr_dev, idx, (long long)n, \
                                         (int)channels, (int)nb, bands_dev, dv)
        if (array_dtype == ALP_U8) ALP_GATHER(unsigned char);
        else if (array_dtype == ALP_U16) ALP_GATHER(unsigned short);
        else if (array_dtype == ALP_F32) ALP_GATHER(float);
        else ALP_GATHER(double);
#undef ALP_GATHER
        if (agg == ALP_AGG_MEAN)
            rc = run_rasterize<AGG_MEAN>(dx, dy, dv, n, (int)nb, x_min, y_max, resolution, (int)width, (int)height, sweeps, nodata, acc, cnt, ra, rb, out_dev);
        else if (agg == ALP_AGG_MAX)
            rc = run_rasterize<AGG_MAX>(dx, dy, dv, n, (int)nb, x_min, y_max, resolution, (int)width, (int)height, sweeps, nodata, acc, cnt, ra, rb, out_dev);
        else if (agg == ALP_AGG_MIN)
            rc = run_rasterize<AGG_MIN>(dx, dy, dv, n, (int)nb, x_min, y_max, resolution, (int)width, (int)height, sweeps, nodata, acc, cnt, ra, rb, out_dev);
        else
            rc = run_rasterize_median(dx, dy, dv, n, (int)nb, x_min, y_max, resolution, (int)width, (int)height, sweeps, nodata, ra, rb, out_dev);
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
                                 uint8_t *out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(x && y && (values || cols) && out, "NULL argument");
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
    // x | y | values | acc (f64) | cnt (u32) | raster a | raster b | out (u8); the columns are staged in acc | cnt
    // (12 bytes per band-cell, cleared afterwards) when they fit, else behind out
    const bool stage_in_acc = cols && pts_bytes * nb <= total * 12;
    const size_t bytes = pts_bytes * (2 + nb) + total * (8 + 4 + 4 + 4 + 1) + 64 + ((cols && !stage_in_acc) ? pts_bytes * nb + 64 : 0);
    ALP_HIP(hipMalloc((void **)&dev, bytes));
    double *dx = (double *)dev, *dy = dx + n, *dv = dy + n;
    double *acc = dv + (size_t)n * nb;
    unsigned *cnt = (unsigned *)(acc + total);
    float *ra = (float *)(cnt + total), *rb = ra + total;
    unsigned char *out_dev = (unsigned char *)(rb + total);
    double *planar = stage_in_acc ? acc : (double *)(((uintptr_t)(out_dev + total) + 63) & ~(uintptr_t)63);
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
        if (agg == ALP_AGG_MEAN)
            rc = run_rasterize<AGG_MEAN>(dx, dy, dv, n, (int)nb, x_min, y_max, resolution, (int)width, (int)height, sweeps,
                                         nodata, acc, cnt, ra, rb, out_dev);
        else if (agg == ALP_AGG_MAX)
            rc = run_rasterize<AGG_MAX>(dx, dy, dv, n, (int)nb, x_min, y_max, resolution, (int)width, (int)height, sweeps,
                                        nodata, acc, cnt, ra, rb, out_dev);
        else if (agg == ALP_AGG_MIN)
            rc = run_rasterize<AGG_MIN>(dx, dy, dv, n, (int)nb, x_min, y_max, resolution, (int)width, (int)height, sweeps,
                                        nodata, acc, cnt, ra, rb, out_dev);
        else
            rc = run_rasterize_median(dx, dy, dv, n, (int)nb, x_min, y_max, resolution, (int)width, (int)height, sweeps,
                                      nodata, ra, rb, out_dev);
    }
    if (e == hipSuccess && rc == ALP_OK) e = hipMemcpyAsync(out, out_dev, total, hipMemcpyDeviceToHost, st);
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

extern "C" int alp_rasterize_columns(const double *x, const double *y, const double *const *columns, int64_t n, int64_t nb,
                                     double x_min, double y_max, double resolution, int64_t width, int64_t height,
                                     int agg, int sweeps, int nodata, uint8_t *out) {
    return alp::rasterize_host_points("alp_rasterize_columns", x, y, nullptr, columns, n, nb, x_min, y_max, resolution, width, height,
                                      agg, sweeps, nodata, out);
}
