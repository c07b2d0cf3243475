// Part of alp_rasterize.hip (one translation unit, included inside namespace alp in the order given there; not a
// stand-alone header): from the float32 raster of aggregates to bytes -- the focal sweeps and the byte conversion as separate passes, and
// the fused tail over the tiles that hold points (tile list, NaN fill, sweeps in LDS).
#pragma once

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

// one sweep of the median's NaN-only 3x3 focal fill (the separate-pass form)
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
