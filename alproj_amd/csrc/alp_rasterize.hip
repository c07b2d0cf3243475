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
// Kernels (round 4: sort-based, no atomics, no accumulators; rasterize_runs.h, rasterize_median.h, rasterize_tail.h): the
// cell of every point, ONE stable radix sort of (cell, point) shared by all bands, then per run of equal cells pandas' aggregate:
//   * band values of any kind, mean: a thread per run walks its points IN THEIR ORIGINAL ORDER with the Kahan-compensated
//     float64 sum of libgroupby's group_mean -- the float64 value, its float32 cast and the truncated byte are the reference's
//     (the atomics of rounds 2-3 added in arrival order: exact for integers below 2^53, one ulp off for general floats);
//   * max / min, and the mean of integer-valued bands: order-free pieces of 16 sorted positions + a join;
//   * byte-valued bands (at most four, no NaN -- a uint8 photograph, the reference's use): the values ride the sort as its
//     32-bit payload instead of the point index, nothing is gathered, and the median selects from the same one sort;
//   * any other median: one sort of a composite 64-bit key (cell : value) per band.
// Then the fused tail (float32 raster -> focal sweeps in LDS -> bytes) over the tiles that hold points or, for more than
// RZ_SMAX sweeps, separate sweep / conversion kernels.  The 3x3 mean adds its window in numpy's order (pairwise block of 8,
// then the ninth).
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

// The kernels, in dependency order (one translation unit):
#include "rasterize_runs.h"
#include "rasterize_tail.h"
#include "rasterize_median.h"

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

// What the band values are, as far as the choice of kernels needs to know (one kernel over all of them, one wait of the host).
// median: kinds[band] of rz_median_check_kernel; the other aggregates: kind of rz_integer_check_kernel, or 3 ("anything")
// where nothing depends on it.  `values` interleaved (n x nb) or, planar, a table's columns as they lie; flags_dev: 64 words.
struct RzLook {
    unsigned kind = 3;
    unsigned kinds[64];
    bool bytes(int agg, int nb) const {          // every value an integer in [0, 255], none NaN
        if (agg != AGG_MEDIAN) return !(kind & 2u);
        for (int b = 0; b < nb; ++b)
            if (kinds[b] & (4u | 8u)) return false;
        return true;
    }
};
static int rz_look_at(int agg, const double *values, long long n, int nb, bool planar, unsigned *flags_dev, RzLook *look) {
    hipStream_t st = ctx().stream;
    const long long count = n * nb;
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((count + 255) / 256, (long long)ctx().cu_count * 8));
    if (agg == AGG_MEDIAN) {
        ALP_HIP(hipMemsetAsync(flags_dev, 0, (size_t)nb * sizeof(unsigned), st));
        hipLaunchKernelGGL(rz_median_check_kernel, dim3(grid), dim3(256), 0, st, values, count, nb, planar ? n : 0ll, flags_dev);
        ALP_HIP(hipMemcpyAsync(look->kinds, flags_dev, (size_t)nb * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        ALP_HIP(hipStreamSynchronize(st));
    } else if (rz_pieces_fit(agg, n, nb) && (agg == AGG_MEAN || nb <= 4)) {
        ALP_HIP(hipMemsetAsync(flags_dev, 0, sizeof(unsigned), st));
        hipLaunchKernelGGL(rz_integer_check_kernel, dim3(grid), dim3(256), 0, st, values, count, rz_integer_limit(n), flags_dev);
        ALP_HIP(hipMemcpyAsync(&look->kind, flags_dev, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        ALP_HIP(hipStreamSynchronize(st));
    }
    return ALP_OK;
}

static int run_rasterize(int agg, const double *dx, const double *dy, const double *dv, long long n, int nb, double x_min,
                         double y_max, double res, int width, int height, int sweeps, int nodata, float *ra, float *rb,
                         unsigned char *out_dev, char *sort_area, float **f32_out = nullptr, bool packed_ready = false,
                         const RzLook *look = nullptr) {
    // packed_ready: the bands are bytes (nb <= 4) and the caller has already put their packed words into the sort's payload slot
    // (rz_payload_slot); dv is not read then.  look: what the caller already knows about the values (else found out here)
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
    RzLook mine;
    if (!packed_ready) {
        if (!look) {                // (the flag words sit at the head of the sorted-key buffer: no sort has run yet)
            if (int rc = rz_look_at(agg, dv, n, nb, false, (unsigned *)key_s, &mine)) return rc;
            look = &mine;
        }
        packed = rz_can_pack(agg, n, nb) && look->bytes(agg, nb);
        if (agg == AGG_MEAN && (look->kind & 1u)) pieces = false;        // a float-valued mean: pandas' order (rz_runs_kernel)
        if (packed) hipLaunchKernelGGL(rz_pack_kernel, dim3(grid(n)), dim3(256), 0, st, dv, n, nb, (long long)nb, 1ll, idx);
    }
    const unsigned *kinds = look ? look->kinds : nullptr;              // (read by the median's per-band sorts only: never NULL there)
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
            const unsigned gs = (unsigned)((nseg + 255) / 256);          // a thread per segment (at most 2^31 / 16 / 256 workgroups); the dispatcher evens them out
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
            const unsigned turns = (unsigned)std::min<long long>((n + RZ_MED_TURN - 1) / RZ_MED_TURN, (long long)cu * 256);
            hipLaunchKernelGGL(rz_median_packed_kernel<RZ_MED_GROUPS>, dim3(turns), dim3(64), 0, st, cell_s, idx_s, n, nb, hw, ra);
            e = hipGetLastError();
        }
    } else if (e == hipSuccess) {
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
    // x | y | values | raster a | raster b | out (u8) | flag words | sort buffers; the columns are staged in the sort buffers when
    // they fit (they are interleaved -- or, byte-valued, packed -- into `values` before the first sort), else behind them
    const size_t sort_bytes = rz_sort_bytes(n, nullptr);
    const bool stage_in_sort = cols && pts_bytes * nb <= sort_bytes;
    const size_t bytes = pts_bytes * (2 + nb) + total * (4 + 4 + 1) + 512 + 256 + sort_bytes + ((cols && !stage_in_sort) ? pts_bytes * nb + 64 : 0);
    ALP_HIP(hipMalloc((void **)&dev, bytes));
    double *dx = (double *)dev, *dy = dx + n, *dv = dy + n;
    float *ra = (float *)(dv + (size_t)n * nb), *rb = ra + total;
    unsigned char *out_dev = (unsigned char *)(rb + total);
    unsigned *look_dev = (unsigned *)(((uintptr_t)(out_dev + total) + 255) & ~(uintptr_t)255);        // 64 words
    char *sort_area = (char *)(look_dev + 64);
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
        RzLook look;
        bool packed = false;
        if (cols) {
            // the columns as they lie: one look at them says whether they are an image's bytes -- then they are packed into the
            // sort's payload straight from the columns (to_geotiff on a uint8 photograph, project.py:364: the usual case) and
            // never interleaved (0.17 ms for the 100 M-vertex frame's table, and the look at the interleaved copy 0.07)
            const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, (long long)ctx().cu_count * 8);
            rc = rz_look_at(agg, planar, n, (int)nb, true, look_dev, &look);
            packed = rc == ALP_OK && rz_can_pack(agg, n, (int)nb) && look.bytes(agg, (int)nb);
            if (packed) {
                hipLaunchKernelGGL(rz_pack_kernel, dim3(grid), dim3(256), 0, st, planar, (long long)n, (int)nb, 1ll, (long long)n, (unsigned *)dv);
                e = hipMemcpyAsync(rz_payload_slot(sort_area, n), dv, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToDevice, st);   // (the slot may lie under the columns)
            } else if (rc == ALP_OK) {
                hipLaunchKernelGGL(rz_interleave_kernel, dim3(grid), dim3(256), 0, st, planar, (long long)n, (int)nb, dv);
            }
        }
        float *f32_dev = nullptr;
        if (rc == ALP_OK && e == hipSuccess)
            rc = run_rasterize(agg, dx, dy, dv, n, (int)nb, x_min, y_max, resolution, (int)width, (int)height, sweeps, nodata, ra, rb,
                               out_dev, sort_area, out_f32 ? &f32_dev : nullptr, packed, cols ? &look : nullptr);
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
