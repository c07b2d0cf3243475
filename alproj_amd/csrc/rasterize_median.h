// Part of alp_rasterize.hip (one translation unit, included inside namespace alp in the order given there; not a
// stand-alone header): the median aggregate -- composite-key sorts per band (cell : value), the selection from ONE cell sort for
// byte-valued bands, and the two-sort fallback.
#pragma once

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
// planar_n: 0 for interleaved rows (element i belongs to band i % nb), n for a table's columns as they lie (band i / n)
__global__ __launch_bounds__(256) void rz_median_check_kernel(const double *__restrict__ values, long long count, int nb,
                                                              long long planar_n, unsigned *__restrict__ flag) {
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
            unsigned *f = flag + (planar_n ? (int)(i / planar_n) : (int)(i % nb));
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
// of up to 16 points (92 % of the frame's runs; its cells hold 6.7 points on average) is sorted in the registers of one lane (a bitonic network over its bytes, padded with 256); a longer one (the 100 M-vertex frame: 138 000 of 1.74 M runs,
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
__device__ __forceinline__ float rz_middle(unsigned lo, unsigned hi, long long len) {       // pandas' median of a group: its middle
    return (float)((len & 1) ? (double)lo : ((double)lo + (double)hi) / 2);                 // value, or the mean of the two
}
template <int N>
__device__ __forceinline__ void rz_median_small(const unsigned *__restrict__ pay_s, long long i, int len, int nb, unsigned cell,
                                                long long hw, float *__restrict__ raster) {
    unsigned w[N];
#pragma unroll
    for (int u = 0; u < N; ++u) w[u] = u < len ? pay_s[i + u] : 0u;
    // The N - len places the run does not fill are padded HALF below every byte (0), half above (257; the bytes themselves
    // count from 1): floor((N - len) / 2) pads sort in front of the run's values, so its middle lands on places N / 2 - 1 and
    // N / 2 of the network's output whatever len is -- an odd run's median is place N / 2 - 1, an even run's the mean of the
    // two.  (Padding behind only, and picking places (len - 1) / 2 and len / 2: the compiler kept the sorted array in scratch
    // memory to index it, 80 bytes per lane.)
    const int low = (N - len) >> 1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g >= nb) break;
        unsigned a[N];
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] = u < len ? ((w[u] >> (8 * g)) & 0xFFu) + 1u : (u - len < low ? 0u : 257u);
        rz_sort_small<N>(a);
        raster[(long long)g * hw + cell] = rz_middle(a[N / 2 - 1] - 1u, a[N / 2] - 1u, len);
    }
}

// One wave (a workgroup of 64) per turn of RZ_MED_GROUPS x 64 consecutive sorted positions.  First it finds the heads of the
// runs that start there and files them by length -- 1-2, 3-4, 5-8, 9-16, longer -- in LDS; then each file is worked through
// with all 64 lanes on runs of one kind.  (A head per lane as the positions lie -- every seventh position of the frame is a
// head, and a wave ran the networks of all four sizes for its nine heads -- took 0.27 ms for the frame's 11.7 M points: the
// issue rate of the sorting networks.  Filed, and the turn's cells loaded in one go: 0.185 ms at 8 groups of 64 positions per
// turn; 2 groups 0.237, 4 groups 0.195, 16 groups 0.226 -- fewer, longer turns leave the chip short of waves.)
constexpr int RZ_MED_GROUPS = 8, RZ_MED_TURN = RZ_MED_GROUPS * 64;
template <int GROUPS>
__global__ __launch_bounds__(64) void rz_median_packed_kernel(const unsigned *__restrict__ cell_s, const unsigned *__restrict__ pay_s,
                                                              long long n, int nb, long long hw, float *__restrict__ raster) {
    constexpr int TURN = GROUPS * 64;
    __shared__ __attribute__((aligned(16))) unsigned hist[4][256];       // read back as uint4 (ds_read_b128) by the long-run median
    // an entry: the head's offset in the turn (10 bits) | the run's length << 10 (at most 16); a file cannot hold more heads
    // than the turn has positions / the shortest run of its kind
    __shared__ unsigned short file0[TURN], file1[TURN / 3 + 4], file2[TURN / 5 + 4], file3[TURN / 9 + 4];
    __shared__ unsigned long_off[TURN / 17 + 4], long_len[TURN / 17 + 4];
    const int lane = (int)threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    const long long turns = (n + TURN - 1) / TURN;
    for (long long turn = blockIdx.x; turn < turns; turn += gridDim.x) {
        const long long base = turn * TURN;
        int cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0, cntl = 0;          // wave-uniform
        unsigned cc[GROUPS];                                     // the turn's cells: all loads in flight together
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const long long i = base + g * 64 + lane;
            cc[g] = i < n ? cell_s[i] : 0xFFFFFFFFu;
        }
        unsigned carry = base > 0 ? cell_s[base - 1] : 0xFFFFFFFFu;    // the cell before the group's first position
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const long long gbase = base + g * 64, i = gbase + lane;
            const unsigned c = cc[g];
            unsigned prev = __shfl_up(c, 1);
            if (lane == 0) prev = carry;
            carry = __shfl(c, 63);
            const bool head = i < n && (i == 0 || prev != c);
            const unsigned long long heads = __ballot(head);
            long long hi = i + 1;
            if (head) {
                const unsigned long long later = lane < 63 ? heads >> (lane + 1) : 0ull;
                if (later) hi = i + 1 + __builtin_ctzll(later);           // the next head among the 64
                else {                                                      // the run reaches the end of the 64: gallop, then bisect (rz_runs_kernel)
                    long long lo = gbase + 63 < n ? gbase + 63 : n - 1, step = 1;
                    while (lo + step < n && cell_s[lo + step] == c) { lo += step; step <<= 1; }
                    hi = lo + step < n ? lo + step : n;
                    while (hi - lo > 1) {
                        const long long mid = lo + ((hi - lo) >> 1);
                        if (cell_s[mid] == c) lo = mid; else hi = mid;
                    }
                }
            }
            const unsigned len = head ? (unsigned)(hi - i) : 0u;
            const unsigned off = (unsigned)(g * 64 + lane);
            const unsigned short entry = (unsigned short)(off | (len << 10));      // (meaningful for len <= 16 only)
            const unsigned long long m0 = __ballot(len >= 1u && len <= 2u), m1 = __ballot(len >= 3u && len <= 4u),
                                     m2 = __ballot(len >= 5u && len <= 8u), m3 = __ballot(len >= 9u && len <= 16u), ml = __ballot(len > 16u);
            if (len >= 1u && len <= 2u) file0[cnt0 + __popcll(m0 & below)] = entry;
            else if (len >= 3u && len <= 4u) file1[cnt1 + __popcll(m1 & below)] = entry;
            else if (len >= 5u && len <= 8u) file2[cnt2 + __popcll(m2 & below)] = entry;
            else if (len >= 9u && len <= 16u) file3[cnt3 + __popcll(m3 & below)] = entry;
            else if (len > 16u) {
                const int at = cntl + __popcll(ml & below);
                long_off[at] = off;
                long_len[at] = len;
            }
            cnt0 += __popcll(m0); cnt1 += __popcll(m1); cnt2 += __popcll(m2); cnt3 += __popcll(m3); cntl += __popcll(ml);
        }
        __syncthreads();
        for (int h = lane; h < cnt0; h += 64) {
            const unsigned e = file0[h], len = e >> 10;
            const long long i = base + (e & 1023u);
            const unsigned c = cell_s[i], w0 = pay_s[i];
            if (len == 1u) {
                for (int g = 0; g < nb; ++g) raster[(long long)g * hw + c] = (float)((w0 >> (8 * g)) & 0xFFu);
            } else {
                const unsigned w1 = pay_s[i + 1];
                for (int g = 0; g < nb; ++g) raster[(long long)g * hw + c] = rz_middle((w0 >> (8 * g)) & 0xFFu, (w1 >> (8 * g)) & 0xFFu, 2);
            }
        }
        for (int h = lane; h < cnt1; h += 64) {
            const unsigned e = file1[h];
            const long long i = base + (e & 1023u);
            rz_median_small<4>(pay_s, i, (int)(e >> 10), nb, cell_s[i], hw, raster);
        }
        for (int h = lane; h < cnt2; h += 64) {
            const unsigned e = file2[h];
            const long long i = base + (e & 1023u);
            rz_median_small<8>(pay_s, i, (int)(e >> 10), nb, cell_s[i], hw, raster);
        }
        for (int h = lane; h < cnt3; h += 64) {
            const unsigned e = file3[h];
            const long long i = base + (e & 1023u);
            rz_median_small<16>(pay_s, i, (int)(e >> 10), nb, cell_s[i], hw, raster);
        }
        // the longer runs, one after the other, by the whole wave
        for (int q = 0; q < cntl; ++q) {
            const long long ri = base + long_off[q];
            const unsigned rlen = long_len[q], rcell = cell_s[ri];
            for (int k = lane; k < nb * 256; k += 64) (&hist[0][0])[k] = 0u;
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
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const unsigned k = k2 == 0 ? (rlen - 1u) >> 1 : rlen >> 1;      // the k-th smallest (from 0)
                    unsigned v = 4u * (unsigned)lane, r = k - before;
                    if (r >= cnt.x) { r -= cnt.x; ++v; if (r >= cnt.y) { r -= cnt.y; ++v; if (r >= cnt.z) ++v; } }
                    mid[k2] = __shfl(v, __builtin_ctzll(__ballot(before <= k && k < upto)));
                }
                if (lane == 0) raster[(long long)g * hw + rcell] = rz_middle(mid[0], mid[1], (long long)rlen);
            }
            __syncthreads();                                                // the histograms are read no more
        }
        __syncthreads();                                                    // the files are read no more
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
