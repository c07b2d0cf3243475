// Part of alp_raster.hip (one translation unit, included inside namespace alp in the order given there; not a
// stand-alone header): what runs on a finished frame or around a mesh: uint8 conversion, pixel gathers, the x > 0 compaction of reverse_proj,
// recognition of regular-grid index arrays, typed uploads.
#pragma once

// ------------------------------------------------------------------ reverse_proj post-processing
// src/alproj/project.py:361-373 on the device: keep the pixels whose first rendered channel
// (offset-relative x) is > 0 (quirk Q13), in row-major pixel order, and return their linear
// index and x, y, z = channels 0, 2, 1 (+ offsets, added in float64 like the reference does).
// Pass 1 counts per chunk of COMPACT_CHUNK pixels, a one-workgroup scan turns the counts into
// offsets, pass 2 writes (order-preserving stream compaction).
constexpr int COMPACT_CHUNK = 4096;

// set_gcp (src/alproj/gcp.py:644-648) against the resident coordinate image instead of a merge
// with the reverse_proj table: pixel (u[i], v[i]) -> x, y, z = channels (0, 2, 1) + offsets,
// NaN where the pixel is outside the image or does not see the surface (x <= 0, project.py:369)
// sim_image's tail (project.py:322-324): (raw * 255).astype(uint8), RGB -> BGR.  numpy's float32 -> uint8 cast is the
// x86 truncating conversion to int32 followed by a wrap to 8 bits (NaN and out-of-range give 0x80000000 -> 0).
__global__ __launch_bounds__(256) void image_u8_kernel(const float *__restrict__ img, long long npix, float scale, int reverse,
                                                       unsigned char *__restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        unsigned char b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = img[3 * i + c] * scale;
            const int q = (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : (int)0x80000000;   // false for NaN too
            b[c] = (unsigned char)(q & 0xFF);
        }
        out[3 * i + 0] = reverse ? b[2] : b[0];
        out[3 * i + 1] = b[1];
        out[3 * i + 2] = reverse ? b[0] : b[2];
    }
}

__global__ __launch_bounds__(256) void gather_pixels_kernel(const float *__restrict__ image, int w, int h,
                                                            const int *__restrict__ u, const int *__restrict__ v,
                                                            long long n, double o0, double o1, double o2,
                                                            double *__restrict__ xyz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double nan = __builtin_nan("");
    double x = nan, y = nan, z = nan;
    const int uu = u[i], vv = v[i];
    if (uu >= 0 && uu < w && vv >= 0 && vv < h) {
        const float *px = image + 3 * ((long long)vv * w + uu);
        if (px[0] > 0.0f) {
            x = (double)px[0] + o0;
            y = (double)px[2] + o2;
            z = (double)px[1] + o1;
        }
    }
    xyz[3 * i + 0] = x;
    xyz[3 * i + 1] = y;
    xyz[3 * i + 2] = z;
}

// The remaining columns of reverse_proj's table (project.py:361-368) for the M surviving pixels idx[i]: the table's labels (the
// linear pixel index, int64), u = column and v = row as int16 (the reference's meshgrid().astype("int16"), wrapping like
// numpy's cast), and the caller's image channels at that pixel as float64 rows chan[c][i] (numpy's cast on assignment).
template <typename A>
__global__ __launch_bounds__(256) void table_columns_kernel(const unsigned *__restrict__ idx, long long M, int w, const A *__restrict__ array,
                                                            int channels, long long *__restrict__ index64, short *__restrict__ u16,
                                                            short *__restrict__ v16, double *__restrict__ chan) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const unsigned p = idx[i];
    const unsigned v = p / (unsigned)w, u = p - v * (unsigned)w;
    index64[i] = (long long)p;
    u16[i] = (short)u;
    v16[i] = (short)v;
    const A *px = array + (long long)p * channels;
    for (int c = 0; c < channels; ++c) chan[(long long)c * M + i] = (double)px[c];
}

// filter_gcp_distance, src/alproj/gcp.py:711-724: rows with a NaN coordinate are dropped, the others kept when their
// distance from the camera -- sqrt(dx^2 + dy^2 + dz^2) in numpy's order, no contraction -- lies in [lo, hi] (a NaN bound
// is an absent one; a NaN distance fails every comparison, like numpy's)
__global__ __launch_bounds__(256) void distance_mask_kernel(const double *__restrict__ xyz, long long n, double cx, double cy,
                                                            double cz, double lo, double hi, unsigned char *__restrict__ keep) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = xyz[3 * i + 0], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    bool k = x == x && y == y && z == z;
    const double dx = x - cx, dy = y - cy, dz = z - cz;
    const double d = sqrt(dx * dx + dy * dy + dz * dz);
    if (lo == lo) k = k && d >= lo;
    if (hi == hi) k = k && d <= hi;
    keep[i] = k ? 1 : 0;
}

// also, per chunk, the extent of the survivors' channels 0 and 2 (min x, max x, min y, max y -> span[4 * chunk ..]; +-inf
// for a chunk without one): to_geotiff's bounds (project.py:420-423) are these plus the offsets -- x -> (double)x + o only
// grows with x, so the minimum of the sums is the sum at the minimum -- and cost no pass of their own over the table
__global__ __launch_bounds__(256) void valid_count_kernel(const float *__restrict__ img, long long npix,
                                                          unsigned *__restrict__ counts, float *__restrict__ span) {
    __shared__ unsigned s[4];
    __shared__ float s_span[4][4];
    const long long base = (long long)blockIdx.x * COMPACT_CHUNK;
    unsigned c = 0;
    float x_lo = INFINITY, x_hi = -INFINITY, y_lo = INFINITY, y_hi = -INFINITY;
    for (int k = threadIdx.x; k < COMPACT_CHUNK; k += 256) {
        const long long p = base + k;
        if (p >= npix) continue;
        const float x = img[p * 3];
        if (x > 0.0f) {
            const float y = img[p * 3 + 2];
            ++c;
            x_lo = fminf(x_lo, x); x_hi = fmaxf(x_hi, x);
            y_lo = fminf(y_lo, y); y_hi = fmaxf(y_hi, y);
        }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        c += __shfl_xor(c, m, 64);
        x_lo = fminf(x_lo, __shfl_xor(x_lo, m, 64)); x_hi = fmaxf(x_hi, __shfl_xor(x_hi, m, 64));
        y_lo = fminf(y_lo, __shfl_xor(y_lo, m, 64)); y_hi = fmaxf(y_hi, __shfl_xor(y_hi, m, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        s[w] = c;
        s_span[w][0] = x_lo; s_span[w][1] = x_hi; s_span[w][2] = y_lo; s_span[w][3] = y_hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x < 4) {
        const int k = (int)threadIdx.x;
        const float a = s_span[0][k], b = s_span[1][k], d = s_span[2][k], e = s_span[3][k];
        span[4 * (long long)blockIdx.x + k] = (k & 1) ? fmaxf(fmaxf(a, b), fmaxf(d, e)) : fminf(fminf(a, b), fminf(d, e));
    }
}

// exclusive scan of n counts (n up to a few ten thousand) by one workgroup; total -> offsets[n]; the chunks' extents
// joined -> the four floats behind it (offsets[n + 1], [n + 2])
__global__ __launch_bounds__(1024) void scan_counts_kernel(const unsigned *__restrict__ counts, int n,
                                                           unsigned long long *__restrict__ offsets, const float *__restrict__ span) {
    __shared__ unsigned long long s[1024];
    __shared__ float s_span[16][4];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    unsigned long long sum = 0;
    float e[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int i = lo; i < hi; ++i) {
        sum += counts[i];
        const float4 sp = *(const float4 *)(span + 4 * (long long)i);
        e[0] = fminf(e[0], sp.x); e[1] = fmaxf(e[1], sp.y); e[2] = fminf(e[2], sp.z); e[3] = fmaxf(e[3], sp.w);
    }
    for (int m = 32; m >= 1; m >>= 1) {
        e[0] = fminf(e[0], __shfl_xor(e[0], m, 64)); e[1] = fmaxf(e[1], __shfl_xor(e[1], m, 64));
        e[2] = fminf(e[2], __shfl_xor(e[2], m, 64)); e[3] = fmaxf(e[3], __shfl_xor(e[3], m, 64));
    }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 4; ++k) s_span[threadIdx.x >> 6][k] = e[k];
    s[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = (int)threadIdx.x;
        float r = s_span[0][k];
        for (int w = 1; w < 16; ++w) r = (k & 1) ? fmaxf(r, s_span[w][k]) : fminf(r, s_span[w][k]);
        ((float *)(offsets + n + 1))[k] = r;
    }
    for (int d = 1; d < 1024; d <<= 1) {                 // Hillis-Steele inclusive scan
        unsigned long long t = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned long long run = threadIdx.x ? s[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        offsets[i] = run;
        run += counts[i];
    }
    if (threadIdx.x == 1023) offsets[n] = s[1023];
}

__global__ __launch_bounds__(256) void valid_write_kernel(const float *__restrict__ img, long long npix,
                                                          const unsigned long long *__restrict__ offsets,
                                                          double o0, double o1, double o2,
                                                          unsigned *__restrict__ idx_out,
                                                          double *__restrict__ xyz_out, long long elem_stride,
                                                          long long plane_stride) {      // (3, 1): rows of x, y, z; (1, M): three planes
    __shared__ unsigned s_wave[4];
    const long long base = (long long)blockIdx.x * COMPACT_CHUNK;
    unsigned long long out = offsets[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k0 = 0; k0 < COMPACT_CHUNK; k0 += 256) {   // consecutive pixels per pass keep the order
        const long long p = base + k0 + threadIdx.x;
        float c0 = 0, c1 = 0, c2 = 0;
        bool valid = false;
        if (p < npix) {
            c0 = img[p * 3];
            valid = c0 > 0.0f;
            if (valid) { c1 = img[p * 3 + 1]; c2 = img[p * 3 + 2]; }
        }
        const unsigned long long m = __ballot(valid);
        const unsigned before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        unsigned wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_wave[w];
        const unsigned total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (valid) {
            const unsigned long long o = out + wbase + before;
            idx_out[o] = (unsigned)p;
            xyz_out[o * elem_stride] = (double)c0 + o0;                        // x  (channel 0 + offsets[0])
            xyz_out[o * elem_stride + plane_stride] = (double)c2 + o2;         // y  (channel 2 + offsets[2])
            xyz_out[o * elem_stride + 2 * plane_stride] = (double)c1 + o1;     // z  (channel 1 + offsets[1])
        }
        out += total;
        __syncthreads();
    }
}

// does an index array spell out exactly the regular grid of surface.py:194-201 with gw columns?  One staged chunk of
// the caller's array (int32 or int64 as it came): triangles [t0, t0 + n_tri) of the grid.
template <typename I>
__global__ __launch_bounds__(256) void check_grid_chunk_kernel(const I *__restrict__ chunk, long long n_tri, long long t0, long long gw,
                                                               unsigned *__restrict__ mismatch) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    bool bad = false;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n_tri; t += stride) {
        const Idx3 e = tri_vertices<true>(nullptr, gw, t0 + t);
        bad |= (long long)chunk[3 * t] != e.a || (long long)chunk[3 * t + 1] != e.b || (long long)chunk[3 * t + 2] != e.c;
    }
    if (bad) *mismatch = 1u;
}

// Is an index array a FILTERED regular grid (the triangles of surface.py:194-201 in their order, some
// removed -- what get_colored_surface returns for a DSM with nodata, surface.py:203-205)?  Pass 1, per
// triangle of the array: it must be a grid triangle, later in grid order than its predecessor; its bit is
// set in `present`, its vertices are marked.  Pass 2, per grid triangle NOT in the array: one of its
// vertices must be unmarked -- then "draw the triangles whose three vertices are marked" draws exactly the
// array, and the mesh is rendered by the implicit-grid kernels with that vertex mask.
__device__ __forceinline__ long long subgrid_id(const int *__restrict__ ind, long long t, long long gw, long long gh) {
    const long long a = ind[3 * t], b = ind[3 * t + 1], c = ind[3 * t + 2];
    int type;
    if (b == a + gw && c == a + gw + 1) type = 0;
    else if (b == a + gw + 1 && c == a + 1) type = 1;
    else return -1;
    if (a < 0) return -1;
    const unsigned row = (unsigned)a / (unsigned)gw, col = (unsigned)a - row * (unsigned)gw;      // fewer than 2^31 vertices: 32-bit division
    if ((long long)row >= gh - 1 || (long long)col >= gw - 1) return -1;
    return 2 * ((long long)row * (gw - 1) + col) + type;
}

// One lane per triangle of the array, consecutive lanes = consecutive triangles: the predecessor's id comes from the
// neighbouring lane, and because the ids of a valid array increase, the lanes that fall into one 32-bit word of `present`
// are a contiguous run -- its bits are OR-ed inside the wave and the run's first lane sends ONE atomic (the version with an
// atomic per triangle spent 10.8 ms on the 2e8 triangles of a 100 M-vertex DSM with nodata, 32 atomics per word).
__global__ __launch_bounds__(256) void subgrid_mark_kernel(const int *__restrict__ ind, long long n_tri, long long gw, long long gh,
                                                           unsigned *__restrict__ present, unsigned char *__restrict__ mark,
                                                           unsigned *__restrict__ mismatch) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long rounds = (n_tri + stride - 1) / stride;              // every lane makes every round: the shuffles are wave-wide
    const int lane = (int)(threadIdx.x & 63);
    bool bad = false;
    for (long long k = 0; k < rounds; ++k) {
        const long long t = k * stride + (long long)blockIdx.x * blockDim.x + threadIdx.x;
        const bool in = t < n_tri;
        const long long id = in ? subgrid_id(ind, t, gw, gh) : -1;
        long long prev = __shfl_up(id, 1);
        if (lane == 0) prev = (in && t > 0) ? subgrid_id(ind, t - 1, gw, gh) : -1;
        const bool ok = in && id >= 0 && !(t > 0 && prev >= id);
        bad |= in && !ok;
        // runs of equal word index among the lanes (lanes that are out or invalid: runs of their own, nothing to store)
        const long long word = ok ? (id >> 5) : (-1 - lane);
        const long long wprev = __shfl_up(word, 1);
        const bool head = lane == 0 || wprev != word;
        const unsigned long long heads = __ballot(head);
        const int run = __popcll(heads & (~0ull >> (63 - lane)));
        unsigned bits = ok ? (1u << (id & 31)) : 0u;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const unsigned b2 = __shfl_down(bits, 1 << j);
            const int r2 = __shfl_down(run, 1 << j);
            if ((lane + (1 << j) < 64) && r2 == run) bits |= b2;
        }
        if (ok) {
            if (head) atomicOr(&present[word], bits);
            mark[ind[3 * t]] = 1;
            mark[ind[3 * t + 1]] = 1;
            mark[ind[3 * t + 2]] = 1;
        }
    }
    if (bad) *mismatch = 1u;
}

__global__ __launch_bounds__(256) void subgrid_absent_kernel(long long n_grid_tri, long long gw, const unsigned *__restrict__ present,
                                                             const unsigned char *__restrict__ mark, unsigned *__restrict__ mismatch) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    bool bad = false;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n_grid_tri; id += stride) {
        if (present[id >> 5] >> (id & 31) & 1u) continue;
        const Idx3 e = tri_vertices<true>(nullptr, gw, id);
        bad |= mark[e.a] && mark[e.b] && mark[e.c];
    }
    if (bad) *mismatch = 1u;
}

// rank[w] = number of set bits in present[0 .. w): block sums, then (after the host scanned them) the words
__global__ __launch_bounds__(256) void subgrid_blocksum_kernel(const unsigned *__restrict__ present, long long n_words,
                                                               unsigned *__restrict__ block_sums) {
    __shared__ unsigned s[4];
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    unsigned c = w < n_words ? (unsigned)__popc(present[w]) : 0u;
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void subgrid_rank_kernel(const unsigned *__restrict__ present, long long n_words,
                                                           const unsigned *__restrict__ block_offsets, unsigned *__restrict__ rank) {
    __shared__ unsigned s[256];
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    s[threadIdx.x] = w < n_words ? (unsigned)__popc(present[w]) : 0u;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                  // Hillis-Steele inclusive scan
        const unsigned t = threadIdx.x >= (unsigned)d ? s[threadIdx.x - d] : 0u;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    if (w < n_words) rank[w] = block_offsets[blockIdx.x] + (threadIdx.x ? s[threadIdx.x - 1] : 0u);
}

// visibility words with the grid's triangle ids -> positions in the caller's (filtered) index array
__global__ __launch_bounds__(256) void vis_translate_kernel(const unsigned long long *__restrict__ vis, long long npix,
                                                            const unsigned *__restrict__ present, const unsigned *__restrict__ rank,
                                                            unsigned long long *__restrict__ out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    unsigned long long key = vis[p];
    if (key) {
        const unsigned id = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
        const unsigned pos = rank[id >> 5] + (unsigned)__popc(present[id >> 5] & ((1u << (id & 31)) - 1u));
        key = (key & 0xFFFFFFFF00000000ull) | (unsigned long long)(0xFFFFFFFFu - pos);
    }
    out[p] = key;
}

// valid = derived AND (user mask or all ones)
__global__ __launch_bounds__(256) void mask_and_kernel(const unsigned char *__restrict__ derived, const unsigned char *__restrict__ user,
                                                       long long n, unsigned char *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = derived[i] && (!user || user[i]);
}

// alp_mesh_create with float64 vertices / values (what get_colored_surface returns, surface.py:189-193): the cast of
// project.py:213-214 (``astype("f4")``: round to nearest even) on the device, chunk by chunk during the upload
__global__ __launch_bounds__(256) void cast_f64_f32_kernel(const double *__restrict__ src, long long count, long long dst_off,
                                                           float *__restrict__ dst) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[dst_off + i] = (float)src[i];
}

// an out-of-range index would fault in the raster kernels: counted on the device (the host loop over 6e8
// indices of a 100 M-vertex mesh took longer than their upload)
__global__ __launch_bounds__(256) void check_index_range_kernel(const int *__restrict__ ind, long long count, long long n_vert,
                                                                unsigned *__restrict__ bad) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned mine = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const int v = ind[i];
        mine += (v < 0 || v >= n_vert) ? 1u : 0u;
    }
    if (mine) atomicAdd(bad, mine);
}

__global__ __launch_bounds__(256) void narrow_indices_kernel(const long long *__restrict__ src, long long count,
                                                             long long dst_off, int *__restrict__ dst, long long n_vert,
                                                             unsigned *__restrict__ bad) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned mine = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const long long v = src[i];
        mine += (v < 0 || v >= n_vert) ? 1u : 0u;
        dst[dst_off + i] = (int)v;
    }
    if (mine) atomicAdd(bad, mine);
}
