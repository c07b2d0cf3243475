// Part of alp_raster.hip (one translation unit, included inside namespace alp in the order given there; not a
// stand-alone header): the resolve stage (perspective-correct interpolation + lens-distortion remap + min_distance) and the stand-alone distort.
#pragma once

// ------------------------------------------------------------------ kernel 4: resolve + remap
__device__ __forceinline__ bool remap_source(const RemapCoef &c, int w, int h, int x, int y, int &sx, int &sy,
                                             float *mapx = nullptr, float *mapy = nullptr) {
    const double x1 = (x - c.c0) / c.c0, y1 = (y - c.c1) / c.c1;
    const double r = __builtin_sqrt(x1 * x1 + y1 * y1), r2 = r * r, r4 = r2 * r2, r6 = r4 * r2;
    const double xd = x1 * (1 + c.k1 * r2 + c.k2 * r4 + c.k3 * r6) / (1 + c.k4 * r2 + c.k5 * r4 + c.k6 * r6) +
                      2 * c.p1 * x1 * y1 + c.p2 * (r2 * 2 * x1 * x1) + c.s1 * r2 + c.s2 * r4;
    const double yd = y1 * (1 + c.a1 + c.k1 * r2 + c.k2 * r4 + c.k3 * r6) / (1 + c.a2 + c.k4 * r2 + c.k5 * r4 + c.k6 * r6) +
                      2 * c.p1 * x1 * y1 + c.p2 * (r2 * 2 * y1 * y1) + c.s3 * r2 + c.s4 * r4;
    const float mx = (float)(xd * c.c0 + c.c0), my = (float)(yd * c.c1 + c.c1);
    if (mapx) { *mapx = mx; *mapy = my; }
    const double rx = __builtin_rint((double)mx), ry = __builtin_rint((double)my);
    if (!(rx >= 0 && rx < w && ry >= 0 && ry < h)) return false;
    sx = (int)rx;
    sy = (int)ry;
    return true;
}

// the float32 source map itself (what distort() hands to cv2.remap, project.py:140-141)
__global__ __launch_bounds__(256) void distort_map_kernel(int w, int h, RemapCoef rc, float *__restrict__ map_x,
                                                          float *__restrict__ map_y) {
    const long long npix = (long long)w * h;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        const int y = (int)(p / w), x = (int)(p - (long long)y * w);
        int sx, sy;
        remap_source(rc, w, h, x, y, sx, sy, map_x + p, map_y + p);
    }
}

// (Measured and not kept, round 2: a two-stage software pipeline -- the loads of a thread's next pixel in
// flight during the float64 interpolation of the current one.  The kernel without its float64 arithmetic
// takes 115 us, with it 176 us; the pipelined form needs 111 VGPRs (4 waves per SIMD) and takes 167 us,
// 220 us at 5 waves and 500 us at 6 (spills).)
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void resolve_kernel(const float *__restrict__ vert, const float *__restrict__ value,
                                                      const int *__restrict__ ind, long long gw, View v,
                                                      RemapCoef rc, int identity_remap, double min_distance,
                                                      const unsigned long long *__restrict__ vis,
                                                      float *__restrict__ out, const unsigned *__restrict__ frame_counts,
                                                      unsigned *__restrict__ host_counts) {
    // the frame's queue counters go to pinned host memory for finish_frame (a copy node of its own costs 5 us)
    if (host_counts && blockIdx.x == 0 && threadIdx.x < 2 * QC_STRIDE) host_counts[threadIdx.x] = frame_counts[threadIdx.x];
    const long long npix = (long long)v.w * v.h;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        const int y = (int)((unsigned)p / (unsigned)v.w), x = (int)((unsigned)p - (unsigned)y * (unsigned)v.w);   // w * h <= 2^30
        float o[3] = {0.0f, 0.0f, 0.0f};
        int sx, sy;
        // no distortion at all (a1 = a2 = 1, everything else 0): the float64 map returns the
        // pixel itself for every image size (checked exhaustively up to 32768), skip it
        if (identity_remap ? (sx = x, sy = y, true) : remap_source(rc, v.w, v.h, x, y, sx, sy)) {
            const int j = v.h - 1 - sy;                               // flipud: image row -> GL row
            const unsigned long long key = vis[(size_t)j * v.w + sx];
            if (key) {
                const long long t = (long long)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
                const Idx3 id = tri_vertices<IMPLICIT>(ind, gw, t);
                const long long ids[3] = {id.a, id.b, id.c};
                // float64 view-space vertices: Rd (v - camd)
                double Q[3][3];
                float P[3][3];                     // the vertices themselves: they are the values when no value array is given
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float *pp = vert + 3 * ids[k];
                    P[k][0] = pp[0];
                    P[k][1] = pp[1];
                    P[k][2] = pp[2];
                    const double d0 = (double)P[k][0] - v.camd[0], d1 = (double)P[k][1] - v.camd[1], d2 = (double)P[k][2] - v.camd[2];
#pragma unroll
                    for (int c = 0; c < 3; ++c) Q[k][c] = (v.Rd[c][0] * d0 + v.Rd[c][1] * d1) + v.Rd[c][2] * d2;
                }
                const double *A = Q[0], *B = Q[1], *C = Q[2];
                const double r[3] = {(((double)sx + 0.5) * v.kx - 1.0) * v.ifx,
                                     (((double)j + 0.5) * v.ky - 1.0) * v.ify, 1.0};
                const double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
                const double pv[3] = {r[1] * e2[2] - r[2] * e2[1], r[2] * e2[0] - r[0] * e2[2], r[0] * e2[1] - r[1] * e2[0]};
                const double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
                const double tv[3] = {-A[0], -A[1], -A[2]};
                const double inv_det = 1.0 / det;      // the one division of the interpolation
                const double beta = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv_det;
                const double qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
                const double gamma = (r[0] * qv[0] + r[1] * qv[1] + r[2] * qv[2]) * inv_det;
                const double alpha = 1.0 - beta - gamma;
                bool masked = false;
                if (min_distance > 0) {
                    const double dA = __builtin_sqrt(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]);
                    const double dB = __builtin_sqrt(B[0] * B[0] + B[1] * B[1] + B[2] * B[2]);
                    const double dC = __builtin_sqrt(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
                    masked = alpha * dA + beta * dB + gamma * dC < min_distance;
                }
                if (!masked) {
                    if (value) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float *pv_ = value + 3 * ids[k];
                            P[k][0] = pv_[0];
                            P[k][1] = pv_[1];
                            P[k][2] = pv_[2];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) o[c] = (float)(alpha * P[0][c] + beta * P[1][c] + gamma * P[2][c]);
                }
            }
        }
        out[p * 3 + 0] = o[0];
        out[p * 3 + 1] = o[1];
        out[p * 3 + 2] = o[2];
    }
}

// stand-alone distort(): gather of an h x w x c image
__global__ __launch_bounds__(256) void distort_image_kernel(const float *__restrict__ img, int w, int h, int c,
                                                            RemapCoef rc, float *__restrict__ out) {
    const long long npix = (long long)w * h;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        const int y = (int)(p / w), x = (int)(p - (long long)y * w);
        int sx, sy;
        const bool ok = remap_source(rc, w, h, x, y, sx, sy);
        for (int k = 0; k < c; ++k) out[p * c + k] = ok ? img[((long long)sy * w + sx) * c + k] : 0.0f;
    }
}
