// Part of alp_raster.hip (one translation unit, included inside namespace alp in the order given there; not a
// stand-alone header): the per-triangle kernels of explicit index arrays (raster_kernel) and of the rare cases set aside by every path
// (raster_general_kernel: near-plane crossings, triangles of 64 px and more).
#pragma once

// ------------------------------------------------------------------ kernel 2: per-triangle raster
// One thread per triangle, three gathered vertices.  Like raster_grid_kernel it finishes only the
// common case itself (all vertices in front and in range, under 64 px) and sets the rest aside
// for raster_general_kernel.
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void raster_kernel(const float *__restrict__ vert, const int *__restrict__ ind,
                                                     const unsigned char *__restrict__ valid,
                                                     long long n_tri, long long gw, View v,
                                                     unsigned long long *__restrict__ vis,
                                                     unsigned *__restrict__ gqueue, unsigned *__restrict__ gcount,
                                                     unsigned gcap) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long rounds = (n_tri + stride - 1) / stride;       // every lane makes every round: coop_drain is wave-wide
    for (long long k = 0; k < rounds; ++k) {
        const long long t = k * stride + (long long)blockIdx.x * blockDim.x + threadIdx.x;
        Deferred park;
        int code = EMIT_DONE;
        bool draw = t < n_tri;
        if (draw && valid) {
            const Idx3 id = tri_vertices<IMPLICIT>(ind, gw, t);
            draw = valid[id.a] && valid[id.b] && valid[id.c];
        }
        if (draw) {
            float q[3][3];
            load_view_tri<IMPLICIT>(v, vert, ind, gw, t, q);
            const bool in0 = q[0][2] >= 1.0f, in1 = q[1][2] >= 1.0f, in2 = q[2][2] >= 1.0f;
            if (in0 && in1 && in2) {
                float xw[3], yw[3], iw[3];
                bool ok = true;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    to_window(v, q[c], xw[c], yw[c], iw[c]);
                    ok = ok && fabsf(xw[c]) < COORD_LIMIT && fabsf(yw[c]) < COORD_LIMIT;
                }
                if (ok) {
                    const int X[3] = {snap(xw[0]), snap(xw[1]), snap(xw[2])};
                    const int Y[3] = {snap(yw[0]), snap(yw[1]), snap(yw[2])};
                    // without the cell fast path in front of it, parking pays from 4 columns / 16 centres (measured)
                    code = emit_small(v, X, Y, iw, 0, 1, 2, (unsigned)t, vis, &park, true, 4, 16);
                } else {
                    code = EMIT_GENERAL;
                }
            } else if (in0 || in1 || in2) {
                code = EMIT_GENERAL;
            }
            if (code == EMIT_GENERAL) {
                const unsigned slot = atomicAdd(gcount, 1u);
                if (slot < gcap) gqueue[slot] = (unsigned)t;
            }
        }
        coop_drain(v, code == EMIT_PARKED, park, vis);
    }
}

// The triangles raster_grid_kernel set aside (near-plane crossings, 64 px and more): one thread
// per entry of the general queue, the same path as raster_kernel.  The entry count is read on
// the device, so no host round trip separates the passes.
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void raster_general_kernel(const float *__restrict__ vert,
                                                             const int *__restrict__ ind, long long gw, View v,
                                                             unsigned long long *__restrict__ vis,
                                                             const unsigned *__restrict__ gqueue,
                                                             const unsigned *__restrict__ gcount, unsigned gcap,
                                                             WorkItem *__restrict__ queue,
                                                             unsigned *__restrict__ qcount, unsigned qcap) {
    const unsigned n = min(*gcount, gcap);
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned it = blockIdx.x * blockDim.x + threadIdx.x; it < n; it += stride) {
        const long long t = gqueue[it];
        float q[3][3];
        load_view_tri<IMPLICIT>(v, vert, ind, gw, t, q);
        emit_general(v, q, t, vis, queue, qcount, qcap);
    }
}
