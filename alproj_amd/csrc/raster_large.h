// Part of alp_raster.hip (one translation unit, included inside namespace alp in the order given there; not a
// stand-alone header): raster_large_kernel: the 64 x 64-pixel work items of large triangles.
#pragma once

// ------------------------------------------------------------------ kernel 3: large triangles
// one wave per (triangle, 64x64-pixel tile): lane = pixel column, loop over the rows
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void raster_large_kernel(const float *__restrict__ vert,
                                                           const int *__restrict__ ind, long long gw, View v,
                                                           unsigned long long *__restrict__ vis,
                                                           const WorkItem *__restrict__ queue,
                                                           const unsigned *__restrict__ qcount, unsigned qcap) {
    const unsigned count = min(*qcount, qcap);
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned it = wave; it < count; it += nwaves) {
        const WorkItem wi = queue[it];
        float q[3][3];
        load_view_tri<IMPLICIT>(v, vert, ind, gw, (long long)wi.tri, q);
        if (wi.sub == 0xFFFF) {
            raster_big(v, q, wi.tri, vis, lane);
            continue;
        }
        float xw[4], yw[4], iw[4];
        bool big;
        const int ntri = clip_project(v, q, xw, yw, iw, big);
        const int f = wi.sub;
        if (f >= ntri) continue;
        const float x3[3] = {xw[0], xw[f + 1], xw[f + 2]}, y3[3] = {yw[0], yw[f + 1], yw[f + 2]},
                    i3[3] = {iw[0], iw[f + 1], iw[f + 2]};
        const TriSetup s = setup_tri(v, x3, y3, i3);
        if (!s.valid) continue;
        const int i = wi.tx * TILE + lane;
        int ja = wi.ty * TILE, jb = ja + TILE - 1;
        ja = ja < s.j0 ? s.j0 : ja;
        jb = jb > s.j1 ? s.j1 : jb;
        if (i < s.i0 || i > s.i1) continue;
        for (int j = ja; j <= jb; ++j) {
            const unsigned long long key = pixel_key(s, i, j, wi.tri);
            if (key) vis_max(vis, v, i, j, key);
        }
    }
}
