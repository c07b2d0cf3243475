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

#ifdef INDEX_LDS_LAB
// ------------------------------------------------------------------ LAB (round 5, not kept): per-triangle raster, vertices shared through LDS
// tools/build_variant.sh index_lds -DINDEX_LDS_LAB (a development switch: needs -DALP_DEV, release builds do not contain it).
// Bit-exact -- same image hash at 100 M vertices, 161 render tests green with it forced -- and SLOWER on both a grid-ordered
// array (2.57 against 1.91 ms per frame) and a shuffled one (11.9 against 11.0): profiles/r05_raster_index_lds_not_kept.txt.
// VERDICT round 4, task 6.  A workgroup's 256 consecutive triangles of a surface mesh name ~260 distinct vertices
// with 768 indices; raster_kernel transforms, projects and snaps each of them three times over (and 90 % of its time is vector
// instructions, profiles/r04_raster_index_two_pass_not_kept.txt).  Here the block's indices are entered into an LDS hash set
// (open addressing, atomicCAS; the thread whose CAS created an entry appends it to a list), the LISTED vertices are transformed
// once each by consecutive lanes -- the same to_view / to_window / snap, so the same integers -- and the triangles are set up
// from the LDS records.  Whether an index array shares enough to pay for the hashing is measured once per mesh
// (index_sharing_kernel: distinct vertices per block over a sample of blocks); an array without locality keeps raster_kernel.
constexpr int IDX_SLOTS = 1024;            // hash slots per block of 256 triangles (768 references at most)
constexpr float INDEX_SHARING_MAX = 0.6f;  // distinct vertices / references of a block up to which the LDS path is taken (a grid: 0.34)
struct alignas(16) SnapRec {
    int X, Y;
    float iw;
    int flags;                             // bit 0: in front of the near plane, bit 1: inside the fixed-point range
};

__device__ __forceinline__ int idx_enter(int *__restrict__ keys, int idx, bool &created) {
    unsigned h = ((unsigned)idx * 2654435761u) >> 22;                        // 10 bits
    for (;;) {
        const int old = atomicCAS(&keys[h], -1, idx);
        if (old == -1) { created = true; return (int)h; }
        if (old == idx) { created = false; return (int)h; }
        h = (h + 1) & (IDX_SLOTS - 1);
    }
}

__global__ __launch_bounds__(256) void raster_index_lds_kernel(const float *__restrict__ vert, const int *__restrict__ ind,
                                                               const unsigned char *__restrict__ valid, long long n_tri, View v,
                                                               unsigned long long *__restrict__ vis, unsigned *__restrict__ gqueue,
                                                               unsigned *__restrict__ gcount, unsigned gcap) {
    __shared__ int keys[IDX_SLOTS];
    __shared__ SnapRec rec[IDX_SLOTS];
    __shared__ unsigned short list[768];
    __shared__ unsigned nlist;
    const long long blocks = (n_tri + 255) / 256;
    for (long long b = blockIdx.x; b < blocks; b += gridDim.x) {
        for (int i = threadIdx.x; i < IDX_SLOTS; i += 256) keys[i] = -1;
        if (threadIdx.x == 0) nlist = 0;
        __syncthreads();
        const long long t = b * 256 + threadIdx.x;
        bool draw = t < n_tri;
        int id[3] = {0, 0, 0}, slot[3] = {0, 0, 0};
        if (draw) {
            id[0] = ind[t * 3 + 0];
            id[1] = ind[t * 3 + 1];
            id[2] = ind[t * 3 + 2];
            if (valid) draw = valid[id[0]] && valid[id[1]] && valid[id[2]];
        }
        if (draw) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                bool created;
                slot[k] = idx_enter(keys, id[k], created);
                if (created) list[atomicAdd(&nlist, 1u)] = (unsigned short)slot[k];
            }
        }
        __syncthreads();
        const unsigned n = nlist;
        for (unsigned i = threadIdx.x; i < n; i += 256) {            // one vertex per lane, dense
            const int s = list[i];
            const float *p = vert + 3ll * keys[s];
            float q[3];
            to_view(v, p[0], p[1], p[2], q);
            SnapRec r = {0, 0, 0.0f, 0};
            if (q[2] >= 1.0f) {
                float xw, yw;
                to_window(v, q, xw, yw, r.iw);
                r.flags = 1 | ((fabsf(xw) < COORD_LIMIT && fabsf(yw) < COORD_LIMIT) ? 2 : 0);
                r.X = snap(xw);
                r.Y = snap(yw);
            }
            rec[s] = r;
        }
        __syncthreads();
        Deferred park;
        int code = EMIT_DONE;
        if (draw) {
            const SnapRec r0 = rec[slot[0]], r1 = rec[slot[1]], r2 = rec[slot[2]];
            const int all = r0.flags & r1.flags & r2.flags, any = r0.flags | r1.flags | r2.flags;
            if (all & 1) {
                if (all & 2) {
                    const int X[3] = {r0.X, r1.X, r2.X}, Y[3] = {r0.Y, r1.Y, r2.Y};
                    const float iw[3] = {r0.iw, r1.iw, r2.iw};
                    code = emit_small(v, X, Y, iw, 0, 1, 2, (unsigned)t, vis, &park, true, 4, 16);
                } else {
                    code = EMIT_GENERAL;
                }
            } else if (any & 1) {
                code = EMIT_GENERAL;
            }
            if (code == EMIT_GENERAL) {
                const unsigned at = atomicAdd(gcount, 1u);
                if (at < gcap) gqueue[at] = (unsigned)t;
            }
        }
        coop_drain(v, code == EMIT_PARKED, park, vis);
        __syncthreads();                                               // the records are read; the next block clears them
    }
}

// distinct vertices per block of 256 triangles, summed over every `step`-th block: sums[0] += distinct, sums[1] += references
__global__ __launch_bounds__(256) void index_sharing_kernel(const int *__restrict__ ind, long long n_tri, long long step,
                                                            unsigned long long *__restrict__ sums) {
    __shared__ int keys[IDX_SLOTS];
    __shared__ unsigned distinct, refs;
    const long long blocks = (n_tri + 255) / 256;
    for (long long b = (long long)blockIdx.x * step; b < blocks; b += (long long)gridDim.x * step) {
        for (int i = threadIdx.x; i < IDX_SLOTS; i += 256) keys[i] = -1;
        if (threadIdx.x == 0) distinct = refs = 0;
        __syncthreads();
        const long long t = b * 256 + threadIdx.x;
        if (t < n_tri) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                bool created;
                idx_enter(keys, ind[t * 3 + k], created);
                if (created) atomicAdd(&distinct, 1u);
            }
            atomicAdd(&refs, 3u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(&sums[0], (unsigned long long)distinct);
            atomicAdd(&sums[1], (unsigned long long)refs);
        }
        __syncthreads();
    }
}
#endif   // INDEX_LDS_LAB

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
