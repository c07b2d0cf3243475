// libalproj_hip.so -- mesh-level entry points that are not part of a frame: construction of the
// implicit-grid mesh from DSM / aerial rasters on the device (get_colored_surface after its file
// I/O, src/alproj/surface.py:173-212), the per-vertex validity mask, the value-source selector
// and the read-back used by the parity tests.  Compiled with -ffp-contract=off like alp_raster.hip:
// the float64 coordinate arithmetic of surface.py must not be contracted into FMAs.
#include "alp_raster_internal.h"

#include <cstring>

namespace alp {

// ------------------------------------------------------------------ mesh construction from rasters
// get_colored_surface after its raster I/O (src/alproj/surface.py:173-212) on the device: the
// DSM and the aerial bands go up once (4 + 3..12 B per vertex instead of 24 B of float32 vert +
// col and 48 B of int64 indices), vertices / colours / the nodata mask are built in HBM and the
// index array is never formed (implicit grid + per-vertex mask).
template <typename Z>
__device__ __forceinline__ double surface_z(const Z *dsm, long long i, double z_max) {
    double z = (double)dsm[i];
    if (z < 0) z = 0;                        // surface.py:175
    if (z > z_max) z = z_max;                // surface.py:176
    return z;
}

// min over the clamped elevations (>= 0, so the float64 bit patterns order like the values)
template <typename Z>
__global__ __launch_bounds__(256) void surface_zmin_kernel(const Z *__restrict__ dsm, long long n, double z_max,
                                                           unsigned long long *__restrict__ out) {
    double m = __builtin_inf();
    const long long stride = (long long)gridDim.x * blockDim.x;
    // 16 bytes per lane and turn (one dword per lane ran at 2.1 TB/s: 0.19 ms for the 400 MB of a 10 000 x 10 000 DSM)
    constexpr int V = 16 / (int)sizeof(Z);
    struct alignas(16) Pack { Z v[V]; };
    const long long nv = n / V;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nv; k += stride) {
        const Pack p = reinterpret_cast<const Pack *>(dsm)[k];       // hipMalloc'ed: 256-byte aligned
#pragma unroll
        for (int j = 0; j < V; ++j) {
            double z = (double)p.v[j];
            if (z < 0) z = 0;                // surface.py:175
            if (z > z_max) z = z_max;        // surface.py:176
            m = z < m ? z : m;               // a NaN elevation never becomes the minimum (numpy would return NaN)
        }
    }
    for (long long i = nv * V + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double z = surface_z(dsm, i, z_max);
        m = z < m ? z : m;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const double o = __shfl_xor(m, d);
        m = o < m ? o : m;
    }
    // one atomic per workgroup: 8 192 waves on one address took longer than reading the DSM
    __shared__ double s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = s_m[w] < m ? s_m[w] : m;
        atomicMin(out, (unsigned long long)__double_as_longlong(m));
    }
}

template <typename Z, typename A>
__global__ __launch_bounds__(256) void surface_build_kernel(const Z *__restrict__ dsm, const A *__restrict__ aerial,
                                                            const unsigned char *__restrict__ nodata,
                                                            long long rows, long long cols, double t0, double t2,
                                                            double t4, double t5, double z_max, double color_div,
                                                            double ox, double oz, double oy,
                                                            float *__restrict__ vert, float *__restrict__ value,
                                                            unsigned char *__restrict__ valid) {
    // (measured, not kept: the workgroup's 768 + 768 floats staged in LDS and stored as whole float4 lines instead of the
    // 12-byte-strided dword stores below -- 0.895-0.898 against 0.882-0.903 ms for zmin + build at 100 M vertices, alternating
    // on one box: L2 merges the strided stores into whole lines anyway)
    const long long n = rows * cols;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long r = i / cols, c = i - r * cols;
    // surface.py:179-180, 189, 211-212: float64 coordinates minus the float64 offsets, then the
    // float32 cast of persp_proj (project.py:213); this unit is compiled without fp contraction
    const double x = (double)c * t0 + t2, y = (double)r * t4 + t5, z = surface_z(dsm, i, z_max);
    vert[3 * i + 0] = (float)(x - ox);
    vert[3 * i + 1] = (float)(z - oz);
    vert[3 * i + 2] = (float)(y - oy);
#pragma unroll
    for (int b = 0; b < 3; ++b) {            // _normalize_aerial, surface.py:44-66
        double a = (double)aerial[b * n + i];
        if (color_div > 0) a /= color_div;
        a = a < 0 ? 0 : a;                   // np.clip keeps a NaN
        a = a > 1 ? 1 : a;
        value[3 * i + b] = (float)a;
    }
    valid[i] = nodata ? (nodata[i] ? 0 : 1) : 1;
}

}  // namespace alp

using namespace alp;

extern "C" {

int alp_mesh_set_value_source(alp_mesh_t *m, int source) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m, "mesh handle is NULL");
    ALP_REQUIRE(source == ALP_VALUE_STORED || source == ALP_VALUE_VERTICES, "source must be ALP_VALUE_STORED or ALP_VALUE_VERTICES");
    m->coords_as_value = source == ALP_VALUE_VERTICES;
    return ALP_OK;
}

int alp_mesh_set_value(alp_mesh_t *m, const void *value, int value_dtype) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m, "mesh handle is NULL");
    if (!value) {
        if (m->value) {
            ALP_HIP(hipStreamSynchronize(ctx().stream));      // a resolve that reads it may be in flight
            hipFree(m->value);
        }
        m->value = nullptr;
        return ALP_OK;
    }
    ALP_REQUIRE(value_dtype == ALP_F32 || value_dtype == ALP_F64, "value_dtype must be ALP_F32 or ALP_F64");
    const bool fresh = !m->value;
    if (fresh) ALP_HIP(hipMalloc((void **)&m->value, (size_t)m->n_vert * 12));
    const int rc = upload_f32(m->value, value, value_dtype, m->n_vert);    // stream-ordered behind any resolve in flight
    if (rc && fresh) {               // never leave an allocated, unwritten value buffer behind: it would render as colours
        hipStreamSynchronize(ctx().stream);
        hipFree(m->value);
        m->value = nullptr;
    }
    return rc;
}

int alp_mesh_frame_counts(alp_mesh_t *m, int64_t counts[2]) {
    ALP_REQUIRE(m && counts, "NULL argument");
    counts[0] = m->frames_full;
    counts[1] = m->frames_resolve_only;
    return ALP_OK;
}

int alp_mesh_set_valid(alp_mesh_t *m, const uint8_t *valid) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m, "mesh handle is NULL");
    m->vis_current = false;
    if (m->valid_derived) return apply_derived_mask(m, valid);   // filtered grid: its own mask stays in force
    if (!valid) {
        if (m->valid) hipFree(m->valid);
        m->valid = nullptr;
        return ALP_OK;
    }
    if (!m->valid) ALP_HIP(hipMalloc((void **)&m->valid, (size_t)m->n_vert));
    return upload_chunked(m->valid, valid, (size_t)m->n_vert);
}

int alp_mesh_from_rasters(const void *dsm, int dsm_dtype, int64_t rows, int64_t cols, const double transform[6],
                          double z_max, const void *aerial, int aerial_dtype, double color_div,
                          const uint8_t *nodata, double offsets_out[3], alp_mesh_t **out) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(out, "out is NULL");
    *out = nullptr;
    ALP_REQUIRE(dsm && aerial && transform && offsets_out, "NULL argument");
    ALP_REQUIRE(dsm_dtype == ALP_F32 || dsm_dtype == ALP_F64, "dsm_dtype must be ALP_F32 or ALP_F64");
    ALP_REQUIRE(aerial_dtype == ALP_F32 || aerial_dtype == ALP_U8 || aerial_dtype == ALP_U16,
                "aerial_dtype must be ALP_U8, ALP_U16 or ALP_F32");
    ALP_REQUIRE(rows >= 2 && cols >= 2, "the raster needs at least 2 x 2 cells");
    ALP_REQUIRE(rows * cols < ((int64_t)1 << 31), "more than 2^31 vertices");
    ALP_REQUIRE(z_max >= 0, "z_max is negative");
    const int64_t n = rows * cols;
    alp_mesh *m = new alp_mesh();
    m->n_vert = n;
    m->grid_h = rows;
    m->grid_w = cols;
    m->n_tri = 2 * (rows - 1) * (cols - 1);
    m->implicit = true;
    int rc = ALP_OK;
    void *dsm_dev = nullptr, *aer_dev = nullptr;
    unsigned char *nod_dev = nullptr;
    unsigned long long *zmin_dev = nullptr;
    auto bail = [&](int code) {
        for (void *p : {dsm_dev, aer_dev, (void *)nod_dev, (void *)zmin_dev})
            if (p) hipFree(p);
        alp_mesh_destroy(m);
        return code;
    };
    const size_t zsize = dsm_dtype == ALP_F32 ? 4 : 8;
    const size_t asize = aerial_dtype == ALP_U8 ? 1 : aerial_dtype == ALP_U16 ? 2 : 4;
    if (hipMalloc(&dsm_dev, (size_t)n * zsize) != hipSuccess || hipMalloc(&aer_dev, (size_t)n * 3 * asize) != hipSuccess ||
        hipMalloc((void **)&zmin_dev, 8) != hipSuccess || hipMalloc((void **)&m->vert, (size_t)n * 12) != hipSuccess ||
        hipMalloc((void **)&m->value, (size_t)n * 12) != hipSuccess || hipMalloc((void **)&m->valid, (size_t)n) != hipSuccess ||
        (nodata && hipMalloc((void **)&nod_dev, (size_t)n) != hipSuccess))
        return bail(fail(ALP_EHIP, "alp_mesh_from_rasters: hipMalloc"));
    if ((rc = upload_chunked(dsm_dev, dsm, (size_t)n * zsize))) return bail(rc);
    if ((rc = upload_chunked(aer_dev, aerial, (size_t)n * 3 * asize))) return bail(rc);
    if (nodata && (rc = upload_chunked(nod_dev, nodata, (size_t)n))) return bail(rc);
    hipStream_t st = ctx().stream;
    // offsets = vert.min(axis=0) (surface.py:211): x and y from the two coordinate vectors on the
    // host (same float64 mul + add), z by a device reduction
    double ox = __builtin_inf(), oy = __builtin_inf();
    for (int64_t c = 0; c < cols; ++c) {
        const double x = (double)c * transform[0] + transform[2];
        ox = x < ox ? x : ox;
    }
    for (int64_t r = 0; r < rows; ++r) {
        const double y = (double)r * transform[4] + transform[5];
        oy = y < oy ? y : oy;
    }
    const unsigned long long inf_bits = 0x7FF0000000000000ull;
    hipError_t e = hipMemcpyAsync(zmin_dev, &inf_bits, 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        KTimeScope kt;
        const dim3 grid((unsigned)(ctx().cu_count * 4));
        if (dsm_dtype == ALP_F32)
            hipLaunchKernelGGL(surface_zmin_kernel<float>, grid, dim3(256), 0, st, (const float *)dsm_dev, (long long)n, z_max, zmin_dev);
        else
            hipLaunchKernelGGL(surface_zmin_kernel<double>, grid, dim3(256), 0, st, (const double *)dsm_dev, (long long)n, z_max, zmin_dev);
        e = hipGetLastError();
    }
    double oz = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&oz, zmin_dev, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return bail(fail(ALP_EHIP, "alp_mesh_from_rasters: %s", hipGetErrorString(e)));
    // (measured, not kept, round 4: four vertices per thread with 16-byte loads and whole-vector stores -- 0.98 against 0.73 ms
    // for zmin + build at 100 M vertices: a quarter of the threads, each with twelve float64 divisions in flight, hide less)
    const dim3 grid((unsigned)((n + 255) / 256));
    ktime_begin();
#define ALP_BUILD(Z, A)                                                                                              \
    hipLaunchKernelGGL((surface_build_kernel<Z, A>), grid, dim3(256), 0, st, (const Z *)dsm_dev, (const A *)aer_dev, \
                       (const unsigned char *)nod_dev, (long long)rows, (long long)cols, transform[0], transform[2], \
                       transform[4], transform[5], z_max, color_div, ox, oz, oy, m->vert, m->value, m->valid)
    if (dsm_dtype == ALP_F32) {
        if (aerial_dtype == ALP_U8) ALP_BUILD(float, unsigned char);
        else if (aerial_dtype == ALP_U16) ALP_BUILD(float, unsigned short);
        else ALP_BUILD(float, float);
    } else {
        if (aerial_dtype == ALP_U8) ALP_BUILD(double, unsigned char);
        else if (aerial_dtype == ALP_U16) ALP_BUILD(double, unsigned short);
        else ALP_BUILD(double, float);
    }
#undef ALP_BUILD
    ktime_end();
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return bail(fail(ALP_EHIP, "alp_mesh_from_rasters: %s", hipGetErrorString(e)));
    if (!nodata) {                           // nothing masked: plain grid
        hipFree(m->valid);
        m->valid = nullptr;
    }
    for (void *p : {dsm_dev, aer_dev, (void *)nod_dev, (void *)zmin_dev})
        if (p) hipFree(p);
    dsm_dev = aer_dev = nullptr;
    nod_dev = nullptr;
    zmin_dev = nullptr;
    if (hipMalloc((void **)&m->qcount_dev, QC_TOTAL * sizeof(unsigned)) != hipSuccess ||
        hipHostMalloc((void **)&m->qcount_host, QC_TOTAL * sizeof(unsigned), hipHostMallocDefault) != hipSuccess)
        return bail(fail(ALP_EHIP, "hipMalloc queue counter"));
    if ((rc = ensure_queue(m, initial_queue_cap()))) return bail(rc);
    if ((rc = ensure_gqueue(m, initial_queue_cap()))) return bail(rc);
    offsets_out[0] = ox;                     // X, Z, Y like `vert` (surface.py:189)
    offsets_out[1] = oz;
    offsets_out[2] = oy;
    *out = m;
    return ALP_OK;
}

int alp_mesh_fetch(alp_mesh_t *m, float *vert, float *value, uint8_t *valid) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(m, "mesh handle is NULL");
    hipStream_t st = ctx().stream;
    if (vert) ALP_HIP(hipMemcpyAsync(vert, m->vert, (size_t)m->n_vert * 12, hipMemcpyDeviceToHost, st));
    if (value) ALP_HIP(hipMemcpyAsync(value, m->value ? m->value : m->vert, (size_t)m->n_vert * 12, hipMemcpyDeviceToHost, st));
    if (valid) {
        if (m->valid) ALP_HIP(hipMemcpyAsync(valid, m->valid, (size_t)m->n_vert, hipMemcpyDeviceToHost, st));
        else memset(valid, 1, (size_t)m->n_vert);
    }
    ALP_HIP(hipStreamSynchronize(st));
    return ALP_OK;
}

}  // extern "C"
