// Shared between the render translation units (alp_raster.hip: rasterisation, resolve, frame
// post-processing; alp_mesh.hip: mesh construction from rasters and mesh-level accessors).
#pragma once
#include "alp_internal.h"

namespace alp {

struct View {
    float R[3][3];
    float camf[3], caml[3];
    float fx, fy, sx, sy;
    int w, h;
    double fxd, fyd;
    double kx, ky, ifx, ify;     // 1/sx, 1/sy, 1/fx, 1/fy in float64 (ray of a pixel centre, resolve_kernel)
    double Rd[3][3], camd[3];    // float64 rotation and camera position: the interpolation of resolve_kernel
};

struct RemapCoef {     // inverted coefficients of project.py:136-137, float64
    double a1, a2, k1, k2, k3, k4, k5, k6, p1, p2, s1, s2, s3, s4, c0, c1;
};

struct WorkItem { unsigned tri; unsigned short sub, tx, ty, pad; };   // sub: fan triangle 0/1

// A snapped window-space triangle set aside ("parked") for a cooperative rasterisation: by the rest of its
// wave (raster_kernel: coop_drain) or, from raster_grid_kernel, through the device queues of
// raster_coop4_kernel / raster_coop_kernel.
struct Deferred {
    int X[3], Y[3];
    float iw[3];
    unsigned t;
};

// A whole grid cell parked by raster_grid_kernel for raster_cell_kernel: the snapped corners a, b, c, d
// (vertex ids a, a + gw, a + gw + 1, a + 1: triangles (a, b, c) and (a, c, d), surface.py:194-201).
struct alignas(16) ParkedCell {
    int X[4], Y[4];
    float iw[4];
    unsigned cell;
    unsigned pad[3];
};

// What ONE tile (workgroup) of raster_grid_kernel parked, for raster_parked_tiles_kernel: its contiguous ranges in the
// three queues and the pixel box (clamped to the viewport) of everything in them.  The consumer covers that box with
// PT_BIN x PT_BIN-pixel LDS depth patches; a unit = (tile record, slot): the bins slot, slot + nslots, ... of the tile.
struct ParkedTile {
    unsigned base[3];          // first entry: small triangles, large triangles, cells
    unsigned n[3];
    int i0, j0, i1, j1;
    unsigned pad[2];
};
struct ParkedUnit {
    unsigned rec;
    unsigned short slot, nslots;
};
constexpr int PT_BIN = 64;            // 4096 words of 8 bytes = 32 KB of LDS per workgroup
constexpr int PT_MAX_UNITS = 32;      // a tile's bins are dealt to at most this many units (the nearest tiles cover 100+ bins)

}  // namespace alp

struct alp_mesh {
    int64_t n_vert = 0, n_tri = 0, grid_h = 0, grid_w = 0;
    bool implicit = false;
    float *vert = nullptr, *value = nullptr;
    int *ind = nullptr;
    unsigned char *valid = nullptr;    // optional, per vertex: 0 = nodata, its triangles are not drawn
    float ind_sharing = -1.0f;         // INDEX_LDS_LAB builds only: distinct vertices / references per block of 256 triangles
    // A filtered index array of the regular grid (surface.py:203-205: the triangles of nodata vertices removed)
    // recognised at creation: rendered as the implicit grid with the vertex mask it implies; these map the
    // grid's triangle ids back to positions in the caller's array (alp_render_fetch_visibility)
    unsigned char *valid_derived = nullptr;   // the mask implied by the index array (valid = derived AND the caller's)
    unsigned *tri_present = nullptr;          // bit per grid triangle: present in the caller's array
    unsigned *tri_rank = nullptr;             // per 32-bit word of tri_present: number of present triangles before it
    float *tile_bounds = nullptr;      // implicit grid: bounding box (centre, half extent) per raster_grid_kernel tile
    unsigned *tile_lists = nullptr;    // implicit grid: near / far / surviving-far tile ids of the current frame + counters
    unsigned *hiz = nullptr;           // depth pyramid of the current frame size (levels 8 .. 256 px)
    bool coords_as_value = false;      // render the vertices themselves (reverse_proj) although values are stored
    // per-render state (sized on first use)
    int w = 0, h = 0;
    unsigned long long *vis = nullptr;
    float *image = nullptr;
    alp::WorkItem *queue = nullptr;
    unsigned qcap = 0;
    unsigned *gqueue = nullptr;        // general queue: triangle ids set aside by raster_grid_kernel
    unsigned gcap = 0;
    alp::Deferred *park_small = nullptr, *park_large = nullptr;   // implicit grid: parked triangles (one allocation)
    alp::ParkedCell *park_cell = nullptr;                         // implicit grid: parked cells
    unsigned park_cap[3] = {0, 0, 0};                             // small, large, cells: first round
    unsigned park_cap_b[3] = {0, 0, 0};                           // second round (its entries follow the first round's)
    alp::ParkedTile *park_tiles = nullptr;                        // first round: one record per tile that parked something
    alp::ParkedUnit *park_units = nullptr;
    unsigned park_tiles_cap = 0, park_units_cap = 0;
    // per round (2 rounds x QC_STRIDE) [0] work items, [1] general entries, [2] small parked, [3] large parked,
    // [4] parked cells; then the three tile-list lengths of the frame plan
    unsigned *qcount_dev = nullptr;
    unsigned *qcount_host = nullptr;   // pinned copy of the queue counters of the last frame
    bool unchecked = false;            // last frame enqueued, its queue counters not yet checked (finish_frame)
    alp::View last_v;
    alp::RemapCoef last_rc;
    double last_min_distance = 0;
    bool rendered = false;
    // Visibility cache: `vis` holds the finished visibility buffer of view `last_v` for the mesh's current vertices
    // and mask.  A render with the same view then runs the resolve alone (new value source, lens coefficients or
    // min_distance: the reference's sim_image + reverse_proj pair at one pose, example.py:28,31).  Cleared by
    // everything that changes what a raster pass would draw (alp_mesh_set_valid, alp_render_load, a new frame size).
    bool vis_current = false;
    int64_t frames_full = 0, frames_resolve_only = 0;
    // reverse_proj compaction scratch
    unsigned *compact_counts = nullptr;
    unsigned long long *compact_offsets = nullptr;
    int compact_cap = 0;
    float valid_span[4] = {0, 0, 0, 0};      // frame_valid_count: min / max of channel 0, min / max of channel 2 over the pixels that see the surface
    int64_t valid_total = -1;
    int64_t valid_total_planes = 0;     // plane length of the planar form of frame_valid_write
    // alp_render_rasterize_plan -> alp_render_rasterize: the compacted points of the current frame (device):
    // x[M] | y[M] float64, then the pixel index idx[M] uint32
    char *rz_points = nullptr;
    size_t rz_cap = 0;
    int64_t rz_n = -1;                 // -1: no plan for the current frame
    // work area of alp_render_rasterize (values, accumulators, raster, the caller's image): kept between calls, grow-only --
    // 5.3 GB for the 100 M-vertex frame at 1 m; allocating and freeing it per call cost 2 ms, and now and then 0.25-0.4 s
    char *rz_work = nullptr;
    size_t rz_work_cap = 0;
    // HIP events around the launches of the last alp_render_enqueue (alp_mesh_frame_ms); created on first use
    hipEvent_t ev_frame[2] = {nullptr, nullptr};
};

namespace alp {
// defined in alp_raster.hip
int upload_chunked(void *dst, const void *src, size_t bytes);
int upload_f32(float *dst, const void *src, int dtype, int64_t n_vert);   // n x 3 float32 / float64 host -> float32 device
int ensure_queue(alp_mesh *m, unsigned cap);
int ensure_gqueue(alp_mesh *m, unsigned cap);
int ensure_park(alp_mesh *m, unsigned cap_small, unsigned cap_large, unsigned cap_cell);
constexpr int QC_STRIDE = 8;           // counters per round
constexpr int QC_TOTAL = 2 * QC_STRIDE + 8;   // two rounds of queue counters, three tile-list counters (+1), the FAR tiles' screen region (4)
unsigned initial_queue_cap();
int frame_valid_count(alp_mesh *m, int64_t *count);
int frame_valid_write(alp_mesh *m, const double *offsets, unsigned *idx_dev, double *xyz_dev, bool planar);
// valid = (mask implied by a filtered grid index array) AND `user` (host, n_vert bytes; NULL = all ones)
int apply_derived_mask(alp_mesh *m, const unsigned char *user);
}  // namespace alp
