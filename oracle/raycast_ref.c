/*
 * raycast_ref.c -- float64 brute-force ray caster: an IMPLEMENTATION-INDEPENDENT check of the
 * depth-buffered render (the reference's persp_proj(), src/alproj/project.py:145-294).
 *
 * TEST INFRASTRUCTURE ONLY (tests/ and nothing else).  Nothing under alproj_amd/ links,
 * includes or calls this file, and this file shares nothing with oracle/raster_ref.c: no
 * sub-pixel snapping, no integer edge functions, no tie rule, no float32 depth expression, no
 * clipping code.  It answers one question per pixel centre, in float64: which front-facing
 * triangle does the ray through the centre hit first, at what depth, with what interpolated
 * value -- what ANY conformant OpenGL rasteriser must produce away from triangle edges and away
 * from depth ties.  For every pixel it also reports how far the centre is from the nearest
 * projected triangle edge and the depth of the second-nearest hit, so that a test can restrict
 * itself to the pixels where the answer does not depend on implementation-defined rules.
 *
 * Geometry restated from the reference:
 *   project.py:203-207   camera position minus offsets (X,Z,Y order, quirk Q16)
 *   project.py:81-109    modelview: view = Rz(roll) Rx(tilt) Ry(360-pan) T(-x,-z,-y)
 *   project.py:13-54,257,262   projection matrix built WITHOUT cx,cy and uploaded untransposed
 *                        => clip = (fx vx, fy vy, -1, vz), ndc = (fx vx/vz, fy vy/vz): the ray of
 *                        ndc (X, Y) is view-space direction (X/fx, Y/fy, 1); clip-space
 *                        -w <= z <= w  <=>  vz >= 1 (near plane at 1, no far plane; Q10, Q11)
 *   project.py:211-212   depth test (nearest wins), back faces culled, CCW front
 *   project.py:217-253   varyings interpolated perspective-correctly = barycentric at the 3-D hit
 *   project.py:269-281   window pixel (i, j), centre (i+0.5, j+0.5), j = 0 at the bottom
 * Vertex data are the float32 values the reference uploads (project.py:213-214), promoted to
 * float64; everything else is float64 from the parameters.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double R[3][3], cam[3];
    double fx, fy;
    int w, h;
} rc_view;

static void rc_setup(const double *p, const double *offsets, rc_view *v) {
    double x = p[0], y = p[1], z = p[2];
    if (offsets) { x -= offsets[0]; y -= offsets[2]; z -= offsets[1]; }
    const double deg = M_PI / 180.0;
    const double pan = (360.0 - p[4]) * deg, tilt = p[5] * deg, roll = p[6] * deg;
    const double cp = cos(pan), sp = sin(pan), ct = cos(tilt), st = sin(tilt), cr = cos(roll), sr = sin(roll);
    /* Rz(roll) * Rx(tilt) * Ry(pan), multiplied out by hand */
    const double rxry[3][3] = {{cp, 0, sp}, {st * sp, ct, -st * cp}, {-ct * sp, st, ct * cp}};
    for (int j = 0; j < 3; ++j) {
        v->R[0][j] = cr * rxry[0][j] - sr * rxry[1][j];
        v->R[1][j] = sr * rxry[0][j] + cr * rxry[1][j];
        v->R[2][j] = rxry[2][j];
    }
    v->cam[0] = x; v->cam[1] = z; v->cam[2] = y;      /* vertices are X, Z(up), Y */
    const double fov_x = p[3] * deg, fov_y = fov_x * p[22] / p[21];
    v->fx = 1.0 / tan(fov_x / 2);
    v->fy = 1.0 / tan(fov_y / 2);
    v->w = (int)p[21];
    v->h = (int)p[22];
}

static void rc_to_view(const rc_view *v, const float *p, double out[3]) {
    const double d[3] = {(double)p[0] - v->cam[0], (double)p[1] - v->cam[1], (double)p[2] - v->cam[2]};
    for (int i = 0; i < 3; ++i) out[i] = v->R[i][0] * d[0] + v->R[i][1] * d[1] + v->R[i][2] * d[2];
}

/* window coordinates (pixels) of a view-space point with vz > 0 */
static void rc_window(const rc_view *v, const double q[3], double *xw, double *yw) {
    *xw = (v->fx * q[0] / q[2] + 1.0) * 0.5 * v->w;
    *yw = (v->fy * q[1] / q[2] + 1.0) * 0.5 * v->h;
}

static void rc_tri(const void *ind, int ind_i64, int64_t gw, int64_t t, int64_t idx[3]) {
    if (ind) {
        for (int k = 0; k < 3; ++k)
            idx[k] = ind_i64 ? ((const int64_t *)ind)[3 * t + k] : (int64_t)((const int32_t *)ind)[3 * t + k];
    } else {                                    /* surface.py:194-201 */
        const int64_t cell = t >> 1, r = cell / (gw - 1), c = cell - r * (gw - 1), a = r * gw + c;
        idx[0] = a;
        idx[1] = (t & 1) ? a + gw + 1 : a + gw;
        idx[2] = (t & 1) ? a + 1 : a + gw + 1;
    }
}

static double seg_dist(double px, double py, double ax, double ay, double bx, double by) {
    const double dx = bx - ax, dy = by - ay, l2 = dx * dx + dy * dy;
    double s = l2 > 0 ? ((px - ax) * dx + (py - ay) * dy) / l2 : 0.0;
    s = s < 0 ? 0 : (s > 1 ? 1 : s);
    const double ex = px - (ax + s * dx), ey = py - (ay + s * dy);
    return sqrt(ex * ex + ey * ey);
}

/*
 * Outputs, all h x w in WINDOW orientation (row 0 = bottom):
 *   tri_out    index of the nearest front-facing triangle hit with vz >= 1, or -1
 *   depth_out  its view depth vz (inf if none);  depth2_out  vz of the second-nearest hit (inf)
 *   edge_out   distance in pixels from the centre to the nearest projected edge (part with
 *              vz >= 1) of ANY triangle, either facing (inf if none within reach)
 *   value_out  h x w x 3: value interpolated at the hit (0 if none)
 * `value` NULL means value == vert.
 */
int alp_raycast(const float *vert, const float *value, int64_t n_vert, const void *ind, int ind_i64,
                int64_t n_tri, int64_t grid_h, int64_t grid_w, const double *params, const double *offsets,
                int32_t *tri_out, double *depth_out, double *depth2_out, double *edge_out, double *value_out) {
    rc_view v;
    rc_setup(params, offsets, &v);
    (void)grid_h;
    const size_t npix = (size_t)v.w * v.h;
    for (size_t p = 0; p < npix; ++p) {
        tri_out[p] = -1;
        depth_out[p] = depth2_out[p] = edge_out[p] = INFINITY;
    }
    memset(value_out, 0, sizeof(double) * npix * 3);
    double *bary = (double *)malloc(sizeof(double) * npix * 2);
    if (!bary) return -2;
    const float *val = value ? value : vert;
    for (int64_t t = 0; t < n_tri; ++t) {
        int64_t id[3];
        rc_tri(ind, ind_i64, grid_w, t, id);
        for (int k = 0; k < 3; ++k)
            if (id[k] < 0 || id[k] >= n_vert) { free(bary); return -1; }
        double q[3][3];
        for (int k = 0; k < 3; ++k) rc_to_view(&v, vert + 3 * id[k], q[k]);
        /* the part of each edge in front of the near plane, in window space */
        double sx[3][2], sy[3][2];
        int has[3] = {0, 0, 0};
        double bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY;
        for (int k = 0; k < 3; ++k) {
            const double *a = q[k], *b = q[(k + 1) % 3];
            double pa[3] = {a[0], a[1], a[2]}, pb[3] = {b[0], b[1], b[2]};
            if (pa[2] < 1.0 && pb[2] < 1.0) continue;
            if (pa[2] < 1.0 || pb[2] < 1.0) {
                double *lo = pa[2] < 1.0 ? pa : pb, *hi = pa[2] < 1.0 ? pb : pa;
                const double s = (1.0 - lo[2]) / (hi[2] - lo[2]);
                for (int c = 0; c < 3; ++c) lo[c] = lo[c] + s * (hi[c] - lo[c]);
                lo[2] = 1.0;
            }
            rc_window(&v, pa, &sx[k][0], &sy[k][0]);
            rc_window(&v, pb, &sx[k][1], &sy[k][1]);
            has[k] = 1;
            for (int e = 0; e < 2; ++e) {
                if (sx[k][e] < bx0) bx0 = sx[k][e];
                if (sx[k][e] > bx1) bx1 = sx[k][e];
                if (sy[k][e] < by0) by0 = sy[k][e];
                if (sy[k][e] > by1) by1 = sy[k][e];
            }
        }
        if (!(has[0] || has[1] || has[2])) continue;          /* entirely behind the near plane */
        /* pixels whose centre can be within 1 px of the clipped outline (the outline bounds the
         * visible part of the triangle: its vertices are the polygon's vertices) */
        double fi0 = floor(bx0 - 1.5), fi1 = ceil(bx1 + 0.5), fj0 = floor(by0 - 1.5), fj1 = ceil(by1 + 0.5);
        if (fi0 < 0) fi0 = 0;
        if (fj0 < 0) fj0 = 0;
        if (fi1 > v.w - 1) fi1 = v.w - 1;
        if (fj1 > v.h - 1) fj1 = v.h - 1;
        if (!(fi0 <= fi1 && fj0 <= fj1)) continue;
        /* orientation: the projected triangle is counter-clockwise in the window iff the triple
         * product A . (B x C) of its view-space vertices is positive (projection divides by
         * positive depths and scales by positive factors) */
        const double *A = q[0], *B = q[1], *C = q[2];
        const double triple = A[0] * (B[1] * C[2] - B[2] * C[1]) - A[1] * (B[0] * C[2] - B[2] * C[0]) +
                              A[2] * (B[0] * C[1] - B[1] * C[0]);
        const double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
        for (int j = (int)fj0; j <= (int)fj1; ++j)
            for (int i = (int)fi0; i <= (int)fi1; ++i) {
                const size_t p = (size_t)j * v.w + i;
                const double pxc = i + 0.5, pyc = j + 0.5;
                for (int k = 0; k < 3; ++k)
                    if (has[k]) {
                        const double d = seg_dist(pxc, pyc, sx[k][0], sy[k][0], sx[k][1], sy[k][1]);
                        if (d < edge_out[p]) edge_out[p] = d;
                    }
                if (!(triple > 0)) continue;                  /* back face or edge-on: culled */
                /* ray o + s r, o = 0, r = (X/fx, Y/fy, 1): Moeller-Trumbore, s = view depth */
                const double r[3] = {(pxc / (0.5 * v.w) - 1.0) / v.fx, (pyc / (0.5 * v.h) - 1.0) / v.fy, 1.0};
                const double pv[3] = {r[1] * e2[2] - r[2] * e2[1], r[2] * e2[0] - r[0] * e2[2], r[0] * e2[1] - r[1] * e2[0]};
                const double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
                if (det == 0.0) continue;
                const double tv[3] = {-A[0], -A[1], -A[2]};
                const double bb = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) / det;
                const double qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
                const double cc = (r[0] * qv[0] + r[1] * qv[1] + r[2] * qv[2]) / det;
                const double s = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) / det;
                if (bb < 0 || cc < 0 || bb + cc > 1 || !(s >= 1.0)) continue;
                if (s < depth_out[p]) {
                    depth2_out[p] = depth_out[p];
                    depth_out[p] = s;
                    tri_out[p] = (int32_t)t;
                    bary[2 * p] = bb;
                    bary[2 * p + 1] = cc;
                } else if (s < depth2_out[p]) {
                    depth2_out[p] = s;
                }
            }
    }
    for (size_t p = 0; p < npix; ++p) {
        if (tri_out[p] < 0) continue;
        int64_t id[3];
        rc_tri(ind, ind_i64, grid_w, tri_out[p], id);
        const double bb = bary[2 * p], cc = bary[2 * p + 1], aa = 1.0 - bb - cc;
        for (int c = 0; c < 3; ++c)
            value_out[3 * p + c] = aa * val[3 * id[0] + c] + bb * val[3 * id[1] + c] + cc * val[3 * id[2] + c];
    }
    free(bary);
    return 0;
}
