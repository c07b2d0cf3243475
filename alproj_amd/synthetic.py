"""Synthetic DSM, camera and GCP workloads; used by tests, smoke
and bench.py.  Pure numpy, no device work.

The surface mimics the layout produced by the reference's ``get_colored_surface``
(src/alproj/surface.py:179-211): a square grid of ``n x n`` vertices, row 0 = north, vertex
id = row * n + col, ``vert`` in X, Z(up), Y order relative to ``offsets = vert.min(0)``, and
two triangles per cell ``(a, a+n, a+n+1), (a, a+n+1, a+1)``.
"""
import math

import numpy as np

ABS_ORIGIN_XZY = np.array([732000.0, 2000.0, 4048000.0])   # X, Z, Y like `offsets`
SEED = 20260220

BASE_CAMERA = dict(fov=75.0, pan=95.0, tilt=0.0, roll=0.0, a1=1.0, a2=1.0,
                   k1=0.0, k2=0.0, k3=0.0, k4=0.0, k5=0.0, k6=0.0, p1=0.0, p2=0.0,
                   s1=0.0, s2=0.0, s3=0.0, s4=0.0, w=5616, h=3744, cx=2808.0, cy=1872.0)

TARGETS_D9 = ["x", "y", "z", "fov", "pan", "tilt", "roll", "a1", "a2"]
TARGETS_D21 = TARGETS_D9 + ["k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "s1", "s2", "s3", "s4"]


def grid_side(n_vertices):
    return int(math.ceil(math.sqrt(n_vertices)))


def _elevation(X, Y):
    return 300.0 * np.sin(X / 700.0) * np.cos(Y / 500.0) + 40.0 * np.sin(X / 37.0) * np.sin(Y / 53.0)


def row_noise(n_side, row, seed=SEED, out=None):
    """The N(0, 0.5^2) elevation noise of ONE grid row (float32, n_side values): a counter-based
    Philox stream keyed by (seed, row), so that any rank generates exactly its own rows -- the same
    values whatever the sharding -- without drawing the other ranks' share of the field."""
    bits = np.random.Philox(key=[seed, 0], counter=[0, int(row), 0, 0])
    out = np.random.Generator(bits).standard_normal(n_side, dtype=np.float32, out=out)
    out *= np.float32(0.5)
    return out


def dsm_rows(n_side, row0, row1, res=1.0, seed=SEED):
    """Vertices of grid rows [row0, row1): float32 (count, 3) in X, Z, Y order relative to
    the surface minimum being (0, zmin, 0) -- i.e. X = col*res, Y = (n-1-row)*res, Z raw."""
    cols = np.arange(n_side, dtype=np.float64) * res
    cols32 = cols.astype(np.float32)
    out = np.empty(((row1 - row0) * n_side, 3), dtype=np.float32)
    step = max(1, (1 << 22) // n_side)
    noise = np.empty((step, n_side), dtype=np.float32)
    for r in range(row0, row1, step):
        r2 = min(row1, r + step)
        Y = ((n_side - 1 - np.arange(r, r2, dtype=np.float64)) * res)[:, None]
        for k, rr in enumerate(range(r, r2)):
            row_noise(n_side, rr, seed, out=noise[k])
        # _elevation(cols, Y), separably: the same floating-point operations in the same order, with the
        # sines / cosines evaluated once per column and row instead of once per vertex
        Z = (300.0 * np.sin(cols / 700.0))[None, :] * np.cos(Y / 500.0) + (40.0 * np.sin(cols / 37.0))[None, :] * np.sin(Y / 53.0)
        Z += noise[:r2 - r]
        blk = out[(r - row0) * n_side:(r2 - row0) * n_side].reshape(r2 - r, n_side, 3)
        blk[:, :, 0] = cols32[None, :]
        blk[:, :, 1] = Z
        blk[:, :, 2] = Y
    return out


def z_floor(n_side, res=1.0):
    """Lower bound used as the Z offset (the analytic minimum minus noise head-room), so
    that shards generated independently agree on `offsets` without a global min pass."""
    return -345.0


def surface(n_side, res=1.0, seed=SEED, rows=None):
    """-> dict(vert (N,3) f32 X,Z,Y relative to offsets, offsets (3,) f64, n_side, row0, row1)."""
    row0, row1 = (0, n_side) if rows is None else rows
    v = dsm_rows(n_side, row0, row1, res, seed)
    zf = z_floor(n_side, res)
    v[:, 1] -= np.float32(zf)
    offsets = ABS_ORIGIN_XZY + np.array([0.0, zf, 0.0])
    return dict(vert=v, offsets=offsets, n_side=n_side, row0=row0, row1=row1, res=res)


def grid_indices(n_side, dtype=np.int64):
    """Triangle index array of the full regular grid (surface.py:194-201 on a square grid), built in
    `dtype` band by band (the int64 form of a 10000 x 10000 grid is 4.8 GB)."""
    m = n_side - 1
    out = np.empty((2 * m * m, 3), dtype=dtype)
    view = out.reshape(m, m, 6)
    cols = np.arange(m, dtype=dtype)
    band = max(1, (1 << 22) // max(m, 1))
    offs = np.array([0, n_side, n_side + 1, 0, n_side + 1, 1], dtype=dtype)
    for r in range(0, m, band):
        r2 = min(m, r + band)
        a = cols[None, :] + (np.arange(r, r2, dtype=dtype) * dtype(n_side))[:, None]
        view[r:r2] = a[:, :, None] + offs[None, None, :]
    return out


def colors(n_vertices, seed=SEED):
    return np.random.default_rng(seed + 1).random((n_vertices, 3)).astype(np.float32)


def base_params(n_side, res=1.0):
    """Camera at the centre of the grid's west edge, 50 m above the local terrain, absolute
    coordinates (x = easting, y = northing, z = elevation)."""
    xl = 0.0
    yl = (n_side - 1) * res / 2.0
    zl = float(_elevation(np.float64(xl), np.float64(yl))) + 50.0
    p = dict(BASE_CAMERA)
    p.update(x=ABS_ORIGIN_XZY[0] + xl, y=ABS_ORIGIN_XZY[2] + yl, z=ABS_ORIGIN_XZY[1] + zl)
    return p


def standoff_params(n_side, res=1.0):
    """Camera for the point-set workloads (projection / CMA-ES over EVERY DSM vertex): on the
    grid's east-west centre line, 0.65 grid lengths west of the west edge, 600 m above the zero
    plane.  From there the whole surface lies in front of the camera (depth >= 0.6 L) and
    almost all of it inside the 75-degree image, so every vertex is a well-conditioned GCP.

    (With the camera ON the west edge -- base_params, the render workload -- thousands of
    vertices sit next to the camera plane; the reference's un-culled projection
    (optimize.py:146-149, quirk Q7) turns them into 1e6..1e40-pixel residuals that swamp any
    loss, in float64 as in float32.)"""
    L = n_side * res
    p = dict(BASE_CAMERA)
    p.update(x=ABS_ORIGIN_XZY[0] - 0.65 * L, y=ABS_ORIGIN_XZY[2] + (n_side - 1) * res / 2.0,
             z=ABS_ORIGIN_XZY[1] + 600.0)
    return p


def perturbed(p):
    """Ground-truth pose = p + the offsets of SURVEY.md 8(d)."""
    q = dict(p)
    q.update(x=p["x"] + 5, y=p["y"] - 7, z=p["z"] + 3, fov=p["fov"] - 4, pan=p["pan"] + 3,
             tilt=p["tilt"] + 2, roll=p["roll"] - 1, a1=1.02, a2=0.98, k1=-0.05, k2=0.01,
             p1=1e-3, p2=-2e-3)
    return q


def truth_params(n_side, res=1.0):
    p = base_params(n_side, res)
    p.update(x=p["x"] + 5, y=p["y"] - 7, z=p["z"] + 3, fov=p["fov"] - 4, pan=p["pan"] + 3,
             tilt=p["tilt"] + 2, roll=p["roll"] - 1, a1=1.02, a2=0.98, k1=-0.05, k2=0.01,
             p1=1e-3, p2=-2e-3)
    return p


def vert_to_xyz_abs(vert, offsets):
    """(N,3) X,Z,Y relative -> (N,3) float64 absolute x, y, z (easting, northing, elevation)."""
    v = vert.astype(np.float64)
    return np.stack([v[:, 0] + offsets[0], v[:, 2] + offsets[2], v[:, 1] + offsets[1]], axis=1)


def vert_to_xyz_local(vert):
    """(N,3) f32 X,Z,Y -> (N,3) f32 x, y, z in the frame of `offsets` (no precision loss)."""
    return np.ascontiguousarray(vert[:, [0, 2, 1]])


def local_params(params, offsets):
    """Camera parameters with the position expressed in the frame of `offsets`
    (what persp_proj does at reference project.py:204-207)."""
    p = dict(params)
    p["x"] = params["x"] - offsets[0]
    p["y"] = params["y"] - offsets[2]
    p["z"] = params["z"] - offsets[1]
    return p


def gcp_points(n, params, seed=1, depth=(80.0, 4000.0), margin=0.05):
    """n well-conditioned ground-control-like points (float64, absolute coordinates): random
    pixels of the pinhole image of ``params`` back-projected to random depths.  Everything is
    in front of the camera and inside the image, as produced by the reference's set_gcp.
    Observed pixel coordinates are NOT produced here (they need a projection: the tests use
    the oracle, bench.py the device)."""
    from math import cos, pi, sin, tan
    rng = np.random.default_rng(seed)
    w, h = params["w"], params["h"]
    u = rng.uniform(margin * w, (1 - margin) * w, n)
    v = rng.uniform(margin * h, (1 - margin) * h, n)
    zc = -rng.uniform(depth[0], depth[1], n)     # the reference's camera looks down -Z_cam
    fov_x = params["fov"] * pi / 180
    fx = w / (2 * tan(fov_x / 2))
    fy = h / (2 * tan(fov_x * h / w / 2))
    xc = (w - u - params["cx"]) / fx * zc       # u = w - (fx X/Z + cx)
    yc = (v - params["cy"]) / fy * zc
    a, b, c = params["pan"] * pi / 180, -(params["tilt"] + 90) * pi / 180, -params["roll"] * pi / 180
    rz = np.array([[cos(a), -sin(a), 0], [sin(a), cos(a), 0], [0, 0, 1.0]])
    rx = np.array([[1.0, 0, 0], [0, cos(b), -sin(b)], [0, sin(b), cos(b)]])
    ry = np.array([[cos(c), 0, sin(c)], [0, 1.0, 0], [-sin(c), 0, cos(c)]])
    rot = rx @ ry @ rz
    cam = np.stack([xc, yc, zc], axis=0)
    return (rot.T @ cam).T + np.array([params["x"], params["y"], params["z"]])
