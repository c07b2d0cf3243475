"""The view / projection scalars of a renderer (CPU raster oracle, ray caster, HIP path) against the reference's OWN
matrices: g7 holds ``projection_mat`` / ``modelview_mat`` (project.py:13-109) as the reference computed them for 8 poses
(tests/golden/gen_golden.py).  Nothing can read a renderer's internal view struct, so the check goes through a picture:
one triangle placed at chosen VIEW-space positions -- world vertex = camera + R^T v with R and the camera taken from
g7's modelview matrix -- is rendered, and every pixel is compared with what the reference's matrices, used as OpenGL uses
them (column-major upload, project.py:262-263: clip = proj^T-as-uploaded . view . (v, 1)), say about it: covered iff the
centre is inside the projected triangle (pixels within 0.01 px of an edge are left out), view depth = the window-space
interpolation of clip w."""
import numpy as np

from oracle import ref_numpy as orc

OFFSETS_XZY = np.array([732000.0, 2000.0, 4048000.0])     # g7's cam_offset (x, y, z) in the X, Z, Y order of `offsets`
SHRINK = 8                                               # frames of w / 8 x h / 8: the same aspect, hence the same projection matrix


def scene(g, k):
    """-> (vert (3,3) float32 X,Z,Y relative to OFFSETS_XZY, ind, params dict, expected dict)"""
    p = orc.vector_to_params(g["params"][k])
    assert p["w"] % SHRINK == 0 and p["h"] % SHRINK == 0
    W, H = int(p["w"]) // SHRINK, int(p["h"]) // SHRINK
    p.update(w=W, h=H, cx=W / 2.0, cy=H / 2.0)
    mv = g["view"][k].reshape(4, 4).T                  # as a column-major GLSL mat4 reads the flat array
    pm = g["proj"][k].reshape(4, 4).T
    R, t = mv[:3, :3], mv[:3, 3]
    cam = -R.T @ t                                     # X, Z, Y relative to the offsets
    tx, ty = np.tan(np.radians(p["fov"]) / 2), np.tan(np.radians(p["fov"]) * H / W / 2)
    d = 35.0 + 7.0 * k
    v = np.array([[-0.62 * tx * d, -0.55 * ty * d, d], [0.70 * tx * 1.3 * d, -0.40 * ty * 1.3 * d, 1.3 * d],
                  [0.08 * tx * 0.8 * d, 0.66 * ty * 0.8 * d, 0.8 * d]])
    vert = (cam + v @ R).astype(np.float32)            # R^T v, row-wise
    clip = (pm @ mv @ np.c_[vert.astype(np.float64), np.ones(3)].T).T          # (3, 4)
    win = (clip[:, :2] / clip[:, 3:4] + 1) / 2 * np.array([W, H])
    return vert, np.array([[0, 1, 2]], dtype=np.int32), p, dict(win=win, w=clip[:, 3], W=W, H=H)


def expected(exp, margin=0.01):
    """-> (inside (H, W) bool, outside (H, W) bool, depth (H, W) float64) in GL window orientation"""
    jj, ii = np.mgrid[0:exp["H"], 0:exp["W"]]
    px, py = ii + 0.5, jj + 0.5
    (x0, y0), (x1, y1), (x2, y2) = exp["win"]

    def edge(ax, ay, bx, by):
        e = (bx - ax) * (py - ay) - (by - ay) * (px - ax)
        return e, e / np.hypot(bx - ax, by - ay)

    e0, d0 = edge(x1, y1, x2, y2)
    e1, d1 = edge(x2, y2, x0, y0)
    e2, d2 = edge(x0, y0, x1, y1)
    area = e0 + e1 + e2
    assert area.min() > 0                              # counter-clockwise in the window: front-facing
    dist = np.minimum(np.minimum(d0, d1), d2)
    inv_w = (e0 / exp["w"][0] + e1 / exp["w"][1] + e2 / exp["w"][2]) / area
    return dist > margin, dist < -margin, 1.0 / inv_w


def check(tri, depth, exp, depth_rtol):
    """tri (H, W): 0 where the triangle is seen, -1 elsewhere; depth (H, W): view depth there"""
    inside, outside, want = expected(exp)
    assert inside.sum() > 5000
    assert (tri[inside] == 0).all(), f"{int((tri[inside] != 0).sum())} pixels inside the reference's triangle are not covered"
    assert (tri[outside] == -1).all(), f"{int((tri[outside] != -1).sum())} pixels outside the reference's triangle are covered"
    err = np.abs(depth[inside] - want[inside]) / want[inside]
    assert err.max() < depth_rtol, err.max()
    return float(err.max())
