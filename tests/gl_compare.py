"""Comparison of a render (CPU raster oracle or HIP path) with what the reference's own persp_proj produced
on a real OpenGL (tests/golden/g15_gl_render.npz, Mesa llvmpipe; see tests/golden/gen_golden_gl.py).

What a conformant GL is NOT free to differ in -- and what is asserted on every pixel the float64 ray
caster calls safe (centre > 1/128 px from every projected edge, first and second hit > 1e-4 apart in
relative depth): the triangle seen (GL's gl_PrimitiveID), background staying background, and the
interpolated value.  GL interpolates float32 attributes over the triangle whose vertices it snapped to its
1/256-px sub-pixel grid; at grazing angles that displacement is worth more than float32 rounding, so the
value tolerance is geometric: |ours - GL| must not exceed the spread of the EXACT perspective-correct
interpolation (float64 ray / plane intersection) over the +-1/128 px neighbourhood of the pixel centre,
plus 2e-6 of the largest vertex value (float32 plane equations).  The plain relative difference is
reported next to it."""
import os

import numpy as np

from oracle import raycast as oray
from oracle import ref_numpy as orc
from tests.render_scenes import IMAGE_STRIDE, PRIM_STRIDE

DEPTH24_STEPS = 3.0          # first and second hit closer than this in 24-bit window depth: GL may z-fight
G15 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g15_gl_render.npz")
G16 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g16_gl_c4_frame.npz")      # the c4-sized frame (gen_golden_gl_c4.py)
NO_LENS = dict(a1=1.0, a2=1.0, **{k: 0.0 for k in ("k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "s1", "s2", "s3", "s4")})


def triangles_of(scene):
    if scene["ind"] is not None:
        return np.asarray(scene["ind"], dtype=np.int64)
    from alproj_amd import synthetic as syn
    return syn.grid_indices(scene["grid"][0], np.int64)


def exact_values(scene, tri_ids, px, py):
    """Perspective-correct interpolation of the scene's values on triangles `tri_ids` at window positions
    (px, py) (pixel units, GL window orientation), float64: ray through the pixel against the triangle's
    plane in view space (project.py:217-237: view * vec4(v, 1), proj as uploaded -> x_ndc = fx vx / vz)."""
    p = dict(scene["params"])
    off = scene["offsets"]
    if off is not None:
        p["x"], p["y"], p["z"] = p["x"] - off[0], p["y"] - off[2], p["z"] - off[1]
    w, h = float(p["w"]), float(p["h"])
    mv = np.asarray(orc.modelview_mat(p["pan"], p["tilt"], p["roll"], p["x"], p["y"], p["z"]), dtype=np.float64).reshape(4, 4).T
    pm = np.asarray(orc.projection_mat(p["fov"], p["w"], p["h"]), dtype=np.float64)
    fx, fy = pm[0], pm[5]
    tris = triangles_of(scene)[tri_ids]                                   # (K, 3)
    v = scene["vert"].astype(np.float64)[tris]                            # (K, 3, 3)
    val = v if scene.get("value") is None else scene["value"].astype(np.float64)[tris]
    view = v @ mv[:3, :3].T + mv[:3, 3]                                   # (K, 3, 3)
    ray = np.stack([(2 * px / w - 1) / fx, (2 * py / h - 1) / fy, np.ones_like(px)], axis=-1)
    p0, e1, e2 = view[:, 0], view[:, 1] - view[:, 0], view[:, 2] - view[:, 0]
    nrm = np.cross(e1, e2)
    t = np.einsum("ij,ij->i", nrm, p0) / np.einsum("ij,ij->i", nrm, ray)
    q = ray * t[:, None] - p0
    d11, d12, d22 = (e1 * e1).sum(1), (e1 * e2).sum(1), (e2 * e2).sum(1)
    q1, q2 = (q * e1).sum(1), (q * e2).sum(1)
    den = d11 * d22 - d12 * d12
    b1, b2 = (d22 * q1 - d12 * q2) / den, (d11 * q2 - d12 * q1) / den
    return val[:, 0] * (1 - b1 - b2)[:, None] + val[:, 1] * b1[:, None] + val[:, 2] * b2[:, None]


def compare_with_gl(name, scene, g, tri, img, delta=1.0 / 128):
    """tri: (h, w) triangle index per pixel (window orientation, -1 = background) of the render under test;
    img: (h, w, 3) its image as persp_proj returns it (row 0 = top, identity lens).  -> dict of rates."""
    gl_tri = g[f"{name}_prim_id"].astype(np.int64)
    img = img[::-1]
    gl_img = g[f"{name}_image"][::-1]                  # back to window orientation
    stride = IMAGE_STRIDE.get(name, 1)
    pstride = PRIM_STRIDE.get(name, 1)
    known = np.ones(tri.shape, dtype=bool)             # pixels whose gl_PrimitiveID the fixture holds
    if pstride > 1:                                    # ... a lattice of the window only (the c4-sized frame)
        full = np.full(tri.shape, -2, dtype=np.int64)
        full[::pstride, ::pstride] = gl_tri
        gl_tri = full
        known[:] = False
        known[::pstride, ::pstride] = True
    have_value = np.ones(gl_tri.shape, dtype=bool)
    if stride > 1:                                     # the fixture holds GL's image on a sub-grid of the window only
        full = np.zeros(img.shape, dtype=np.float32)
        full[::stride, ::stride] = gl_img
        gl_img = full
        have_value[:] = False
        have_value[::stride, ::stride] = True
    p = dict(scene["params"], **NO_LENS)
    rc = oray.raycast(scene["vert"], scene.get("value"), scene["ind"], p, scene["offsets"], grid=scene["grid"])
    safe = oray.safe_mask(rc, depth24_steps=DEPTH24_STEPS) & known
    same = (tri == gl_tri) | ~known
    bad = safe & ~same
    assert not bad.any(), f"{name}: {int(bad.sum())} safe pixels show another triangle than OpenGL, first at {np.argwhere(bad)[0]}"
    assert (gl_tri[safe] == rc["tri"][safe]).all()     # and GL itself agrees with the ray caster there
    md = scene.get("min_distance")
    hit = safe & (tri >= 0) & have_value
    jj, ii = np.nonzero(hit)
    ids = tri[hit]
    cx, cy = ii + 0.5, jj + 0.5
    ex = [exact_values(scene, ids, cx + dx, cy + dy) for dx, dy in ((0, 0), (-delta, -delta), (delta, -delta), (-delta, delta), (delta, delta))]
    spread = np.max(ex, axis=0) - np.min(ex, axis=0)
    vmax = np.abs(scene["vert"] if scene.get("value") is None else scene["value"]).max()
    tol = spread + 2e-6 * vmax
    ours, theirs = img[hit].astype(np.float64), gl_img[hit].astype(np.float64)
    keep = np.ones(len(ids), dtype=bool)
    if md is not None:
        # the mask compares the INTERPOLATED |view_pos| with min_dist (project.py:235,247): leave out
        # the pixels within 1e-3 of the threshold, where float32 interpolation decides
        dist = rc["depth"][hit] * np.sqrt(1 + (((cx / (p["w"] / 2) - 1) / _f(p)[0]) ** 2) + (((cy / (p["h"] / 2) - 1) / _f(p)[1]) ** 2))
        keep = np.abs(dist - md) > 1e-3 * md
        black_ours, black_gl = ~ours.any(axis=1), ~theirs.any(axis=1)
        assert (black_ours[keep] == black_gl[keep]).all(), f"{name}: min_distance mask differs from OpenGL"
        assert black_gl[keep].sum() > 100 and (~black_gl[keep]).sum() > 100
        keep &= ~black_gl
    diff = np.abs(ours - theirs)
    over = (diff > tol)[keep]
    assert not over.any(), (f"{name}: {int(over.any(axis=1).sum())} safe pixels differ from OpenGL by more than the "
                            f"+-{delta:.4f} px spread, worst {np.max((diff / np.maximum(tol, 1e-30))[keep]):.2f} x tolerance")
    # GL itself inside the same band around the exact interpolation at the centre
    assert (np.abs(theirs - ex[0]) <= tol)[keep].all()
    assert not img[safe & (tri < 0)].any() and not gl_img[safe & (tri < 0) & have_value].any()
    rel = diff[keep] / np.maximum(np.abs(theirs[keep]), 1.0)
    unsafe = ~safe & known
    # why the differing pixels differ: inside the depth buffer's resolution, or on an edge
    zfight = oray.safe_mask(rc) & ~safe & known
    return dict(differ=int((~same).sum()), differ_depth24=int((~same & zfight).sum()), depth24_unsafe=int(zfight.sum()),
                pixels=int(known.sum()), safe=int(safe.sum()), hits=int(hit.sum()),
                unsafe=int(unsafe.sum()), unsafe_same=int((same & unsafe).sum()),
                all_same_rate=float(same[known].mean()), max_rel=float(rel.max()), max_ratio=float(np.max((diff / np.maximum(tol, 1e-30))[keep])),
                frac_rel_le_1e5=float((rel <= 1e-5).mean()))


def _f(p):
    fx = 1 / np.tan(np.radians(p["fov"]) / 2)
    fy = 1 / np.tan(np.radians(p["fov"]) * p["h"] / p["w"] / 2)
    return fx, fy


def report(name, kind, r):
    print(f"[g15] {name} ({kind}): {r['differ']} of {r['pixels']} pixels show another triangle than OpenGL, {r['differ_depth24']} of them among the "
          f"{r['depth24_unsafe']} pixels whose two nearest surfaces lie within {DEPTH24_STEPS} steps of a 24-bit depth buffer")
    print(f"[g15] {name} ({kind}): all {r['safe']} safe pixels of {r['pixels']} show OpenGL's triangle; unsafe {r['unsafe']}: "
          f"{r['unsafe_same']} equal ({r['unsafe_same'] / max(r['unsafe'], 1):.4f}); whole frame {r['all_same_rate']:.6f}; "
          f"values: max |d|/max(|GL|,1) {r['max_rel']:.1e}, {r['frac_rel_le_1e5']:.4f} within 1e-5, worst {r['max_ratio']:.2f} of the sub-pixel tolerance")


def lens_source(params):
    """(sy, sx, inside): the source pixel of every output pixel of distort() (project.py:128-141): the float32 maps the
    reference computes (restated in oracle.ref_numpy.distort_maps, pinned bit for bit by g13), rounded half to even,
    zero outside the image"""
    w, h = int(params["w"]), int(params["h"])
    mx, my = orc.distort_maps(w, h, [params[k] for k in orc.DIST_KEYS])
    sx, sy = np.rint(mx.astype(np.float64)).astype(np.int64), np.rint(my.astype(np.float64)).astype(np.int64)
    inside = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
    return np.clip(sy, 0, h - 1), np.clip(sx, 0, w - 1), inside


def check_lens_composition(with_lens, without_lens, params):
    """persp_proj ends with flipud (project.py:281) and then distort (:292): the image with the lens must be the nearest gather
    of the image without it, exactly (the raster passes do not see the lens)"""
    sy, sx, inside = lens_source(params)
    want = np.where(inside[..., None], without_lens[sy, sx], 0.0).astype(np.float32)
    np.testing.assert_array_equal(with_lens, want)
    assert 0.5 < inside.mean() < 1.0 and (sx != np.arange(sx.shape[1])[None, :]).mean() > 0.5      # a real lens: most pixels move
