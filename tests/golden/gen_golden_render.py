#!/usr/bin/env python3
"""Golden vectors for the render wrappers, the distortion source map and LsqOptimizer, FROM THE
REFERENCE ITSELF.

Run only in the build container (reference checkout at /root/reference):

    python tests/golden/gen_golden_render.py

The reference's ``src/alproj/project.py`` and ``optimize.py`` are loaded by file path exactly as
in gen_golden.py (absent third-party imports registered as placeholders).  What is replaced, and
only that:

  g12_wrappers   ``persp_proj`` -- the OpenGL render, which cannot run here -- is replaced by a
                 function that hands back a seeded raw image; ``reverse_proj`` (project.py:327-374:
                 channel reorder :361, meshgrid :362-363, concatenate :364, int16 cast :368,
                 ``x > 0`` filter :369, offsets :370-373) and ``sim_image`` (project.py:296-325:
                 ``* 255``, ``astype(uint8)``, RGB->BGR) then run UNMODIFIED on it.
                 ``cv2.cvtColor(raw, COLOR_RGB2BGR)`` is a channel reversal (OpenCV documentation).
  g13_distort_map  ``cv2.remap`` is replaced by a recorder: ``distort`` (project.py:111-143) runs
                 unmodified and the float32 ``map_x`` / ``map_y`` it hands to cv2 are stored --
                 everything of the remap except cv2's nearest rounding / border rule.
  g14_lsq        nothing is replaced: ``LsqOptimizer.optimize`` (optimize.py:467-539, scipy's
                 ``least_squares`` is installed) on seeded GCP sets; returned params + error.

Only data is written: no reference source or bytecode enters this repository.
"""
import os
import sys
import warnings

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import FULL, BASE, PARAM_KEYS, gcp_like, load_reference, pvec  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DIST_KEYS = ("a1", "a2", "k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "s1", "s2", "s3", "s4")


def seeded_coord_image(rng, h, w):
    """What a render of the vertices themselves looks like: X, Z, Y per pixel (float32), zero
    where nothing is seen; plus the edge cases the x > 0 filter must get right."""
    img = np.zeros((h, w, 3), dtype=np.float32)
    seen = rng.random((h, w)) < 0.62
    img[..., 0] = np.where(seen, rng.uniform(0.5, 4000.0, (h, w)), 0.0)
    img[..., 1] = np.where(seen, rng.uniform(0.0, 900.0, (h, w)), 0.0)
    img[..., 2] = np.where(seen, rng.uniform(0.0, 4000.0, (h, w)), 0.0)
    img[0, 0] = [0.0, 12.5, 33.0]            # x == 0 on the surface's min column: dropped (quirk Q13)
    img[0, 1] = [-3.0, 5.0, 7.0]             # negative x: dropped
    img[0, 2] = [np.float32(1e-30), 1.0, 2.0]  # tiny positive x: kept
    img[1, 0] = [np.nan, 1.0, 2.0]           # NaN > 0 is False: dropped
    img[1, 1] = [np.inf, 1.0, 2.0]           # kept
    img[h - 1, w - 1] = [4000.0, 900.0, 4000.0]
    return img


def main():
    opt, prj = load_reference()
    warnings.simplefilter("ignore")
    rng = np.random.default_rng(20260221)
    cv2 = sys.modules["cv2"]
    cv2.COLOR_RGB2BGR = 4
    cv2.INTER_NEAREST = 0
    cv2.cvtColor = lambda raw, code: np.ascontiguousarray(raw[:, :, ::-1])

    # ---- g12: reverse_proj / sim_image post-processing ------------------------------------------
    g12 = {}
    offsets = np.array([732000.0, 1655.0, 4048000.0])        # X, Z, Y like get_colored_surface
    real_persp = prj.persp_proj
    for tag, (h, w, nch) in {"a": (37, 53, 3), "b": (64, 40, 1), "c": (90, 140, 4)}.items():
        raw = seeded_coord_image(rng, h, w)
        array = rng.integers(0, 256, (h, w, nch)).astype(np.uint8)
        chn = ["B", "G", "R", "NIR"][:nch] if nch != 1 else ["gray"]
        prj.persp_proj = lambda *a, _raw=raw, **k: _raw.copy()
        for otag, off in (("off", offsets), ("nooff", None)):
            df = prj.reverse_proj(array, None, None, dict(FULL, w=w, h=h), off, chnames=chn)
            g12[f"{tag}_{otag}_values"] = df.to_numpy(dtype=np.float64)
            g12[f"{tag}_{otag}_index"] = df.index.to_numpy()
            g12[f"{tag}_{otag}_columns"] = np.array(list(df.columns))
            g12[f"{tag}_{otag}_dtypes"] = np.array([str(t) for t in df.dtypes])
        g12[f"{tag}_raw"] = raw
        g12[f"{tag}_array"] = array
        g12[f"{tag}_chnames"] = np.array(chn)
    g12["offsets"] = offsets
    # sim_image: colours in and slightly outside [0, 1] (the uint8 cast of the reference wraps)
    col = rng.uniform(-0.05, 1.05, (45, 61, 3)).astype(np.float32)
    col[0, 0] = [0.0, 1.0, 0.5]
    col[0, 1] = [np.float32(254.999 / 255), np.float32(1 / 255), np.float32(0.999999)]
    col = np.clip(col, 0.0, 1.0)          # get_colored_surface clips colours to [0, 1] (surface.py:66)
    prj.persp_proj = lambda *a, **k: col.copy()
    g12["sim_raw"] = col
    g12["sim_bgr"] = prj.sim_image(None, None, None, dict(FULL, w=61, h=45))
    prj.persp_proj = real_persp
    np.savez_compressed(f"{OUT}/g12_wrappers.npz", **g12)

    # ---- g13: the source map distort() hands to cv2.remap --------------------------------------
    g13 = {}
    rec = {}

    def remap_recorder(img, map_x, map_y, interpolation=None):
        rec["map_x"], rec["map_y"], rec["interp"] = map_x.copy(), map_y.copy(), interpolation
        return img

    cv2.remap = remap_recorder
    coeff_sets = {
        "identity": [1, 1] + [0] * 12,
        "aonly": [1.1, 0.9] + [0] * 12,
        "radial": [1, 1, -0.05, 0.01, 0.002, 0.003, -0.001, 0.0005, 0, 0, 0, 0, 0, 0],
        "full": [FULL[k] for k in DIST_KEYS],
        "strong": [1.02, 0.98, 0.3, -0.1, 0.02, 0.05, 0.01, -0.004, 0.02, -0.015, 0.01, -0.01, 0.008, 0.004],
    }
    for (h, w) in ((48, 64), (97, 131), (187, 281)):
        img = np.zeros((h, w, 3), dtype=np.float32)
        for name, c in coeff_sets.items():
            prj.distort(img, np.array(c, dtype=np.float64))
            assert rec["map_x"].dtype == np.float32 and rec["map_x"].shape == (h, w)
            g13[f"mapx_{name}_{h}x{w}"] = rec["map_x"]
            g13[f"mapy_{name}_{h}x{w}"] = rec["map_y"]
            g13[f"coeffs_{name}"] = np.array(c, dtype=np.float64)
    g13["interpolation_flag_is_INTER_NEAREST"] = np.array(rec["interp"] == cv2.INTER_NEAREST)
    np.savez_compressed(f"{OUT}/g13_distort_map.npz", **g13)

    # ---- g14: LsqOptimizer.optimize ------------------------------------------------------------
    # Well-posed problems only (the targets can explain the data, so the optimum is sharp): with a
    # model that cannot fit, least_squares stops somewhere in a flat valley and the reference's own
    # result moves by 1e-3 relative under a 1e-13 perturbation of the residuals (measured).
    g14 = {}
    truth = dict(FULL, x=FULL["x"] + 5, y=FULL["y"] - 7, z=FULL["z"] + 3)
    xyz = gcp_like(opt, rng, 400, truth)
    dfx = pd.DataFrame(xyz, columns=["x", "y", "z"])
    uv_a = opt.project(dfx, truth).to_numpy() + rng.normal(0, 0.7, (len(xyz), 2))
    uv_b = uv_a.copy()
    uv_b[::37] += rng.normal(0, 40.0, uv_b[::37].shape)       # a few outliers for the robust losses
    ang = dict(pan=truth["pan"] + 1.5, tilt=truth["tilt"] - 1.0, fov=truth["fov"] + 2.0, roll=truth["roll"] + 0.5)
    init_ang = dict(truth, **ang)
    init_pose = dict(truth, x=truth["x"] + 2.0, y=truth["y"] - 1.5, z=truth["z"] + 1.0, **ang)
    init_dist = dict(truth, a1=1.0, a2=1.0, k1=0.0, k2=0.0, p1=0.0, p2=0.0)
    g14.update(xyz=xyz, uv_a=uv_a, uv_b=uv_b, param_keys=np.array(PARAM_KEYS))
    pose7 = ["x", "y", "z", "fov", "pan", "tilt", "roll"]
    cases = {
        "trf_linear_d7": dict(uv="a", init=init_pose, targets=pose7, kw=dict(method="trf")),
        "trf_huber_d7": dict(uv="b", init=init_pose, targets=pose7, kw=dict(method="trf", loss="huber", f_scale=5.0)),
        "dogbox_softl1_d4": dict(uv="b", init=init_ang, targets=["fov", "pan", "tilt", "roll"],
                                 kw=dict(method="dogbox", loss="soft_l1", f_scale=3.0,
                                         bound_widths={"fov": 10, "pan": 10, "tilt": 10, "roll": 10})),
        "trf_cauchy_d4": dict(uv="b", init=init_ang, targets=["fov", "pan", "tilt", "roll"],
                              kw=dict(method="trf", loss="cauchy", f_scale=2.0)),
        "lm_d4": dict(uv="a", init=init_ang, targets=["fov", "pan", "tilt", "roll"], kw=dict(method="lm")),
        "trf_linear_dist_d6": dict(uv="a", init=init_dist, targets=["a1", "a2", "k1", "k2", "p1", "p2"],
                                   kw=dict(method="trf")),
    }
    for name, c in cases.items():
        dfu = pd.DataFrame(uv_a if c["uv"] == "a" else uv_b, columns=["u", "v"])
        o = opt.LsqOptimizer(dfx, dfu, dict(c["init"]))
        o.set_target(c["targets"])
        params, err = o.optimize(**c["kw"])
        g14[f"{name}_uv"] = np.array(c["uv"])
        g14[f"{name}_init"] = pvec(c["init"])
        g14[f"{name}_targets"] = np.array(c["targets"])
        g14[f"{name}_params"] = pvec(params)
        g14[f"{name}_error"] = np.float64(err)
    np.savez_compressed(f"{OUT}/g14_lsq.npz", **g14)
    print("g12, g13, g14 written to", OUT)


if __name__ == "__main__":
    main()
