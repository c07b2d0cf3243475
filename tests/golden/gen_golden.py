#!/usr/bin/env python3
"""Generate the golden vectors in this directory FROM THE REFERENCE ITSELF.

Run only in the build container, where the reference checkout lives at /root/reference:

    python tests/golden/gen_golden.py

It loads the reference's ``src/alproj/optimize.py`` and ``src/alproj/project.py`` by file path
(their third-party imports that are absent here -- cmaes, cv2, moderngl, rasterio -- are
registered as empty placeholder modules; none of the functions exercised below touches them),
feeds them seeded synthetic inputs, and stores inputs + outputs as ``*.npz``.  Only data is
written: no reference source or bytecode enters this repository.

Fixture groups:
  g1_matrices      intrinsic_mat / extrinsic_mat                  optimize.py:8-96
  g2_distort       _distort on grids + random points              optimize.py:98-120
  g3_project       project, UTM-magnitude points, 8 param sets    optimize.py:122-155
  g4_losses        rmse / huber_loss incl. r == f_scale boundary  optimize.py:157-212
  g5_population    CMAOptimizer._loss_function over (P,D) X       optimize.py:329-357
  g6_bounds        bounds_to_array defaults/overrides/fallback    optimize.py:249-276
  g7_gl_matrices   projection_mat / modelview_mat                 project.py:13-109
  g8_residuals     compute_residuals                              optimize.py:215-237
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import pandas as pd

REF = "/root/reference/src/alproj"
OUT = os.path.dirname(os.path.abspath(__file__))

PARAM_KEYS = ("x", "y", "z", "fov", "pan", "tilt", "roll", "a1", "a2",
              "k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2",
              "s1", "s2", "s3", "s4", "w", "h", "cx", "cy")


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    _placeholder("cmaes", CMA=object)
    spec = importlib.util.spec_from_file_location("alproj.optimize", f"{REF}/optimize.py")
    opt = importlib.util.module_from_spec(spec)
    pkg = _placeholder("alproj")
    pkg.__path__ = []
    sys.modules["alproj.optimize"] = opt
    spec.loader.exec_module(opt)
    pkg.optimize = opt
    _placeholder("moderngl")
    _placeholder("cv2")
    _placeholder("rasterio")
    _placeholder("rasterio.transform", from_bounds=None)
    spec = importlib.util.spec_from_file_location("alproj.project", f"{REF}/project.py")
    prj = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(prj)
    return opt, prj


def pvec(p):
    return np.array([float(p[k]) for k in PARAM_KEYS])


BASE = dict(x=732731.0, y=4051171.0, z=2458.0, fov=75.0, pan=95.0, tilt=0.0, roll=0.0,
            a1=1.0, a2=1.0, k1=0.0, k2=0.0, k3=0.0, k4=0.0, k5=0.0, k6=0.0, p1=0.0, p2=0.0,
            s1=0.0, s2=0.0, s3=0.0, s4=0.0, w=5616, h=3744, cx=2808.0, cy=1872.0)
FULL = dict(BASE, tilt=3.0, roll=1.0, a1=1.02, a2=0.98, k1=-0.05, k2=0.01, k3=0.002,
            k4=0.003, k5=-0.001, k6=0.0005, p1=0.001, p2=-0.002, s1=0.0005, s2=-0.0002,
            s3=-0.0003, s4=0.0001, cx=2800.0, cy=1880.0)


def param_sets():
    sets = [BASE, FULL,
            dict(FULL, tilt=-10.0, roll=5.0, pan=200.0, fov=50.0),
            dict(BASE, a1=1.1, a2=0.9),
            dict(BASE, k1=0.08, k2=-0.02, k3=0.004, k4=0.01, k5=0.002, k6=-0.001),
            dict(FULL, w=1920, h=1080, cx=955.0, cy=545.0, fov=60.0, pan=10.0, tilt=25.0),
            dict(FULL, x=732700.5, y=4051200.25, z=2440.75, pan=359.0, tilt=-45.0, roll=-30.0),
            dict(BASE, fov=110.0, pan=-80.0, tilt=12.0, roll=90.0, p1=0.01, p2=0.01,
                 s1=-0.01, s4=0.02)]
    return sets


def points_utm(rng, n, cam):
    """n points in a 4 km box around the camera (some behind it), + special cases."""
    pts = np.empty((n, 3))
    pts[:, 0] = cam["x"] + rng.uniform(-2000, 4000, n)
    pts[:, 1] = cam["y"] + rng.uniform(-3000, 3000, n)
    pts[:, 2] = cam["z"] + rng.uniform(-800, 900, n)
    # a near point, a point behind the camera, the camera position itself (NaN case, Q7)
    pts[0] = [cam["x"] + 9.3125, cam["y"] - 9.25, cam["z"] - 4.644531]
    pts[1] = [cam["x"] - 1000.0, cam["y"], cam["z"]]
    pts[2] = [cam["x"], cam["y"], cam["z"]]
    return pts


def gcp_like(opt, rng, n, p, depth=(60.0, 3500.0)):
    """n points inside the view of pose p (what set_gcp yields): random pixels of the pinhole
    image back-projected to random depths through the reference's own K and E matrices."""
    K = opt.intrinsic_mat(p["fov"], p["w"], p["h"], p["cx"], p["cy"])
    E = opt.extrinsic_mat(p["pan"], p["tilt"], p["roll"], p["x"], p["y"], p["z"])
    u = rng.uniform(0.04 * p["w"], 0.96 * p["w"], n)
    v = rng.uniform(0.04 * p["h"], 0.96 * p["h"], n)
    z = -rng.uniform(depth[0], depth[1], n)                 # in front: the camera looks down -Z_cam
    img = np.stack([(p["w"] - u) * z, v * z, z])            # u = w - x/z  (optimize.py:147)
    cam = np.linalg.solve(K, img)
    return (E[:3, :3].T @ (cam - E[:3, 3:4])).T


def main():
    opt, prj = load_reference()
    warnings.simplefilter("ignore")
    rng = np.random.default_rng(20260220)
    sets = param_sets()

    # ---- g1: camera matrices -------------------------------------------------------------
    K, E = [], []
    for p in sets:
        K.append(opt.intrinsic_mat(p["fov"], p["w"], p["h"], p["cx"], p["cy"]))
        E.append(opt.extrinsic_mat(p["pan"], p["tilt"], p["roll"], p["x"], p["y"], p["z"]))
    K_default = opt.intrinsic_mat(75, 5616, 3744)
    np.savez(f"{OUT}/g1_matrices.npz", params=np.array([pvec(p) for p in sets]),
             K=np.array(K), E=np.array(E), K_default_75_5616_3744=K_default)

    # ---- g2: _distort -----------------------------------------------------------------------
    coeff_sets = {
        "zero": [1, 1] + [0] * 12,
        "aonly": [1.1, 0.9] + [0] * 12,
        "radial": [1, 1, -0.05, 0.01, 0.002, 0.003, -0.001, 0.0005, 0, 0, 0, 0, 0, 0],
        "full": [FULL[k] for k in ("a1", "a2", "k1", "k2", "k3", "k4", "k5", "k6",
                                   "p1", "p2", "s1", "s2", "s3", "s4")],
    }
    g2 = {}
    for (w, h) in ((5616, 3744), (641, 479)):
        gx, gy = np.meshgrid(np.linspace(0, w - 1, 9), np.linspace(0, h - 1, 7))
        pts = np.vstack([np.stack([gx.ravel(), gy.ravel()]).T,
                         np.stack([rng.uniform(-500, w + 500, 200),
                                   rng.uniform(-500, h + 500, 200)]).T])
        g2[f"pts_{w}x{h}"] = pts
        for name, c in coeff_sets.items():
            g2[f"coeffs_{name}"] = np.array(c, dtype=float)
            g2[f"out_{name}_{w}x{h}"] = opt._distort(pts, w, h, *c)
    np.savez(f"{OUT}/g2_distort.npz", **g2)

    # ---- g3: project ---------------------------------------------------------------------
    g3 = {"params": np.array([pvec(p) for p in sets])}
    for i, p in enumerate(sets):
        pts = points_utm(rng, 1000, p)
        df = pd.DataFrame(pts, columns=["x", "y", "z"])
        g3[f"xyz_{i}"] = pts
        g3[f"uv_{i}"] = opt.project(df, p).to_numpy()
        inview = gcp_like(opt, rng, 600, p)
        g3[f"xyz_inview_{i}"] = inview
        g3[f"uv_inview_{i}"] = opt.project(pd.DataFrame(inview, columns=["x", "y", "z"]), p).to_numpy()
    # SURVEY 8.2 known answers
    ka = np.array([[733731, 4051071, 2500], [734200.3125, 4050691.75, 2988.827881],
                   [732740.3125, 4051161.75, 2453.355469], [731731, 4051171, 2458],
                   [732731, 4051171, 2458]], dtype=float)
    g3["xyz_known"] = ka
    g3["uv_known"] = opt.project(pd.DataFrame(ka, columns=["x", "y", "z"]), FULL).to_numpy()
    g3["params_known"] = pvec(FULL)
    np.savez(f"{OUT}/g3_project.npz", **g3)

    # ---- g4: losses --------------------------------------------------------------------------
    n = 500
    proj = np.stack([rng.uniform(0, 5616, n), rng.uniform(0, 3744, n)]).T
    obs = proj + rng.normal(0, 8, (n, 2))
    # exact boundary r == f_scale (3-4-5 triangles scaled): r = 10 and r = 1000
    obs[0] = proj[0] + [6.0, 8.0]
    obs[1] = proj[1] + [600.0, 800.0]
    obs[2] = proj[2]  # r == 0
    dfo = pd.DataFrame(obs, columns=["u", "v"])
    dfp = pd.DataFrame(proj, columns=["u", "v"])
    g4 = dict(obs=obs, proj=proj, rmse=opt.rmse(dfo, dfp),
              huber_default=opt.huber_loss(dfo, dfp),
              huber_10=opt.huber_loss(dfo, dfp, 10.0),
              huber_1000=opt.huber_loss(dfo, dfp, 1000.0),
              huber_0p5=opt.huber_loss(dfo, dfp, 0.5))
    np.savez(f"{OUT}/g4_losses.npz", **g4)

    # ---- g5: population loss (the closure of CMAOptimizer._loss_function) -----------------------
    truth = dict(FULL, x=FULL["x"] + 5, y=FULL["y"] - 7, z=FULL["z"] + 3)
    g5_sets = {}
    # "gcp": GCP-like points inside the image (the optimiser's real input);
    # "wild": a box around the camera, mostly outside the image, where the distortion
    #         polynomial explodes (losses ~1e13) -- a float64-only stress case.
    pts = gcp_like(opt, rng, 1127, truth)
    wild = points_utm(rng, 1200, truth)[3:]
    wild = wild[wild[:, 0] > truth["x"] + 50]
    for tag, pp in (("gcp", pts), ("wild", wild)):
        dfx_ = pd.DataFrame(pp, columns=["x", "y", "z"])
        uvo = opt.project(dfx_, truth).to_numpy() + rng.normal(0, 1.0, (len(pp), 2))
        g5_sets[tag] = (pp, uvo)
    pts, uv_obs = g5_sets["gcp"]
    dfx = pd.DataFrame(pts, columns=["x", "y", "z"])
    dfu = pd.DataFrame(uv_obs, columns=["u", "v"])
    g5 = dict(xyz=pts, uv_obs=uv_obs, params_init=pvec(FULL),
              wild_xyz=g5_sets["wild"][0], wild_uv_obs=g5_sets["wild"][1])
    targets = {
        "d9": ["x", "y", "z", "fov", "pan", "tilt", "roll", "a1", "a2"],
        "d12": ["k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "s1", "s2", "s3", "s4"],
        "d21": ["x", "y", "z", "fov", "pan", "tilt", "roll", "a1", "a2", "k1", "k2", "k3",
                "k4", "k5", "k6", "p1", "p2", "s1", "s2", "s3", "s4"],
    }
    for name, tgt in targets.items():
        o = opt.CMAOptimizer(dfx, dfu, dict(FULL))
        o.set_target(tgt)
        bounds = opt.bounds_to_array(o.params_init, tgt, None)
        P = 24
        X = rng.uniform(0.3, 0.7, (P, len(tgt)))
        X[0] = 0.5
        X[7] = X[3]                      # an exact tie: argmin must be the first index
        g5[f"{name}_X"] = X
        g5[f"{name}_bounds"] = bounds
        g5[f"{name}_targets"] = np.array(tgt)
        for tag, fs in (("md", None), ("hub", 10.0)):
            f = o._loss_function(bounds, fs)
            g5[f"{name}_{tag}"] = np.array([f(x) for x in X])
        if name == "d21":
            ow = opt.CMAOptimizer(pd.DataFrame(g5_sets["wild"][0], columns=["x", "y", "z"]),
                                  pd.DataFrame(g5_sets["wild"][1], columns=["u", "v"]), dict(FULL))
            ow.set_target(tgt)
            for tag, fs in (("md", None), ("hub", 10.0)):
                f = ow._loss_function(bounds, fs)
                g5[f"wild_{name}_{tag}"] = np.array([f(x) for x in X])
    np.savez(f"{OUT}/g5_population.npz", **g5)

    # ---- g6: bounds_to_array ----------------------------------------------------------------------
    tgt = ["x", "fov", "k1", "cx"]                 # cx: not in the width table -> 0.2 fallback
    g6 = dict(params=pvec(FULL), targets=np.array(tgt),
              default=opt.bounds_to_array(FULL, tgt, None),
              override=opt.bounds_to_array(FULL, tgt, {"fov": 10, "cx": 7.5}),
              all21=opt.bounds_to_array(FULL, targets["d21"], None),
              all21_targets=np.array(targets["d21"]))
    np.savez(f"{OUT}/g6_bounds.npz", **g6)

    # ---- g7: GL matrices -----------------------------------------------------------------------
    PM, MV = [], []
    for p in sets:
        PM.append(prj.projection_mat(p["fov"], p["w"], p["h"]))
        MV.append(prj.modelview_mat(p["pan"], p["tilt"], p["roll"],
                                    p["x"] - 732000.0, p["y"] - 4048000.0, p["z"] - 2000.0))
    pm_cxcy = prj.projection_mat(75, 5616, 3744, near=0.5, far=5000.0, cx=2800.0, cy=1880.0)
    np.savez(f"{OUT}/g7_gl_matrices.npz", params=np.array([pvec(p) for p in sets]),
             proj=np.array(PM), view=np.array(MV), proj_cxcy_near_far=pm_cxcy,
             cam_offset=np.array([732000.0, 4048000.0, 2000.0]))

    # ---- g8: compute_residuals -------------------------------------------------------------------
    res = opt.compute_residuals(dfx, dfu, FULL)
    np.savez(f"{OUT}/g8_residuals.npz", xyz=pts, uv_obs=uv_obs, params=pvec(FULL), residuals=res)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
