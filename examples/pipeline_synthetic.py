#!/usr/bin/env python3
"""The reference's example.py (georectification of one photograph), end to end on synthetic
data with every stage on the GPU through alproj_amd:

  rasters -> colored_surface_mesh -> sim_image -> reverse_proj_device -> (synthetic matches)
  -> set_gcp -> filter_gcp_distance -> CMAOptimizer phase 1 (pose) -> phase 2 (distortion)
  -> LsqOptimizer polish -> reverse_proj of the "photograph" -> rasterize (to_geotiff's raster)
  (the last two also without the table in between: ReverseProjection.rasterize)

The only step replaced is `image_match` (CNN feature matching, out of scope): the "photograph"
is a render with a hidden true camera, and a match is made for a pixel of the simulated image
by projecting the world point it sees through the true camera (+ pixel noise, + outliers).

    python examples/pipeline_synthetic.py [grid_side] [image_width]
"""
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import synthetic as syn                                   # noqa: E402
from alproj_amd.gcp import filter_gcp_distance, set_gcp                   # noqa: E402
from alproj_amd.optimize import CMAOptimizer, LsqOptimizer, project       # noqa: E402
from alproj_amd.project import rasterize, reverse_proj, reverse_proj_device, sim_image   # noqa: E402
from alproj_amd.surface import colored_surface_mesh                       # noqa: E402


def synthetic_rasters(n, res=1.0, seed=7):
    """A filled DSM, three uint8 aerial bands, the affine transform and a nodata mask."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64) * res
    dsm = (1800 + 160 * np.sin(xx / 310.0) * np.cos(yy / 230.0) + 25 * np.sin(xx / 41.0) * np.sin(yy / 57.0)
           + rng.normal(0, 0.15, (n, n))).astype(np.float32)
    tex = (127 + 60 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + 50 * np.sin((xx + yy) / 31.0))
    aerial = np.clip(np.stack([tex + rng.normal(0, 8, (n, n)), tex * 0.8 + 30 + rng.normal(0, 8, (n, n)),
                               255 - tex + rng.normal(0, 8, (n, n))]), 0, 255).astype(np.uint8)
    nodata = np.zeros((n, n), dtype=bool)
    nodata[n // 3:n // 3 + 6, n // 2:n // 2 + 40] = True           # a hole in the DSM
    transform = (res, 0.0, 732000.0, 0.0, -res, 4048000.0 + n * res)    # north-up
    return aerial, dsm, transform, nodata


def synthetic_matches(rp, true_params, n, w, h, rng, noise_px=0.7, outliers=0.03):
    """What image_match would deliver: pixel pairs (photograph, simulated image)."""
    u_sim = rng.integers(0, w, 4 * n)
    v_sim = rng.integers(0, h, 4 * n)
    xyz = rp.lookup(u_sim, v_sim)
    ok = np.isfinite(xyz[:, 0])
    u_sim, v_sim, xyz = u_sim[ok][:n], v_sim[ok][:n], xyz[ok][:n]
    uv = project(pd.DataFrame(xyz, columns=["x", "y", "z"]), true_params).to_numpy()
    uv += rng.normal(0, noise_px, uv.shape)
    bad = rng.random(len(uv)) < outliers
    uv[bad] += rng.normal(0, 80, (int(bad.sum()), 2))
    inside = (uv[:, 0] >= 0) & (uv[:, 0] < w) & (uv[:, 1] >= 0) & (uv[:, 1] < h)
    return pd.DataFrame({"u_org": np.rint(uv[inside, 0]).astype(int), "v_org": np.rint(uv[inside, 1]).astype(int),
                         "u_sim": u_sim[inside], "v_sim": v_sim[inside]})


def run(n=1024, w=1404, h=936, generations=150, seed=1, verbose=True):
    t = {}
    rng = np.random.default_rng(seed)

    def tick(name, t0):
        t[name] = time.perf_counter() - t0
        if verbose:
            print(f"  {name:32s} {t[name] * 1e3:9.1f} ms", flush=True)

    aerial, dsm, transform, nodata = synthetic_rasters(n)
    t0 = time.perf_counter()
    mesh, offsets = colored_surface_mesh(aerial, dsm, transform, nodata, aerial.dtype)
    tick("colored_surface_mesh", t0)

    cam = dict(syn.BASE_CAMERA)
    cam.update(x=732000.0 + 20.0, y=4048000.0 + n / 2, z=float(dsm[n // 2, 20]) + 60.0, pan=92.0, tilt=-8.0, fov=70.0,
               w=w, h=h, cx=w / 2, cy=h / 2)
    true = dict(cam)
    true.update(x=cam["x"] + 6, y=cam["y"] - 9, z=cam["z"] + 4, pan=cam["pan"] + 2.5, tilt=cam["tilt"] + 1.5,
                roll=0.8, fov=cam["fov"] - 3, a1=1.01, a2=0.99, k1=-0.06, k2=0.015, p1=8e-4, p2=-6e-4)

    t0 = time.perf_counter()
    photo = sim_image(mesh, None, None, true, offsets)                    # the "photograph"
    sim = sim_image(mesh, None, None, cam, offsets, min_distance=50)      # example.py:33
    tick("sim_image x2", t0)

    # ---- phase 1: pose
    t0 = time.perf_counter()
    rp = reverse_proj_device(mesh, None, cam, offsets)                    # example.py:36, table stays in HBM
    match = synthetic_matches(rp, true, 1500, w, h, rng)
    gcps = filter_gcp_distance(set_gcp(match, rp), cam, min_distance=50)  # example.py:50-53
    tick("reverse_proj + set_gcp (device)", t0)
    t0 = time.perf_counter()
    opt = CMAOptimizer(gcps[["x", "y", "z"]], gcps[["u", "v"]], cam)
    opt.set_target(["x", "y", "z", "fov", "pan", "tilt", "roll", "a1", "a2"])
    p2, err1 = opt.optimize(generation=generations, sigma=1.0, population_size=50, f_scale=10.0, seed=seed, progress=False)
    tick(f"CMA phase 1 ({generations} gen x 50)", t0)

    # ---- phase 2: distortion, from a fresh simulation with the phase-1 pose
    t0 = time.perf_counter()
    rp2 = reverse_proj_device(mesh, None, p2, offsets)
    match2 = synthetic_matches(rp2, true, 1500, w, h, rng)
    gcps2 = filter_gcp_distance(set_gcp(match2, rp2), p2, min_distance=50)
    tick("reverse_proj + set_gcp (device)#2", t0)
    t0 = time.perf_counter()
    opt = CMAOptimizer(gcps2[["x", "y", "z"]], gcps2[["u", "v"]], p2)
    opt.set_target(["k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "s1", "s2", "s3", "s4"])
    p3, err2 = opt.optimize(generation=generations, sigma=1.0, population_size=50, f_scale=10.0, seed=seed, progress=False)
    tick(f"CMA phase 2 ({generations} gen x 50)", t0)
    t0 = time.perf_counter()
    lsq = LsqOptimizer(gcps2[["x", "y", "z"]], gcps2[["u", "v"]], p3)
    lsq.set_target(["x", "y", "z", "fov", "pan", "tilt", "roll", "a1", "a2", "k1", "k2", "p1", "p2"])
    p4, err3 = lsq.optimize(method="trf", loss="huber", f_scale=10.0, max_nfev=200)
    tick("LsqOptimizer polish", t0)

    # ---- georectification of the photograph (example.py:103-118)
    t0 = time.perf_counter()
    geo = reverse_proj(photo, mesh, None, p4, offsets)
    tick("reverse_proj (DataFrame)", t0)
    t0 = time.perf_counter()
    raster, bounds = rasterize(geo, resolution=2.0, bands=["R", "G", "B"], interpolate=True, max_dist=2.0)
    tick("rasterize (to_geotiff compute)", t0)
    # the same two steps without the table in between: the coordinate image stays in HBM and is binned there
    t0 = time.perf_counter()
    with reverse_proj_device(mesh, None, p4, offsets) as rp4:
        raster_dev, bounds_dev = rp4.rasterize(photo, resolution=2.0, bands=["R", "G", "B"], interpolate=True, max_dist=2.0)
    tick("reverse_proj + rasterize (device)", t0)
    assert bounds_dev == bounds and np.array_equal(raster_dev, raster)

    # ---- quality: reprojection error of clean world points under the estimated camera
    chk = rp2.lookup(rng.integers(0, w, 4000), rng.integers(0, h, 4000))
    chk = pd.DataFrame(chk[np.isfinite(chk[:, 0])], columns=["x", "y", "z"])
    d = project(chk, p4).to_numpy() - project(chk, true).to_numpy()
    inside = np.all(np.isfinite(d), axis=1)
    reproj = float(np.median(np.hypot(d[inside, 0], d[inside, 1])))
    d0 = project(chk, cam).to_numpy() - project(chk, true).to_numpy()
    reproj0 = float(np.nanmedian(np.hypot(d0[:, 0], d0[:, 1])))
    rp.close()
    rp2.close()
    mesh.close()
    out = dict(times=t, gcps=(len(gcps), len(gcps2)), errors=(float(err1), float(err2), float(err3)),
               reproj_px_initial=reproj0, reproj_px_final=reproj, raster_shape=raster.shape,
               raster_filled=float((raster[0] != 255).mean()), georectified_rows=len(geo), params=p4, true=true)
    if verbose:
        print(f"GCPs {out['gcps']}, optimiser errors {out['errors']}")
        print(f"median reprojection error vs the true camera: {reproj0:.1f} px initially -> {reproj:.2f} px")
        print(f"georectified table {len(geo)} rows -> raster {raster.shape}, {out['raster_filled']:.0%} filled")
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 1404
    run(n=n, w=w, h=w * 2 // 3)
