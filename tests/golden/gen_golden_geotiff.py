#!/usr/bin/env python3
"""Golden vectors for the COMPUTE part of the reference's to_geotiff (src/alproj/project.py:
376-503): rasterisation by (row, col) group + 3x3 NaN-aware focal fill + uint8 conversion.

Run in the build container only.  The reference's project.py is loaded by file path; rasterio
is absent, so `rasterio.open` is replaced by an object that CAPTURES what the reference hands to
the GeoTIFF writer (height, width, count, nodata, transform and every band array) -- the file
format itself is outside the path.  `from_bounds` is a stand-in that records its arguments.
Only data is written to tests/golden/g9_geotiff.npz.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import pandas as pd

REF = "/root/reference/src/alproj"
OUT = os.path.dirname(os.path.abspath(__file__))


class _Capture:
    def __init__(self):
        self.meta, self.bands = None, {}

    def open(self, path, mode, **kw):
        self.meta = kw
        cap = self

        class _W:
            def __enter__(self_w):
                return self_w

            def __exit__(self_w, *a):
                return False

            def write(self_w, arr, idx):
                cap.bands[idx] = np.array(arr)
        return _W()


def load():
    cap = _Capture()
    for name in ("moderngl", "cv2", "cmaes"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["cmaes"].CMA = object
    ras = types.ModuleType("rasterio")
    ras.open = cap.open
    sys.modules["rasterio"] = ras
    tr = types.ModuleType("rasterio.transform")
    tr.from_bounds = lambda *a: ("from_bounds",) + tuple(a)
    sys.modules["rasterio.transform"] = tr
    pkg = types.ModuleType("alproj")
    pkg.__path__ = []
    sys.modules["alproj"] = pkg
    spec = importlib.util.spec_from_file_location("alproj.optimize", f"{REF}/optimize.py")
    opt = importlib.util.module_from_spec(spec)
    sys.modules["alproj.optimize"] = opt
    spec.loader.exec_module(opt)
    spec = importlib.util.spec_from_file_location("alproj.project", f"{REF}/project.py")
    prj = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(prj)
    return prj, cap


def synthetic_df(rng, n, span_x, span_y, holes=True, integer=True):
    x = 732000.0 + rng.uniform(0, span_x, n)
    y = 4048000.0 + rng.uniform(0, span_y, n)
    if holes:                       # a band without points -> NaN cells for the focal fill
        keep = ~((x - 732000.0 > 0.4 * span_x) & (x - 732000.0 < 0.47 * span_x))
        x, y = x[keep], y[keep]
    vals = rng.integers(0, 256, (len(x), 3)).astype(np.float64)
    if not integer:
        vals += rng.uniform(0, 0.9, vals.shape)
    return pd.DataFrame({"x": x, "y": y, "R": vals[:, 0], "G": vals[:, 1], "B": vals[:, 2]})


def main():
    prj, cap = load()
    rng = np.random.default_rng(20260220)
    out = {}
    cases = {
        "mean_int": dict(df=synthetic_df(rng, 6000, 60, 40), kw=dict(resolution=1.0, agg_func="mean")),
        "mean_float_res2": dict(df=synthetic_df(rng, 4000, 90, 70, integer=False),
                                kw=dict(resolution=2.0, agg_func="mean", max_dist=4.0)),
        "max_nointerp": dict(df=synthetic_df(rng, 3000, 50, 50), kw=dict(resolution=1.0, agg_func="max", interpolate=False)),
        "min_sparse": dict(df=synthetic_df(rng, 700, 40, 40), kw=dict(resolution=1.0, agg_func="min", max_dist=2.0, nodata=0)),
        "median_small": dict(df=synthetic_df(rng, 2500, 30, 30), kw=dict(resolution=1.5, agg_func="median")),
    }
    for name, c in cases.items():
        cap.meta, cap.bands = None, {}
        prj.to_geotiff(c["df"], "/dev/null", **c["kw"])
        out[f"{name}_x"] = c["df"]["x"].to_numpy()
        out[f"{name}_y"] = c["df"]["y"].to_numpy()
        out[f"{name}_vals"] = c["df"][["R", "G", "B"]].to_numpy()
        out[f"{name}_raster"] = np.stack([cap.bands[i + 1] for i in range(3)])
        out[f"{name}_hw"] = np.array([cap.meta["height"], cap.meta["width"]])
        out[f"{name}_bounds"] = np.array(cap.meta["transform"][1:5], dtype=np.float64)
        out[f"{name}_kw"] = np.array(repr(c["kw"]))
    np.savez_compressed(f"{OUT}/g9_geotiff.npz", **out)
    for name in cases:
        r = out[f"{name}_raster"]
        print(name, r.shape, r.dtype, "nodata frac", float((r == cases[name]["kw"].get("nodata", 255)).mean()))


if __name__ == "__main__":
    main()
