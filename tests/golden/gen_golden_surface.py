#!/usr/bin/env python3
"""Golden vectors for the mesh construction of get_colored_surface FROM THE REFERENCE ITSELF.

Run only in the build container (reference checkout at /root/reference):

    python tests/golden/gen_golden_surface.py

rasterio (GDAL) is not installed here, so the reference's raster I/O cannot run: this script
loads ``src/alproj/surface.py`` by file path with stand-in modules for ``rasterio``,
``rasterio.merge``, ``rasterio.enums`` and ``rasterio.fill`` whose ``merge`` hands back seeded
synthetic rasters (what reading + resampling real files would deliver) and whose ``fillnodata``
fills the holes with a fixed value.  Everything after the I/O -- the clamps, the coordinate
grid, ``_normalize_aerial``, the index array and its nodata filter, the offsets
(surface.py:168-212) -- is the reference's own code, and its outputs are stored next to the
arrays that entered it.  Only data is written.
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/src/alproj/surface.py"
OUT = os.path.dirname(os.path.abspath(__file__))


class FakeDataset:
    def __init__(self, data, nodata, transform):
        self.data, self.nodata, self.transform = data, nodata, transform
        self.dtypes = tuple(str(data.dtype) for _ in range(data.shape[0]))
        self.bounds = None


STATE = {}


def load_reference():
    def merge(datasets, bounds=None, res=None, resampling=None):
        ds = datasets[0]
        return ds.data.copy(), ds.transform

    def fillnodata(arr, mask, max_search_distance=None):
        out = arr.copy()
        out[~mask] = STATE["fill_value"]
        STATE["filled"] = out.copy()
        return out

    rio = types.ModuleType("rasterio")
    rio_merge = types.ModuleType("rasterio.merge")
    rio_merge.merge = merge
    rio_enums = types.ModuleType("rasterio.enums")
    rio_enums.Resampling = types.SimpleNamespace(cubic_spline="cubic_spline")
    rio_fill = types.ModuleType("rasterio.fill")
    rio_fill.fillnodata = fillnodata
    sys.modules.update({"rasterio": rio, "rasterio.merge": rio_merge, "rasterio.enums": rio_enums,
                        "rasterio.fill": rio_fill})
    spec = importlib.util.spec_from_file_location("alproj_ref_surface", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def scene(rng, n, aerial_dtype, aerial_scale, holes, neg):
    t = (1.0, 0.0, 732000.25 + rng.uniform(0, 100), 0.0, -1.0, 4048000.75 + n + rng.uniform(0, 100))   # north-up, res 1
    yy, xx = np.mgrid[0:n, 0:n]
    dsm = (1500 + 6 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + rng.normal(0, 0.3, (n, n))).astype(np.float32)
    if neg:
        dsm[rng.random((n, n)) < 0.01] = -3.0           # "still negative after filling"
    if holes:
        mask = rng.random((n, n)) < 0.03
        mask[5:9, 10:20] = True
        dsm[mask] = np.nan
    if np.issubdtype(np.dtype(aerial_dtype), np.integer):
        aerial = rng.integers(0, int(aerial_scale) + 1, (4, n, n)).astype(aerial_dtype)     # 4 bands: the 4th is dropped
    else:
        aerial = (rng.random((4, n, n)) * aerial_scale).astype(aerial_dtype)
    return t, dsm[np.newaxis], aerial


def main():
    ref = load_reference()
    rng = np.random.default_rng(20260220)
    cases = {
        "u8_holes": dict(n=48, aerial_dtype=np.uint8, aerial_scale=255, holes=True, neg=False, color_max=None),
        "u16_full": dict(n=33, aerial_dtype=np.uint16, aerial_scale=65535, holes=False, neg=True, color_max=None),
        "f32_unit": dict(n=40, aerial_dtype=np.float32, aerial_scale=1.0, holes=True, neg=False, color_max=None),
        "f32_255": dict(n=40, aerial_dtype=np.float32, aerial_scale=250.0, holes=False, neg=False, color_max=None),
        "u16_cmax": dict(n=36, aerial_dtype=np.uint16, aerial_scale=4095, holes=True, neg=False, color_max=4095.0),
    }
    out = {}
    for name, c in cases.items():
        t, dsm, aerial = scene(rng, c["n"], c["aerial_dtype"], c["aerial_scale"], c["holes"], c["neg"])
        STATE["fill_value"] = 1490.0
        a_ds = FakeDataset(aerial, None, t)
        d_ds = FakeDataset(dsm, None, t)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            vert, col, ind, offsets = ref.get_colored_surface(a_ds, d_ds, {"x": 0.0, "y": 0.0}, distance=10, res=1.0,
                                                              color_max=c["color_max"])
        nodata = np.isnan(dsm[0])
        valid = dsm[0][~nodata]
        out.update({f"{name}_aerial": aerial, f"{name}_filled": STATE["filled"], f"{name}_transform": np.array(t),
                    f"{name}_nodata": nodata, f"{name}_zmax": np.float64(valid.max()),
                    f"{name}_color_max": np.float64(np.nan if c["color_max"] is None else c["color_max"]),
                    f"{name}_vert": vert, f"{name}_col": col, f"{name}_ind": ind, f"{name}_offsets": offsets})
        print(name, vert.shape, col.dtype, ind.shape, offsets)
    out["names"] = np.array(list(cases))
    np.savez_compressed(os.path.join(OUT, "g11_surface.npz"), **out)


if __name__ == "__main__":
    main()
