#!/usr/bin/env python3
"""Golden rasters of the reference's to_geotiff (src/alproj/project.py:376-503) for a MILLION FLOAT-VALUED points.

Run in the build container only:    python tests/golden/gen_golden_geotiff_float.py      (about a minute)

Same set-up as gen_golden_geotiff.py (the reference's project.py loaded by file path, `rasterio.open` replaced by an object
that captures what the reference hands to the GeoTIFF writer).  The inputs come from tests/rasterize_cases.float_points
(seeded; NOT stored): clustered points -- hundreds per cell at one edge -- with fractional band values of mixed magnitude,
the case in which the ORDER of a cell's float64 sum can change its float32 mean and, next to an integer, its byte.  pandas
sums a group in row order with Kahan compensation; the device path does the same since round 4.
Only the reference's uint8 rasters and their geometry are written to tests/golden/g17_geotiff_float.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_golden_geotiff as gg                     # noqa: E402
from tests.rasterize_cases import FLOAT_CASES, float_points        # noqa: E402


def main():
    prj, cap = gg.load()
    df = float_points()
    out = {"n_points": np.array(len(df))}
    for name, kw in FLOAT_CASES.items():
        cap.meta, cap.bands = None, {}
        prj.to_geotiff(df, "/dev/null", **kw)
        out[f"{name}_raster"] = np.stack([cap.bands[i + 1] for i in range(3)])
        out[f"{name}_hw"] = np.array([cap.meta["height"], cap.meta["width"]])
        out[f"{name}_bounds"] = np.array(cap.meta["transform"][1:5], dtype=np.float64)
        r = out[f"{name}_raster"]
        print(name, r.shape, "nodata frac", float((r == 255).mean()), flush=True)
    np.savez_compressed(os.path.join(HERE, "g17_geotiff_float.npz"), **out)
    print("wrote g17_geotiff_float.npz", os.path.getsize(os.path.join(HERE, "g17_geotiff_float.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
