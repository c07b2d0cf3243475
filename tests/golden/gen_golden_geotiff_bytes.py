#!/usr/bin/env python3
"""Golden rasters of the reference's to_geotiff (src/alproj/project.py:376-503) for a MILLION BYTE-VALUED points.

Run in the build container only:    python tests/golden/gen_golden_geotiff_bytes.py      (about a minute)

Same set-up as gen_golden_geotiff.py / gen_golden_geotiff_float.py (the reference's project.py loaded by file path,
`rasterio.open` replaced by an object that captures what the reference hands to the GeoTIFF writer).  The inputs come from
tests/rasterize_cases.byte_points (seeded; NOT stored): the clustered points of g17 carrying a photograph's bytes, the
reference's own use of the function -- the case the device path takes through its packed kernels (the band values ride the cell
sort as its payload; mean / max / min from order-free pieces, the median selected from the same one sort: runs of up to 16
points in registers, longer ones through LDS histograms).  All four aggregates, one to three bands, two resolutions.
Only the reference's uint8 rasters and their geometry are written to tests/golden/g18_geotiff_bytes.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_golden_geotiff as gg                     # noqa: E402
from tests.rasterize_cases import BYTE_CASES, byte_points        # noqa: E402


def main():
    prj, cap = gg.load()
    df = byte_points()
    out = {"n_points": np.array(len(df))}
    for name, kw in BYTE_CASES.items():
        cap.meta, cap.bands = None, {}
        prj.to_geotiff(df, "/dev/null", **kw)
        nb = len(kw.get("bands", ["R", "G", "B"]))
        out[f"{name}_raster"] = np.stack([cap.bands[i + 1] for i in range(nb)])
        out[f"{name}_hw"] = np.array([cap.meta["height"], cap.meta["width"]])
        out[f"{name}_bounds"] = np.array(cap.meta["transform"][1:5], dtype=np.float64)
        r = out[f"{name}_raster"]
        print(name, r.shape, "nodata frac", float((r == kw.get("nodata", 255)).mean()), flush=True)
    np.savez_compressed(os.path.join(HERE, "g18_geotiff_bytes.npz"), **out)
    print("wrote g18_geotiff_bytes.npz", os.path.getsize(os.path.join(HERE, "g18_geotiff_bytes.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
