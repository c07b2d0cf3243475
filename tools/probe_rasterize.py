#!/usr/bin/env python3
"""Development probe of to_geotiff's compute on the device-resident frame: the 100 M-vertex frame's ~11.7 M
surface pixels -> 3 x 8088 x 9786 raster.   python3 tools/probe_rasterize.py [N] [agg ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import project as aproj     # noqa: E402
from alproj_amd import synthetic as syn     # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
aggs = sys.argv[2:] or ["mean", "median"]
L.init(0)
n = syn.grid_side(N)
s = syn.surface(n)
cam = syn.base_params(n)
img = np.random.default_rng(0).integers(0, 256, (int(cam["h"]), int(cam["w"]), 3), dtype=np.uint8)
with aproj.reverse_proj_device(s["vert"], None, cam, s["offsets"], grid_shape=(n, n)) as rp:
    for agg in aggs:
        rp.rasterize(img, ["B", "G", "R"], 1.0, ["B", "G", "R"], True, 1.0, agg)
        L.kernel_timing(True)
        for rep in range(3):
            L.kernel_time_ms()
            t = time.perf_counter()
            ras, b = rp.rasterize(img, ["B", "G", "R"], 1.0, ["B", "G", "R"], True, 1.0, agg)
            wall = time.perf_counter() - t
            k, sec = L.kernel_time_ms()
            print(f"{agg}: call {wall * 1e3:.1f} ms, kernels {k:.3f} ms in {sec} sections, raster {ras.shape}", flush=True)
        L.kernel_timing(False)
