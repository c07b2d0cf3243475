#!/usr/bin/env python3
"""Development: randomised differential test of to_geotiff's compute (alp_rasterize_columns through
alproj_amd.project.rasterize, and the device-fed ReverseProjection.rasterize) against the pandas / scipy restatement of the
reference (oracle.ref_numpy.rasterize_points): clustered points (long runs of one raster cell inside a wave, runs across
wave and workgroup boundaries), NaN values, 1-4 bands, all four aggregates, 0-9 focal sweeps, several resolutions; a third of
the cases once more with the points as the surface pixels of a resident frame and the bands from a uint8 / uint16 / float32 /
float64 image.  Byte-exact.   python3 tools/fuzz_rasterize.py [seconds] [seed]"""
import os
import sys
import time
import warnings

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import project as prj       # noqa: E402
from oracle import ref_numpy as orc         # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
L.init(0)
warnings.simplefilter("ignore")
t_end = time.time() + budget
n_cases = n_f32 = n_dev = 0
names = ["R", "G", "B", "N"]
while time.time() < t_end:
    n = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 5000, 40_000]))
    ext_x, ext_y = float(rng.uniform(3, 90)), float(rng.uniform(3, 70))
    # clusters: consecutive points share a cell (like the pixels of a camera row near the camera), then jump
    centres = rng.uniform(0, 1, (max(1, n // int(rng.choice([1, 3, 20, 200]))), 2)) * [ext_x, ext_y]
    which = np.sort(rng.integers(0, len(centres), n)) if rng.random() < 0.7 else rng.integers(0, len(centres), n)
    jitter = rng.normal(0, float(rng.choice([0.0, 0.05, 0.8])), (n, 2))
    xy = centres[which] + jitter
    x, y = 1000.0 + xy[:, 0], 5000.0 + xy[:, 1]
    nb = int(rng.integers(1, 5))
    kind = rng.random()
    if kind < 0.45:
        vals = rng.integers(0, 256, (n, nb)).astype(np.float64)                    # image bytes
    elif kind < 0.6:
        vals = rng.integers(0, 65536, (n, nb)).astype(np.float64)                  # 16-bit samples
    elif kind < 0.8:
        vals = rng.uniform(-20, 300, (n, nb))
    elif kind < 0.9:
        vals = rng.uniform(-20, 300, (n, nb)).astype(np.float32).astype(np.float64)   # float32 values in float64 columns
    else:
        vals = 10.0 ** rng.uniform(-6, 6, (n, nb)) * rng.choice([-1.0, 1.0], (n, nb))  # mixed magnitudes: the ORDER of a sum shows
    if rng.random() < 0.4 and n > 3:
        vals[rng.integers(0, n, max(1, n // 20)), rng.integers(0, nb)] = np.nan
    res = float(rng.choice([0.5, 1.0, 2.0, 3.3]))
    agg = str(rng.choice(["mean", "max", "min", "median"]))
    interp = bool(rng.random() < 0.7)
    max_dist = float(rng.choice([0.5, 1.0, 2.0, 3.0, 3.0, 8.0, 9.0])) * res      # 8 sweeps: the fused tail's last; 9: the separate passes
    nodata = int(rng.choice([255, 0, 7]))
    df = pd.DataFrame({"x": x, "y": y, **{names[b]: vals[:, b] for b in range(nb)}})
    bands = [names[b] for b in rng.permutation(nb)]
    try:
        want, wb = orc.rasterize_points(x, y, df[bands].to_numpy(), res, interp, max_dist, agg, nodata)
    except ValueError:
        with_error = True
    else:
        with_error = False
    try:
        got, gb = prj.rasterize(df, resolution=res, bands=bands, interpolate=interp, max_dist=max_dist, agg_func=agg, nodata=nodata)
    except ValueError:
        assert with_error, "the device path refused what the oracle accepts"
        n_cases += 1
        continue
    assert not with_error, "the device path accepted what the oracle refuses"
    if gb != wb or not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        print(f"MISMATCH: n {n} nb {nb} res {res} agg {agg} interp {interp} max_dist {max_dist} nodata {nodata}: {len(bad)} bytes differ, first {bad[:3].tolist()} "
              f"got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}", flush=True)
        sys.exit(1)
    if rng.random() < 0.5:          # sharper than bytes: the float32 raster before the byte conversion, every bit
        wf, _ = orc.rasterize_points(x, y, df[bands].to_numpy(), res, interp, max_dist, agg, nodata, return_float=True)
        gf, _ = L.rasterize_points_f32(x, y, df[bands].to_numpy(), res, interp, max_dist, agg)
        same = (gf.view(np.uint32) == wf.view(np.uint32)) | (np.isnan(gf) & np.isnan(wf))
        if not same.all():
            bad = np.argwhere(~same)
            print(f"FLOAT32 MISMATCH: n {n} nb {nb} res {res} agg {agg} interp {interp} max_dist {max_dist}: {len(bad)} cells differ, first {bad[:3].tolist()} "
                  f"got {gf[tuple(bad[0])]!r} want {wf[tuple(bad[0])]!r}", flush=True)
            sys.exit(1)
        n_f32 += 1
    if rng.random() < 0.35:         # the same points as the surface pixels of a resident frame: ReverseProjection.rasterize
        off = np.array([732000.0, 1655.0, 4048000.0])
        w_img = int(rng.integers(max(1, -(-n // 30000)), 60))         # (a frame is at most 32768 pixels tall)
        h_img = -(-n // w_img) + int(rng.integers(0, 3))
        slots = np.sort(rng.choice(w_img * h_img, n, replace=False))
        raw = np.zeros((h_img * w_img, 3), dtype=np.float32)
        raw[slots, 0] = np.maximum(xy[:, 0] + 10.0, 0.5).astype(np.float32)      # channel 0 > 0: the pixel sees the surface
        raw[slots, 2] = (xy[:, 1] + 10.0).astype(np.float32)
        raw[slots, 1] = 5.0
        dtype = [np.uint8, np.uint8, np.uint16, np.float32, np.float64][int(rng.integers(0, 5))]
        arr = np.zeros((h_img * w_img, nb), dtype=dtype)
        if dtype == np.uint8:
            arr[slots] = rng.integers(0, 256, (n, nb))
        elif dtype == np.uint16:
            arr[slots] = rng.integers(0, 65536, (n, nb))
        else:
            arr[slots] = rng.uniform(-20, 300, (n, nb))
            if rng.random() < 0.4:
                arr[slots[rng.integers(0, n, max(1, n // 20))], rng.integers(0, nb)] = np.nan
        xd = raw[slots, 0].astype(np.float64) + off[0]
        yd = raw[slots, 2].astype(np.float64) + off[2]
        vd = arr[slots].astype(np.float64)
        chn = names[:nb]
        order = [chn.index(b) for b in bands]
        try:
            want_d, wb_d = orc.rasterize_points(xd, yd, vd[:, order], res, interp, max_dist, agg, nodata)
        except ValueError:
            want_d = None
        vert = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1], [1, 0, 1]], dtype=np.float32)
        with L.Mesh(vert, None, None, grid=(2, 2)) as m:
            m.load_image(raw.reshape(h_img, w_img, 3))
            rp = prj.ReverseProjection(m, off, w_img, h_img, False, None)
            try:
                got_d, gb_d = rp.rasterize(arr.reshape(h_img, w_img, nb), chn, resolution=res, bands=bands, interpolate=interp,
                                           max_dist=max_dist, agg_func=agg, nodata=nodata)
            except ValueError:
                assert want_d is None, "the device-fed path refused what the oracle accepts"
                got_d = None
        if got_d is not None:
            assert want_d is not None, "the device-fed path accepted what the oracle refuses"
            if gb_d != wb_d or not np.array_equal(got_d, want_d):
                bad = np.argwhere(got_d != want_d)
                print(f"DEVICE-FED MISMATCH: n {n} nb {nb} dtype {np.dtype(dtype).name} res {res} agg {agg} interp {interp} max_dist {max_dist}: "
                      f"{len(bad)} bytes differ, first {bad[:3].tolist()}", flush=True)
                sys.exit(1)
        n_dev += 1
    n_cases += 1
print(f"fuzz_rasterize: {n_dev} of the cases also as the surface pixels of a resident frame (ReverseProjection.rasterize; uint8 / uint16 / float32 / float64 images): byte-identical")
print(f"fuzz_rasterize: {n_cases} random cases, every raster byte-identical to the pandas / scipy restatement of the reference; "
      f"{n_f32} of them also compared as float32 rasters before the byte conversion: every bit equal")
