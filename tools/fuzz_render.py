#!/usr/bin/env python3
"""Development: randomised differential test of the HIP render against the frozen C raster oracle -- random terrain grids,
cell sizes, camera poses (also low over the ground, tilted, rolled, wide), frame sizes, lens coefficients, and every way a
mesh can be handed over (implicit grid, full-grid index array int32 / int64 recognised on the host or on the device,
nodata-filtered index array, shuffled index array on the index kernels, user masks), each rendered twice (second time from
the visibility cache with another value source).  Visibility words must be identical, images equal to 1e-6.
   python3 tools/fuzz_render.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import synthetic as syn     # noqa: E402
from oracle import raster as orast         # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
L.init(0)
t_end = time.time() + budget
n_cases = n_pixels = 0
kinds = {}
while time.time() < t_end:
    n = int(rng.integers(20, 420))
    res = float(rng.choice([0.5, 1.0, 2.0, 4.0]))
    s = syn.surface(n, res=res, seed=int(rng.integers(1, 1 << 30)))
    vert = s["vert"]
    gh = gw = n
    if rng.random() < 0.35:                    # a non-square part of the grid (vertex id = row * gw + col)
        gh, gw = int(rng.integers(2, n + 1)), int(rng.integers(2, n + 1))
        vert = np.ascontiguousarray(vert.reshape(n, n, 3)[:gh, :gw].reshape(-1, 3))

    def rect_indices(dtype):
        a = (np.arange(gh - 1, dtype=np.int64)[:, None] * gw + np.arange(gw - 1, dtype=np.int64)[None, :]).reshape(-1)
        out = np.empty((len(a), 2, 3), dtype=np.int64)
        out[:, 0] = np.stack([a, a + gw, a + gw + 1], 1)
        out[:, 1] = np.stack([a, a + gw + 1, a + 1], 1)
        return out.reshape(-1, 3).astype(dtype)

    w = int(rng.integers(48, 1300)); h = int(rng.integers(40, 900))
    p = dict(syn.base_params(n, res), w=w, h=h, cx=w / 2.0, cy=h / 2.0)
    L_ = n * res
    p["x"] += float(rng.uniform(-0.1, 0.6) * L_)
    p["y"] += float(rng.uniform(-0.4, 0.4) * L_)
    p["z"] += float(rng.choice([-49.0, -45.0, -20.0, 0.0, 60.0, 300.0])) + float(rng.uniform(0, 3))
    p.update(pan=float(rng.uniform(0, 360)), tilt=float(rng.uniform(-60, 15)), roll=float(rng.uniform(-25, 25)),
             fov=float(rng.uniform(20, 89)))
    if rng.random() < 0.4:
        p.update(a1=float(rng.uniform(0.95, 1.05)), a2=float(rng.uniform(0.95, 1.05)), k1=float(rng.uniform(-0.1, 0.1)),
                 k2=float(rng.uniform(-0.02, 0.02)), p1=float(rng.uniform(-2e-3, 2e-3)), p2=float(rng.uniform(-2e-3, 2e-3)),
                 s1=float(rng.uniform(-1e-3, 1e-3)), s3=float(rng.uniform(-1e-3, 1e-3)))
    offsets = s["offsets"] if rng.random() < 0.8 else None
    if offsets is None:
        p = syn.local_params(p, s["offsets"])
    kind = str(rng.choice(["implicit", "full_i32", "full_i64_host", "full_i64_dev", "nodata", "shuffled", "masked"]))
    os.environ.pop("ALP_HOST_THREADS", None)
    ind = grid = valid = None
    ref_ind = None
    if kind == "implicit":
        grid = (gh, gw)
    elif kind.startswith("full"):
        ind = rect_indices(np.int32 if kind == "full_i32" else np.int64)
        os.environ["ALP_HOST_THREADS"] = "0" if kind == "full_i64_dev" else "3"
    elif kind == "nodata":
        full = rect_indices(np.int64)
        bad = rng.random(gh * gw) < 0.01
        ind = ref_ind = full[~bad[full].any(axis=1)]
        if len(ind) < 2:
            continue
    elif kind == "shuffled":
        full = rect_indices(np.int32)
        ind = ref_ind = full[rng.permutation(len(full))[: max(2, len(full) // 2)]]
    else:
        grid = (gh, gw)
        valid = rng.random(gh * gw) > 0.02
        full = rect_indices(np.int64)
        ref_ind = full[valid[full].all(axis=1)]
    col = syn.colors(gh * gw, seed=7)
    md = float(rng.uniform(5, 200)) if rng.random() < 0.3 else None
    ref_vis = orast.visibility(vert, ref_ind, p, offsets, grid=None if ref_ind is not None else (gh, gw))
    ref_img = orast.render(vert, col, ref_ind, p, offsets, md, grid=None if ref_ind is not None else (gh, gw))
    ref_crd = orast.render(vert, None, ref_ind, p, offsets, None, grid=None if ref_ind is not None else (gh, gw))
    with L.Mesh(vert.astype(np.float64) if rng.random() < 0.5 else vert, col, ind, grid) as m:
        if valid is not None:
            m.set_valid(valid)
        m.render_enqueue(L.params_vector(p), offsets, md)
        vis = m.fetch_visibility()
        img = m.fetch()
        m.render_enqueue(L.params_vector(p), offsets, None, coords=True)       # from the visibility cache
        crd = m.fetch()
        counts = m.frame_counts()
    hit = ref_vis != 0
    ok = np.array_equal(vis != 0, hit) and np.array_equal(vis[hit] >> np.uint64(32), ref_vis[hit] >> np.uint64(32))
    if ok and kind not in ("nodata", "masked"):
        ok = np.array_equal(vis, ref_vis)                # triangle numbering is the caller's for every other kind
    ok = ok and np.allclose(img, ref_img, rtol=1e-6, atol=1e-6) and np.allclose(crd, ref_crd, rtol=1e-6, atol=1e-6) and counts == (1, 1)
    n_cases += 1
    n_pixels += w * h
    kinds[kind] = kinds.get(kind, 0) + 1
    if not ok:
        print(f"MISMATCH: kind {kind} n {n} grid {gh}x{gw} res {res} frame {w}x{h} md {md} offsets {offsets is not None} counts {counts}\n  params {p}", flush=True)
        sys.exit(1)
print(f"fuzz_render: {n_cases} random scenes, {n_pixels / 1e6:.1f} M pixels, all identical to the oracle; by kind {kinds}")
