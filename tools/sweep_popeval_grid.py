#!/usr/bin/env python3
"""Development: launch-shape sweep of the population kernel for populations of a few candidate tiles
(ALP_POP_GRID = "stripes,ytiles"; the losses do not depend on it).   python3 tools/sweep_popeval_grid.py [N] [P ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import synthetic as syn     # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Ps = [int(a) for a in sys.argv[2:]] or [256]
L.init(0)
n_side = syn.grid_side(N)
s = syn.surface(n_side)
xyz = syn.vert_to_xyz_local(s["vert"])
base = syn.local_params(syn.standoff_params(n_side), s["offsets"])
truth = syn.local_params(syn.perturbed(syn.standoff_params(n_side)), s["offsets"])
pts = L.Points(xyz, [base["x"], base["y"], base["z"]], "f32")
pts.project(L.params_vector(truth))
u, v = pts.fetch(np.float32)
obs = np.stack([u, v], 1) + np.random.default_rng(1).normal(0, 1, (len(u), 2)).astype(np.float32)
obs[~np.isfinite(obs)] = 0
pts.set_observed(obs)
rows = (len(xyz) + 255) // 256
for P in Ps:
    rng = np.random.default_rng(0)
    cols = [L.PARAM_KEYS.index(t) for t in syn.TARGETS_D9]
    cand = np.tile(L.params_vector(base), (P, 1))
    cand[:, cols] += rng.uniform(-0.1, 0.1, (P, 9)) * np.array([30, 30, 30, 45, 45, 45, 45, 0.2, 0.2])
    tiles = (P + 127) // 128
    G = int(os.environ.get("SWEEP_GROUP_ROWS", "6"))      # rows of a full group: 6 for the general variant, 8 for the lens-free one
    shapes = [None] + [(1024 * r, 1) for r in (1, 2, 3, 4)] + [((rows + G * k - 1) // (G * k), tiles) for k in (1, 2, 3, 4, 6)] + \
             [((rows + G * k - 1) // (G * k), 1) for k in (1, 2, 3, 4)]
    ref = None
    for shp in shapes:
        if shp is None:
            os.environ.pop("ALP_POP_GRID", None)
        else:
            os.environ["ALP_POP_GRID"] = f"{shp[0]},{shp[1]}"
        best = 1e9
        for r in range(6):
            L.event_record(0)
            pts.eval_population_enqueue(cand, L.LOSS_HUBER, 10.0)
            L.event_record(1)
            losses, amin = pts.eval_population_wait(P)
            best = min(best, L.event_elapsed_ms(0, 1))
        if ref is None:
            ref = losses
        dev = np.nanmax(np.abs(losses - ref) / np.abs(ref))
        print(f"N={len(xyz)} P={P} {pts.eval_population_info()} grid={'shipped' if shp is None else shp}: {best:.3f} ms  {len(xyz) * P / best / 1e6:.0f} G evals/s  "
              f"(losses vs shipped: {dev:.1e})", flush=True)
