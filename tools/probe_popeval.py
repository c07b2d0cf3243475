#!/usr/bin/env python3
"""Micro-probe of the population-evaluation kernel (development tool, not a test):
N vertices x P candidates, prints kernel ms from HIP events.  Run it under rocprofv3 for
counters.   python3 tools/probe_popeval.py [N] [P] [reps] [precision]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import synthetic as syn     # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
prec = sys.argv[4] if len(sys.argv) > 4 else "f32"

L.init(0)
n_side = syn.grid_side(N)
s = syn.surface(n_side)
xyz = syn.vert_to_xyz_local(s["vert"])
base = syn.local_params(syn.standoff_params(n_side), s["offsets"])
truth = syn.local_params(syn.perturbed(syn.standoff_params(n_side)), s["offsets"])
pts = L.Points(xyz, [base["x"], base["y"], base["z"]], prec)
pts.project(L.params_vector(truth))
u, v = pts.fetch(np.float32)
obs = np.stack([u, v], 1) + np.random.default_rng(1).normal(0, 1, (len(u), 2)).astype(np.float32)
obs[~np.isfinite(obs)] = 0
pts.set_observed(obs)
rng = np.random.default_rng(0)
cols = [L.PARAM_KEYS.index(t) for t in syn.TARGETS_D21]
cand = np.tile(L.params_vector(base), (P, 1))
w = np.array([30, 30, 30, 45, 45, 45, 45] + [0.2] * 14)
if len(sys.argv) > 5 and sys.argv[5] == "distonly":
    w[:7] = 0          # distortion-only population: every candidate shares the pose
if len(sys.argv) > 5 and sys.argv[5] == "d9":
    w[9:] = 0          # the reference's first phase (x, y, z, fov, pan, tilt, roll, a1, a2): a lens-free population
cand[:, cols] += rng.uniform(-0.1, 0.1, (P, 21)) * w
best = 1e9
for r in range(reps):
    L.event_record(0)
    pts.eval_population_enqueue(cand, L.LOSS_HUBER, 10.0)
    L.event_record(1)
    losses, amin = pts.eval_population_wait(P)
    ms = L.event_elapsed_ms(0, 1)
    best = min(best, ms)
    print(f"rep {r}: {ms:.3f} ms")
ev = len(xyz) * P
print(f"variant {pts.eval_population_info()}")
print(f"N={len(xyz)} P={P} {prec}: best {best:.3f} ms  {ev / best / 1e6:.1f} G evals/s  "
      f"{ev * 77 / best / 1e9:.1f} TFLOP/s(77/eval)  finite losses: {np.isfinite(losses).sum()}/{P}")
