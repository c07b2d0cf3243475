#!/usr/bin/env python3
"""Development: where a CMA-ES generation goes at the REFERENCE's own problem size (1127 GCPs, pop 50, D 9, float64 point set:
docs/usage.md:335, example.py:51-54), where the device work is microseconds: ask, candidate matrix, eval_population (fold + copies +
two launches + wait; the kernels alone from the library's HIP events), tell."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import synthetic as syn     # noqa: E402
from alproj_amd.cma import CMA              # noqa: E402
from alproj_amd.optimize import bounds_to_array   # noqa: E402

P, D = 50, 9
L.init(0)
tp = syn.truth_params(316)
gx = syn.gcp_points(1127, tp, seed=3)
pts = L.Points(gx, [tp["x"], tp["y"], tp["z"]], "f64")
pts.project(L.params_vector(tp))
u, v = pts.fetch()
pts.set_observed(np.stack([u, v], 1) + np.random.default_rng(3).normal(0, 1.0, (1127, 2)))
targets = syn.TARGETS_D9
b = bounds_to_array(tp, targets)
lo, hi = b[:, 0], b[:, 1]
cols = [L.PARAM_KEYS.index(t) for t in targets]
basev = L.params_vector(tp)
opt = CMA(mean=np.full(D, 0.5), sigma=1.0, bounds=np.column_stack([np.zeros(D), np.ones(D)]), population_size=P, n_max_resampling=100, seed=1)
T = np.zeros(5)
G = 300
for g in range(G + 20):
    t0 = time.perf_counter(); X = opt.ask_population()
    t1 = time.perf_counter(); cand = np.tile(basev, (P, 1)); cand[:, cols] = X * (hi - lo) + lo
    t2 = time.perf_counter(); losses, amin = pts.eval_population(cand, L.LOSS_HUBER, 10.0, want_argmin=False)
    t3 = time.perf_counter(); opt.tell_population(X, losses)
    t4 = time.perf_counter()
    if g >= 20:
        T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, pts.eval_population_timing()[0] / 1e3]
T = T / G * 1e6
print(f"GCP scale (1127 points, pop {P}, D {D}, f64), us per generation over {G}: ask {T[0]:.0f}, candidate matrix {T[1]:.0f}, eval call {T[2]:.0f} "
      f"(of which the two kernels between HIP events {T[4]:.0f}), tell {T[3]:.0f}  -> {T[:4].sum():.0f} us; a graph of the call's 2 launches + 2 copies could remove "
      f"at most eval call - kernels - one wait = ~{max(T[2] - T[4] - 15, 0):.0f} us of it")
