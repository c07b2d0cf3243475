#!/usr/bin/env python3
"""Development: host-side cost of one CMA-ES generation at pop P (tiny point set, so the kernel is
negligible): ask, candidate matrix, eval_population (fold + H2D + launch + D2H + wait), tell."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import synthetic as syn     # noqa: E402
from alproj_amd.cma import CMA              # noqa: E402
from alproj_amd.optimize import bounds_to_array   # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L.init(0)
n = 64
s = syn.surface(n)
xyz = syn.vert_to_xyz_local(s["vert"])
base = syn.local_params(syn.standoff_params(n), s["offsets"])
pts = L.Points(xyz, [base["x"], base["y"], base["z"]], "f32")
pts.set_observed(np.zeros((len(xyz), 2), np.float32))
targets = syn.TARGETS_D21
b = bounds_to_array(base, targets)
lo, hi = b[:, 0], b[:, 1]
cols = [L.PARAM_KEYS.index(t) for t in targets]
basev = L.params_vector(base)
opt = CMA(mean=np.full(21, 0.5), sigma=1.0, bounds=np.column_stack([np.zeros(21), np.ones(21)]), population_size=P,
          n_max_resampling=100, seed=1)
T = np.zeros(4)
for g in range(12):
    t0 = time.perf_counter(); X = opt.ask_population()
    t1 = time.perf_counter(); cand = np.tile(basev, (P, 1)); cand[:, cols] = X * (hi - lo) + lo
    t2 = time.perf_counter(); losses, amin = pts.eval_population(cand, L.LOSS_HUBER, 10.0)
    t3 = time.perf_counter(); opt.tell_population(X, losses)
    t4 = time.perf_counter()
    if g >= 4:
        T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]
print(f"P={P}: ask {T[0]/8*1e3:.2f} ms, candidate matrix {T[1]/8*1e3:.2f}, eval (fold+copies+launch+wait) {T[2]/8*1e3:.2f}, "
      f"tell {T[3]/8*1e3:.2f}  -> {T.sum()/8*1e3:.2f} ms per generation of host-side work")
