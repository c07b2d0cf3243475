#!/usr/bin/env python3
"""Development (round 6): how many candidates of a CMA-ES generation have a non-finite loss, and what the kernel's second walk
(popeval_kernel: a wave re-walks its share for candidates whose sum is not finite) costs in those generations.
   python3 tools/probe_cma_nonfinite.py [N] [pop] [dims 9|21] [generations] [precision]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                # noqa: E402
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import synthetic as syn     # noqa: E402
from alproj_amd.cma import CMA              # noqa: E402
from alproj_amd.optimize import bounds_to_array  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pop = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dims = int(sys.argv[3]) if len(sys.argv) > 3 else 21
gens = int(sys.argv[4]) if len(sys.argv) > 4 else 12
prec = sys.argv[5] if len(sys.argv) > 5 else "f32"
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
targets = syn.TARGETS_D21 if dims == 21 else syn.TARGETS_D9
bounds = bounds_to_array(base, targets)
lower, upper = bounds[:, 0], bounds[:, 1]
cols = [L.PARAM_KEYS.index(t) for t in targets]
opt = CMA(mean=np.full(len(targets), 0.5), sigma=1.0, bounds=np.column_stack([np.zeros(len(targets)), np.ones(len(targets))]),
          population_size=pop, n_max_resampling=100, seed=1234, sampler=L.cma_sample)
for g in range(gens):
    X = opt.ask_population()
    cand = np.tile(L.params_vector(base), (pop, 1))
    cand[:, cols] = X * (upper - lower) + lower
    losses, _ = pts.eval_population(cand, L.LOSS_HUBER, 10.0, want_argmin=False)
    ms, _ = pts.eval_population_timing()
    print(f"generation {g}: kernel {ms:8.3f} ms  {pts.eval_population_info()}  inf {int(np.isinf(losses).sum()):5d}  NaN {int(np.isnan(losses).sum()):5d}  "
          f"of {pop}; best {np.nanmin(losses):.4g}", flush=True)
    opt.tell_population(X, losses)
