#!/usr/bin/env python3
"""Development probe: how far the float64 population kernel of the loaded library (ALPROJ_HIP_LIB) is from the REFERENCE's
own losses (tests/golden/g5_population.npz: CMAOptimizer._loss_function of the reference on GCP-like and "wild" points)
-- one line per set, and the losses themselves to a file so that two builds can be compared with each other.
   python3 tools/probe_popeval_parity.py [out.npz]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alproj_amd import _lib as L            # noqa: E402
from oracle import ref_numpy as orc         # noqa: E402  (development probe: the checker's parameter packing only)

g = np.load(os.path.join(ROOT, "tests", "golden", "g5_population.npz"))
L.init(0)
init = orc.vector_to_params(g["params_init"])
out = {}
worst = 0.0
for pts_key, obs_key, names in (("xyz", "uv_obs", ("d9", "d12", "d21")), ("wild_xyz", "wild_uv_obs", ("wild_d21",))):
    with L.Points(g[pts_key], [init["x"], init["y"], init["z"]], "f64") as pts:
        pts.set_observed(g[obs_key])
        for name in names:
            base = name.replace("wild_", "")
            tgt = [str(t) for t in g[f"{base}_targets"]]
            b = g[f"{base}_bounds"]
            cand = np.tile(L.params_vector(init), (len(g[f"{base}_X"]), 1))
            cand[:, [L.PARAM_KEYS.index(t) for t in tgt]] = g[f"{base}_X"] * (b[:, 1] - b[:, 0]) + b[:, 0]
            for tag, kind, fs in (("md", L.LOSS_MEAN_DIST, 0.0), ("hub", L.LOSS_HUBER, 10.0)):
                losses, amin = pts.eval_population(cand, kind, fs)
                ref = g[f"{name}_{tag}"]
                rel = np.abs(losses - ref) / np.abs(ref)
                out[f"{name}_{tag}"] = losses
                worst = max(worst, rel.max()) if not name.startswith("wild") else worst
                print(f"{name:9s} {tag:3s}: max |rel err| vs the reference {rel.max():.3e}  median {np.median(rel):.3e}  "
                      f"argmin {'same' if amin == int(np.argmin(ref)) else 'DIFFERENT'}")
print(f"worst over the GCP-like sets: {worst:.3e}")
if len(sys.argv) > 1:
    np.savez(sys.argv[1], **out)
