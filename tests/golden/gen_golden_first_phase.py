#!/usr/bin/env python3
"""g19_first_phase.npz: the reference's FIRST optimisation phase, evaluated by the reference itself (build container only).

    python tests/golden/gen_golden_first_phase.py

example.py:19-22 + :51-54: a camera WITHOUT lens coefficients (k1..k6 = p1 = p2 = s1..s4 = 0, a1 = a2 = 1) and the targets
x, y, z, fov, pan, tilt, roll, a1, a2.  Every candidate of such a population is lens-free, which is what the round-6 kernel
variant (alp_eval_population_info: ALP_POP_LENS_FREE) keys on; g5's populations all sit around a camera WITH a lens.  Stored:
1 127 GCP-like points, their noisy observations, the (P, 9) normalised candidates (P = 140: two candidate tiles, the second
ragged; row 0 = the initial camera, rows 3 and 7 identical: a tie), the bounds `bounds_to_array` gave, and the losses
`CMAOptimizer._loss_function` (optimize.py:329-357) returned for every row with both losses.  Data only."""
import os
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import BASE, gcp_like, load_reference, pvec  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    opt, _ = load_reference()
    rng = np.random.default_rng(20261005)
    init = dict(BASE)                                   # example.py:19-22: no lens
    truth = dict(BASE, x=BASE["x"] + 5, y=BASE["y"] - 7, z=BASE["z"] + 3, fov=71.0, pan=98.0, tilt=2.0, roll=-1.0, a1=1.02, a2=0.98)
    pts = gcp_like(opt, rng, 1127, truth)
    uv_obs = opt.project(pd.DataFrame(pts, columns=["x", "y", "z"]), truth).to_numpy() + rng.normal(0, 1.0, (len(pts), 2))
    tgt = ["x", "y", "z", "fov", "pan", "tilt", "roll", "a1", "a2"]
    o = opt.CMAOptimizer(pd.DataFrame(pts, columns=["x", "y", "z"]), pd.DataFrame(uv_obs, columns=["u", "v"]), dict(init))
    o.set_target(tgt)
    bounds = opt.bounds_to_array(o.params_init, tgt, None)
    P = 140
    X = rng.uniform(0.4, 0.6, (P, len(tgt)))
    X[0] = 0.5
    X[7] = X[3]
    X[11] = (np.array([truth[k] for k in tgt]) - bounds[:, 0]) / (bounds[:, 1] - bounds[:, 0])     # the truth: the winner
    g = dict(xyz=pts, uv_obs=uv_obs, params_init=pvec(init), X=X, bounds=bounds, targets=np.array(tgt))
    for tag, fs in (("md", None), ("hub", 10.0)):
        f = o._loss_function(bounds, fs)
        g[tag] = np.array([f(x) for x in X])
    np.savez(f"{OUT}/g19_first_phase.npz", **g)
    print("g19_first_phase: P", P, "argmin md", int(np.argmin(g["md"])), "hub", int(np.argmin(g["hub"])), "losses", g["md"][:4])


if __name__ == "__main__":
    main()
