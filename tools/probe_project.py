#!/usr/bin/env python3
"""Development probe: single-pose projection of an N-vertex synthetic DSM, kernel ms from HIP
events.  python3 tools/probe_project.py [N] [reps] [precision]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import synthetic as syn     # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
L.init(0)
n = syn.grid_side(N)
s = syn.surface(n)
xyz = syn.vert_to_xyz_local(s["vert"])
base = syn.local_params(syn.standoff_params(n), s["offsets"])
truth = syn.local_params(syn.perturbed(syn.standoff_params(n)), s["offsets"])
pts = L.Points(xyz, [base["x"], base["y"], base["z"]], prec)
pv = L.params_vector(truth)
best = 1e9
for r in range(reps):
    L.event_record(0)
    pts.project(pv)
    L.event_record(1)
    L.synchronize()
    best = min(best, L.event_elapsed_ms(0, 1))
bpv = 20 if prec == "f32" else 40
print(f"N={n * n} {prec}: best {best:.4f} ms  {n * n / best / 1e6:.1f} Gpts/s  {n * n * bpv / best / 1e6:.0f} GB/s "
      f"({n * n * bpv / best / 1e6 / 8000:.3f} of 8 TB/s)")
