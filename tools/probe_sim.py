#!/usr/bin/env python3
"""Development: render time with a per-vertex value array (sim_image's colours) next to the coordinate render."""
import os, sys
os.environ.setdefault("ALP_NO_VIS_CACHE", "1")     # every frame of a probe is drawn (no visibility cache)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L
from alproj_amd import synthetic as syn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
L.init(0)
n = syn.grid_side(N)
s = syn.surface(n)
p = syn.base_params(n)
pv = L.params_vector(p)
col = np.random.default_rng(1).random((n * n, 3), dtype=np.float32)
for name, value in (("coordinates (value = vert)", None), ("colours", col)):
    mesh = L.Mesh(s["vert"], value, None, grid=(n, n))
    best = 1e9
    for r in range(6):
        L.event_record(0)
        mesh.render_enqueue(pv, s["offsets"])
        L.event_record(1)
        L.synchronize()
        best = min(best, L.event_elapsed_ms(0, 1))
    print(f"{name}: best {best:.3f} ms/frame")
    mesh.close()
