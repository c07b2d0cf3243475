#!/usr/bin/env python3
"""Development probe of the render path: N-vertex synthetic DSM onto the 5616x3744 frame.
   python3 tools/probe_raster.py [N] [reps] [explicit|implicit|shuffled] [distorted]
   (shuffled: the index array's rows in random order -- an array without locality; ALP_INDEX_LDS=0|1 forces either index kernel)"""
import os
os.environ.setdefault("ALP_NO_VIS_CACHE", "1")     # every frame of a probe is drawn (no visibility cache)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import synthetic as syn     # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "implicit"
distorted = len(sys.argv) > 4

L.init(0)
n = syn.grid_side(N)
t = time.time()
s = syn.surface(n)
print(f"surface {n}x{n} generated in {time.time() - t:.1f}s", flush=True)
p = syn.base_params(n)
if distorted:
    p.update(k1=-0.05, k2=0.01, a1=1.02, a2=0.98, p1=1e-3, p2=-2e-3)
pv = L.params_vector(p)
t = time.time()
if mode in ("explicit", "shuffled"):
    os.environ["ALP_NO_GRID_DETECT"] = "1"     # time the index-array kernel itself
    ind = syn.grid_indices(n, np.int32)
    if mode == "shuffled":
        ind = ind[np.random.default_rng(0).permutation(len(ind))]
    mesh = L.Mesh(s["vert"], None, ind)
    del ind
else:
    mesh = L.Mesh(s["vert"], None, None, grid=(n, n))
print(f"mesh upload {time.time() - t:.2f}s", flush=True)
best = 1e9
for r in range(reps):
    L.event_record(0)
    mesh.render_enqueue(pv, s["offsets"])
    L.event_record(1)
    L.synchronize()
    ms = L.event_elapsed_ms(0, 1)
    best = min(best, ms)
    print(f"rep {r}: {ms:.3f} ms", flush=True)
img = mesh.fetch()
if mode != "implicit":
    import hashlib
    print("image sha256", hashlib.sha256(img.tobytes()).hexdigest()[:16], "ALP_INDEX_LDS =", os.environ.get("ALP_INDEX_LDS"))
print(f"N={n * n} T={2 * (n - 1) ** 2} {mode}: best {best:.3f} ms/frame  {n * n / best / 1e6:.2f} Gvertices/s  "
      f"covered {float((img[:, :, 0] > 0).mean()):.3f}")
