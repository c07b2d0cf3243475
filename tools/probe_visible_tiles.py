#!/usr/bin/env python3
"""Development: how many tiles of 64 x 16 cells win at least one pixel of the 100 M-vertex bench frame
(compare with the tiles the frame plan draws: ALP_RASTER_STATS prints those)."""
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
mesh = L.Mesh(s["vert"], None, None, grid=(n, n))
mesh.render_enqueue(L.params_vector(p), s["offsets"])
vis = mesh.fetch_visibility()
hit = vis != 0
tri = (0xFFFFFFFF - (vis[hit] & np.uint64(0xFFFFFFFF))).astype(np.int64)
cell = tri >> 1
r, c = cell // (n - 1), cell % (n - 1)
tiles_x = (n - 1 + 63) // 64
tile = (r // 16) * tiles_x + (c // 64)
ut, cnt = np.unique(tile, return_counts=True)
print(f"covered pixels {hit.sum()}, distinct winning triangles {len(np.unique(tri))}, tiles with at least one winning pixel {len(ut)}")
for th in (1, 4, 16, 64, 256):
    print(f"  tiles with >= {th} pixels: {(cnt >= th).sum()}")
