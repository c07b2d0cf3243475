#!/usr/bin/env python3
"""Development (round 6, VERDICT r05 task 7): the REAL keys of to_geotiff's cell sort -- the 100 M-vertex frame's surface pixels in
pixel order -> cell = row * width + col of the 1 m raster (rz_cell_kernel's arithmetic) -- written to a file for
tools/sort_real_keys.hip, with a census of how sorted they already are.   python3 tools/probe_f2_keys.py OUT.bin [N]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import project as aproj     # noqa: E402
from alproj_amd import synthetic as syn     # noqa: E402

out = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
L.init(0)
n = syn.grid_side(N)
s = syn.surface(n)
cam = syn.base_params(n)
with aproj.reverse_proj_device(s["vert"], None, cam, s["offsets"], grid_shape=(n, n)) as rp:
    df = rp.to_frame(np.zeros((int(cam["h"]), int(cam["w"]), 3), np.uint8))
x, y = df["x"].to_numpy(), df["y"].to_numpy()
res = 1.0
x_min, x_max, y_min, y_max = x.min(), x.max(), y.min(), y.max()
width, height = int(np.ceil((x_max - x_min) / res)), int(np.ceil((y_max - y_min) / res))
col = np.clip(((x - x_min) / res).astype(np.int64), 0, width - 1)
row = np.clip(((y_max - y) / res).astype(np.int64), 0, height - 1)
cell = (row * width + col).astype(np.uint32)
cell_t = (col * height + row).astype(np.uint32)          # the transposed key: the same groups, another order
hw = width * height
bits = int(np.ceil(np.log2(hw)))
tile = (row >> 5) * ((width + 63) // 64) + (col >> 6)
tiles_used = np.unique(tile)
rank = np.searchsorted(tiles_used, tile)
compact = (rank * 2048 + (row & 31) * 64 + (col & 63)).astype(np.uint32)
cbits = int(np.ceil(np.log2(len(tiles_used) * 2048)))
with open(out, "wb") as f:
    np.array([len(cell), bits, cbits, 0], dtype=np.uint64).tofile(f)
    cell.tofile(f)
    cell_t.tofile(f)
    compact.tofile(f)


def census(name, k, kb):
    asc = float((k[1:] >= k[:-1]).mean())
    runs_in = int((k[1:] != k[:-1]).sum()) + 1
    distinct = len(np.unique(k))
    top = k >> max(0, kb - 9)
    blk = top[:len(top) // 8192 * 8192].reshape(-1, 8192)
    per_block = np.array([len(np.unique(b)) for b in blk[::16]])
    order = np.argsort(k, kind="stable")
    moved = np.abs(order - np.arange(len(k)))
    print(f"{name}: {len(k)} keys of {kb} bits; adjacent pairs in ascending order {asc:.4f}; runs of equal keys as they lie {runs_in} "
          f"(distinct keys = runs after the sort: {distinct}); distinct top-9-bit digits per block of 8192 consecutive keys: median "
          f"{np.median(per_block):.0f}, max {per_block.max()}; distance a key moves in the stable sort: median {np.median(moved):.0f}, "
          f"90 % {np.percentile(moved, 90):.0f}, max {moved.max()}")


print(f"raster {height} x {width} = {hw} cells ({bits} bits); tiles of 64 x 32 that hold a point: {len(tiles_used)} of "
      f"{((width + 63) // 64) * ((height + 31) // 32)} -> compacted key {cbits} bits")
census("row-major cell (shipped)", cell, bits)
census("column-major cell", cell_t, bits)
census("tile-rank compacted", compact, cbits)
