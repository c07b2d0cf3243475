#!/usr/bin/env python3
"""Development: the headline kernel's launches in a rocprofv3 --kernel-trace database of a bench.py run, BY LAUNCH SIZE -- the
per-kernel average of --stats mixes the 100 M-vertex launches (`value`) with the 10 M-vertex ones of the config-2 leg and the
probes; this lists each grid size apart so that the 100 M line can be held against roofline.kernel_ms.
   python3 tools/rocpd_headline_kernel.py gpurun_out/prof/p_results.db [kernel-substring]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "project_kernel<float>"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
size_cols = [c for c in ("grid_x", "grid_size_x", "grid_size", "workgroup_count_x") if c in cols]
groups = collections.defaultdict(list)
if size_cols:
    for name, s, e, g in db.execute(f"select name, start, end, {size_cols[0]} from kernels"):
        if flt in name:
            groups[int(g)].append((e - s) / 1e3)
else:                                         # no size column in this rocprofv3: cluster by duration (a 100 M launch is > 10 x a 10 M one)
    for name, s, e in db.execute("select name, start, end from kernels"):
        if flt in name:
            d = (e - s) / 1e3
            groups[10 ** len(str(int(d)))].append(d)
print(f"{flt}: launches by {'grid size (' + size_cols[0] + ')' if size_cols else 'duration decade (no grid column in this database)'}")
print("size,launches,avg_us,median_us,min_us,max_us")
for g, d in sorted(groups.items()):
    d.sort()
    print(f"{g},{len(d)},{sum(d) / len(d):.2f},{d[len(d) // 2]:.2f},{d[0]:.2f},{d[-1]:.2f}")
