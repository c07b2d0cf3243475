#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (results.db): per kernel name the number of dispatches and the
average / min / max duration (what `--stats` prints), and, when counters were collected (--pmc), the
mean counter value per dispatch.
   python3 tools/rocpd_summary.py gpurun_out/prof/x_results.db [substring-filter] [--csv out.csv]"""
import collections
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
csv_out = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
if csv_out in args:
    args.remove(csv_out)
db = sqlite3.connect(args[0])
flt = args[1] if len(args) > 1 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end from kernels").fetchall() if {"name", "start", "end"} <= set(cols) else []
agg = collections.defaultdict(list)
for name, s, e in rows:
    if flt in name:
        agg[name].append((e - s) / 1e3)
lines = ["kernel,calls,avg_us,min_us,max_us,total_us"]
for name, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    short = name.split("(")[0][:90]
    lines.append(f"\"{short}\",{len(d)},{sum(d) / len(d):.2f},{min(d):.2f},{max(d):.2f},{sum(d):.2f}")
print("\n".join(lines))
try:
    ccols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    if ccols:
        q = db.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
        if q:
            lines.append("kernel,counter,mean_per_dispatch,dispatches")
            print("kernel,counter,mean_per_dispatch,dispatches")
        for kn, cn, tot, nd in q:
            if flt in kn:
                line = f"\"{kn.split('(')[0][:90]}\",{cn},{tot / max(nd, 1):.0f},{nd}"
                lines.append(line)
                print(line)
except sqlite3.Error as e:
    print("no counter table:", e)
if "--per-dispatch" in sys.argv:
    q = db.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection "
                   "group by dispatch_id, counter_name order by dispatch_id").fetchall()
    last = None
    for kn, did, cn, val in q:
        if flt in kn:
            if did != last:
                print(f"dispatch {did} {kn.split('(')[0][:60]}")
                last = did
            print(f"    {cn:36s} {val:16.0f}")
if csv_out:
    open(csv_out, "w").write("\n".join(lines) + "\n")
