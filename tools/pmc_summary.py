#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection.csv: per kernel name, mean counter value per dispatch."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    if flt and flt not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(disp[k])
    print(k[:110], f"(dispatches {n})")
    for c, val in sorted(v.items()):
        print(f"    {c:32s} {val / n:16.0f}")
