#!/usr/bin/env python3
"""HBM traffic of the kernels of one probe, from the PMC counters, as MI355X_MICROARCH.md prescribes:
two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE costs 3 TCC counters, WRITE_SIZE 2: they do not fit one
pass), FETCH_SIZE doubled on gfx950 (128-B requests are tallied at 64 B), WRITE_SIZE as it is; both are
reported in KiB.  The doubling is calibrated for wide coalesced streaming reads; other access widths
(the gathers and 8-byte atomics of the render) are uncalibrated, which the output says.

    python3 tools/pmc_traffic.py OUT.json FRAME_DIVISOR KERNEL_FILTER -- python3 tools/probe_raster.py 100000000 3

FRAME_DIVISOR = how many frames / launches-of-interest the probe makes (per-frame totals = sum / divisor);
KERNEL_FILTER = comma-separated substrings of kernel names to keep ("" = all but runtime copies/fills).
"""
import collections
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

out_json, divisor, flt = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cmd = sys.argv[sys.argv.index("--") + 1:]
filters = [f for f in flt.split(",") if f]
res = {}
# where the profiler writes: PMC_TRAFFIC_DIR (bench.py's live pass: a temporary directory outside the checkout, which bench.py
# removes whatever happens), else gpurun_out/ of the checkout (a hand-run whose output one wants to keep and look at)
keep = "PMC_TRAFFIC_DIR" not in os.environ
base = os.environ.get("PMC_TRAFFIC_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(base, exist_ok=True)
env = dict(os.environ)
env.setdefault("TMPDIR", "/tmp")
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = tempfile.mkdtemp(prefix="pmc_", dir=base)
    try:
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--"] + cmd, check=True, env=env,
                       stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        db = sqlite3.connect(os.path.join(d, "p_results.db"))
        try:
            q = db.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name = ? "
                           "group by kernel_name", (counter,)).fetchall()
        finally:
            db.close()
    finally:
        if not keep:
            shutil.rmtree(d, ignore_errors=True)
    for kn, tot, nd in q:
        short = kn.split("(")[0].replace("void ", "")
        if "rocclr" in short and not filters:
            continue
        if filters and not any(f in short for f in filters):
            continue
        res.setdefault(short, {})[counter + "_KiB_total"] = tot
        res[short]["dispatches"] = nd
frame = collections.OrderedDict()
total = 0.0
for k, v in sorted(res.items()):
    fetch = v.get("FETCH_SIZE_KiB_total", 0.0) * 1024 * 2        # gfx950 correction
    write = v.get("WRITE_SIZE_KiB_total", 0.0) * 1024
    frame[k] = {"dispatches_in_probe": v["dispatches"], "fetch_bytes_per_frame_x2_corrected": fetch / divisor,
                "write_bytes_per_frame": write / divisor, "hbm_bytes_per_frame": (fetch + write) / divisor}
    total += (fetch + write) / divisor
doc = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) -- " + " ".join(cmd),
       "frames_or_launches_in_probe": divisor,
       "correction": "gfx950: FETCH_SIZE x 2 (128-B requests tallied at 64 B; calibrated for wide coalesced streaming reads, "
                     "uncalibrated for narrower gathers / 8-byte atomics), WRITE_SIZE exact for 16-B-per-lane stores "
                     "(MI355X_MICROARCH.md, section HBM); separate --pmc passes",
       "kernels": frame, "hbm_bytes_per_frame": total}
json.dump(doc, open(out_json, "w"), indent=1)
print(json.dumps(doc, indent=1))
