#!/bin/bash
# Development: durations of every kernel of a frame, in launch order, plus the frame total.
#   tools/probe_frame.sh TAG [N] [env assignments...]
cd "$(dirname "$0")/.."
tag=$1; n=${2:-100000000}; shift; shift
export TMPDIR=/tmp
for a in "$@"; do export "$a"; done
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/frame_$tag -o p -- python3 tools/probe_raster.py $n 5 > gpurun_out/frame_$tag.log 2>&1 </dev/null
python3 - <<PY
import sqlite3
db = sqlite3.connect("gpurun_out/frame_$tag/p_results.db")
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# the last frame: from the last fillBuffer of 168 MB (vis clear) on
names = [r[0].split("(")[0].replace("void ", "").replace("alp::", "") for r in rows]
last = max(i for i, n in enumerate(names) if n.startswith("resolve_kernel"))
first = max(i for i in range(last) if names[i].startswith("tile_plan") or names[i].startswith("raster_kernel"))
while first > 0 and "fillBuffer" in names[first - 1]:
    first -= 1
tot = 0.0
for i in range(first, last + 1):
    d = (rows[i][2] - rows[i][1]) / 1e3
    tot += d
    print(f"   {names[i][:40]:40s} {d:8.1f} us")
print(f"   sum of kernels {tot:.1f} us; first start to last end {(rows[last][2] - rows[first][1]) / 1e3:.1f} us")
PY
tail -1 gpurun_out/frame_$tag.log
