#!/bin/bash
# Development: hardware counters of raster_parked_kernel (first-round dispatch), several rocprofv3 --pmc passes.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INST_LEVEL_SMEM" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES SQ_IFETCH SQC_ICACHE_MISSES SQC_ICACHE_REQ"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/ppmc_$i -o p -- python3 tools/probe_raster.py ${1:-100000000} 3 > gpurun_out/ppmc_$i.log 2>&1 </dev/null
  python3 - <<PY
import sqlite3
db = sqlite3.connect("gpurun_out/ppmc_$i/p_results.db")
rows = db.execute("select dispatch_id, counter_name, sum(value), max(duration) from counters_collection where kernel_name like '%raster_parked%' group by dispatch_id, counter_name order by dispatch_id").fetchall()
big = max((r[3] for r in rows), default=0)
seen = set()
for d, c, v, dur in rows:
    if dur > 0.5 * big and c not in seen:
        seen.add(c)
        print(f"   {c:40s} {v:16.0f}   (dispatch {d}, {dur/1000:.1f} us)")
PY
done
