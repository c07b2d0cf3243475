#!/bin/bash
# Development (round 4): raster_parked_tiles_kernel (-DPARKED_TILES_LAB) against raster_parked_kernel -- stage-skipping builds and
# hardware counters; results: profiles/r04_parked_tiles_lab.txt.  Build first (hipcc only):
#   for v in "nocells -DPT_SKIP_CELLS" "nosmall -DPT_SKIP_SMALL" "nolarge -DPT_SKIP_LARGE" "nowalk -DPT_SKIP_CELLS -DPT_SKIP_SMALL -DPT_SKIP_LARGE" \
#            "wg5 -DPARKED_TILES_WGS_PER_CU=5" "lab"; do set -- $v; n=$1; shift; tools/build_variant.sh pt_$n -DPARKED_TILES_LAB "$@"; done
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r04c
out=gpurun_out/r04c
for variant in nocells nosmall nolarge nowalk wg5; do
  ALPROJ_HIP_LIB=build/abl/libalproj_pt_$variant.so tools/probe_frame.sh r04c_$variant 100000000 ALP_PARKED=tiles 2>&1 | grep "parked_tiles\|sum of" | sed "s/^/[$variant] /"
done
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU" \
           "TCC_ATOMIC_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  for mode in waves tiles; do
    ALPROJ_HIP_LIB=build/abl/libalproj_pt_lab.so ALP_PARKED=$mode timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/pmc_${mode}_$i -o p -- python3 tools/probe_raster.py 100000000 3 > $out/pmc_${mode}_$i.log 2>&1 </dev/null
    python3 - <<PY
import sqlite3
db = sqlite3.connect("$out/pmc_${mode}_$i/p_results.db")
rows = db.execute("select dispatch_id, counter_name, sum(value), max(duration) from counters_collection where kernel_name like '%raster_parked%' group by dispatch_id, counter_name order by dispatch_id").fetchall()
big = max((r[3] for r in rows), default=0)
seen = set()
for d, c, v, dur in rows:
    if dur > 0.5 * big and c not in seen:
        seen.add(c)
        print(f"[$mode] {c:32s} {v:16.0f}   (dispatch {d}, {dur/1000:.1f} us)")
PY
  done
done
rm -rf $out/pmc_*/  gpurun_out/frame_r04c_*/
