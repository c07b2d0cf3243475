#!/bin/bash
# Development: per-kernel durations (+ SQ_INSTS_VALU) of one render configuration under rocprofv3.
#   tools/probe_raster_prof.sh TAG [N] [env assignments...]
cd "$(dirname "$0")/.."
tag=$1; n=${2:-100000000}; shift; shift
export TMPDIR=/tmp
for a in "$@"; do export "$a"; done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d gpurun_out/rprof_$tag -o p -- python3 tools/probe_raster.py $n 6 > gpurun_out/rprof_$tag.log 2>&1 </dev/null
python3 tools/rocpd_summary.py gpurun_out/rprof_$tag/p_results.db 2>&1 | grep -v "fillBuffer\|copyBuffer" | head -30
tail -1 gpurun_out/rprof_$tag.log
