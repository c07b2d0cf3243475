#!/bin/bash
# Development: kernel time of raster_grid_kernel for the stage-stop variants (build/abl/libalproj_stopK.so)
# and the full library, under rocprofv3 --kernel-trace.  Usage: tools/probe_raster_stages.sh TAG [N]
cd "$(dirname "$0")/.."
tag=$1; n=${2:-100000000}
export TMPDIR=/tmp
for v in stop1 stop2 stop3 full; do
  lib=build/abl/libalproj_$v.so; [ $v = full ] && lib=alproj_amd/libalproj_hip.so
  ALPROJ_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d gpurun_out/stages_$tag -o $v -- python3 tools/probe_raster.py $n 4 > gpurun_out/stages_${tag}_$v.log 2>&1 </dev/null
  echo "== $v"; python3 tools/rocpd_summary.py gpurun_out/stages_$tag/${v}_results.db raster_grid 2>&1 | grep -v "^kernel" 
done
