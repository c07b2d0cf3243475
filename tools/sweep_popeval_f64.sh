#!/bin/bash
# Development: the float64 population kernel, one build per variant (tools/build_variant_points.sh), each timed at
# 100 M x 2048 (tools/probe_popeval.py) and held against the reference's losses (tools/probe_popeval_parity.py).
#   tools/sweep_popeval_f64.sh OUT.txt NAME:"-Dflags" ...     (run on the GPU box from the repo root)
cd "$(dirname "$0")/.."
out=$1; shift
: > $out
for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    lib=$(tools/build_variant_points.sh f64_$name $flags | tail -1)
    echo "=== $name  ($flags)" >> $out
    grep -A6 "popeval_kernelIdLi1ENS_6PopCfgIdEELb0EdEE" build/abl/points_f64_$name.log | grep -E "VGPRs:|Occupancy|LDS Size|ScratchSize" | sed 's/.*remark: [^ ]* */    /' | tr '\n' ' ' >> $out
    echo >> $out
    ALPROJ_HIP_LIB=$lib python3 tools/probe_popeval_parity.py gpurun_out/f64_parity_$name.npz 2>&1 | tail -9 >> $out
    env ALPROJ_HIP_LIB=$lib ALP_POP_GRID=${GRID:-0} python3 tools/probe_popeval.py ${NPTS:-100000000} ${POP:-2048} 3 f64 2>&1 | tail -1 >> $out
done
python3 - >> $out <<'PY'
import glob, numpy as np, os
files = sorted(glob.glob("gpurun_out/f64_parity_*.npz"))
if files:
    base = np.load(files[0])
    print(f"--- losses of each build against {os.path.basename(files[0])} (max relative difference over all sets; wild sets apart)")
    for f in files[1:]:
        d = np.load(f)
        gcp = max(np.abs(d[k] / base[k] - 1).max() for k in d.files if not k.startswith("wild"))
        wild = max(np.abs(d[k] / base[k] - 1).max() for k in d.files if k.startswith("wild"))
        print(f"{os.path.basename(f):40s} GCP-like {gcp:.3e}   wild {wild:.3e}")
PY
cat $out
