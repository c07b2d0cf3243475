#!/bin/bash
# Development: the lens-free population kernel, one prebuilt library per variant (tools/build_variant_points.sh lf_* -DPOP_V_LF=..),
# timed at BASELINE config 3's shape (10 M x 256) and at 100 M x 2048, D = 9 (tools/probe_popeval.py ... d9).
#   tools/sweep_popeval_lens_free.sh OUT.txt     (run on the GPU box from the repo root)
cd "$(dirname "$0")/.."
out=$1
: > $out
run() { # name lib precision
    echo "=== $1" >> $out
    env ALPROJ_HIP_LIB=$2 python3 tools/probe_popeval.py 10000000 256 8 $3 d9 2>&1 | tail -2 >> $out
    env ALPROJ_HIP_LIB=$2 python3 tools/probe_popeval.py 100000000 2048 3 $3 d9 2>&1 | tail -1 >> $out
}
run "shipped f32" alproj_amd/libalproj_hip.so f32
for v in 6 10 12 16; do run "f32 V=$v" build/abl/libalproj_lf_v$v.so f32; done
run "f32 TC=64" build/abl/libalproj_lf_tc64.so f32
run "shipped f64" alproj_amd/libalproj_hip.so f64
for v in 4 8 10; do run "f64 V=$v" build/abl/libalproj_lf_vd$v.so f64; done
cat $out
