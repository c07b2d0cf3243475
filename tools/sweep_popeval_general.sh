#!/bin/bash
# Development (round 6): the general float32 population kernel under the stripes x tiles grid, one prebuilt library per
# points-per-lane / occupancy / tile variant (tools/build_variant_points.sh g_* -DPOP_V=.. -DPOP_MINW=.. -DPOP_TC=..),
# timed at 100 M x 2048 and at 10 M x 2048 (tools/probe_popeval.py).   tools/sweep_popeval_general.sh OUT.txt
cd "$(dirname "$0")/.."
out=$1
: > $out
run() {
    echo "=== $1" >> $out
    env ALPROJ_HIP_LIB=$2 python3 tools/probe_popeval.py 100000000 2048 3 f32 2>&1 | tail -2 >> $out
    env ALPROJ_HIP_LIB=$2 python3 tools/probe_popeval.py 10000000 2048 4 f32 2>&1 | tail -1 >> $out
}
run "shipped V=6 TC=128 4 waves" alproj_amd/libalproj_hip.so
for n in g_v4w6 g_v5w5 g_v7w4 g_v8w3 g_tc64 g_v6w5 g_v4w4; do
    [ -f build/abl/libalproj_$n.so ] && run "$n" build/abl/libalproj_$n.so
done
cat $out
