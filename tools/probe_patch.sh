#!/bin/bash
# Development: render time of the 100 M-vertex frame by LDS patch size (near / far round).
for cfg in "0 0" "4096 1024" "4096 0" "0 1024" "2048 1024" "5632 1024" "4096 2048" "4096 512"; do
  set -- $cfg
  echo "== ALP_PATCH_NEAR=$1 ALP_PATCH_FAR=$2"
  ALP_PATCH_NEAR=$1 ALP_PATCH_FAR=$2 timeout 300 python3 tools/probe_raster.py 100000000 6 2>&1 | grep -E "best|Error|error"
done
