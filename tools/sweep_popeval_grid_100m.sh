for g in "" "16276,16" "8138,16" "4069,16" "2035,16" "32552,16" "8138,8" "8138,4" "8138,2"; do
  echo "== grid '$g'"
  if [ -z "$g" ]; then python tools/probe_popeval.py 100000000 2048 3 f32 2>&1 | tail -2; else ALP_POP_GRID=$g python tools/probe_popeval.py 100000000 2048 3 f32 2>&1 | tail -2; fi
done
for g in "" "12208,16" "6104,16" "3052,16" "1526,16"; do
  echo "== lens-free grid '$g'"
  if [ -z "$g" ]; then python tools/probe_popeval.py 100000000 2048 3 f32 d9 2>&1 | tail -2; else ALP_POP_GRID=$g python tools/probe_popeval.py 100000000 2048 3 f32 d9 2>&1 | tail -2; fi
done
