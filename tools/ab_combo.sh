#!/bin/bash
# bench line under combinations of environment settings, alternated: tools/ab_combo.sh REPS "A=1 B=2" "C=3" ...  ("-" = nothing set)
REPS=$1; shift
for rep in $(seq 1 $REPS); do
  i=0
  for combo in "$@"; do
    i=$((i+1))
    ( if [ "$combo" != "-" ]; then export $combo; fi
      timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/abc_${i}_$rep.json 2>/dev/null ) || exit 1
    python3 - <<PY
import json
d=json.load(open("gpurun_out/abc_${i}_$rep.json"))
print("$combo", round(d["value"],1), round(d["device_ms"]["shade_kernel"],3), round(d["device_ms"]["raster_kernel"],3))
PY
  done
done
