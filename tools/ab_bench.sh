#!/bin/bash
# bench line (frames/s, per-pass device ms) of every librdr_hip*.so in renderer_amd/lib
for path in renderer_amd/lib/librdr_hip*.so; do
  lib=$(basename $path)
  RDR_HIP_LIB=$PWD/$path timeout -k 10 200 python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-rest > gpurun_out/ab_$lib.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.load(open("gpurun_out/ab_$lib.json"))
print("$lib", round(d["value"],1), {k: round(v, 3) for k, v in d["device_ms"].items()})
PY
done
