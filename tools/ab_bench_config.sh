#!/bin/bash
# bench.py --config N under every librdr_hip*.so in renderer_amd/lib: tools/ab_bench_config.sh N
for path in renderer_amd/lib/librdr_hip*.so; do
  lib=$(basename $path)
  RDR_HIP_LIB=$PWD/$path timeout -k 10 250 python bench.py --config $1 --no-cpu-baseline > gpurun_out/abc_$lib.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.load(open("gpurun_out/abc_$lib.json"))
print("$lib", round(d["value"],1), {k: round(v, 3) for k, v in d["device_ms"].items()})
PY
done
