#!/bin/bash
# bench line of the library under several values of one environment variable: tools/ab_env.sh NAME v1 v2 ...
NAME=$1; shift
for v in "$@"; do
  env $NAME=$v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-rest > gpurun_out/abenv_$v.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.load(open("gpurun_out/abenv_$v.json"))
print("$NAME=$v", round(d["value"],1), {k: round(v, 3) for k, v in d["device_ms"].items()})
PY
done
