#!/bin/bash
# per-dispatch durations of selected kernels for every librdr_hip*.so in renderer_amd/lib (timing experiments)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for path in renderer_amd/lib/librdr_hip*.so; do
  lib=$(basename $path .so)
  RDR_HIP_LIB=$PWD/$path timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_$lib -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-rest > gpurun_out/kt_$lib.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/kt_$lib/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("$lib", {k: round(sorted(v)[len(v) // 2], 1) for k, v in d.items() if k.startswith(("$1", "$2", "$3", "$4"))})
PY
done
