#!/bin/bash
# kernel-trace statistics of bench.py --config N: tools/kstats_config.sh N
N=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kcfg_$N -- python3 bench.py --config $N --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/kcfg_$N.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/kcfg_$N/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print("%-60s calls %5s avg %9.1f us  %5.2f %%" % (r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
