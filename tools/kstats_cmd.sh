#!/bin/bash
# kernel-trace statistics of an arbitrary python script: tools/kstats_cmd.sh TAG script.py [args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kc_$TAG -- python3 "$@" > gpurun_out/kc_$TAG.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/kc_$TAG/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    print("%-60s calls %5s avg %9.1f us  %5.2f %%" % (r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
