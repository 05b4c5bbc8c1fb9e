#!/bin/bash
# kernel-trace statistics of the headline bench -> gpurun_out/kstats_TAG.txt (per kernel: calls, average us, share)
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_$TAG -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-rest > gpurun_out/kstats_$TAG.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/kstats_$TAG/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/kstats_$TAG.txt", "w") as out:
    for r in rows[:24]:
        line = "%-70s calls %5s avg %9.1f us  %5.2f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
        print(line); out.write(line + "\n")
PY
