#!/bin/bash
# One rocprofv3 --pmc pass of the headline bench; prints the median per dispatch of every counter for the kernels
# whose names start with PREFIX.   tools/pmc_kernel.sh TAG PREFIX COUNTER...   (at most 8 SQ counters per pass)
TAG=$1; PREFIX=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pk_$TAG -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-rest > gpurun_out/pk_$TAG.log 2>&1 || { tail -5 gpurun_out/pk_$TAG.log; exit 1; }
python3 - "$TAG" "$PREFIX" <<'PY'
import csv, glob, collections, sys
tag, prefix = sys.argv[1], sys.argv[2]
def short(n): return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
c = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(f"gpurun_out/pk_{tag}/**/*counter_collection.csv", recursive=True)[0])):
    c[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in c:
    if k.startswith(prefix):
        print(tag, k, {n: "%.4g" % (sorted(v)[len(v) // 2]) for n, v in sorted(c[k].items())})
PY
