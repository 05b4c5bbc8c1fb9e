#!/bin/bash
# For every librdr_hip*.so in renderer_amd/lib: median duration and VALU wave-instructions per dispatch of the kernels
# whose names start with the given prefixes (diagnostic builds, see RDR_DIAG in kernels_shade.hip).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for path in renderer_amd/lib/librdr_hip*.so; do
  lib=$(basename $path .so)
  RDR_HIP_LIB=$PWD/$path timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ds_$lib/kt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/ds_$lib.kt.log 2>&1 || exit 1
  RDR_HIP_LIB=$PWD/$path timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/ds_$lib/pmc -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/ds_$lib.pmc.log 2>&1 || exit 1
  python3 - "$lib" "$@" <<'PY'
import csv, glob, collections, sys
lib, prefixes = sys.argv[1], tuple(sys.argv[2:])
def short(n): return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(f"gpurun_out/ds_{lib}/kt/**/*kernel_trace.csv", recursive=True)[0])):
    d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
c = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(f"gpurun_out/ds_{lib}/pmc/**/*counter_collection.csv", recursive=True)[0])):
    c[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in d:
    if k.startswith(prefixes):
        med = sorted(d[k])[len(d[k]) // 2]
        print(lib, k, "us", round(med, 1), {n: "%.4g" % (sorted(v)[len(v) // 2]) for n, v in sorted(c[k].items())})
PY
done
