#!/bin/bash
# per-kernel times of the serial frame (rasterisers one after the other) for every librdr_hip*.so in renderer_amd/lib
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for path in renderer_amd/lib/librdr_hip*.so; do
  lib=$(basename $path .so)
  export RDR_HIP_LIB=$PWD/$path RDR_SERIAL_RASTER=1
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rv_$lib -- python3 tools/time_frame_serial.py > gpurun_out/rv_$lib.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/rv_$lib/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_raster" in r["Name"]:
        print("%-22s %-28s calls %3s avg %8.1f us" % ("$lib", r["Name"].replace("(anonymous namespace)::", "")[5:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
