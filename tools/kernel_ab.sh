#!/bin/bash
# one kernel alone (frames rendered one at a time) under every librdr_hip*.so in renderer_amd/lib: tools/kernel_ab.sh k_setup
for path in renderer_amd/lib/librdr_hip*.so; do
  lib=$(basename $path .so)
  RDR_HIP_LIB=$PWD/$path bash tools/kstats_cmd.sh kab_$lib tools/time_frame_serial.py > gpurun_out/kab_$lib.txt 2>&1 || exit 1
  echo "$lib $(grep -E "$1" gpurun_out/kab_$lib.txt | head -3 | sed 's/(.*calls/ calls/' | tr '\n' ';')"
done
