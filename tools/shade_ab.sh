#!/bin/bash
# k_shade_frags alone (frames rendered one at a time) under every librdr_hip*.so in renderer_amd/lib: average us per launch
for path in renderer_amd/lib/librdr_hip*.so; do
  lib=$(basename $path .so)
  RDR_HIP_LIB=$PWD/$path bash tools/kstats_cmd.sh sab_$lib tools/time_frame_serial.py > gpurun_out/sab_$lib.txt 2>&1 || exit 1
  echo "$lib $(grep -E 'k_shade_frags' gpurun_out/sab_$lib.txt | head -1 | sed 's/.*calls/calls/')"
done
