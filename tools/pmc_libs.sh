#!/bin/bash
# One rocprofv3 --pmc pass of the headline bench per librdr_hip*.so in renderer_amd/lib (diagnostic builds): median per
# dispatch of every counter for kernels whose names start with PREFIX.   tools/pmc_libs.sh PREFIX COUNTER...
PREFIX=$1; shift
for path in renderer_amd/lib/librdr_hip*.so; do
  lib=$(basename $path .so)
  RDR_HIP_LIB=$PWD/$path bash tools/pmc_kernel.sh $lib "$PREFIX" "$@" || exit 1
done
