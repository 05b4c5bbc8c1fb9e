#!/bin/bash
# Timing experiments: builds renderer_amd/lib/librdr_hip_NAME.so from the current objects with ONE source recompiled
# under extra flags.  tools/build_variant.sh NAME kernels_shade.hip -DRDR_DIAG=1
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../renderer_amd/csrc"
make -s -j8
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $SRC -o /tmp/variant_$NAME.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/librdr_hip_$NAME.so $OBJS /tmp/variant_$NAME.o
echo built librdr_hip_$NAME.so
