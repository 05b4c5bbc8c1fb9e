#!/bin/bash
# Register / scratch / LDS use of every kernel of one source file, from the code object's metadata.
# Usage: tools/kernel_regs.sh kernels_shade [extra hipcc flags]
F=$1; shift
cd "$(dirname "$0")/../renderer_amd/csrc"
mkdir -p /tmp/kregs && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" \
  --save-temps=obj -c $F.hip -o /tmp/kregs/$F.o 2>/dev/null
S=$(ls /tmp/kregs/$F-hip-amdgcn-amd-amdhsa-gfx950.s)
python3 - "$S" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", t, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\d+)", body) or [None, "?"])[1]
    short = re.sub(r"^_ZN\d+_GLOBAL__N_1", "", name)[:60]
    print(f"{short:62s} vgpr {g('next_free_vgpr'):>4s} sgpr {g('next_free_sgpr'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
PY
