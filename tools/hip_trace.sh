#!/bin/bash
# host API calls and kernel executions of the headline bench on one clock -> gpurun_out/ht_TAG (no counters in this run)
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d gpurun_out/ht_$TAG -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ht_$TAG.log 2>&1
ls gpurun_out/ht_$TAG/*/
