#!/bin/bash
# frame timeline of the bench under one environment setting: tools/timeline_env.sh TAG NAME=value
TAG=$1; shift
export "$@"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$TAG -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-rest > gpurun_out/tl_$TAG.log 2>&1
python3 tools/frame_timeline.py gpurun_out/tl_$TAG > gpurun_out/tl_$TAG.txt
tail -1 gpurun_out/tl_$TAG.txt
