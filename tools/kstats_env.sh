#!/bin/bash
# kernel trace of the headline bench with one environment variable set: tools/kstats_env.sh TAG NAME VALUE
TAG=$1
export $2=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_$TAG -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/kstats_$TAG.log 2>&1
python3 tools/frame_timeline.py gpurun_out/kstats_$TAG > gpurun_out/timeline_$TAG.txt
tail -1 gpurun_out/timeline_$TAG.txt
