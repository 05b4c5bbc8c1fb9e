#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace statistics and three PMC passes of the headline bench.
# Usage: tools/profile_gpu.sh TAG      -> gpurun_out/prof_TAG/{stats,pmc_sq,pmc_fetch,pmc_write}
set -e
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
BENCH="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-rest"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT.stats.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT.pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT.pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT.pmc_write.log 2>&1
find $OUT -name "*.csv" | sort
