#!/bin/bash
# counters of bench.py --config N, one pass each: tools/pmc_config.sh N   -> gpurun_out/pmc_cfg_N/{stats,pmc_sq,pmc_fetch,pmc_write}
N=$1
OUT=gpurun_out/pmc_cfg_$N
BENCH="python3 bench.py --config $N --steps 4 --warmup 1 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT.stats.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT.pmc_sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT.pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT.pmc_write.log 2>&1 || exit 1
python3 tools/pmc_summary.py $OUT > $OUT.summary.txt
