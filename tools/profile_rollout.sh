#!/bin/bash
# GPU box: kernel trace + two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_rollout
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --steps 512 --warmup 128"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
python3 $R/tools/pmc_summary.py $OUT dummy_rollout | tee $OUT/summary.json
cat $OUT/trace/*/*kernel_stats.csv | cut -c1-160 | head -8
