#!/bin/bash
# GPU box: the round's profiles. Kernel trace + separate FETCH_SIZE / WRITE_SIZE passes, for the
# Dummy rollout kernel (bench.py) and the YGOPro encode kernel (tools/bench_ygo.py).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r01
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export EP_YGO_CPU_SECONDS=0
ARGS="--no-cpu-baseline --no-ygo --steps 512 --warmup 128"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d_trace -- python3 $R/bench.py $ARGS > $OUT/d_trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/d_fetch -- python3 $R/bench.py $ARGS > $OUT/d_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/d_write -- python3 $R/bench.py $ARGS > $OUT/d_write.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/y_trace -- python3 $R/tools/bench_ygo.py > $OUT/y_trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/y_fetch -- python3 $R/tools/bench_ygo.py > $OUT/y_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/y_write -- python3 $R/tools/bench_ygo.py > $OUT/y_write.log 2>&1
mkdir -p $OUT/sum
python3 $R/tools/pmc_summary.py $OUT dummy_rollout > $OUT/sum/rollout_summary.json
python3 $R/tools/pmc_summary.py $OUT ygo_encode 4000000 > $OUT/sum/ygo_summary.json
cp $OUT/d_trace/*/*kernel_stats.csv $OUT/sum/bench_kernel_stats.csv
cp $OUT/y_trace/*/*kernel_stats.csv $OUT/sum/ygo_kernel_stats.csv
tail -1 $OUT/d_trace.log > $OUT/sum/bench_under_rocprof.json
tail -1 $OUT/y_trace.log > $OUT/sum/ygo_bench_under_rocprof.json
cat $OUT/sum/rollout_summary.json $OUT/sum/ygo_summary.json
cut -c1-150 $OUT/sum/bench_kernel_stats.csv | head -5; cut -c1-150 $OUT/sum/ygo_kernel_stats.csv | head -5
