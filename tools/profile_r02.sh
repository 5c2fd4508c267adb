#!/bin/bash
# GPU box: the round's profiles. Kernel trace + separate FETCH_SIZE / WRITE_SIZE passes for the
# Dummy rollout kernel under the DRIVER's bench command (--steps 20 --warmup 5) and under the
# default one, and for the YGOPro encode kernel (tools/bench_ygo.py).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02
rm -rf $OUT; mkdir -p $OUT/sum
cd /tmp && export TMPDIR=/tmp
export EP_YGO_CPU_SECONDS=0
A20="--no-cpu-baseline --no-ygo --steps 20 --warmup 5"
A2K="--no-cpu-baseline --no-ygo --steps 2048 --warmup 256"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d20_trace -- python3 $R/bench.py $A20 > $OUT/d20_trace.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d_trace -- python3 $R/bench.py $A2K > $OUT/d_trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/d_fetch -- python3 $R/bench.py $A2K > $OUT/d_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/d_write -- python3 $R/bench.py $A2K > $OUT/d_write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/y_trace -- python3 $R/tools/bench_ygo.py > $OUT/y_trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/y_fetch -- python3 $R/tools/bench_ygo.py > $OUT/y_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/y_write -- python3 $R/tools/bench_ygo.py > $OUT/y_write.log 2>&1 || exit 1
mkdir -p $OUT/t20 $OUT/t2k $OUT/ty
cp -r $OUT/d20_trace $OUT/t20/; cp -r $OUT/d_trace $OUT/d_fetch $OUT/d_write $OUT/t2k/; cp -r $OUT/y_trace $OUT/y_fetch $OUT/y_write $OUT/ty/
PMC_SUMMARY_SPLIT_NS=60000 python3 $R/tools/pmc_summary.py $OUT/t20 dummy_rollout > $OUT/sum/rollout_steps20_summary.json
python3 $R/tools/pmc_summary.py $OUT/t2k dummy_rollout > $OUT/sum/rollout_summary.json
python3 $R/tools/pmc_summary.py $OUT/ty ygo_encode 4000000 > $OUT/sum/ygo_summary.json
cp $OUT/d20_trace/*/*kernel_stats.csv $OUT/sum/bench_steps20_kernel_stats.csv
cp $OUT/d_trace/*/*kernel_stats.csv $OUT/sum/bench_kernel_stats.csv
cp $OUT/y_trace/*/*kernel_stats.csv $OUT/sum/ygo_kernel_stats.csv
grep '^{"metric"' $OUT/d20_trace.log > $OUT/sum/bench_steps20_under_rocprof.json
grep '^{"metric"' $OUT/d_trace.log > $OUT/sum/bench_under_rocprof.json
cat $OUT/sum/rollout_steps20_summary.json $OUT/sum/rollout_summary.json $OUT/sum/ygo_summary.json
cut -c1-150 $OUT/sum/bench_steps20_kernel_stats.csv | head -6; cut -c1-150 $OUT/sum/ygo_kernel_stats.csv | head -5
