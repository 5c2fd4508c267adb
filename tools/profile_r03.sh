#!/bin/bash
# GPU box: round 3's profiles. Kernel trace + separate FETCH_SIZE / WRITE_SIZE passes for the Dummy
# rollout kernel under the DRIVER's bench command (--steps 20 --warmup 5) and under the default one,
# a kernel trace of the YGOPro encode bench, and a kernel trace of the multi-player step loop (the
# gap between its two kernels). Every step under its own time limit; the script stops at the first failure.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT/sum
cd /tmp && export TMPDIR=/tmp
export EP_YGO_CPU_SECONDS=0
A20="--no-cpu-baseline --no-ygo --steps 20 --warmup 5"
A2K="--no-cpu-baseline --no-ygo --steps 2048 --warmup 256"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d20_trace -- python3 $R/bench.py $A20 > $OUT/d20_trace.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d_trace -- python3 $R/bench.py $A2K > $OUT/d_trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/d_fetch -- python3 $R/bench.py $A2K > $OUT/d_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/d_write -- python3 $R/bench.py $A2K > $OUT/d_write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/d20_fetch -- python3 $R/bench.py $A20 > $OUT/d20_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/d20_write -- python3 $R/bench.py $A20 > $OUT/d20_write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/y_trace -- python3 $R/tools/bench_ygo.py > $OUT/y_trace.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/y_fetch -- python3 $R/tools/bench_ygo.py > $OUT/y_fetch.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/y_write -- python3 $R/tools/bench_ygo.py > $OUT/y_write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mp_trace -- python3 $R/tools/mp_profile.py > $OUT/mp_trace.log 2>&1 || exit 1
mkdir -p $OUT/t20 $OUT/t2k $OUT/ty
cp -r $OUT/d20_trace $OUT/d20_fetch $OUT/d20_write $OUT/t20/; cp -r $OUT/d_trace $OUT/d_fetch $OUT/d_write $OUT/t2k/; cp -r $OUT/y_trace $OUT/y_fetch $OUT/y_write $OUT/ty/
PMC_SUMMARY_SPLIT_FRAC=0.5 PMC_SUMMARY_SPLIT_NS=60000 python3 $R/tools/pmc_summary.py $OUT/t20 dummy_rollout > $OUT/sum/rollout_steps20_summary.json
python3 $R/tools/pmc_summary.py $OUT/t2k dummy_rollout > $OUT/sum/rollout_summary.json
python3 $R/tools/pmc_summary.py $OUT/ty ygo_encode 4000000 > $OUT/sum/ygo_summary.json
python3 $R/tools/mp_gap.py $OUT/mp_trace > $OUT/sum/mp_gap.json
cp $OUT/d20_trace/*/*kernel_stats.csv $OUT/sum/bench_steps20_kernel_stats.csv
cp $OUT/d_trace/*/*kernel_stats.csv $OUT/sum/bench_kernel_stats.csv
cp $OUT/y_trace/*/*kernel_stats.csv $OUT/sum/ygo_kernel_stats.csv
cp $OUT/mp_trace/*/*kernel_stats.csv $OUT/sum/mp_kernel_stats.csv
grep '^{"metric"' $OUT/d20_trace.log > $OUT/sum/bench_steps20_under_rocprof.json
grep '^{"metric"' $OUT/d_trace.log > $OUT/sum/bench_under_rocprof.json
tail -n 2 $OUT/mp_trace.log > $OUT/sum/mp_profile.txt
# the bulky per-dispatch csv files stay on the box; the summaries travel back
rm -rf $OUT/d20_trace $OUT/d_trace $OUT/d_fetch $OUT/d_write $OUT/d20_fetch $OUT/d20_write $OUT/y_trace $OUT/y_fetch $OUT/y_write $OUT/mp_trace $OUT/t20 $OUT/t2k $OUT/ty
cat $OUT/sum/rollout_steps20_summary.json $OUT/sum/rollout_summary.json $OUT/sum/mp_gap.json
