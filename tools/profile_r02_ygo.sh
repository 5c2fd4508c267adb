#!/bin/bash
# GPU box: the YGOPro part of tools/profile_r02.sh alone (kernel trace + separate FETCH_SIZE /
# WRITE_SIZE passes over tools/bench_ygo.py), after a change to the encode kernel's outputs.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02y
rm -rf $OUT; mkdir -p $OUT/sum
cd /tmp && export TMPDIR=/tmp
export EP_YGO_CPU_SECONDS=0
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/y_trace -- python3 $R/tools/bench_ygo.py > $OUT/y_trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/y_fetch -- python3 $R/tools/bench_ygo.py > $OUT/y_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/y_write -- python3 $R/tools/bench_ygo.py > $OUT/y_write.log 2>&1 || exit 1
mkdir -p $OUT/ty
cp -r $OUT/y_trace $OUT/y_fetch $OUT/y_write $OUT/ty/
python3 $R/tools/pmc_summary.py $OUT/ty ygo_encode 4000000 > $OUT/sum/ygo_summary.json
cp $OUT/y_trace/*/*kernel_stats.csv $OUT/sum/ygo_kernel_stats.csv
cat $OUT/sum/ygo_summary.json; cut -c1-150 $OUT/sum/ygo_kernel_stats.csv | head -4
