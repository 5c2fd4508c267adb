#!/bin/bash
# GPU box: PMC passes for the YGO encode kernel.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_ygo
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export EP_YGO_CPU_SECONDS=0
rocprofv3 -L > $OUT/counters.txt 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- python3 $R/tools/bench_ygo.py > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/p2 -- python3 $R/tools/bench_ygo.py > $OUT/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- python3 $R/tools/bench_ygo.py > $OUT/p3.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -- python3 $R/tools/bench_ygo.py > $OUT/p4.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_ygo.py > $OUT/trace.log 2>&1
python3 $R/tools/pmc_summary.py $OUT ygo_encode | tee $OUT/summary.json
