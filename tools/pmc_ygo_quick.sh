#!/bin/bash
# GPU box: stamps build breakdown + two SQ counter passes for the YGO encode kernel (per-wave averages).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_ygo_q
rm -rf $OUT; mkdir -p $OUT
export EP_YGO_CPU_SECONDS=0
if [ -f $R/build/libenvpool_amd_stamps.so ]; then
  ENVPOOL_AMD_LIB=$R/build/libenvpool_amd_stamps.so timeout -k 10 200 python3 -u $R/tools/bench_ygo.py > $OUT/stamps.log 2>&1 || exit 1
  grep stamps $OUT/stamps.log | tail -8
fi
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- python3 $R/tools/bench_ygo.py > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/p2 -- python3 $R/tools/bench_ygo.py > $OUT/p2.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
for p in sorted(glob.glob('$OUT/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'ygo_encode' in r['Kernel_Name'] and int(r['Grid_Size'])>=4000000:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, sum(v)/len(v)/65536)
PY
