#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_ygo_q
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export EP_YGO_CPU_SECONDS=0
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- python3 $R/tools/bench_ygo.py > $OUT/p1.log 2>&1
python3 - <<PY
import csv,glob,collections
for p in sorted(glob.glob('$OUT/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'ygo_encode' in r['Kernel_Name'] and int(r['Grid_Size'])>=4000000:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, sum(v)/len(v)/65536)
PY
