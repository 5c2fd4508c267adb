#!/bin/bash
# GPU box: vector-L1 / L2 request counters of the encode kernel, default build vs the build without
# hash-table reads (YGO_SKIP=64), to price the probes.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_ygo_mem
rm -rf $OUT; mkdir -p $OUT
export EP_YGO_CPU_SECONDS=0
cd /tmp && export TMPDIR=/tmp
for f in $R/envpool_amd/libenvpool_amd.so $R/build/libenvpool_amd_skip64.so; do
  tag=$(basename $f .so)
  i=0
  for ctrs in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum"; do
    i=$((i+1))
    ENVPOOL_AMD_LIB=$f timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/$tag.$i -- python3 $R/tools/bench_ygo.py > $OUT/$tag.$i.log 2>&1 || { tail -3 $OUT/$tag.$i.log; exit 1; }
  done
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for p in glob.glob('$OUT/$tag.*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(p)):
        if 'ygo_encode' in r['Kernel_Name'] and int(r['Grid_Size'])>=4000000:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('$tag', {k: round(sum(v)/len(v)/65536,1) for k,v in sorted(acc.items())})
PY
done
