#!/bin/bash
shopt -s nullglob
# GPU box: dynamic instruction counts per env for the default build and the phase-skipping builds.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_ygo_skip
rm -rf $OUT; mkdir -p $OUT
export EP_YGO_CPU_SECONDS=0
cd /tmp && export TMPDIR=/tmp
for f in $R/envpool_amd/libenvpool_amd.so $R/build/libenvpool_amd_skip*.so; do
  tag=$(basename $f .so)
  ENVPOOL_AMD_LIB=$f timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/$tag -- python3 $R/tools/bench_ygo.py > $OUT/$tag.log 2>&1 || exit 1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for p in glob.glob('$OUT/$tag/*/*counter_collection.csv'):
    for r in csv.DictReader(open(p)):
        if 'ygo_encode' in r['Kernel_Name'] and int(r['Grid_Size'])>=4000000:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('$tag', {k: round(sum(v)/len(v)/65536,1) for k,v in sorted(acc.items())})
PY
done
