#!/bin/bash
# GPU box: A/B of the default build against every build/libenvpool_amd_*.so, three alternating rounds.
shopt -s nullglob
mkdir -p gpurun_out
export EP_YGO_CPU_SECONDS=0
for round in 1 2 3; do
  for f in envpool_amd/libenvpool_amd.so build/libenvpool_amd_*.so; do
    ENVPOOL_AMD_LIB=$f timeout -k 10 200 python -u tools/bench_ygo.py > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
    echo "$round $f $(tail -1 gpurun_out/ab.log | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["large_batch"]["kernel_us"])')"
  done
done
