#!/bin/bash
shopt -s nullglob
# GPU box: price each phase of the encode kernel by timing diagnostic builds that leave it out.
mkdir -p gpurun_out
export EP_YGO_CPU_SECONDS=0
for f in envpool_amd/libenvpool_amd.so build/libenvpool_amd_skip*.so; do
  ENVPOOL_AMD_LIB=$f timeout -k 10 200 python -u tools/bench_ygo.py > gpurun_out/skip.log 2>&1 || { tail -5 gpurun_out/skip.log; exit 1; }
  echo "$f $(tail -1 gpurun_out/skip.log | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["large_batch"]["kernel_us"])')"
done
