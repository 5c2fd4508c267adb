#!/bin/bash
# GPU box: parity tests with the default build, then the large-batch bench for each wave-count build.
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_ygo_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/ygo_tests.log 2>&1
tail -2 gpurun_out/ygo_tests.log
EP_YGO_CPU_SECONDS=0 timeout -k 10 200 python -u tools/bench_ygo.py > gpurun_out/ygo_bench_w4.log 2>&1
tail -1 gpurun_out/ygo_bench_w4.log | cut -c1-330
for w in 1 2 8 stamps; do
  ENVPOOL_AMD_LIB=build/libenvpool_amd_$( [ $w = stamps ] && echo stamps || echo w$w ).so EP_YGO_CPU_SECONDS=0 timeout -k 10 200 python -u tools/bench_ygo.py > gpurun_out/ygo_bench_$w.log 2>&1
  echo "== $w"; tail -1 gpurun_out/ygo_bench_$w.log | cut -c1-330
done
grep stamps gpurun_out/ygo_bench_stamps.log | tail -8
