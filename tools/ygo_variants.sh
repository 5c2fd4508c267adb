#!/bin/bash
shopt -s nullglob
# GPU box: parity tests with the default build, then the large-batch bench for each variant build.
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_ygo_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/ygo_tests.log 2>&1 || { tail -30 gpurun_out/ygo_tests.log; exit 1; }
tail -2 gpurun_out/ygo_tests.log
export EP_YGO_CPU_SECONDS=0
for f in envpool_amd/libenvpool_amd.so build/libenvpool_amd_*.so; do
  ENVPOOL_AMD_LIB=$f timeout -k 10 200 python -u tools/bench_ygo.py > gpurun_out/ygo_bench_var.log 2>&1 || { tail -5 gpurun_out/ygo_bench_var.log; exit 1; }
  echo "== $f"; grep stamps gpurun_out/ygo_bench_var.log | tail -8 || true; tail -1 gpurun_out/ygo_bench_var.log | cut -c1-420
done
