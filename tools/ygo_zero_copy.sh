#!/bin/bash
# GPU box: parity tests, then the device-resident rate and the host-record pipeline (config 3 / 5
# shapes) with the staging copy and with the kernel reading the pinned slot itself.
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_ygo_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/ygo_tests.log 2>&1 || { tail -30 gpurun_out/ygo_tests.log; exit 1; }
tail -1 gpurun_out/ygo_tests.log
export EP_YGO_CPU_SECONDS=0
for z in 0 1; do
  EP_YGO_ZERO_COPY=$z timeout -k 10 200 python -u tools/bench_ygo.py > gpurun_out/zc$z.log 2>&1 || { tail -5 gpurun_out/zc$z.log; exit 1; }
  echo "zero_copy=$z"; tail -1 gpurun_out/zc$z.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["large_batch"]["kernel_us"]); print({k:{kk:vv for kk,vv in v.items() if kk in ("us_per_step_wall","kernel_us","us_per_step_wall_in_place","h2d_copy_us")} for k,v in d.items() if "pipeline" in k})'
done
