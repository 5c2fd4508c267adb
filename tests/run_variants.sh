#!/bin/bash
# Runs the rollout parity tests and the bench once per rollout-kernel variant (GPU box only).
set -e
for v in 0 1 2; do
  echo "=== EP_ROLLOUT_VARIANT=$v"
  EP_ROLLOUT_VARIANT=$v timeout -k 10 300 python -u -m pytest tests/test_dummy_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "rollout" 2>&1 | tail -3
  EP_ROLLOUT_VARIANT=$v timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 1024 --warmup 128 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.3e' % d['value'], 'kernel_us %.1f' % d['roofline']['kernel_us'], 'GB/s %.0f' % d['roofline']['achieved'], 'frac %.3f' % d['roofline']['frac'])"
done
echo "=== generic"
EP_FORCE_GENERIC=1 timeout -k 10 300 python -u -m pytest tests/test_dummy_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "rollout" 2>&1 | tail -3
