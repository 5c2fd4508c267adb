"""bench.py's contract, end to end: the driver's own command on this GPU must print ONE JSON line
with the fields the contract names, and no leg of it may have failed silently."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _find_errors(obj, path=""):
  found = []
  if isinstance(obj, dict):
    for k, v in obj.items():
      if k == "error":
        found.append((path, v))
      found += _find_errors(v, f"{path}/{k}")
  elif isinstance(obj, list):
    for i, v in enumerate(obj):
      found += _find_errors(v, f"{path}[{i}]")
  return found


def test_driver_command_prints_the_contract_line():
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
  assert len(lines) == 1, out.stdout[-2000:]
  j = json.loads(lines[0])
  for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
    assert k in j, k
  assert j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5 and j["higher_is_better"] is True
  assert j["unit"] == "env-steps/s" and j["dtype"] == "int32" and j["vs_baseline"] is None and "workload" in j["config"]
  # value = units / time, and the time is the one reported
  assert abs(j["value"] - 65536 * 20 / (j["ms_per_step"] * 20e-3)) / j["value"] < 1e-6
  r = j["roofline"]
  assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
  assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
  assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["kernel_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-6
  assert r["kernel_us"] * r["launches"] <= j["ms_per_step"] * 20 * 1e3  # the kernel fits into the timed region
  c = j["cpu_baseline"]
  assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
  assert set(c["reference_architecture"]["value_by_worker_threads"]) >= {"1"}
  ps = j["per_step_launch"]
  assert isinstance(ps, dict) and ps["ms_per_step"] > 0
  for k in ("resident_server", "stream_ordered", "multiplayer_P3", "host_buffers_pcie_inclusive"):
    assert isinstance(ps[k], dict) and ps[k]["ms_per_step"] > 0, k
  y = j["ygo_encode"]
  assert y["roofline"]["bound"] == "hbm" and 0.3 < y["roofline"]["frac"] < 1.0
  assert y["cpu_baseline"]["kind"] == "port"
  assert not _find_errors(j), _find_errors(j)
