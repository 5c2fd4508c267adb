"""The ThreadSanitizer targets that need no GPU, run as part of the CPU suite: the threaded CPU
restatement of the reference's queue + worker architecture (oracle/dummy_pool_mt.c) and the host
logic of ep_ygo_pool against a mock encoder (tools/tsan_ygo_pool.cpp). The sanitizers abort on the
first report (halt_on_error), so a zero exit status is a clean run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(args, timeout=600):
  out = subprocess.run(["make", "-C", ROOT] + args, capture_output=True, text=True, timeout=timeout)
  assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
  return out.stdout


def test_tsan_threaded_restatement():
  assert "no ThreadSanitizer report" in _make(["-C", "oracle", "tsan"])


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or not shutil.which("make"), reason="needs ROCm's clang")
def test_tsan_ygo_pool_host_logic():
  assert "no ThreadSanitizer report" in _make(["tsan-pool"])
