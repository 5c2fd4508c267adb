"""The C ABI used from plain C (examples/dummy_step_loop.c: gcc, -std=c11, only include/envpool_amd.h):
a sync Dummy-v0 pool stepped with host arrays and checked in C against the env's closed form."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "dummy_step_loop")
EXE_YGO = os.path.join(ROOT, "examples", "ygo_pool_loop")


def test_c_example_is_built_and_links_against_the_library():
  for exe in (EXE, EXE_YGO):
    assert os.path.exists(exe), "run `make examples`"
    out = subprocess.run(["ldd", exe], capture_output=True, text=True, check=True).stdout
    line = [l for l in out.splitlines() if "libenvpool_amd.so" in l]
    assert line and "not found" not in line[0], out


@pytest.mark.gpu
def test_c_example_runs():
  r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
  assert r.returncode == 0, r.stdout + r.stderr
  assert "match the closed form" in r.stdout


@pytest.mark.gpu
def test_c_example_ygo_pool_runs():
  """examples/ygo_pool_loop.c: ep_ygo_pool over two C callbacks, every checked key predicted by hand."""
  r = subprocess.run([EXE_YGO], capture_output=True, text=True, timeout=120)
  assert r.returncode == 0, r.stdout + r.stderr
  assert "as predicted" in r.stdout
