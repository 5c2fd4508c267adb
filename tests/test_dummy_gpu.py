"""GPU parity of the Dummy lockstep reset/step + StateBuffer path, through the C ABI,
against the CPU oracle (oracle/dummy_oracle.c). Bit-exact on every fixed-shape key."""
import numpy as np
import pytest

from tests.util import assert_batches_equal, dummy_action

pytestmark = pytest.mark.gpu


def _pair(**kw):
  from envpool_amd.capi import DummyPoolC
  from oracle.dummy_oracle import DummyOracle
  okw = {k: v for k, v in kw.items() if k not in ("ring_depth", "device_id")}
  return DummyPoolC(**kw), DummyOracle(**okw)


@pytest.mark.parametrize("seed", [1, 3, 42])
def test_sync_n8_lockstep(seed):
  g, o = _pair(num_envs=8, seed=seed)
  ids = np.arange(8, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), f"seed {seed} reset")
  for t in range(1, 65):
    a = dummy_action(ids)
    g._send(a); o._send(a)
    assert_batches_equal(g._recv(), o._recv(), f"seed {seed} t={t}")


def test_sync_large_lockstep():
  """BASELINE config 2 shape (N=65536, seed 42) for a few dozen steps."""
  n = 65536
  g, o = _pair(num_envs=n, seed=42)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), "reset")
  for t in range(1, 50):
    a = dummy_action(ids)
    g._send(a); o._send(a)
    assert_batches_equal(g._recv(), o._recv(), f"t={t}")


def test_sync_odd_sizes_and_permutation():
  """Row i of the output is the env at position i of the SENT env_id array;
  sizes that are not multiples of the 256-env tile or of the 16-byte store width."""
  rng = np.random.default_rng(0)
  for n in (1, 3, 63, 257, 1000):
    g, o = _pair(num_envs=n, seed=2)
    ids = np.arange(n, dtype=np.int32)
    g._reset(ids); o._reset(ids)
    assert_batches_equal(g._recv(), o._recv(), f"n={n} reset")
    for t in range(12):
      perm = rng.permutation(n).astype(np.int32)
      a = dummy_action(perm)
      g._send(a); o._send(a)
      assert_batches_equal(g._recv(), o._recv(), f"n={n} t={t}")
