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


def _rollout_case(n, T, seed, ring_depth, rounds=2):
  """ep_pool_rollout (fused lockstep loop, device-resident [T,N] action blocks) vs T oracle steps."""
  import ctypes as C

  import torch

  from envpool_amd import capi

  g, o = _pair(num_envs=n, seed=seed, ring_depth=ring_depth)
  L = capi.lib()
  dev = torch.device("cuda", 0)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), "reset")
  rng = np.random.default_rng(seed)
  for rnd in range(rounds):
    # players.env_id mismatches on ~10% of the rows (pins obs:raw[:,1] == 0 there)
    penv = np.tile(ids, (T, 1))
    penv[rng.random((T, n)) < 0.1] += 1
    la = np.repeat(rng.integers(0, 5, size=(T, n, 1)).astype(np.float64), 6, axis=2)
    t_ids = torch.from_numpy(ids).to(dev)
    t_penv = torch.from_numpy(penv).to(dev)
    t_la = torch.from_numpy(la).to(dev)
    t_z = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_penv.data_ptr(), t_la.data_ptr(), t_z.data_ptr(), t_z.data_ptr())
    dims = (C.c_int64 * 5)(n, n, n, n, n)
    strides = (C.c_int64 * 5)(0, n, n, 0, 0)
    capi._check(L.ep_pool_rollout(g.handle, T, 5, ptrs, dims, strides))
    for t in range(T):
      o._send([ids, penv[t], la[t], ids * 0, ids * 0])
      assert_batches_equal(g._recv(), o._recv(), f"n={n} T={T} round {rnd} t={t}")
  sr, sl, sc = g.episode_stats()
  # episode statistics against the closed form: length of a finished episode of env e is
  # max(1, seed+e); count = finished episodes in rounds*T steps after the reset
  Ls = np.maximum(1, seed + ids.astype(np.int64))
  total = rounds * T
  assert np.array_equal(sc, total // (Ls + 1) + ((total % (Ls + 1)) >= Ls))
  assert np.array_equal(sl[sc > 0], Ls[sc > 0])
  assert not sr.any()


@pytest.mark.parametrize("n,T,depth", [(64, 12, 12), (640, 24, 32), (4096, 64, 128), (1000, 7, 8), (128, 5, 6)])
def test_rollout_matches_stepwise_oracle(n, T, depth):
  _rollout_case(n, T, seed=1, ring_depth=depth)


def test_rollout_reports_action_check():
  """list_action lanes that differ abort the reference (dummy_envpool.h:189-192); here the
  batch comes back and recv reports EP_ERR_ACTION_CHECK."""
  import ctypes as C

  import torch

  from envpool_amd import capi

  n, T = 128, 4
  g, _ = _pair(num_envs=n, seed=50, ring_depth=8)
  L = capi.lib()
  dev = torch.device("cuda", 0)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); g._recv()
  la = np.ones((T, n, 6))
  la[2, 77, 3] = 2.0
  t_ids, t_penv, t_la = (torch.from_numpy(x).to(dev) for x in (ids, np.tile(ids, (T, 1)), la))
  torch.cuda.synchronize()
  ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_penv.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
  capi._check(L.ep_pool_rollout(g.handle, T, 5, ptrs, (C.c_int64 * 5)(n, n, n, n, n), (C.c_int64 * 5)(0, n, n, 0, 0)))
  g._recv(allow=(capi.EP_ERR_ACTION_CHECK,))
  assert g.last_status == capi.EP_ERR_ACTION_CHECK
  for _ in range(T - 1):
    g._recv()
