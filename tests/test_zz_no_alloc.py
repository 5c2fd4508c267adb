"""ep_dummy_send_noalloc (the entry point that allocates nothing) against ep_pool_reset /
ep_pool_send + ep_pool_recv on a pool of the same config: bit-exact on every fixed-shape key, on
two calls in a row, the second with a smaller input. Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest

from tests.util import FIXED_KEYS, KEY_NAMES, dummy_action

pytestmark = pytest.mark.gpu

def _out_arrays(torch, capi, dev, n, P, S):
  """12 caller-owned device arrays in _state_keys order, poisoned so that unwritten rows show."""
  shapes = {0: (n,), 1: (n * P,), 2: (n,), 3: (n,), 4: (n * P,), 5: (n * P,), 6: (n,), 7: (n,), 8: (n * P, S),
            10: (n * P,), 11: (n * P,)}
  dts = {0: torch.int32, 1: torch.int32, 2: torch.int32, 3: torch.uint8, 4: torch.float32, 5: torch.float32,
         6: torch.int32, 7: torch.uint8, 8: torch.int32, 10: torch.uint8, 11: torch.int32}
  tens, arr = {}, (capi.Array * 12)()
  for k, shp in shapes.items():
    t = torch.full(shp, 77, dtype=dts[k], device=dev)
    tens[k] = t
    arr[k].data = t.data_ptr()
    arr[k].ndim = len(shp)
    for i, d in enumerate(shp):
      arr[k].shape[i] = d
  return tens, arr


def _pool_batches(g, count):
  """`count` rows' worth of batches from the pool, concatenated per key."""
  got, rows = None, 0
  while rows < count:
    b = g._recv()
    rows += len(b[0])
    got = b if got is None else [None if x is None else np.concatenate([x, y]) for x, y in zip(got, b)]
  assert rows == count
  return got


@pytest.mark.parametrize("P,N,B,n2", [(3, 5000, 1000, 3000), (1, 4096, 0, 1500), (4, 70000, 35000, 257), (2, 64, 64, 1)])
def test_noalloc_matches_pool(P, N, B, n2):
  import torch

  from envpool_amd import capi
  L = capi.lib()
  dev = torch.device("cuda", 0)
  S, seed = 10, 2
  g = capi.DummyPoolC(num_envs=N, batch_size=B, max_num_players=P, seed=seed, ring_depth=16)
  cfg = capi.DummyConfig(num_envs=N, batch_size=0, num_threads=0, max_num_players=P, thread_affinity_offset=-1,
                         seed=seed, max_episode_steps=2**31 - 1, state_num=S, action_num=6, device_id=0,
                         env_id_offset=0, ring_depth=0)
  env_state = torch.tensor([[0, -1]] * N, dtype=torch.int32, device=dev).contiguous()
  result = torch.zeros(4, dtype=torch.int32, device=dev)
  stream = torch.cuda.current_stream().cuda_stream
  rng = np.random.default_rng(N + P)

  def noalloc(n, ids, penv, la, force_reset):
    tens, arr = _out_arrays(torch, capi, dev, n, P, S)
    t_ids = torch.from_numpy(ids).to(dev)
    t_penv = torch.from_numpy(penv).to(dev) if penv is not None else None
    t_la = torch.from_numpy(la).to(dev) if la is not None else None
    rc = L.ep_dummy_send_noalloc(C.byref(cfg), env_state.data_ptr(), n, t_ids.data_ptr(),
                                 0 if penv is None else len(penv), None if penv is None else t_penv.data_ptr(),
                                 None if la is None else t_la.data_ptr(), 1 if force_reset else 0, arr, 12,
                                 result.data_ptr(), stream)
    assert rc == 0, L.ep_last_error()
    torch.cuda.synchronize()
    res = result.cpu().numpy()
    assert res[1] == 0 and res[2] == 0
    np_rows = int(res[0])
    out = [None] * 12
    for k, t in tens.items():
      rows = np_rows if k in (1, 4, 5, 8, 10, 11) else n
      a = t.cpu().numpy()[:rows]
      out[k] = a.astype(np.bool_) if k in (3, 7, 10) else a
    return out

  def compare(got, want, ctx):
    for k in FIXED_KEYS:
      a, b = np.asarray(got[k]), np.asarray(want[k])
      assert a.dtype == b.dtype and a.shape == b.shape, f"{ctx} key {KEY_NAMES[k]}: {a.dtype}{a.shape} vs {b.dtype}{b.shape}"
      assert a.tobytes() == b.tobytes(), f"{ctx} key {KEY_NAMES[k]} differs"

  ids = np.arange(N, dtype=np.int32)
  # call 1: reset of every env
  g._reset(ids)
  want = _pool_batches(g, N)
  compare(noalloc(N, ids, None, None, True), want, "reset")
  # a few full steps so that episodes end and player counts vary
  for t in range(3):
    penv = np.asarray(want[1])                     # one action row per player the env reported
    sel = rng.permutation(N).astype(np.int32)
    a = dummy_action(sel, players_env_id=penv)
    g._send(a)
    want = _pool_batches(g, N)
    compare(noalloc(N, sel, penv, a[2], False), want, f"step {t}")
  # call with a SMALLER input right after a larger one (the scratch keeps the last call's contents)
  sub = rng.choice(N, size=n2, replace=False).astype(np.int32)
  penv = np.asarray(want[1])
  a = dummy_action(sub, players_env_id=penv)
  g._send(a)
  if g._L.ep_pool_is_sync(g.handle) or n2 % (B or N) == 0:
    want2 = _pool_batches(g, n2)
    compare(noalloc(n2, sub, penv, a[2], False), want2, "smaller call")
  else:
    # the pool's batch is not complete yet: complete it with a second send and compare both calls together
    rest = (B or N) - n2 % (B or N)
    sub2 = rng.choice(N, size=rest, replace=False).astype(np.int32)
    a2 = dummy_action(sub2, players_env_id=penv)
    g._send(a2)
    want2 = _pool_batches(g, n2 + rest)
    o1 = noalloc(n2, sub, penv, a[2], False)
    o2 = noalloc(rest, sub2, penv, a2[2], False)
    both = [None if x is None else np.concatenate([x, y]) for x, y in zip(o1, o2)]
    compare(both, want2, "smaller calls")
  if P > 1:
    # a multi-player step that brings NO player rows, right after steps that did: every env must
    # see action_num = 0 (obs:raw[:,1]), not the previous call's histogram left in the scratch
    empty = np.zeros(0, dtype=np.int32)
    a = dummy_action(ids, players_env_id=empty)
    g._send(a)
    want3 = _pool_batches(g, N)
    got3 = noalloc(N, ids, empty, a[2], False)
    compare(got3, want3, "step without player rows")
    raw = np.asarray(got3[8])
    stepped = raw[:, 0] != 0                    # rows of envs that stepped (a reset writes state 0)
    assert stepped.any() and not raw[stepped, 1].any()
  g.close()


def test_noalloc_rejects_oversize_and_bad_arguments():
  import torch

  from envpool_amd import capi
  L = capi.lib()
  dev = torch.device("cuda", 0)
  cfg = capi.DummyConfig(num_envs=262144 + 1, batch_size=0, num_threads=0, max_num_players=1, thread_affinity_offset=-1,
                         seed=1, max_episode_steps=100, state_num=10, action_num=6, device_id=0, env_id_offset=0,
                         ring_depth=0)
  x = torch.zeros(64, dtype=torch.int32, device=dev)
  tens, arr = _out_arrays(torch, capi, dev, 8, 1, 10)
  rc = L.ep_dummy_send_noalloc(C.byref(cfg), x.data_ptr(), 8, x.data_ptr(), 8, x.data_ptr(), x.data_ptr(), 1, arr, 12,
                               x.data_ptr(), None)
  assert rc == capi.EP_ERR_TOO_LARGE and b"at most 262144" in L.ep_last_error()
  cfg.num_envs = 16
  rc = L.ep_dummy_send_noalloc(C.byref(cfg), x.data_ptr(), 17, x.data_ptr(), 8, x.data_ptr(), x.data_ptr(), 1, arr, 12,
                               x.data_ptr(), None)
  assert rc == capi.EP_ERR_TOO_LARGE
  rc = L.ep_dummy_send_noalloc(C.byref(cfg), x.data_ptr(), 9, x.data_ptr(), 9, x.data_ptr(), x.data_ptr(), 1, arr, 12,
                               x.data_ptr(), None)
  assert rc == capi.EP_ERR_INVALID  # output arrays hold 8 rows
  # obs:raw laid out for another state_num would be overrun by the row stores: refused
  st = torch.zeros(32, dtype=torch.int32, device=dev)
  ids8 = torch.arange(8, dtype=torch.int32, device=dev)
  res = torch.zeros(4, dtype=torch.int32, device=dev)
  rc = L.ep_dummy_send_noalloc(C.byref(cfg), st.data_ptr(), 8, ids8.data_ptr(), 0, None, None, 1, arr, 12,
                               res.data_ptr(), None)
  assert rc == 0, L.ep_last_error()
  torch.cuda.synchronize()
  arr[8].shape[1] = 12
  rc = L.ep_dummy_send_noalloc(C.byref(cfg), x.data_ptr(), 8, x.data_ptr(), 8, x.data_ptr(), x.data_ptr(), 1, arr, 12,
                               x.data_ptr(), None)
  assert rc == capi.EP_ERR_INVALID and b"state_num" in L.ep_last_error()
  arr[8].shape[1] = 10
  arr[0].ndim = 2
  rc = L.ep_dummy_send_noalloc(C.byref(cfg), x.data_ptr(), 8, x.data_ptr(), 8, x.data_ptr(), x.data_ptr(), 1, arr, 12,
                               x.data_ptr(), None)
  assert rc == capi.EP_ERR_INVALID
  torch.cuda.synchronize()
