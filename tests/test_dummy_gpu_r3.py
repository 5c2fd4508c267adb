"""GPU parity, round-3 cases around the resident step server of ep_pool_step: the doorbell written
through the PCIe BAR and the leader wake-up it replaces, the idle watchdog, the launch-path
back-off for callers that synchronise the device, and ep_pool_step_begin / _end."""
import ctypes as C
import time

import numpy as np
import pytest

from tests.util import assert_batches_equal, dummy_action
from tests.test_dummy_gpu_r2 import _pair, _views

pytestmark = pytest.mark.gpu


def _setup(n, seed, dev, torch):
  g, o = _pair(num_envs=n, seed=seed, ring_depth=4)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), "reset")
  t_ids = torch.from_numpy(ids).to(dev)
  t_penv = torch.empty(n, dtype=torch.int32, device=dev)
  t_la = torch.empty((n, 6), dtype=torch.float64, device=dev)
  return g, o, ids, t_ids, t_penv, t_la


@pytest.mark.parametrize("bar", ["1", "0"])
def test_one_server_launch_serves_many_steps(bar, monkeypatch):
  """Steady state of the server: 240 consecutive ep_pool_steps of a 65 536-env pool with only
  STREAM-scoped waits between them are served by ONE resident kernel - the doorbell loop and its
  (k+1)*shard_size counter arithmetic, not the restart path. The actions are rewritten between the
  steps by torch kernels on torch's stream (what the server's sc1 loads have to see); every step's
  rows are copied out of the ring on that stream and compared with the oracle afterwards, so that
  the loop keeps a cadence of some tens of microseconds. bar = "0": the leader wake-up through
  host-mapped memory (what a machine without a large BAR gets)."""
  import torch

  from envpool_amd import capi, interop
  monkeypatch.setenv("EP_SERVER_BAR", bar)
  L = capi.lib()
  dev = torch.device("cuda", 0)
  n, steps = 65536, 240
  g, o, ids, t_ids, t_penv, t_la = _setup(n, 42, dev, torch)
  rng = np.random.default_rng(5)
  penv_all = ids[None, :] + (rng.random((steps, n)) < 0.1).astype(np.int32)
  lane_all = rng.integers(0, 5, size=(steps, n)).astype(np.float64)
  d_penv_all = torch.from_numpy(penv_all).to(dev)
  d_lane_all = torch.from_numpy(lane_all).to(dev)
  keys = [k for k in range(12) if k != 9]
  store = None
  views = {}
  dims = (C.c_int64 * 5)(n, n, n, n, n)
  ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_penv.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
  b = capi.Batch()
  stream = torch.cuda.current_stream(dev)
  torch.cuda.synchronize()
  for t in range(steps):
    t_penv.copy_(d_penv_all[t])
    t_la.copy_(d_lane_all[t].unsqueeze(1).expand(n, 6))
    stream.synchronize()             # stream-scoped: does not wait for the resident kernel; the copies of step t-1 are done too
    rc = L.ep_pool_step(g.handle, 5, ptrs, dims, C.byref(b))
    assert rc == 0, L.ep_last_error()
    assert b.num_rows == n
    for k in keys:
      a = b.arrays[k]
      v = views.get(a.data)
      if v is None:
        shape = tuple(a.shape[i] for i in range(a.ndim))
        v = views[a.data] = interop.as_torch((a.data, shape, np.dtype(capi._NP[a.dtype]).str), device=dev)
      if store is None:
        store = {}
      if k not in store:
        store[k] = torch.empty((steps,) + tuple(v.shape), dtype=v.dtype, device=dev)
      store[k][t].copy_(v)           # on torch's stream; the ring slot is reused 4 steps later
  launches, idle_stops, served, limit = g.server_stats()
  torch.cuda.synchronize()
  host = {k: store[k].cpu().numpy() for k in keys}
  for t in range(steps):
    o._send([ids, penv_all[t], np.repeat(lane_all[t][:, None], 6, axis=1), ids * 0, ids * 0])
    want = o._recv()
    got = [None if k == 9 else host[k][t] for k in range(12)]
    assert_batches_equal(got, want, f"bar={bar} step {t}")
  # (the first steps allocate the result store: an idle stop there is tolerated, a restart per step is not)
  assert served >= steps - 40 and launches <= 3 and launches <= 1 + idle_stops, (launches, idle_stops, served, limit)
  g.close()


def test_device_sync_between_steps_stays_cheap():
  """A caller that synchronises the whole device between two steps (torch.cuda.synchronize()) has
  to wait for the resident kernel to leave. The watchdog bounds that wait (200 us at first) and
  after idle stops in a row the pool serves such a caller through the launch path: 400 steps of a
  65 536-env pool with a device-wide synchronisation before each stay under 100 us per step on
  average (round 2: up to 50 ms each), oracle-equal."""
  import torch

  from envpool_amd import capi
  L = capi.lib()
  dev = torch.device("cuda", 0)
  n, steps = 65536, 400
  g, o, ids, t_ids, t_penv, t_la = _setup(n, 42, dev, torch)
  t_penv.copy_(t_ids)
  t_la.fill_(1.5)
  torch.cuda.synchronize()
  dims = (C.c_int64 * 5)(n, n, n, n, n)
  ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_penv.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
  b = capi.Batch()
  a = dummy_action(ids)
  spent = 0.0
  worst = 0.0
  for t in range(steps):
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    rc = L.ep_pool_step(g.handle, 5, ptrs, dims, C.byref(b))
    dt = time.perf_counter() - t0
    spent += dt
    worst = max(worst, dt)
    assert rc == 0, L.ep_last_error()
    o._send(a)
    want = o._recv()
    if t % 7 == 0 or t > steps - 5:
      assert_batches_equal(_views(b, dev), want, f"step {t}")
  launches, idle_stops, served, limit = g.server_stats()
  per_step_us = spent / steps * 1e6
  assert per_step_us < 100, (per_step_us, worst * 1e6, launches, idle_stops, served)
  assert worst < 0.02, worst          # no 50 ms stall any more
  assert served < steps               # the launch path took over
  g.close()


def test_step_begin_end_and_what_is_refused_in_between(monkeypatch):
  """ep_pool_step_begin rings, ep_pool_step_end waits: two pools stepped at once (what the sharded
  front-end does with one pool per GPU), each against its oracle; between the two halves the pool
  refuses its other entry points. (Both pools sit on the one GPU of this box, where the library
  would keep the resident server out of the way: EP_SERVER_SHARED_DEVICE=1 lets them have it.)"""
  import torch

  from envpool_amd import capi
  monkeypatch.setenv("EP_SERVER_SHARED_DEVICE", "1")
  monkeypatch.setenv("EP_SERVER_MAX_PER_DEVICE", "4")
  L = capi.lib()
  dev = torch.device("cuda", 0)
  n = 4096
  pools = []
  for r in range(2):
    g, o = _pair(num_envs=n, seed=5, ring_depth=4, env_id_offset=r * n)
    ids = np.arange(r * n, (r + 1) * n, dtype=np.int32)
    g._reset(ids); o._reset(ids)
    assert_batches_equal(g._recv(), o._recv(), "reset")
    t_ids = torch.from_numpy(ids).to(dev)
    t_la = torch.full((n, 6), 2.5, dtype=torch.float64, device=dev)
    pools.append((g, o, ids, t_ids, t_la))
  torch.cuda.synchronize()
  dims = (C.c_int64 * 5)(n, n, n, n, n)
  b = capi.Batch()
  for t in range(40):
    for g, o, ids, t_ids, t_la in pools:
      ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_ids.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
      assert L.ep_pool_step_begin(g.handle, 5, ptrs, dims) == 0, L.ep_last_error()
    if t == 0:
      # (the very first step rings the server; later ones may take the launch path, because this
      # loop - an oracle comparison per step - is slow enough for the idle watchdog to step in, and
      # a send + recv step refuses nothing in between)
      g0 = pools[0][0]
      assert L.ep_pool_sync(g0.handle) == capi.EP_ERR_INVALID and b"in flight" in L.ep_last_error()
    if t in (0, 3):
      assert L.ep_pool_step_begin(pools[0][0].handle, 5, ptrs, dims) == capi.EP_ERR_INVALID
    for g, o, ids, t_ids, t_la in pools:
      assert L.ep_pool_step_end(g.handle, C.byref(b)) == 0, L.ep_last_error()
      o._send(dummy_action(ids, value=2.5))
      assert_batches_equal(_views(b, dev), o._recv(), f"step {t}")
  assert L.ep_pool_step_end(pools[0][0].handle, C.byref(b)) == capi.EP_ERR_INVALID
  # a pool the server does not apply to (1000 envs): begin = send, end = recv
  g, o = _pair(num_envs=1000, seed=2, ring_depth=4)
  ids = np.arange(1000, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), "reset")
  t_ids = torch.from_numpy(ids).to(dev)
  t_la = torch.full((1000, 6), 1.0, dtype=torch.float64, device=dev)
  torch.cuda.synchronize()
  d2 = (C.c_int64 * 5)(1000, 1000, 1000, 1000, 1000)
  ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_ids.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
  for t in range(5):
    assert L.ep_pool_step_begin(g.handle, 5, ptrs, d2) == 0, L.ep_last_error()
    assert L.ep_pool_step_end(g.handle, C.byref(b)) == 0, L.ep_last_error()
    o._send(dummy_action(ids, value=1.0))
    assert_batches_equal(_views(b, dev), o._recv(), f"small pool step {t}")
  assert g.server_stats()[2] == 0
  g.close()
  for g, *_ in pools:
    g.close()


def test_server_displaced_rows_and_duplicates():
  """The server requests env i's record together with action row i (speculation) and only rows
  that name another env take the dependent load and the stamp exchange: permuted ids step like the
  oracle; an env named twice - by its own row and a displaced one, or by two displaced rows - is
  reported (EP_ERR_DUP_ENV, the batch still returned)."""
  import torch

  from envpool_amd import capi
  L = capi.lib()
  dev = torch.device("cuda", 0)
  n = 2048
  g, o, ids, t_ids, t_penv, t_la = _setup(n, 3, dev, torch)
  t_la.fill_(1.5)
  dims = (C.c_int64 * 5)(n, n, n, n, n)
  ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_penv.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
  b = capi.Batch()
  rng = np.random.default_rng(9)
  for t in range(12):
    sel = ids if t % 3 == 0 else rng.permutation(ids) if t % 3 == 1 else np.concatenate([ids[:1000], ids[1000:][::-1]])
    sel = np.ascontiguousarray(sel, dtype=np.int32)
    t_ids.copy_(torch.from_numpy(sel).to(dev))
    t_penv.copy_(t_ids)
    torch.cuda.current_stream(dev).synchronize()
    rc = L.ep_pool_step(g.handle, 5, ptrs, dims, C.byref(b))
    assert rc == 0, L.ep_last_error()
    o._send(dummy_action(sel))
    assert_batches_equal(_views(b, dev), o._recv(), f"step {t}")
  assert g.server_stats()[2] >= 1
  for case in ("own row + displaced row", "two displaced rows"):
    sel = ids.copy()
    if case == "own row + displaced row":
      sel[5] = 7                      # env 7 named by row 7 (its own) and by row 5
    else:
      sel[5], sel[9] = 700, 700       # env 700 named by rows 5 and 9 and by its own row ... so displace that one too
      sel[700] = 5
    t_ids.copy_(torch.from_numpy(sel).to(dev))
    t_penv.copy_(t_ids)
    torch.cuda.current_stream(dev).synchronize()
    rc = L.ep_pool_step(g.handle, 5, ptrs, dims, C.byref(b))
    assert rc == capi.EP_ERR_DUP_ENV, (case, rc, L.ep_last_error())
    assert b.num_rows == n
  g.close()


def test_step_on_stream_no_host_wait_matches_oracle():
  """ep_pool_step_on_stream through the C ABI: 200 steps of a 65 536-env pool with NO host wait
  anywhere in the loop - the actions are rewritten by torch kernels on a side stream, the pool's step
  stream waits for that stream, that stream waits for the rows and copies them out of the ring (whose
  4 slots come round 50 times) - compared with the oracle afterwards. Then a step with differing
  list_action lanes: EP_OK at the step, EP_ERR_ACTION_CHECK from ep_pool_poll_errors once the kernel
  has run, and nothing left for ep_pool_sync to report."""
  import torch

  from envpool_amd import capi, interop
  L = capi.lib()
  dev = torch.device("cuda", 0)
  n, steps = 65536, 200
  g, o, ids, t_ids, t_penv, t_la = _setup(n, 11, dev, torch)
  rng = np.random.default_rng(8)
  penv_all = ids[None, :] + (rng.random((steps, n)) < 0.1).astype(np.int32)
  lane_all = rng.integers(0, 5, size=(steps, n)).astype(np.float64)
  d_penv_all = torch.from_numpy(penv_all).to(dev)
  d_lane_all = torch.from_numpy(lane_all).to(dev)
  keys = [k for k in range(12) if k != 9]
  dims = (C.c_int64 * 5)(n, n, n, n, n)
  ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_penv.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
  b = capi.Batch()
  side = torch.cuda.Stream(device=dev)
  store, views = {}, {}
  torch.cuda.synchronize()
  with torch.cuda.stream(side):
    for t in range(steps):
      t_penv.copy_(d_penv_all[t])
      t_la.copy_(d_lane_all[t].unsqueeze(1).expand(n, 6))
      rc = L.ep_pool_step_on_stream(g.handle, C.c_void_p(side.cuda_stream), 5, ptrs, dims, C.byref(b))
      assert rc == 0, L.ep_last_error()
      assert b.num_rows == n
      for k in keys:
        a = b.arrays[k]
        v = views.get(a.data)
        if v is None:
          shape = tuple(a.shape[i] for i in range(a.ndim))
          v = views[a.data] = interop.as_torch((a.data, shape, np.dtype(capi._NP[a.dtype]).str), device=dev)
        if k not in store:
          store[k] = torch.empty((steps,) + tuple(v.shape), dtype=v.dtype, device=dev)
        store[k][t].copy_(v)
  assert len(views) == 4 * len(keys)  # the ring's four slots, again and again
  # the last batch's `terminated`, written by its step kernel beside the slot: no launch, a pointer
  term = C.c_void_p()
  assert L.ep_pool_terminated_of(g.handle, C.byref(b), C.byref(term)) == 0, L.ep_last_error()
  t_term = interop.as_torch((term.value, (n,), "|b1"), device=dev)
  side.synchronize()
  last = {k: views[b.arrays[k].data].cpu().numpy() for k in (3, 7)}
  assert t_term.cpu().numpy().tobytes() == (last[3] & ~last[7]).tobytes()
  assert L.ep_pool_poll_errors(g.handle) == 0, L.ep_last_error()
  host = {k: store[k].cpu().numpy() for k in keys}
  for t in range(steps):
    o._send([ids, penv_all[t], np.repeat(lane_all[t][:, None], 6, axis=1), ids * 0, ids * 0])
    want = o._recv()
    got = [None if k == 9 else host[k][t] for k in range(12)]
    assert_batches_equal(got, want, f"step {t}")
  # a flagged step: reported by the poll after the kernel has run, not by the step
  with torch.cuda.stream(side):
    t_la[77, 2] = 9.0
    rc = L.ep_pool_step_on_stream(g.handle, C.c_void_p(side.cuda_stream), 5, ptrs, dims, C.byref(b))
    assert rc == 0, L.ep_last_error()
  side.synchronize()
  assert L.ep_pool_poll_errors(g.handle) == capi.EP_ERR_ACTION_CHECK
  assert b"self-check" in L.ep_last_error()
  assert L.ep_pool_poll_errors(g.handle) == 0
  assert L.ep_pool_sync(g.handle) == 0, L.ep_last_error()
  # the same loop with the side stream ADOPTED by the pool (ep_pool_set_stream): its kernels run on
  # that stream, in order with the torch kernels around them, with no event between the two
  o._send([ids, penv_all[-1], np.repeat(lane_all[-1][:, None], 6, axis=1), ids * 0, ids * 0])
  o._recv()  # (the flagged step: same state change, the oracle is given clean lanes)
  assert L.ep_pool_set_stream(g.handle, C.c_void_p(side.cuda_stream), 0) == 0, L.ep_last_error()
  assert L.ep_pool_stream(g.handle) == side.cuda_stream
  with torch.cuda.stream(side):
    for t in range(60):
      t_penv.copy_(d_penv_all[t])
      t_la.copy_(d_lane_all[t].unsqueeze(1).expand(n, 6))
      rc = L.ep_pool_step_on_stream(g.handle, C.c_void_p(side.cuda_stream), 5, ptrs, dims, C.byref(b))
      assert rc == 0, L.ep_last_error()
      for k in keys:
        store[k][t].copy_(views[b.arrays[k].data])
  side.synchronize()
  assert L.ep_pool_poll_errors(g.handle) == 0, L.ep_last_error()
  host = {k: store[k][:60].cpu().numpy() for k in keys}
  for t in range(60):
    o._send([ids, penv_all[t], np.repeat(lane_all[t][:, None], 6, axis=1), ids * 0, ids * 0])
    want = o._recv()
    assert_batches_equal([None if k == 9 else host[k][t] for k in range(12)], want, f"adopted stream, step {t}")
  # the resident kernel of ep_pool_step writes the array as well (several steps, so that envs finish);
  # a rollout's batches carry none
  assert L.ep_pool_set_stream(g.handle, None, 1) == 0, L.ep_last_error()
  t_penv.copy_(t_ids); torch.cuda.synchronize()
  seen = 0
  for t in range(40):
    assert L.ep_pool_step(g.handle, 5, ptrs, dims, C.byref(b)) == 0, L.ep_last_error()
    o._send([ids, ids, np.repeat(lane_all[59][:, None], 6, axis=1), ids * 0, ids * 0])
    want = o._recv()
    assert L.ep_pool_terminated_of(g.handle, C.byref(b), C.byref(term)) == 0, L.ep_last_error()
    got = interop.as_torch((term.value, (n,), "|b1"), device=dev).cpu().numpy()
    assert got.tobytes() == (want[3] & ~want[7]).tobytes(), t
    seen += int(got.sum())
  assert seen > 0 and g.server_stats()[2] > 0
  strides = (C.c_int64 * 5)(0, 0, 0, 0, 0)
  assert L.ep_pool_rollout(g.handle, 2, 5, ptrs, dims, strides) == 0, L.ep_last_error()
  two = (capi.Batch * 2)()
  assert L.ep_pool_recv_many(g.handle, capi.EP_MEM_DEVICE, 2, two) == 0, L.ep_last_error()
  assert L.ep_pool_terminated_of(g.handle, C.byref(two[1]), C.byref(term)) == capi.EP_ERR_INVALID and term.value is None
  for t in range(2):
    o._send([ids, ids, np.repeat(lane_all[59][:, None], 6, axis=1), ids * 0, ids * 0]); o._recv()
  # back on its own stream the blocking calls are what they were
  assert L.ep_pool_set_stream(g.handle, None, 1) == 0, L.ep_last_error()
  assert L.ep_pool_stream(g.handle) != side.cuda_stream
  a = dummy_action(ids)
  g._send(a); o._send(a)
  assert_batches_equal(g._recv(), o._recv(), "numpy step after returning to the pool's own stream")
  g.close()
  # multi-player pools are refused, as by ep_pool_recv_on_stream
  g3, _ = _pair(num_envs=256, seed=1, max_num_players=3)
  assert L.ep_pool_step_on_stream(g3.handle, None, 5, ptrs, dims, C.byref(b)) == capi.EP_ERR_INVALID
  g3.close()


def test_step_torch_stream_ordered_front_end():
  """env.step_torch(..., stream_ordered=True) of the unsharded Python front-end: the pool adopts
  torch's current stream, results come from the per-slot cache (python/streamstep.py) with
  `terminated` refreshed by ep_terminated_many; compared with the oracle at every step (reading a
  result on the host synchronises the stream), then 300 unread steps in a row, then a flagged step."""
  import torch

  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  n, seed = 4096, 4
  env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=seed)
  one = DummyOracle(num_envs=n, seed=seed)
  ids = np.arange(n, dtype=np.int32)
  dev = torch.device("cuda", 0)
  env.reset_torch()
  one._reset(ids); one._recv()
  la = torch.full((n, 6), 1.5, dtype=torch.float64, device=dev)
  penv = torch.from_numpy(ids).to(dev)
  act = {"list_action": la, "players.action": penv, "players.id": penv, "players.env_id": penv}
  rng = np.random.default_rng(1)

  def check(res, want, ctx):
    obs, rew, term, trunc, info = res
    assert obs["raw"].cpu().numpy().tobytes() == want[8].tobytes(), ctx
    assert rew.cpu().numpy().tobytes() == want[4].tobytes() and trunc.cpu().numpy().tobytes() == want[7].tobytes(), ctx
    assert term.cpu().numpy().tobytes() == (want[3] & ~want[7]).tobytes(), ctx
    assert info["env_id"].cpu().numpy().tobytes() == want[0].tobytes() and info["elapsed_step"].cpu().numpy().tobytes() == want[2].tobytes(), ctx

  seen_term = False
  for t in range(40):
    pe = ids + (rng.random(n) < 0.1).astype(np.int32)
    penv.copy_(torch.from_numpy(pe).to(dev))
    res = env.step_torch(act, stream_ordered=True)
    one._send(dummy_action(ids, players_env_id=pe))
    want = one._recv()
    check(res, want, f"step {t}")
    seen_term = seen_term or bool((want[3] & ~want[7]).any())
  for t in range(300):
    res = env.step_torch(act, stream_ordered=True)
    assert "mine" not in res[-1] and "mine" not in res[0]  # (the dicts around the kept views are new ones every time)
    res[-1]["mine"] = res[0]["mine"] = t
    one._send(dummy_action(ids, players_env_id=pe))
    want = one._recv()
    seen_term = seen_term or bool((want[3] & ~want[7]).any())
  check(res, want, "after 300 unread steps")
  assert seen_term  # (the comparison of `terminated` saw ones, not only zeros)
  la[5, 3] = 2.5
  env.step_torch(act, stream_ordered=True)
  one._send(dummy_action(ids, players_env_id=pe)); one._recv()
  torch.cuda.current_stream().synchronize()
  la[5, 3] = 1.5
  with pytest.raises(RuntimeError, match="self-check"):
    env.step_torch(act, stream_ordered=True)
  # the blocking form and the numpy boundary still work on the same pool (its stream stays adopted)
  res = env.step_torch(act, stream_ordered=False)
  one._send(dummy_action(ids, players_env_id=pe))
  check(res, one._recv(), "blocking step afterwards")
  a = dummy_action(ids)
  env._send(a); one._send(a)
  assert_batches_equal(env._recv(), one._recv(), "numpy step afterwards")
  env.close()


def test_terminated_many_odd_shapes():
  """ep_terminated_many through the C ABI: 20 arrays in one call (two launches of 16 and 4), lengths
  from 0 to 70 001 bytes, pointers 16-byte aligned and not (the bytewise path and the tails), against
  numpy; bytes past each array's end stay untouched; bad arguments are refused."""
  import torch

  from envpool_amd import capi
  L = capi.lib()
  dev = torch.device("cuda", 0)
  rng = np.random.default_rng(12)
  lens = [0, 1, 15, 16, 17, 255, 256, 4095, 4096, 4097, 65536, 70001, 33, 64, 1000, 31, 48, 7, 12345, 2]
  offs = [0, 0, 0, 0, 0, 3, 0, 1, 0, 16, 0, 5, 8, 0, 2, 0, 0, 9, 0, 1]
  k = len(lens)
  pad = 64
  bufs, want, outs = [], [], []
  dptr, tptr, optr = (C.c_void_p * k)(), (C.c_void_p * k)(), (C.c_void_p * k)()
  n = (C.c_int64 * k)(*lens)
  for i, (ln, off) in enumerate(zip(lens, offs)):
    d = rng.integers(0, 2, ln, dtype=np.uint8)
    t = (rng.integers(0, 2, ln, dtype=np.uint8) & d) if i % 3 else rng.integers(0, 2, ln, dtype=np.uint8)  # trunc without done too
    td = torch.zeros(ln + off + pad, dtype=torch.uint8, device=dev); td[off:off + ln] = torch.from_numpy(d).to(dev)
    tt = torch.zeros(ln + off + pad, dtype=torch.uint8, device=dev); tt[off:off + ln] = torch.from_numpy(t).to(dev)
    to = torch.full((ln + off + pad,), 0x5A, dtype=torch.uint8, device=dev)
    bufs.append((td, tt))
    outs.append(to)
    want.append(d & ~t & 1)
    dptr[i], tptr[i], optr[i] = td.data_ptr() + off, tt.data_ptr() + off, to.data_ptr() + off
  torch.cuda.synchronize()
  stream = torch.cuda.current_stream(dev).cuda_stream
  assert L.ep_terminated_many(k, dptr, tptr, optr, n, 0, C.c_void_p(stream)) == 0, L.ep_last_error()
  torch.cuda.synchronize()
  for i, (ln, off) in enumerate(zip(lens, offs)):
    got = outs[i].cpu().numpy()
    assert got[off:off + ln].tobytes() == want[i].astype(np.uint8).tobytes(), i
    assert (got[:off] == 0x5A).all() and (got[off + ln:] == 0x5A).all(), i
  assert L.ep_terminated_many(0, None, None, None, None, -1, None) == 0
  assert L.ep_terminated_many(-1, dptr, tptr, optr, n, -1, None) == capi.EP_ERR_INVALID
  bad = (C.c_int64 * k)(*([-1] + lens[1:]))
  assert L.ep_terminated_many(k, dptr, tptr, optr, bad, -1, None) == capi.EP_ERR_INVALID


def test_entry_points_leave_the_current_device_alone():
  """A pool or encoder call runs on ITS device and leaves the calling thread's current HIP device
  as it found it (torch's notion of the current device is HIP's). One GPU here: the calls are made
  with the device set to 0 and with whatever the library would have to switch from; what can be
  checked on one device is that nothing raises and the current device reads back unchanged."""
  import torch

  from envpool_amd import capi
  L = capi.lib()
  L.hipGetDevice.argtypes = [C.POINTER(C.c_int)]
  cur = C.c_int(-1)
  assert L.hipGetDevice(C.byref(cur)) == 0
  before = cur.value
  g, o = _pair(num_envs=256, seed=2)
  ids = np.arange(256, dtype=np.int32)
  g._reset(ids); g._recv()
  g._send(dummy_action(ids)); g._recv()
  assert L.ep_stream_synchronize(0, None) == 0 and L.ep_pool_poll_errors(g.handle) == 0
  g.close()
  assert L.hipGetDevice(C.byref(cur)) == 0 and cur.value == before
  assert torch.cuda.current_device() == before


@pytest.mark.parametrize("host_const", ["1", "0"])
def test_host_memory_batches_with_and_without_host_written_arrays(host_const, monkeypatch):
  """A batch whose rows go to host memory has four of its eleven arrays written by the sending thread
  (reward and info:players.id: zero for every env; info:env_id / info:players.env_id: the ids that were
  sent) instead of by the kernel over PCIe (EP_HOST_CONST=1, the default; =0: the kernel writes all
  eleven). Both against the oracle: whole-pool and partial sends, an env named twice in one send (split
  launches), a reset of a subset, a device-memory send into the same kind of batch, an async pool whose
  sends straddle batches."""
  monkeypatch.setenv("EP_HOST_CONST", host_const)
  rng = np.random.default_rng(8)
  for kw in (dict(num_envs=40000, seed=5), dict(num_envs=5000, batch_size=1300, seed=6)):
    g, o = _pair(**kw)
    n = kw["num_envs"]
    B = kw.get("batch_size", 0) or n
    ids = np.arange(n, dtype=np.int32)
    g._reset(ids); o._reset(ids)
    for it in range(14 if B == n else 40):   # (every batch received is sent back: one is always there)
      got, want = g._recv(), o._recv()
      assert_batches_equal(got, want, f"{kw} it={it}")
      back = np.asarray(got[0])
      if B == n:
        if it % 4 == 1:
          back = rng.permutation(back)
        elif it % 4 == 2:
          back = np.concatenate([back[:100], back[:3], back[100:n - 3]])   # three envs named twice, three left out
        elif it % 4 == 3:
          sub = np.sort(rng.choice(n, size=n // 3, replace=False)).astype(np.int32)
          g._reset(sub); o._reset(sub)
          assert_batches_equal(g._recv(), o._recv(), f"{kw} it={it} subset reset")
      pe = back + (rng.random(back.size) < 0.1).astype(np.int32)
      a = dummy_action(back, players_env_id=pe, value=float(it))
      if it % 5 == 4 and B == n and len(set(back.tolist())) == back.size:
        g._send_via_device(a)
      else:
        g._send(a)
      o._send(a)
