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


def test_rollout_at_bench_size():
  """BASELINE config 2 as bench.py runs it: 65 536 envs, seed 42, T = 128 steps per launch, ring
  depth 256 - every step of the fused loop against the oracle, plus the episode statistics."""
  _rollout_case(65536, 128, seed=42, ring_depth=256, rounds=1)


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
  # the failure is reported with the batch of the step that failed (t = 2), not with the launch
  for t in range(T):
    g._recv(allow=(capi.EP_ERR_ACTION_CHECK,))
    assert (g.last_status == capi.EP_ERR_ACTION_CHECK) == (t == 2), t


def test_async_mode_fifo_batches():
  """batch_size < num_envs: rows follow the send order (the reference's arrival order with one
  worker); batches straddle sends; recv reports would-block instead of deadlocking."""
  from envpool_amd import capi
  rng = np.random.default_rng(3)
  g, o = _pair(num_envs=64, batch_size=24, seed=4)
  g.set_recv_timeout(0)  # the default waits like StateBuffer::Wait; this test wants the refusal
  ids = np.arange(64, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  outstanding = 64
  for it in range(60):
    while outstanding >= 24:
      assert_batches_equal(g._recv(), o._recv(), f"it {it}")
      outstanding -= 24
    with pytest.raises(capi.EpError) as e:
      g._recv()
    assert e.value.code == capi.EP_ERR_WOULD_BLOCK
    k = int(rng.integers(1, 40))
    sel = rng.choice(64, size=k, replace=False).astype(np.int32)
    a = dummy_action(sel)
    g._send(a); o._send(a)
    outstanding += k


def test_sync_partial_and_quirks():
  g, o = _pair(num_envs=8, seed=5)
  ids = np.arange(8, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), "reset")
  for sel in ([5, 1, 6], list(range(8)), [7], [2, 3, 4, 0]):
    a = dummy_action(np.array(sel, np.int32))
    g._send(a); o._send(a)
    assert_batches_equal(g._recv(), o._recv(), f"partial {sel}")
  # forced reset of a subset mid-episode
  g._reset(ids[[3, 0]]); o._reset(ids[[3, 0]])
  assert_batches_equal(g._recv(), o._recv(), "subset reset")
  # two partial sends, one recv: same rows, the tail rows must read as zero
  for sel in ([0, 1, 2], [3, 4, 5]):
    a = dummy_action(np.array(sel, np.int32))
    g._send(a); o._send(a)
  assert_batches_equal(g._recv(), o._recv(), "two sends one recv")
  # 5 + 5 of 8: the second send straddles two StateBuffers
  for sel in ([0, 1, 2, 3, 4], [5, 6, 7, 0, 1]):
    a = dummy_action(np.array(sel, np.int32))
    g._send(a); o._send(a)
  assert_batches_equal(g._recv(), o._recv(), "straddle 1")
  assert_batches_equal(g._recv(), o._recv(), "straddle 2")
  # recv with nothing outstanding returns an empty batch
  assert_batches_equal(g._recv(), o._recv(), "empty")


def test_trunc_seed_clamp_mismatch_and_state_num():
  for kw in (dict(num_envs=16, seed=1, max_episode_steps=3), dict(num_envs=8, seed=-5),
             dict(num_envs=33, seed=0, max_episode_steps=1, state_num=7), dict(num_envs=70, seed=2, state_num=3)):
    g, o = _pair(**kw)
    n = kw["num_envs"]
    ids = np.arange(n, dtype=np.int32)
    g._reset(ids); o._reset(ids)
    assert_batches_equal(g._recv(), o._recv(), f"{kw} reset")
    for t in range(20):
      pe = ids.copy()
      pe[t % n] = -1  # players.env_id != env_id -> obs:raw[:,1] = 0 for that row
      a = dummy_action(ids, players_env_id=pe)
      g._send(a); o._send(a)
      assert_batches_equal(g._recv(), o._recv(), f"{kw} t={t}")


def test_sharded_pool_uses_global_ids():
  g, o = _pair(num_envs=256, seed=42, env_id_offset=65536 * 3)
  ids = np.arange(256, dtype=np.int32) + 65536 * 3
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), "reset")
  for t in range(5):
    a = dummy_action(ids)
    g._send(a); o._send(a)
    assert_batches_equal(g._recv(), o._recv(), f"t={t}")
  from envpool_amd import capi
  with pytest.raises(capi.EpError) as e:
    g._send(dummy_action(ids - 1))
  assert e.value.code == capi.EP_ERR_INVALID


def test_ring_full_and_action_check():
  from envpool_amd import capi
  g, o = _pair(num_envs=2, seed=3)
  ids = np.arange(2, dtype=np.int32)
  for _ in range(o.queue_size):
    g._send(dummy_action(ids)); assert o._send(dummy_action(ids)) == 0
  with pytest.raises(capi.EpError) as e:
    g._send(dummy_action(ids))
  assert e.value.code == capi.EP_ERR_RING_FULL and o._send(dummy_action(ids)) == 5
  g2, _ = _pair(num_envs=4, seed=9)
  g2._reset(np.arange(4, dtype=np.int32)); g2._recv()
  a = dummy_action(np.arange(4, dtype=np.int32))
  a[2][1, 4] = 7.0
  g2._send(a)
  g2._recv(allow=(capi.EP_ERR_ACTION_CHECK,))
  assert g2.last_status == capi.EP_ERR_ACTION_CHECK


@pytest.mark.parametrize("P,n,B", [(3, 4, 0), (2, 16, 0), (4, 300, 0), (3, 40, 16), (5, 2000, 512), (3, 40000, 0),
                                   (4, 70000, 30000)])
def test_multiplayer_ragged_player_dim(P, n, B):
  """max_num_players > 1: player rows are a device prefix sum."""
  _mp_ragged(P, n, B)


def _mp_ragged(P, n, B, via_device=False):
  g, o = _pair(num_envs=n, batch_size=B, max_num_players=P, seed=2)
  ids = np.arange(n, dtype=np.int32)
  bsz = B or n
  g._reset(ids); o._reset(ids)
  outstanding, last_pe = n, {}
  rng = np.random.default_rng(P)
  for it in range(25):
    got_ids = []
    while outstanding >= bsz:
      gr, orr = g._recv(), o._recv()
      assert_batches_equal(gr, orr, f"P={P} it={it}")
      got_ids.append(orr[1])
      outstanding -= bsz
    # send back every env we have seen, with one action row per player it reported
    pe = np.concatenate(got_ids) if got_ids else np.zeros(0, np.int32)
    sel = np.unique(pe) if pe.size else ids[:0]
    if sel.size == 0:
      sel = ids[: bsz - outstanding]; pe = sel
    perm = rng.permutation(sel.size)
    sel = sel[perm].astype(np.int32)
    a = dummy_action(sel, players_env_id=pe)
    (g._send_via_device if via_device else g._send)(a); o._send(a)
    outstanding += sel.size
  return g.episode_stats()


@pytest.mark.parametrize("P,n,B", [(3, 40, 0), (3, 40, 16), (4, 5000, 1500), (3, 16384, 0)])
def test_multiplayer_device_memory_sends(P, n, B):
  """The same with every send from DEVICE memory: the compile-time device case of the fused kernel
  (mp_step_fused<true>: record, episode return, histogram count, action rows and the duplicate
  stamp's exchange requested together), episode statistics included. Then what only that path
  reports: an env named twice in one send (EP_ERR_DUP_ENV at the recv of that batch) and differing
  list_action lanes (EP_ERR_ACTION_CHECK)."""
  from envpool_amd import capi
  # (the same sends through host memory take the generic kernel: the episode statistics must agree)
  for x, y in zip(_mp_ragged(P, n, B, via_device=True), _mp_ragged(P, n, B)):
    assert x.tobytes() == y.tobytes()
  if B:
    return
  g, o = _pair(num_envs=n, max_num_players=P, seed=4)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  want = o._recv()
  assert_batches_equal(g._recv(), want, "reset")
  pe = np.asarray(want[1]).astype(np.int32)
  a = dummy_action(ids, players_env_id=pe)
  a[2][n // 2, 3] += 1.0                      # lanes differ for one env
  g._send_via_device(a)
  g._recv(allow=(capi.EP_ERR_ACTION_CHECK,))
  assert g.last_status == capi.EP_ERR_ACTION_CHECK
  twice = ids.copy(); twice[1] = twice[0]     # one env twice, another not at all
  g._send_via_device(dummy_action(twice, players_env_id=pe))
  g._recv(allow=(capi.EP_ERR_DUP_ENV,))
  assert g.last_status == capi.EP_ERR_DUP_ENV


def test_multiplayer_ticket_tile_ids(monkeypatch):
  """The look-back scan with tile ids from the ticket counter (what a launch too large for the
  device takes) gives the same rows as with static tile ids."""
  monkeypatch.setenv("EP_MP_FORCE_TICKET", "1")
  test_multiplayer_ragged_player_dim(3, 3000, 0)
  test_multiplayer_ragged_player_dim(4, 1500, 500)


def test_multiplayer_folded_histogram(monkeypatch):
  """EP_MP_FOLD=1: the histogram as a phase of the fused kernel behind a grid-wide wait (one launch
  per send where the send is one launch of at most two workgroups per CU; DESIGN.md section 3.4: built,
  measured, slower than two launches, kept behind the switch). Same rows as the oracle: sends that
  fold (sync pools, 4 and 32 768 envs), sends that do not (an async pool whose sends straddle batches),
  and a batch whose rows all name one env."""
  monkeypatch.setenv("EP_MP_FOLD", "1")
  test_multiplayer_ragged_player_dim(3, 4, 0)
  test_multiplayer_ragged_player_dim(3, 32768, 0)
  test_multiplayer_ragged_player_dim(5, 2000, 512)
  test_multiplayer_all_rows_name_one_env()


def test_multiplayer_all_rows_name_one_env():
  """Every players.env_id row names the same env: that env's action_num is the whole row count
  (env.h:128-132), everyone else's 0 - and the row histogram must not crawl on one counter."""
  n, P = 4096, 3
  g, o = _pair(num_envs=n, max_num_players=P, seed=5)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), "reset")
  import time
  for t, target in enumerate((7, 0, n - 1)):
    pe = np.full(2 * n, target, dtype=np.int32)
    a = dummy_action(ids, players_env_id=pe)
    t0 = time.perf_counter()
    g._send(a); gr = g._recv()
    assert time.perf_counter() - t0 < 2.0
    o._send(a)
    assert_batches_equal(gr, o._recv(), f"t={t}")


def test_checkpoint_roundtrip():
  import ctypes as C
  from envpool_amd import capi
  g, o = _pair(num_envs=100, seed=3)
  L = capi.lib()
  ids = np.arange(100, dtype=np.int32)
  g._reset(ids); o._reset(ids); g._recv(); o._recv()
  for _ in range(7):
    g._send(dummy_action(ids)); o._send(dummy_action(ids)); g._recv(); o._recv()
  nbytes = L.ep_pool_checkpoint_bytes(g.handle)
  buf = C.create_string_buffer(nbytes)
  capi._check(L.ep_pool_checkpoint_save(g.handle, buf))
  want = []
  for _ in range(9):
    g._send(dummy_action(ids)); o._send(dummy_action(ids)); want.append(o._recv()); assert_batches_equal(g._recv(), want[-1])
  capi._check(L.ep_pool_checkpoint_restore(g.handle, buf))
  for w in want:
    g._send(dummy_action(ids)); assert_batches_equal(g._recv(), w, "after restore")


def test_recv_on_stream_policy_loop():
  """ep_pool_recv_on_stream: a GPU-resident policy (torch, running ON the pool's step stream) reads
  each batch as device views and sends actions computed from it; the host never waits inside the
  loop. Every step against the oracle driven by the same policy in numpy."""
  import ctypes as C

  import torch

  from envpool_amd import capi, interop

  n, T = 4096, 40
  g, o = _pair(num_envs=n, seed=7, ring_depth=4)   # far fewer slots than steps: the views are consumed in order
  L = capi.lib()
  dev = torch.device("cuda", 0)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  t_ids = torch.from_numpy(ids).to(dev)
  t_z = torch.zeros(n, dtype=torch.int32, device=dev)
  t_penv = torch.empty(n, dtype=torch.int32, device=dev)
  t_la = torch.empty((n, 6), dtype=torch.float64, device=dev)
  torch.cuda.synchronize()
  ext = torch.cuda.ExternalStream(L.ep_pool_stream(g.handle))
  ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_penv.data_ptr(), t_la.data_ptr(), t_z.data_ptr(), t_z.data_ptr())
  dims = (C.c_int64 * 5)(n, n, n, n, n)
  b = capi.Batch()
  logged = []

  def view(k):
    a = b.arrays[k]
    shape = tuple(a.shape[i] for i in range(a.ndim))
    return interop.as_torch((a.data, shape, np.dtype(capi._NP[a.dtype]).str), device=dev)

  with torch.cuda.stream(ext):
    for t in range(T):
      capi._check(L.ep_pool_recv_on_stream(g.handle, None, C.byref(b)))
      assert b.num_rows == n
      elapsed, raw = view(2), view(8)
      logged.append([view(k).clone() for k in range(b.num_keys) if b.arrays[k].dtype != capi.EP_CONTAINER_I32])
      t_penv.copy_(t_ids + (elapsed % 3 == 0).to(torch.int32))   # a mismatch on some rows
      t_la.copy_(((elapsed + raw[:, 0]) % 5).to(torch.float64).unsqueeze(1).expand(n, 6))
      capi._check(L.ep_pool_send(g.handle, 5, ptrs, dims, capi.EP_MEM_DEVICE))
  capi._check(L.ep_pool_sync(g.handle))
  fixed = [k for k in range(b.num_keys) if b.arrays[k].dtype != capi.EP_CONTAINER_I32]
  for t in range(T):
    want = o._recv()
    for j, k in enumerate(fixed):
      got = logged[t][j].cpu().numpy()
      assert got.tobytes() == np.asarray(want[k]).tobytes(), f"t={t} key {k}"
    el, rw = np.asarray(want[2]), np.asarray(want[8])
    la = np.repeat(((el + rw[:, 0]) % 5).astype(np.float64)[:, None], 6, axis=1)
    o._send([ids, ids + (el % 3 == 0).astype(np.int32), la, np.zeros(n, np.int32), np.zeros(n, np.int32)])
  # the batch of the last send is still outstanding: the blocking form drains it
  assert_batches_equal(g._recv(), o._recv(), "drain")
  # multi-player pools cannot be received this way
  g2, _ = _pair(num_envs=8, seed=1, max_num_players=2)
  g2._reset(np.arange(8, dtype=np.int32))
  assert L.ep_pool_recv_on_stream(g2.handle, None, C.byref(b)) == capi.EP_ERR_INVALID


def test_direct_host_mode_and_mixed_recv_modes():
  """A whole sync batch sent from host buffers is stepped by kernels that read the pinned staging
  area and write a pinned result block over PCIe themselves (no copy engine). Received as host
  arrays, as blocking device views and as stream-ordered device views in turn - the last two find
  the batch in the host block and bring it into the ring - plus a device-resident send in between
  and an action-check failure: every step against the oracle."""
  import ctypes as C

  import torch

  from envpool_amd import capi, interop

  n = 8192
  g, o = _pair(num_envs=n, seed=11, ring_depth=3)
  L = capi.lib()
  dev = torch.device("cuda", 0)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), "reset")
  b = capi.Batch()
  rng = np.random.default_rng(5)

  def device_batch():
    out = []
    for k in range(b.num_keys):
      a = b.arrays[k]
      if a.dtype == capi.EP_CONTAINER_I32:
        out.append(None)
        continue
      shape = tuple(a.shape[i] for i in range(a.ndim))
      out.append(interop.as_torch((a.data, shape, np.dtype(capi._NP[a.dtype]).str), device=dev).cpu().numpy())
    return out

  for t in range(12):
    penv = ids + (rng.random(n) < 0.1).astype(np.int32)
    la = np.repeat(rng.integers(0, 5, size=(n, 1)).astype(np.float64), 6, axis=1)
    bad = t == 7
    if bad:
      la[123, 2] += 1.0  # lanes differ: the reference aborts, here the batch is flagged
    acts = [ids, penv, la, np.zeros(n, np.int32), np.zeros(n, np.int32)]
    if t % 4 == 3:  # device-resident send between the host ones
      tens = [torch.from_numpy(x).to(dev) for x in acts]
      torch.cuda.synchronize()
      capi._check(L.ep_pool_send(g.handle, 5, (C.c_void_p * 5)(*[x.data_ptr() for x in tens]), (C.c_int64 * 5)(*[n] * 5), capi.EP_MEM_DEVICE))
    else:
      g._send(acts)
    want = None
    if not bad:
      o._send(acts)
      want = o._recv()
    mode = t % 3
    if mode == 0:
      got = g._recv(allow=(capi.EP_ERR_ACTION_CHECK,))
      assert (g.last_status == capi.EP_ERR_ACTION_CHECK) == bad
    elif mode == 1:
      rc = L.ep_pool_recv(g.handle, capi.EP_MEM_DEVICE, C.byref(b))
      assert rc == (capi.EP_ERR_ACTION_CHECK if bad else 0)
      got = device_batch()
    else:
      capi._check(L.ep_pool_recv_on_stream(g.handle, None, C.byref(b)))
      rc = L.ep_pool_sync(g.handle)
      assert rc == (capi.EP_ERR_ACTION_CHECK if bad else 0)
      got = device_batch()
    if bad:
      break  # the oracle (like the reference) cannot go past a failed self-check
    assert_batches_equal(got, want, f"t={t} recv mode {mode}")
