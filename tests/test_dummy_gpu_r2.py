"""GPU parity, round-2 cases: an env named twice in one send, the threading contract of
send / recv, blocking recv, unaligned device views, per-step error attribution."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from tests.util import assert_batches_equal, dummy_action

pytestmark = pytest.mark.gpu


def _pair(**kw):
  from envpool_amd.capi import DummyPoolC
  from oracle.dummy_oracle import DummyOracle
  okw = {k: v for k, v in kw.items() if k not in ("ring_depth", "device_id")}
  return DummyPoolC(**kw), DummyOracle(**okw)


@pytest.mark.parametrize("kw", [dict(num_envs=8, seed=3), dict(num_envs=8, batch_size=3, seed=3),
                                dict(num_envs=300, batch_size=7, seed=2), dict(num_envs=5000, seed=1),
                                dict(num_envs=12, max_num_players=3, seed=2)])
def test_env_named_twice_in_one_host_send_is_stepped_in_order(kw):
  """The reference's single worker steps an env that a send names twice twice, in order
  (async_envpool.h:57-80, 116-129); host-memory sends are split into launches at every repeat."""
  g, o = _pair(**kw)
  g.set_recv_timeout(0)
  n = kw["num_envs"]
  B = kw.get("batch_size", 0) or n
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  rng = np.random.default_rng(n)
  outstanding = n
  sends = [[3, 5, 3], [1, 1, 1, 1], [0, 2, 4, 0, 2, 4, 6]]
  for it in range(30):
    while outstanding >= B:
      assert_batches_equal(g._recv(), o._recv(), f"{kw} it {it}")
      outstanding -= B
    sel = np.array(sends[it] if it < len(sends) else rng.integers(0, n, size=int(rng.integers(1, min(n, 600) + 1))), np.int32)
    if g._L.ep_pool_is_sync(g.handle):
      # a sync pool hands the batch out whatever its size; keep the sends within one batch
      sel = sel[:B]
    a = dummy_action(sel)
    g._send(a); o._send(a)
    outstanding += len(sel)
    if g._L.ep_pool_is_sync(g.handle):
      assert_batches_equal(g._recv(), o._recv(), f"{kw} it {it} sync")
      outstanding = 0


def test_env_named_twice_in_a_device_send_is_reported():
  """Device-memory ids cannot be looked at by the host: the items of one launch run concurrently and
  the kernels' per-env launch stamps flag the repeat (EP_ERR_DUP_ENV at recv)."""
  import torch

  from envpool_amd import capi
  L = capi.lib()
  dev = torch.device("cuda", 0)
  for kw in (dict(num_envs=16, seed=3), dict(num_envs=16, batch_size=3, seed=3, ring_depth=32),
             dict(num_envs=15, batch_size=3, max_num_players=2, seed=3, ring_depth=32)):
    g, _ = _pair(**kw)
    ids = np.arange(kw["num_envs"], dtype=np.int32)
    g._reset(ids)
    L.ep_pool_set_recv_timeout(g.handle, 0)
    for _ in range(64):   # drain the reset rows (a sync pool answers an idle recv with an empty batch)
      b = capi.Batch()
      if L.ep_pool_recv(g.handle, capi.EP_MEM_DEVICE, C.byref(b)) != 0 or b.num_rows == 0:
        break
    sends = [([3, 5, 3], capi.EP_ERR_DUP_ENV), ([3, 5, 7], 0), ([2, 2, 2], capi.EP_ERR_DUP_ENV), ([3, 5, 7], 0)]
    if kw["num_envs"] % kw.get("batch_size", 1) == 1:
      # 16 reset rows = 5 batches + 1 row: two more rows put the queue on a batch boundary, so that
      # every 3-item send below is ONE launch (a send that straddles two batches is two launches,
      # in order, and a repeat across them is no race)
      g._send(dummy_action(np.array([0, 1], np.int32)))
      assert L.ep_pool_recv(g.handle, capi.EP_MEM_DEVICE, C.byref(b)) == 0 and b.num_rows == 3
    for sel, want in sends:
      t_ids = torch.tensor(sel, dtype=torch.int32, device=dev)
      t_la = torch.ones((3, 6), dtype=torch.float64, device=dev)
      t_z = torch.zeros(3, dtype=torch.int32, device=dev)
      torch.cuda.synchronize()
      ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_ids.data_ptr(), t_la.data_ptr(), t_z.data_ptr(), t_z.data_ptr())
      capi._check(L.ep_pool_send(g.handle, 5, ptrs, (C.c_int64 * 5)(3, 3, 3, 3, 3), capi.EP_MEM_DEVICE))
      b = capi.Batch()
      rc = L.ep_pool_recv(g.handle, capi.EP_MEM_DEVICE, C.byref(b))
      assert rc == want, (kw, sel, rc, L.ep_last_error())
      assert b.num_rows == 3
    g.close()
  # the fused rollout checks once per launch
  n, T = 128, 3
  g, _ = _pair(num_envs=n, seed=9, ring_depth=8)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); g._recv()
  dup = ids.copy(); dup[77] = 5
  t_ids = torch.from_numpy(dup).to(dev)
  t_la = torch.ones((n, 6), dtype=torch.float64, device=dev)
  torch.cuda.synchronize()
  ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_ids.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
  capi._check(L.ep_pool_rollout(g.handle, T, 5, ptrs, (C.c_int64 * 5)(n, n, n, n, n), (C.c_int64 * 5)(0, 0, 0, 0, 0)))
  g._recv(allow=(capi.EP_ERR_DUP_ENV,))
  assert g.last_status == capi.EP_ERR_DUP_ENV
  for _ in range(T - 1):
    g._recv()


def test_recv_blocks_until_another_thread_sends():
  """The async actor/learner pattern: the consumer sits in recv (StateBuffer::Wait,
  state_buffer.h:135-140) while another thread's sends fill the batch. recv must neither raise
  nor keep the sender out."""
  g, o = _pair(num_envs=64, batch_size=48, seed=5)
  ids = np.arange(64, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), "first")   # 48 of the 64 reset rows; 16 stay queued
  got = []

  def consumer():
    got.append(g._recv())          # 16 rows there, 32 more needed: blocks

  th = threading.Thread(target=consumer)
  th.start()
  time.sleep(0.3)
  assert th.is_alive(), "recv returned before its batch was complete"
  for part in (ids[:10], ids[10:20], ids[20:40]):   # 40 more rows: the batch completes during the third send
    a = dummy_action(part)
    g._send(a); o._send(a)
  th.join(timeout=20)
  assert not th.is_alive(), "recv did not wake up"
  assert_batches_equal(got[0], o._recv(), "blocking recv")
  # a bounded wait gives up with EP_ERR_WOULD_BLOCK
  from envpool_amd import capi
  g.set_recv_timeout(50)
  t0 = time.time()
  with pytest.raises(capi.EpError) as e:
    g._recv()
  assert e.value.code == capi.EP_ERR_WOULD_BLOCK and time.time() - t0 >= 0.04


def test_concurrent_senders_one_consumer():
  """Threading contract (action_buffer_queue.h:59-69, state_buffer_queue.h:131-138): several
  threads send disjoint env blocks of an async pool at once, one consumer receives. The arrival
  order is up to the threads, so the result is compared as sets keyed by info:env_id."""
  n, B, nthreads, rounds = 4096, 512, 4, 6
  # ring deep enough for everything the senders push before the consumer catches up (the reference
  # throws out_of_range from StateBuffer::Allocate when its queue overflows, state_buffer.h:111)
  g, o = _pair(num_envs=n, batch_size=B, seed=7, ring_depth=rounds * n // B + 8)
  g.set_recv_timeout(20000)   # a failure must end the test, not hang it
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)

  def drain(pool, k):
    rows = {}
    for _ in range(k):
      r = pool._recv()
      for j, e in enumerate(np.asarray(r[0]).tolist()):
        rows.setdefault(e, []).append(tuple(np.asarray(r[key])[j].tobytes() for key in (1, 2, 3, 4, 5, 6, 7, 8, 10, 11)))
    return rows

  want = drain(o, n // B)
  got = drain(g, n // B)
  assert got == want
  blocks = np.array_split(ids, nthreads)
  errors = []

  def sender(block):
    try:
      for _ in range(rounds):
        for part in np.array_split(block, 4):
          g._send(dummy_action(part))
    except Exception as ex:  # noqa: BLE001
      errors.append(ex)

  ths = [threading.Thread(target=sender, args=(b,)) for b in blocks]
  res = {}
  cons = threading.Thread(target=lambda: res.update(drain(g, rounds * n // B)))
  cons.start()
  for t in ths:
    t.start()
  for t in ths:
    t.join(timeout=60)
  cons.join(timeout=60)
  assert not errors, errors
  assert not cons.is_alive()
  want = {}
  for _ in range(rounds):   # (the oracle's queue has the reference's depth: drain it round by round)
    for block in blocks:
      for part in np.array_split(block, 4):
        o._send(dummy_action(part))
    for e, rows in drain(o, n // B).items():
      want.setdefault(e, []).extend(rows)
  assert res == want   # per env: the same rows in the same order (each env has ONE sender thread)


def test_rollout_with_an_8_byte_aligned_view():
  """list_action handed over as a view that is only 8-byte aligned: the LDS-DMA kernel needs 16,
  so the generic rollout kernel runs; results are the same."""
  import torch

  from envpool_amd import capi
  L = capi.lib()
  dev = torch.device("cuda", 0)
  n, T = 256, 6
  g, o = _pair(num_envs=n, seed=4, ring_depth=8)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), "reset")
  rng = np.random.default_rng(0)
  la = np.repeat(rng.integers(0, 5, size=(T, n, 1)).astype(np.float64), 6, axis=2)
  flat = torch.zeros(T * n * 6 + 1, dtype=torch.float64, device=dev)
  view = flat[1:]
  view.copy_(torch.from_numpy(la.reshape(-1)).to(dev))
  assert view.data_ptr() % 16 == 8
  t_ids = torch.from_numpy(ids).to(dev)
  torch.cuda.synchronize()
  ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_ids.data_ptr(), view.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
  capi._check(L.ep_pool_rollout(g.handle, T, 5, ptrs, (C.c_int64 * 5)(n, n, n, n, n), (C.c_int64 * 5)(0, 0, n, 0, 0)))
  for t in range(T):
    o._send([ids, ids, la[t], ids * 0, ids * 0])
    assert_batches_equal(g._recv(), o._recv(), f"t={t}")
  # one-launch-per-step sends accept the same view
  capi._check(L.ep_pool_send(g.handle, 5, ptrs, (C.c_int64 * 5)(n, n, n, n, n), capi.EP_MEM_DEVICE))
  o._send([ids, ids, la[0], ids * 0, ids * 0])
  assert_batches_equal(g._recv(), o._recv(), "send")
  # NULL arrays are refused, not dereferenced
  bad = (C.c_void_p * 5)(t_ids.data_ptr(), None, view.data_ptr(), None, None)
  assert L.ep_pool_send(g.handle, 5, bad, (C.c_int64 * 5)(n, n, n, n, n), capi.EP_MEM_DEVICE) == capi.EP_ERR_INVALID
  assert L.ep_pool_rollout(g.handle, 2, 5, bad, (C.c_int64 * 5)(n, n, n, n, n), (C.c_int64 * 5)(0, 0, 0, 0, 0)) == capi.EP_ERR_INVALID


def test_stream_ordered_error_words_survive_slot_reuse():
  """A batch handed out by ep_pool_recv_on_stream keeps its error word until ep_pool_sync even when
  its ring slot has been refilled and received again in between (error words are per batch
  generation, not per slot)."""
  import torch

  from envpool_amd import capi
  L = capi.lib()
  dev = torch.device("cuda", 0)
  n = 512
  g, _ = _pair(num_envs=n, seed=100, ring_depth=2)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); g._recv()
  t_ids = torch.from_numpy(ids).to(dev)
  good = torch.ones((n, 6), dtype=torch.float64, device=dev)
  bad = good.clone(); bad[17, 3] = 2.0
  torch.cuda.synchronize()
  dims = (C.c_int64 * 5)(n, n, n, n, n)

  def send(la):
    ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_ids.data_ptr(), la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
    capi._check(L.ep_pool_send(g.handle, 5, ptrs, dims, capi.EP_MEM_DEVICE))

  b = capi.Batch()
  send(bad)
  capi._check(L.ep_pool_recv_on_stream(g.handle, None, C.byref(b)))   # the failing batch: not looked at yet
  for _ in range(5):   # its slot and its word's index both come round again
    send(good)
    assert L.ep_pool_recv(g.handle, capi.EP_MEM_DEVICE, C.byref(b)) == 0   # the later batches are clean
  assert L.ep_pool_sync(g.handle) == capi.EP_ERR_ACTION_CHECK
  assert L.ep_pool_sync(g.handle) == 0


def _views(b, dev):
  import torch  # noqa: F401

  from envpool_amd import capi, interop
  out = []
  for k in range(b.num_keys):
    a = b.arrays[k]
    if a.dtype == capi.EP_CONTAINER_I32:
      out.append(None)
      continue
    shape = tuple(a.shape[i] for i in range(a.ndim))
    out.append(interop.as_torch((a.data, shape, np.dtype(capi._NP[a.dtype]).str), device=dev).cpu().numpy())
  return out


@pytest.mark.parametrize("n,steps", [(65536, 160), (1024, 400), (768, 60), (1000, 20)])
def test_ep_pool_step_resident_server(n, steps):
  """ep_pool_step: a blocking step with no kernel launch (resident server kernel, doorbell in
  host-mapped memory). The action arrays are rewritten between steps by OTHER kernels (torch, on its
  own stream) - what the server's sc1 loads have to see - and every step's rows are compared with
  the oracle. In between: a launch-per-step send/recv, a fused rollout, a checkpoint, a change of
  action arrays and an idle pause long enough for the server to leave by itself.
  n = 1000 is not a multiple of 256: ep_pool_step takes the send + recv path there."""
  import torch

  from envpool_amd import capi
  L = capi.lib()
  dev = torch.device("cuda", 0)
  g, o = _pair(num_envs=n, seed=42 if n == 65536 else 3, ring_depth=4)
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  assert_batches_equal(g._recv(), o._recv(), "reset")
  rng = np.random.default_rng(n)
  t_ids = torch.from_numpy(ids).to(dev)
  bufs = [(torch.empty(n, dtype=torch.int32, device=dev), torch.empty((n, 6), dtype=torch.float64, device=dev)) for _ in range(2)]
  dims = (C.c_int64 * 5)(n, n, n, n, n)
  b = capi.Batch()
  which = 0
  for t in range(steps):
    if t == steps // 2:
      which = 1                      # other action arrays: the server is restarted for them
    t_penv, t_la = bufs[which]
    penv = ids + (rng.random(n) < 0.1).astype(np.int32)
    lane = rng.integers(0, 5, size=(n, 1)).astype(np.float64)
    # written on the device by torch kernels (H2D copy + elementwise kernels on torch's stream)
    t_penv.copy_(torch.from_numpy(penv).to(dev) + 0)
    t_la.copy_(torch.from_numpy(lane).to(dev).expand(n, 6) * 1.0)
    torch.cuda.synchronize()
    ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_penv.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
    rc = L.ep_pool_step(g.handle, 5, ptrs, dims, C.byref(b))
    assert rc == 0, L.ep_last_error()
    assert b.num_rows == n
    o._send([ids, penv, np.repeat(lane, 6, axis=1), ids * 0, ids * 0])
    assert_batches_equal(_views(b, dev), o._recv(), f"n={n} step {t}")
    if t == 7:       # the launch-per-step path in between (stops the server)
      a = dummy_action(rng.permutation(ids))
      g._send(a); o._send(a)
      assert_batches_equal(g._recv(), o._recv(), "send/recv in between")
    if t == 11:      # a checkpoint reads the env state the server wrote
      nb = L.ep_pool_checkpoint_bytes(g.handle)
      buf = C.create_string_buffer(nb)
      capi._check(L.ep_pool_checkpoint_save(g.handle, buf))
      rec = np.frombuffer(buf.raw[: 8 * n], np.int32).reshape(n, 2)
      want = o._recv()   # empty batch; the oracle's env state is checked through the next steps
      assert len(want[0]) == 0 and rec[:, 1].min() >= 0
    if t == 13:
      time.sleep(0.25)   # longer than the server's idle limit: it leaves, the next step starts a new one
  sr, sl, sc = g.episode_stats()
  assert (sc >= 0).all() and not sr.any()
  # a failed self-check is reported by the step it happened in
  t_penv, t_la = bufs[which]
  t_la.fill_(1.0)
  t_la[5, 2] = 2.0
  torch.cuda.synchronize()
  ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_penv.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
  rc = L.ep_pool_step(g.handle, 5, ptrs, dims, C.byref(b))
  assert rc in (capi.EP_ERR_ACTION_CHECK, 0)   # (0 when env 5 happened to be on its auto-reset step)
  g.close()


@pytest.mark.parametrize("host_first", [True, False])
def test_host_and_device_sends_into_one_batch(host_first):
  """Host mode gives a batch a pinned result block when its first rows come from host memory, and
  every later kernel that writes rows of that batch must write into the block too - a device-memory
  send included; a batch begun by a device-memory send stays in the ring and a host recv copies it
  out. Either way the batch equals the oracle's, and so does a device-view recv of a block-backed
  batch (the block is copied into the ring first)."""
  import torch

  from envpool_amd import capi
  L = capi.lib()
  dev = torch.device("cuda", 0)
  N, B = 64, 16
  g, o = _pair(num_envs=N, batch_size=B, seed=3, ring_depth=16)
  ids = np.arange(N, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  for it in range(12):
    got, want = g._recv(), o._recv()
    assert_batches_equal(got, want, f"it {it}")
    sel = np.asarray(want[0]).astype(np.int32)
    halves = [sel[:B // 2], sel[B // 2:]]
    for k, part in enumerate(halves):
      from_host = (k == 0) == host_first
      if from_host:
        g._send(dummy_action(part))
      else:
        t_ids = torch.from_numpy(part.copy()).to(dev)
        t_la = torch.full((len(part), 6), 1.5, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ptrs = (C.c_void_p * 5)(t_ids.data_ptr(), t_ids.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr())
        n = len(part)
        capi._check(L.ep_pool_send(g.handle, 5, ptrs, (C.c_int64 * 5)(n, n, n, n, n), capi.EP_MEM_DEVICE))
      o._send(dummy_action(part))
  # a block-backed batch wanted as device views: send from host, receive on the device
  for it in range(3):
    b = capi.Batch()
    capi._check(L.ep_pool_recv(g.handle, capi.EP_MEM_DEVICE, C.byref(b)))
    want = o._recv()
    assert b.num_rows == B
    hip_ids = np.zeros(B, np.int32)
    assert L.hipMemcpy(C.c_void_p(hip_ids.ctypes.data), C.c_void_p(b.arrays[0].data), C.c_size_t(4 * B), C.c_int(2)) == 0
    hip_el = np.zeros(B, np.int32)
    assert L.hipMemcpy(C.c_void_p(hip_el.ctypes.data), C.c_void_p(b.arrays[2].data), C.c_size_t(4 * B), C.c_int(2)) == 0
    raw = np.zeros((B, 10), np.int32)
    assert L.hipMemcpy(C.c_void_p(raw.ctypes.data), C.c_void_p(b.arrays[8].data), C.c_size_t(40 * B), C.c_int(2)) == 0
    assert hip_ids.tobytes() == want[0].tobytes() and hip_el.tobytes() == want[2].tobytes() and raw.tobytes() == want[8].tobytes()
    sel = np.asarray(want[0]).astype(np.int32)
    g._send(dummy_action(sel)); o._send(dummy_action(sel))
  g.close()


def test_refused_calls_leave_the_pool_as_it_was(monkeypatch):
  """tools/dummy_abi_fuzz.py, a few rounds: wrong array counts, null pointers, disagreeing / negative /
  oversized dims, unknown memory kinds, rollouts that do not fit - each refused, and the valid steps
  around them still equal the oracle's."""
  from tools import dummy_abi_fuzz
  monkeypatch.setenv("FUZZ_SEED", "4")
  dummy_abi_fuzz.main(rounds=6)
