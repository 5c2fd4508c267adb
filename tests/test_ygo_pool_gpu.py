"""ep_ygo_pool: AsyncEnvPool<YGOProEnv> over a duel engine behind callbacks
(envpool_amd/csrc/ygo_pool.hip; envpool2/core/async_envpool.h:57-189). The engine here is the
deterministic stand-in of tools/ygo_synth.c; every received row is compared bit for bit with the
oracle's encoding of the state that engine produced for that env at that point, with the pool's own
bookkeeping (auto-reset, elapsed_step, env_id) modelled by tools/ygo_synth.EnginePoolModel."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from envpool_amd import capi, capi_ygo  # noqa: E402

NK = 18  # 16 state keys + aux:h_card_ids + aux:action_mask


def _expect_rows(cfg, table, raws):
  from oracle import ygo_oracle
  arr = (capi_ygo.RawState * len(raws))()
  for i, r in enumerate(raws):
    C.memmove(C.byref(arr[i]), C.byref(r), C.sizeof(capi_ygo.RawState))
  want, status = ygo_oracle.write_batch(cfg, table, arr, len(raws))
  return want, status


def _row(out, j):
  return tuple(np.asarray(out[k][j]).tobytes() for k in range(NK))


@pytest.mark.parametrize("threads", [1, 5])
def test_sync_pool_rows_in_sent_order(threads):
  """batch_size == num_envs: row i of the batch is position i of the SENT env_id array
  (async_envpool.h:69, state_buffer.h:92-95), partial sends and forced resets included."""
  from tools import ygo_synth
  N = 96
  cfg = capi_ygo.default_config(max_batch=N, ring_depth=0)
  table = ygo_synth.make_table(1000, seed=0)
  eng = ygo_synth.SynthEngine(cfg, table, N, seed=5)
  model = ygo_synth.EnginePoolModel(cfg, table, N, seed=5)
  pool = capi_ygo.YgoPoolC(cfg, eng.engine(), N, N, table, len(table), num_threads=threads)
  assert pool.num_threads == threads
  pool.set_recv_timeout(30000)
  rng = np.random.default_rng(threads)
  ids = np.arange(N, dtype=np.int32)
  checked = 0
  for it in range(14):
    sel = ids if it % 4 == 0 else rng.permutation(N)[: int(rng.integers(1, N + 1))].astype(np.int32)
    if it in (0, 6):
      pool.reset(sel)
      raws = [model.advance(int(e)) for e in sel]
    else:
      act = rng.integers(0, 16, size=sel.size).astype(np.int32)
      pool.send(sel, act)
      raws = [model.advance(int(e), int(a)) for e, a in zip(sel, act)]
    out = pool.recv(allow=(capi.EP_ERR_ACTION_CHECK,))
    assert out[0].tolist() == sel.tolist() and pool.last_env_ids().tolist() == sel.tolist()
    want, status = _expect_rows(cfg, table, raws)
    for j in range(sel.size):
      if status[j] == 0:
        assert _row(out, j) == _row(want, j), (it, j, int(sel[j]))
        checked += 1
  assert checked > 300
  # a sync pool takes one send per recv; an env may appear once per send
  pool.send(ids[:4], np.zeros(4, np.int32))
  with pytest.raises(capi.EpError) as e:
    pool.send(ids[4:8], np.zeros(4, np.int32))
  assert e.value.code == capi.EP_ERR_INVALID
  pool.recv(allow=(capi.EP_ERR_ACTION_CHECK,))
  with pytest.raises(capi.EpError) as e:
    pool.send(np.array([3, 5, 3], np.int32), np.zeros(3, np.int32))
  assert e.value.code == capi.EP_ERR_DUP_ENV
  pool.set_recv_timeout(0)
  with pytest.raises(capi.EpError) as e:
    pool.recv()
  assert e.value.code == capi.EP_ERR_WOULD_BLOCK
  pool.close(); eng.close()


def test_sync_pool_config3_shape():
  """BASELINE config 3 as a POOL at its own size: 1024 envs, sync, 4 workers. Resets, full sends
  and partial sends sized so that batches go to the encoder in 1, 2, 3 and 4 parts
  (ygo_pool.hip: parts of at least 128 rows, at most 4: ep_ygo_commit_slot_rows); every row against
  the oracle's encoding of the engine model's state (ygopro.h:1828-1893), rows in SENT order
  (async_envpool.h:57-80, state_buffer.h:92-95)."""
  from tools import ygo_synth
  N = 1024
  cfg = capi_ygo.default_config(max_batch=N, ring_depth=0)
  table = ygo_synth.make_table(1000, seed=0)
  eng = ygo_synth.SynthEngine(cfg, table, N, seed=11)
  model = ygo_synth.EnginePoolModel(cfg, table, N, seed=11)
  pool = capi_ygo.YgoPoolC(cfg, eng.engine(), N, N, table, len(table), num_threads=4)
  assert pool.num_threads == 4
  pool.set_recv_timeout(60000)
  rng = np.random.default_rng(33)
  ids = np.arange(N, dtype=np.int32)
  # (rows, forced reset): 1024 / 700 / 520 -> 4 parts, 400 -> 3, 300 -> 2, 130 / 17 -> 1
  plan = [(N, True), (N, False), (300, False), (N, False), (130, False), (700, False), (400, True), (520, False),
          (17, False), (N, False)]
  checked = 0
  for it, (n, forced) in enumerate(plan):
    sel = ids if n == N and it % 2 == 0 else rng.permutation(N)[:n].astype(np.int32)
    if forced:
      pool.reset(sel)
      raws = [model.advance(int(e)) for e in sel]
    else:
      act = rng.integers(0, 16, size=sel.size).astype(np.int32)
      pool.send(sel, act)
      raws = [model.advance(int(e), int(a)) for e, a in zip(sel, act)]
    out = pool.recv(allow=(capi.EP_ERR_ACTION_CHECK,))
    assert out[0].shape[0] == n and out[0].tolist() == sel.tolist() and pool.last_env_ids().tolist() == sel.tolist()
    want, status = _expect_rows(cfg, table, raws)
    for k in range(NK):
      assert np.asarray(out[k]).shape == np.asarray(want[k]).shape, (it, k)
    good = np.flatnonzero(np.asarray(status) == 0)
    for k in range(NK):
      a, b = np.asarray(out[k])[good], np.asarray(want[k])[good]
      if a.tobytes() != b.tobytes():
        bad = [int(j) for j in good if np.asarray(out[k][j]).tobytes() != np.asarray(want[k][j]).tobytes()]
        raise AssertionError(f"send {it} ({n} rows): key {k} differs in rows {bad[:8]} (envs {[int(sel[j]) for j in bad[:8]]})")
    checked += good.size
  assert checked > 0.9 * sum(n for n, _ in plan)
  pool.close(); eng.close()


@pytest.mark.parametrize("threads", [1, 6])
def test_async_pool_config5_shape(threads):
  """BASELINE config 5 per GPU (1024 envs, batches of 512) as a POOL: reset all, then keep sending
  back whatever arrives. Rows come in arrival order: compared per env, in the order each env's
  rows arrive (an env has at most one step in flight)."""
  from tools import ygo_synth
  N, B = 1024, 512
  cfg = capi_ygo.default_config(max_batch=B, ring_depth=0)
  table = ygo_synth.make_table(1000, seed=0)
  eng = ygo_synth.SynthEngine(cfg, table, N, seed=9, engine_us=1.0)
  model = ygo_synth.EnginePoolModel(cfg, table, N, seed=9)
  pool = capi_ygo.YgoPoolC(cfg, eng.engine(), N, B, table, len(table), num_threads=threads)
  pool.set_recv_timeout(30000)
  expect = {}
  ids = np.arange(N, dtype=np.int32)
  pool.reset(ids)
  for e in ids:
    expect[int(e)] = model.advance(int(e))
  rng = np.random.default_rng(3)
  checked = 0
  for it in range(10):
    out = pool.recv(allow=(capi.EP_ERR_ACTION_CHECK,))
    got_ids = out[0]
    assert got_ids.shape[0] == B and len(set(got_ids.tolist())) == B
    assert pool.last_env_ids().tolist() == got_ids.tolist()   # the host-side note of who arrived, row by row
    raws = [expect.pop(int(e)) for e in got_ids]
    want, status = _expect_rows(cfg, table, raws)
    for j in range(B):
      if status[j] == 0:
        assert _row(out, j) == _row(want, j), (it, j, int(got_ids[j]))
        checked += 1
    act = rng.integers(0, 16, size=B).astype(np.int32)
    pool.send(got_ids, act)
    for e, a in zip(got_ids, act):
      expect[int(e)] = model.advance(int(e), int(a))
  assert checked > 0.9 * 10 * B
  pool.close(); eng.close()


def test_env_id_offset_shard():
  """One pool per GPU owning a block of global env ids (config 5: 8192 envs = 8 x 1024): send takes
  global ids, info:env_id is global, ids of other shards are refused."""
  from tools import ygo_synth
  N, off = 48, 5 * 1024
  cfg = capi_ygo.default_config(max_batch=N, ring_depth=0)
  table = ygo_synth.make_table(1000, seed=0)
  eng = ygo_synth.SynthEngine(cfg, table, N, seed=6)
  eng.set_env_id_offset(off)
  model = ygo_synth.EnginePoolModel(cfg, table, N, seed=6)
  pool = capi_ygo.YgoPoolC(cfg, eng.engine(), N, N, table, len(table), num_threads=3)
  pool.set_env_id_offset(off)
  pool.set_recv_timeout(30000)
  ids = np.arange(off, off + N, dtype=np.int32)
  with pytest.raises(capi.EpError):
    pool.reset(np.arange(N, dtype=np.int32))
  pool.reset(ids)
  raws = [model.advance(int(e) - off) for e in ids]
  for r, e in zip(raws, ids):
    r.env_id = int(e)
  out = pool.recv(allow=(capi.EP_ERR_ACTION_CHECK,))
  assert out[0].tolist() == ids.tolist() and out[1].tolist() == ids.tolist()
  want, status = _expect_rows(cfg, table, raws)
  assert sum(_row(out, j) == _row(want, j) for j in range(N) if status[j] == 0) == int((status == 0).sum()) > N // 2
  with pytest.raises(capi.EpError):
    pool.set_env_id_offset(0)   # too late
  pool.close(); eng.close()


def test_destroy_with_a_full_ring_and_nobody_receiving():
  """Workers asleep on a full ring (the consumer went away) must not keep the pool from closing."""
  import threading
  from tools import ygo_synth
  N, B = 64, 16
  cfg = capi_ygo.default_config(max_batch=B, ring_depth=2)
  table = ygo_synth.make_table(1000, seed=0)
  eng = ygo_synth.SynthEngine(cfg, table, N, seed=4)
  pool = capi_ygo.YgoPoolC(cfg, eng.engine(), N, B, table, len(table), num_threads=3)
  pool.reset(np.arange(N, dtype=np.int32))   # 4 batches into a ring of 2: the workers end up waiting
  import time
  time.sleep(0.3)
  t = threading.Thread(target=pool.close)
  t.start()
  t.join(timeout=20)
  assert not t.is_alive()
  eng.close()


def test_engine_failure_is_reported_beside_the_batch():
  from tools import ygo_synth
  N = 8
  cfg = capi_ygo.default_config(max_batch=N, ring_depth=0)
  table = ygo_synth.make_table(1000, seed=0)
  eng = ygo_synth.SynthEngine(cfg, table, N, seed=1)
  pool = capi_ygo.YgoPoolC(cfg, eng.engine(), N, N, table, len(table), num_threads=2)
  pool.set_recv_timeout(30000)
  ids = np.arange(N, dtype=np.int32)
  eng.set_fail_env(3)
  pool.reset(ids)
  out = pool.recv(allow=(capi.EP_ERR_INVALID,))
  assert pool.last_status == capi.EP_ERR_INVALID
  assert out[0].tolist() == ids.tolist()
  assert bool(out[3][3]) and int(out[12][3]) == 1 and not out[8][3].any()   # done, num_options = 1, empty obs
  eng.set_fail_env(-1)
  pool.send(ids, np.zeros(N, np.int32))
  pool.recv(allow=(capi.EP_ERR_ACTION_CHECK,))
  pool.close(); eng.close()


def test_make_ygopro_with_init_module():
  """`make("YGOPro-v0")` after `init_module` (the reference's module-level init, ygopro.h:1030-1064):
  the gym-flavoured front-end over the raw `_send / _recv / _reset` boundary."""
  import envpool_amd
  from envpool_amd import ygopro
  from tools import ygo_synth
  N = 32
  cfg = capi_ygo.default_config(max_batch=N, ring_depth=0)
  table = ygo_synth.make_table(1000, seed=0)
  eng = ygo_synth.SynthEngine(cfg, table, N, seed=2)
  ygopro.init_module(eng.factory, table)
  try:
    env = envpool_amd.make("YGOPro-v0", "raw", num_envs=N, num_threads=3)
    obs, info = env.reset()
    assert obs["cards_"].shape == (N, 150, 39) and obs["actions_"].shape == (N, 16, 19)
    assert info["env_id"].tolist() == list(range(N)) and (info["elapsed_step"] == 0).all()
    obs, rew, term, trunc, info = env.step(np.ones(N, dtype=np.int32))
    assert (info["elapsed_step"] == 1).all() and rew.shape == (N,) and (rew == 1.0).all()
    assert (info["win_reason"] == 1).all()   # the stand-in engine echoes the action there
    env.send(np.full(5, 2, np.int32), np.array([9, 1, 4, 30, 2], np.int32))
    obs, rew, term, trunc, info = env.recv()
    assert info["env_id"].tolist() == [9, 1, 4, 30, 2] and (info["win_reason"] == 2).all()
    # observations for a policy on the GPU: torch views of the encoder's output, actions back as numpy
    env.send(np.full(N, 3, np.int32))
    obs_t, rew_t, term_t, trunc_t, info_t = env.recv_torch()
    assert obs_t["cards_"].is_cuda and tuple(obs_t["cards_"].shape) == (N, 150, 39)
    assert env.last_env_ids.tolist() == info_t["env_id"].cpu().tolist()
    assert (info_t["win_reason"] == 3).all().item() and info_t["env_id"].cpu().tolist() == list(range(N))
    m = info_t["action_mask"]
    assert m.is_cuda and tuple(m.shape) == (N, 16)
    assert (m.sum(1).cpu() == info_t["num_options"].cpu().clamp(max=16)).all()
    import torch
    obs, rew, term, trunc, info = env.step(torch.zeros(N, dtype=torch.int32, device="cuda"), env_id=info_t["env_id"])
    assert info["action_mask"].shape == (N, 16) and (info["action_mask"].sum(1) == np.minimum(info["num_options"], 16)).all()
    env.close()
  finally:
    ygopro.init_module(None, None)
    eng.close()
