"""CPU: pins oracle/dummy_oracle.c against the reference-produced golden traces and the closed
form recorded in SURVEY.md §8a, and checks the queue semantics restated from
state_buffer(_queue).h / async_envpool.h."""
import json
import os

import numpy as np
import pytest

from oracle.dummy_oracle import DummyOracle, OracleError, ThreadedDummyOracle
from tests.util import dummy_action

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dummy_survey_traces.json")))


def test_golden_sync_trace():
  g = GOLD["sync_seed3_n8"]
  o = DummyOracle(num_envs=8, seed=3)
  ids = np.arange(8, dtype=np.int32)
  o._reset(ids)
  o._recv()
  for t in range(1, 13):
    o._send(dummy_action(ids, value=1.5))
    r = o._recv()
    if str(t) in g["elapsed_step"]:
      assert r[2].tolist() == g["elapsed_step"][str(t)], t
      assert r[8][:, 0].tolist() == g["elapsed_step"][str(t)], t  # obs:raw[:,0] == state_ == elapsed here
    if str(t) in g["done"]:
      assert r[3].astype(int).tolist() == g["done"][str(t)], t


def test_golden_multiplayer_trace():
  g = GOLD["multiplayer_seed2_p3_n4"]
  o = DummyOracle(num_envs=4, seed=2, max_num_players=3)
  ids = np.arange(4, dtype=np.int32)
  o._reset(ids)
  r = o._recv()
  for t in range(1, 4):
    o._send(dummy_action(ids, players_env_id=r[1]))
    r = o._recv()
    if t == 1:
      assert r[1].tolist() == g["t1"]["info:players.env_id"]
      assert r[11].tolist() == g["t1"]["info:players.id"]
      assert r[4].tolist() == g["t1"]["reward"]
      assert r[8][:, 1].tolist() == g["t1"]["obs:raw_col1"]
    if t == 3:
      assert r[1].tolist() == g["t3"]["info:players.env_id"]


@pytest.mark.parametrize("seed,n,max_steps", [(42, 64, 2**31 - 1), (1, 16, 3), (-5, 8, 2**31 - 1), (0, 5, 1)])
def test_closed_form(seed, n, max_steps):
  """SURVEY §8a closed form: L = max(1, seed+env_id), p = t mod (L+1)."""
  o = DummyOracle(num_envs=n, seed=seed, max_episode_steps=max_steps)
  ids = np.arange(n, dtype=np.int32)
  L = np.maximum(1, seed + ids.astype(np.int64))
  o._reset(ids)
  r = o._recv()
  assert not r[2].any() and not r[3].any() and (r[6] == 0).all()
  for t in range(1, 200):
    o._send(dummy_action(ids))
    r = o._recv()
    p = t % (L + 1)
    done = p == L
    assert np.array_equal(r[0], ids) and np.array_equal(r[1], ids)
    assert np.array_equal(r[2], p) and np.array_equal(r[8][:, 0], p)
    assert np.array_equal(r[3], done) and np.array_equal(r[10], done)
    assert np.array_equal(r[6], np.where(p == 0, 0, np.where(done, 2, 1)))
    assert np.array_equal(r[5], (~done).astype(np.float32))
    assert np.array_equal(r[7], done & (p >= max_steps))
    assert np.array_equal(r[8][:, 1], (p != 0).astype(np.int32))
    assert not r[8][:, 2:].any()
    assert r[4].tobytes() == np.zeros(n, np.float32).tobytes()  # +0.0f, not -0.0f
    assert not r[11].any()


def test_players_env_id_mismatch_zeroes_action_num():
  o = DummyOracle(num_envs=4, seed=10)
  ids = np.arange(4, dtype=np.int32)
  o._reset(ids); o._recv()
  o._send(dummy_action(ids, players_env_id=[0, 9, 2, 9]))
  assert o._recv()[8][:, 1].tolist() == [1, 0, 1, 0]


def test_partial_sync_send_truncates_and_realigns():
  o = DummyOracle(num_envs=8, seed=5)
  ids = np.arange(8, dtype=np.int32)
  o._reset(ids); o._recv()
  o._send(dummy_action(ids[[5, 1, 6]]))
  r = o._recv()
  assert r[0].tolist() == [5, 1, 6] and r[8].shape == (3, 10)
  o._send(dummy_action(ids))  # next batch starts in a fresh buffer
  assert o._recv()[0].tolist() == list(range(8))


def test_sync_two_sends_one_recv_quirk():
  """Two partial sync sends before one recv land on the SAME rows (order = index in the sent
  batch, state_buffer.h:92-95); rows beyond stay zero. Deterministic under FIFO execution."""
  o = DummyOracle(num_envs=8, seed=5)
  ids = np.arange(8, dtype=np.int32)
  o._reset(ids); o._recv()
  o._send(dummy_action(ids[:3])); o._send(dummy_action(ids[3:6]))
  r = o._recv()
  assert r[0].tolist() == [3, 4, 5, 0, 0, 0]
  assert not r[8][3:].any()


def test_async_batches_fifo_and_would_block():
  o = DummyOracle(num_envs=8, batch_size=4, seed=3)
  assert not o.is_sync and o.queue_size == (8 // 4 + 2) * 2
  ids = np.arange(8, dtype=np.int32)
  o._reset(ids)
  assert o._recv()[0].tolist() == [0, 1, 2, 3]
  assert o._recv()[0].tolist() == [4, 5, 6, 7]
  with pytest.raises(OracleError) as e:
    o._recv()
  assert e.value.code == 4
  o._send(dummy_action(ids[[7, 2]]))
  with pytest.raises(OracleError):
    o._recv()
  o._send(dummy_action(ids[[0, 5]]))
  assert o._recv()[0].tolist() == [7, 2, 0, 5]


def test_ring_full():
  o = DummyOracle(num_envs=2, seed=3)  # queue of (2/2+2)*2 = 6 buffers
  ids = np.arange(2, dtype=np.int32)
  for _ in range(6):
    assert o._send(dummy_action(ids)) == 0
  assert o._send(dummy_action(ids)) == 5


def test_sharded_oracles_equal_one_pool():
  """env_id_offset: two 4-env shards reproduce rows of one 8-env pool (seed_ uses the GLOBAL id)."""
  whole = DummyOracle(num_envs=8, seed=2)
  parts = [DummyOracle(num_envs=4, seed=2, env_id_offset=4 * r) for r in range(2)]
  ids = np.arange(8, dtype=np.int32)
  whole._reset(ids); [p._reset(ids[4 * r:4 * r + 4]) for r, p in enumerate(parts)]
  for t in range(30):
    w = whole._recv()
    ps = [p._recv() for p in parts]
    for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11):
      assert np.array_equal(w[k], np.concatenate([p[k] for p in ps])), (t, k)
    whole._send(dummy_action(ids)); [p._send(dummy_action(ids[4 * r:4 * r + 4])) for r, p in enumerate(parts)]


def _same(ra, rb):
  for k, (x, y) in enumerate(zip(ra, rb)):
    if isinstance(x, tuple):
      assert all(np.array_equal(u, v) for u, v in zip(x, y)), k
    else:
      assert x.dtype == y.dtype and np.array_equal(x, y), k


@pytest.mark.parametrize("threads", [1, 4])
def test_threaded_architecture_equals_sequential_oracle_sync(threads):
  """oracle/dummy_pool_mt.c (action queue -> worker threads -> state-buffer queue, the shape of
  async_envpool.h:116-129) against the sequential restatement: in sync mode `order` fixes every
  row, so any thread interleaving must give the same batches, partial sends and forced resets
  included."""
  n = 600
  a = DummyOracle(num_envs=n, seed=3)
  b = ThreadedDummyOracle(num_envs=n, seed=3, num_threads=threads)
  ids = np.arange(n, dtype=np.int32)
  a._reset(ids); b._reset(ids)
  _same(a._recv(), b._recv())
  rng = np.random.default_rng(threads)
  for t in range(60):
    perm = rng.permutation(n).astype(np.int32)[: n if t % 3 else n // 2]
    if t % 7 == 6:
      a._reset(perm); b._reset(perm)
    else:
      act = dummy_action(perm)
      a._send(act); b._send(act)
    _same(a._recv(), b._recv())
  assert not b.action_check_failed
  b.close()


def test_threaded_architecture_async_rows_by_env():
  """Async mode: rows arrive in whatever order the workers finish; per env the sequence of rows
  must be the sequential oracle's."""
  n, bsz = 64, 16
  a = DummyOracle(num_envs=n, batch_size=bsz, seed=2)
  b = ThreadedDummyOracle(num_envs=n, batch_size=bsz, seed=2, num_threads=4)
  ids = np.arange(n, dtype=np.int32)
  a._reset(ids); b._reset(ids)
  seq_a, seq_b = {i: [] for i in range(n)}, {i: [] for i in range(n)}
  pend_a, pend_b = n, n

  def take(pool, seq):
    r = pool._recv()
    for j, e in enumerate(r[0]):
      seq[int(e)].append((int(r[2][j]), bool(r[3][j]), int(r[6][j]), int(r[8][j, 0]), int(r[8][j, 1])))
    return r[0]

  for _ in range(40):
    ea = take(a, seq_a); eb = take(b, seq_b)
    assert len(ea) == bsz and len(eb) == bsz
    a._send(dummy_action(ea)); b._send(dummy_action(eb))
  for _ in range(n // bsz):  # drain what is still in flight
    take(a, seq_a); take(b, seq_b)
  for i in range(n):
    m = min(len(seq_a[i]), len(seq_b[i]))
    assert m > 0 and seq_a[i][:m] == seq_b[i][:m], i
  b.close()


def test_threaded_architecture_multiplayer_one_worker_is_fifo():
  """One worker = strictly FIFO = the sequential oracle, ragged player rows included."""
  n = 9
  a = DummyOracle(num_envs=n, max_num_players=3, seed=2)
  b = ThreadedDummyOracle(num_envs=n, max_num_players=3, seed=2, num_threads=1)
  ids = np.arange(n, dtype=np.int32)
  a._reset(ids); b._reset(ids)
  ra, rb = a._recv(), b._recv()
  _same(ra, rb)
  for _ in range(12):
    act = dummy_action(ra[0], players_env_id=ra[1])
    a._send(act); b._send(act)
    ra, rb = a._recv(), b._recv()
    _same(ra, rb)
  b.close()
