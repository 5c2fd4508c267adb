"""CPU: the C-ABI library loads and exports every symbol include/*.h declares; the pybind spec
surface matches the reference's; host front-end logic; no-GPU failure
is loud; the N>1 sharding path under gloo with world_size 2."""
import ctypes
import glob
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "dummy_survey_traces.json")))["spec"]


def _declared_symbols():
  names = []
  for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
    text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
    names += re.findall(r"\b(ep_[a-z0-9_]+)\s*\(", text)
  return sorted(set(names))


def test_library_exports_every_declared_symbol():
  from envpool_amd._loader import load_library
  lib = load_library()
  syms = _declared_symbols()
  assert len(syms) >= 20
  for s in syms:
    assert hasattr(lib, s), f"libenvpool_amd.so does not export {s}"


def test_capi_binding_matches_header():
  from envpool_amd import capi
  declared = set(_declared_symbols())
  assert set(capi.SYMBOLS) <= declared
  L = capi.lib()
  assert b"gfx950" in L.ep_version()


def test_no_gpu_fails_loudly():
  from envpool_amd import capi
  L = capi.lib()
  if L.ep_device_count() > 0:
    pytest.skip("a GPU is visible")
  with pytest.raises(capi.EpError) as e:
    capi.DummyPoolC(num_envs=8)
  assert e.value.code == capi.EP_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_invalid_configs_rejected_before_touching_a_device():
  from envpool_amd import capi
  with pytest.raises(capi.EpError) as e:
    capi.DummyPoolC(num_envs=4, batch_size=8)
  assert e.value.code == capi.EP_ERR_INVALID and "batch_size <= num_envs" in str(e.value)
  with pytest.raises(capi.EpError):
    capi.DummyPoolC(num_envs=0)


def test_spec_surface_matches_reference():
  from envpool_amd.dummy.dummy_envpool import _DummyEnvPool, _DummyEnvSpec
  assert type(_DummyEnvSpec).__name__ == "ABCMeta" and type(_DummyEnvPool).__name__ == "ABCMeta"
  assert list(_DummyEnvSpec._config_keys)[:11] == GOLD["config_keys"]
  assert list(_DummyEnvSpec._default_config_values)[:11] == GOLD["default_config_values"]
  assert list(_DummyEnvSpec._state_keys) == GOLD["state_keys"] == list(_DummyEnvPool._state_keys)
  assert list(_DummyEnvSpec._action_keys) == GOLD["action_keys"] == list(_DummyEnvPool._action_keys)
  s = _DummyEnvSpec((8, 0, 0, 1, -1, "envpool2", 42, False, 2**31 - 1, 10, 6))  # the reference's 11-tuple
  assert s._config_values[1] == 8  # batch_size 0 -> num_envs (env_spec.h:80-82)
  st = dict(zip(s._state_keys, s._state_spec))
  i32, f32 = np.dtype(np.int32), np.dtype(np.float32)
  assert st["info:env_id"] == (i32, [], (-2**31, 2**31 - 1), ([], []))
  assert st["reward"] == (f32, [-1], (GOLD["float_min"], float(np.finfo(np.float32).max)), ([], []))
  assert st["discount"][2] == (0.0, 1.0)
  assert st["done"] == (np.dtype(bool), [], (False, True), ([], []))
  assert st["obs:raw"][:2] == (i32, [-1, 10])
  assert st["obs:dyn"][:2] == (i32, ([-1], [-1, 10]))
  assert st["info:players.id"][2] == (0, 1)
  ac = dict(zip(s._action_keys, s._action_spec))
  assert ac["list_action"] == (np.dtype(np.float64), [6], (GOLD["double_min"], float(np.finfo(np.float64).max)), ([], []))
  with pytest.raises(ValueError, match="batch_size <= num_envs"):
    _DummyEnvSpec((4, 8, 0, 1, -1, "x", 42, False, 100, 10, 6))


def test_make_spec_and_registry_checks():
  import envpool_amd
  assert "Dummy-v0" in envpool_amd.list_all_envs()
  spec = envpool_amd.make_spec("Dummy-v0", num_envs=16, batch_size=4, seed=7, state_num=12)
  assert spec.config.num_envs == 16 and spec.config.batch_size == 4 and spec.config.state_num == 12
  assert spec.state_array_spec["obs:raw"].shape == [-1, 12]
  assert spec.action_array_spec["list_action"].dtype == np.float64
  with pytest.raises(AssertionError):
    envpool_amd.make_spec("Dummy-v0", num_envs=4, batch_size=5)
  with pytest.raises(AssertionError):
    envpool_amd.make_spec("Dummy-v0", num_envs=4, seed=2**31)
  with pytest.raises(ValueError, match="gym_reset_return_info"):
    envpool_amd.make("Dummy-v0", "gym", num_envs=2, gym_reset_return_info=False)


class _FakeRaw:
  """Stands in for the pybind pool: records what the front-end hands to _send/_reset."""
  _state_keys = GOLD["state_keys"]
  _action_keys = GOLD["action_keys"]

  def __init__(self, spec):
    self._spec = spec
    self.sent, self.reset_ids = [], []

  def _send(self, arrays):
    self.sent.append(arrays)

  def _reset(self, ids):
    self.reset_ids.append(ids)

  def _recv(self):
    n = 3
    z = np.zeros(n, np.int32)
    return [np.arange(n, dtype=np.int32), z, z + 5, np.array([True, True, False]), np.zeros(n, np.float32),
            np.ones(n, np.float32), z, np.array([True, False, False]), np.zeros((n, 10), np.int32), None,
            np.zeros(n, bool), z]


def _front(kind):
  import envpool_amd
  from envpool_amd.python import pool as P
  spec = envpool_amd.make_spec("Dummy-v0", num_envs=3, gym_reset_return_info=True)
  front = {"raw": P.EnvPoolFrontEnd, "gym": P.GymFrontEnd, "gymnasium": P.GymnasiumFrontEnd, "dm": P.DMFrontEnd}[kind]
  return type("T", (front, _FakeRaw), {})(spec)


def test_front_end_action_marshalling():
  env = _front("raw")
  la = np.full((3, 6), 2.0)
  env.send({"list_action": la, "players": {"action": np.zeros(3, np.int32), "id": np.zeros(3, np.int32)}})
  a = env.sent[-1]
  assert [x.dtype for x in a] == [np.int32, np.int32, np.float64, np.int32, np.int32]
  assert a[0].tolist() == [0, 1, 2] and a[1] is a[0]  # env_id defaults to all ids, players.env_id to env_id
  env.send({"list_action": la[:2], "players.action": np.zeros(2, np.int32), "players.id": np.zeros(2, np.int32)},
           env_id=np.array([2, 0]))
  assert env.sent[-1][0].tolist() == [2, 0] and env.sent[-1][0].dtype == np.int32
  env2 = _front("raw")
  with pytest.raises(RuntimeError, match="Expected dtype"):
    env2.send({"list_action": la.astype(np.float32), "players.action": np.zeros(3, np.int32), "players.id": np.zeros(3, np.int32)})
  assert env.config["num_envs"] == 3 and len(env) == 3 and not env.is_async


@pytest.mark.parametrize("kind", ["raw", "gym", "gymnasium"])
def test_front_end_step_output(kind):
  env = _front(kind)
  obs, rew, term, trunc, info = env.step({"list_action": np.zeros((3, 6)), "players.action": np.zeros(3, np.int32),
                                          "players.id": np.zeros(3, np.int32)})
  assert set(obs) == {"raw", "dyn"} and obs["raw"].shape == (3, 10)
  assert term.tolist() == [False, True, False] and trunc.tolist() == [True, False, False]  # done & ~trunc
  assert info["env_id"].tolist() == [0, 1, 2] and set(info["players"]) == {"env_id", "done", "id"}
  assert info["elapsed_step"].tolist() == [5, 5, 5]
  obs, info = env.reset()
  assert env.reset_ids[-1].tolist() == [0, 1, 2] and "elapsed_step" in info


def test_front_end_dm_timestep():
  ts = _front("dm").step({"list_action": np.zeros((3, 6)), "players.action": np.zeros(3, np.int32),
                          "players.id": np.zeros(3, np.int32)})
  assert ts.step_type.tolist() == [0, 0, 0] and ts.first().all()
  assert ts.observation.raw.shape == (3, 10) and ts.observation.players.env_id.shape == (3,)
  assert ts.observation.env_id.tolist() == [0, 1, 2]


def test_lazy_dyn():
  from envpool_amd.python.lazy import LazyDyn
  d = LazyDyn(np.array([0, 2, 70000], np.int32), 10)
  assert len(d) == 3 and d[1].shape == (3, 10) and (d[1] == 2).all() and d[1].dtype == np.int32
  assert d[2].shape == (70001, 10)
  m = LazyDyn(np.array([1, 0], np.int32), 4).materialize()
  assert m.dtype == object and m[0].shape == (2, 4) and m[1].tolist() == [[0, 0, 0, 0]]


def test_shard_helpers():
  from envpool_amd.shard import owner_of, shard_range, split_by_owner
  assert shard_range(3, 8, 524288) == (196608, 65536)
  with pytest.raises(ValueError):
    shard_range(0, 3, 8)
  r, l = owner_of(np.array([0, 65535, 65536, 524287]), 8, 524288)
  assert r.tolist() == [0, 0, 1, 7] and l.tolist() == [0, 65535, 0, 65535]
  parts = split_by_owner([5, 0, 7, 2], 2, 8)
  assert parts[0].tolist() == [1, 3] and parts[1].tolist() == [0, 2]


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["EP_ROOT"])
import numpy as np, torch, torch.distributed as dist
from envpool_amd.shard import shard_range, all_gather_episode_stats
from oracle.dummy_oracle import DummyOracle
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
G, seed, steps = 16, 2, 40
off, n = shard_range(rank, world, G)
o = DummyOracle(num_envs=n, seed=seed, env_id_offset=off)
ids = np.arange(off, off + n, dtype=np.int32)
la = np.full((n, 6), 1.5); z = np.zeros(n, np.int32)
o._reset(ids); o._recv()
fin_len = np.zeros(n, np.int32); fin_ret = np.zeros(n, np.float32); acc = np.zeros(n, np.float32)
for t in range(steps):
  o._send([ids, ids, la, z, z]); r = o._recv()
  acc = np.where(r[2] == 0, 0, acc) + r[4]
  fin_len = np.where(r[3], r[2], fin_len); fin_ret = np.where(r[3], acc, fin_ret)
g_ret, g_len = all_gather_episode_stats(torch.from_numpy(fin_ret), torch.from_numpy(fin_len))
# every rank must hold the stats of the whole pool, ordered by global env_id
L = np.maximum(1, seed + np.arange(G))
want = np.where(L <= steps, L, 0)
assert g_len.numpy().tolist() == want.tolist(), (g_len, want)
assert g_ret.numpy().tolist() == [0.0] * G
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_stats_all_gather_gloo_world2(tmp_path):
  """N>1 path on CPU: two ranks own env_id blocks, step their shards, and all-gather the
  episode stats (gloo stands in for RCCL)."""
  script = tmp_path / "worker.py"
  script.write_text(_GLOO_WORKER)
  env = dict(os.environ, EP_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
  out = subprocess.run(
    [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
     "--master-port", "29533", str(script)], env=env, capture_output=True, text=True, timeout=240)
  assert out.returncode == 0, out.stdout + out.stderr
  assert out.stdout.count("ok") == 2


_BARRIER_WORKER = r"""
import os, sys, time
sys.path.insert(0, os.environ["EP_ROOT"])
import torch.distributed as dist
from bench import NodeBarrier
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
b = NodeBarrier(rank, world, dist)
for i in range(300):
  if i % world == rank and i % 10 == 0:
    time.sleep(0.002)                        # a straggler: nobody may pass generation i+1 before it arrives
  b()
  assert int(b.slots.min()) >= i + 1, (i, b.slots)
  assert int(b.slots.max()) <= i + 2, (i, b.slots)   # and nobody can be more than one barrier ahead
assert not os.path.exists(b.path)            # rank 0 unlinked it before its first barrier call
b.close(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_bench_node_barrier_gloo_world2(tmp_path):
  """bench.py's shared-memory barrier (the N>1 timed region is bracketed by it)."""
  script = tmp_path / "worker.py"
  script.write_text(_BARRIER_WORKER)
  env = dict(os.environ, EP_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
  out = subprocess.run(
    [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
     "--master-port", "29537", str(script)], env=env, capture_output=True, text=True, timeout=240)
  assert out.returncode == 0, out.stdout + out.stderr
  assert out.stdout.count("ok") == 2


def test_ygopro_spec_surface():
  """ygopro.h:1177-1207 + env_spec.h:25-42: keys, defaults, dtypes, shapes, bounds."""
  import envpool_amd
  from envpool_amd.ygopro import YGOProEnvSpec, _YGOProEnvSpec
  assert "YGOPro-v0" in envpool_amd.list_all_envs()
  assert _YGOProEnvSpec._config_keys[9:] == ["deck1", "deck2", "player", "play_mode", "verbose", "max_options",
                                              "max_cards", "n_history_actions", "max_multi_select"]
  assert _YGOProEnvSpec._default_config_values[9:] == ("OldSchool", "OldSchool", -1, "bot", False, 16, 75, 16, 5)
  spec = envpool_amd.make_spec("YGOPro-v0", num_envs=1024)
  assert isinstance(spec, YGOProEnvSpec) and spec.config.batch_size == 1024 and spec.config.max_cards == 75
  st = spec.state_array_spec
  assert list(st)[:8] == GOLD["state_keys"][:8]
  u8 = np.dtype(np.uint8)
  assert (st["obs:cards_"].dtype, st["obs:cards_"].shape) == (u8, [150, 39])
  assert st["obs:global_"].shape == [8] and st["obs:actions_"].shape == [16, 19] and st["obs:h_actions_"].shape == [16, 19]
  assert (st["info:num_options"].minimum, st["info:num_options"].maximum) == (0, 15)
  assert (st["info:win_reason"].minimum, st["info:win_reason"].maximum) == (-1, 1)
  ac = spec.action_array_spec
  assert list(ac) == ["env_id", "players.env_id", "action"] and (ac["action"].minimum, ac["action"].maximum) == (0, 15)
  s2 = envpool_amd.make_spec("YGOPro-v0", num_envs=8, max_multi_select=4, max_cards=60, n_history_actions=8)
  assert s2.state_array_spec["obs:actions_"].shape == [16, 17] and s2.state_array_spec["obs:cards_"].shape == [120, 39]
  assert s2.state_array_spec["obs:h_actions_"].shape == [8, 17]
  with pytest.raises(RuntimeError, match="duel engine"):
    envpool_amd.make("YGOPro-v0", "raw", num_envs=2)


class _OracleShard:
  """A shard pool for the GPU-less container: the CPU oracle behind the raw boundary."""

  def __init__(self, rank, n, b, off, dev, **kw):
    from oracle.dummy_oracle import DummyOracle
    self.o = DummyOracle(num_envs=n, batch_size=b, env_id_offset=off, **kw)

  def _send(self, a):
    assert self.o._send(a) == 0

  def _reset(self, ids):
    assert self.o._reset(ids) == 0

  def _recv(self):
    out = self.o._recv()
    out[9] = None
    return out


def _rows_by_env(batch, into):
  for j, e in enumerate(np.asarray(batch[0]).tolist()):
    into.setdefault(e, []).append(tuple(np.asarray(batch[k])[j].tobytes() for k in (1, 2, 3, 4, 5, 6, 7, 8, 10, 11)))


def test_sharded_front_end_stitches_in_sent_order():
  """the front-end splits a sent batch by owner and puts the shards' rows back in SENT
  order (async_envpool.h:69 + state_buffer.h:92-95). Four shard pools against ONE pool of 4N envs:
  permuted ids, partial sends, subset resets - bit-exact. (CPU oracle pools stand in for the GPU
  pools here; tests/test_sharded_gpu.py runs the same against the product.)"""
  from oracle.dummy_oracle import DummyOracle
  from envpool_amd.python.sharded import ShardedRawPool
  from tests.util import assert_batches_equal, dummy_action
  K, n, seed, off = 4, 48, 3, 1000
  sh = ShardedRawPool(K * n, 0, 1, 10, K, lambda *a: _OracleShard(*a, seed=seed), env_id_offset=off)
  one = DummyOracle(num_envs=K * n, seed=seed, env_id_offset=off)
  ids = np.arange(off, off + K * n, dtype=np.int32)
  rng = np.random.default_rng(0)

  def same(ctx):
    got, want = sh._recv(), one._recv()
    got[9] = want[9] = None
    assert_batches_equal(got, want, ctx)

  sh._reset(ids); one._reset(ids); same("reset")
  for t in range(30):
    if t % 7 == 3:      # partial send: a random subset in random order (some shards may get nothing)
      sel = rng.permutation(ids)[: int(rng.integers(1, K * n))]
    elif t % 7 == 5:    # everything that one shard owns, nothing else
      sel = ids[n:2 * n][::-1].copy()
    elif t % 7 == 0:    # every env in order: the path of slices
      sel = ids
    elif t % 7 == 1:    # an ascending subset (some shards may get nothing): slices as well
      sel = np.sort(rng.permutation(ids)[: int(rng.integers(1, K * n))])
    elif t % 7 == 2:    # ascending, one shard only
      sel = ids[2 * n + 5:3 * n - 7]
    else:
      sel = rng.permutation(ids)
    pe = sel + (rng.random(sel.size) < 0.1).astype(np.int32)
    a = dummy_action(sel, players_env_id=pe)
    sh._send(a); one._send(a); same(f"t={t}")
    if t == 11:
      sub = rng.permutation(ids)[:37]
      sh._reset(sub); one._reset(sub); same("subset reset")
  sh._recv_empty = sh._recv()   # nothing outstanding: an empty batch, like the reference
  assert len(sh._recv_empty[0]) == 0
  with pytest.raises(RuntimeError, match="one send"):
    sh._send(dummy_action(ids)); sh._send(dummy_action(ids))
  with pytest.raises(ValueError, match="outside this pool"):
    ShardedRawPool(8, 0, 1, 10, 2, lambda *a: _OracleShard(*a, seed=1))._send(dummy_action(np.array([8], np.int32)))


@pytest.mark.parametrize("P", [1, 3])
def test_sharded_front_end_async_batches(P):
  """Async mode: every shard runs with batch_size / K; a received batch is one batch of each shard
  (BASELINE config 5's shape). Compared with one big pool per env, as sets keyed by info:env_id."""
  from oracle.dummy_oracle import DummyOracle
  from envpool_amd.python.sharded import ShardedRawPool
  from tests.util import dummy_action
  K, n, b, seed = 2, 32, 8, 2
  sh = ShardedRawPool(K * n, K * b, P, 10, K, lambda *a: _OracleShard(*a, seed=seed, max_num_players=P))
  one = DummyOracle(num_envs=K * n, batch_size=K * b, seed=seed, max_num_players=P)
  ids = np.arange(K * n, dtype=np.int32)
  sh._reset(ids); one._reset(ids)
  got, want = {}, {}
  for it in range(12):
    gb, wb = [], []
    for _ in range(n // b):
      gb.append(sh._recv()); wb.append(one._recv())
      assert len(gb[-1][0]) == K * b
      _rows_by_env(gb[-1], got) if P == 1 else None
      _rows_by_env(wb[-1], want) if P == 1 else None
    # send every env back with one action row per player it reported
    pe_g = np.concatenate([x[1] for x in gb]); pe_w = np.concatenate([x[1] for x in wb])
    assert sorted(pe_g.tolist()) == sorted(pe_w.tolist())
    for x, rows in ((gb, got), (wb, want)):
      if P > 1:
        for bt in x:
          for e in np.asarray(bt[0]).tolist():
            rows.setdefault(e, []).append(tuple(np.asarray(bt[k])[np.asarray(bt[1]) == e].tobytes() for k in (4, 5, 8, 10, 11)))
    if P == 1:   # single player: action row i belongs to item i
      pe_g = pe_w = ids
    sh._send(dummy_action(ids, players_env_id=pe_g)); one._send(dummy_action(ids, players_env_id=pe_w))
  assert got == want


class _PollableShard:
  """A shard pool that can be polled: batches of `b` rows become complete only when the test says
  so (a shard whose duel engines are slow)."""

  def __init__(self, rank, n, b, off, dev):
    self.rank, self.b, self.off = rank, b, off
    self.ready = 0       # complete batches not yet handed out
    self.handed = 0
    self.blocking_recvs = 0

  def _batch(self):
    ids = (self.off + self.handed * self.b + np.arange(self.b)).astype(np.int32)
    self.handed += 1
    z = np.zeros(self.b, np.int32)
    return [ids, ids, z, z.astype(bool), z.astype(np.float32), z.astype(np.float32), z, z.astype(bool),
            np.zeros((self.b, 10), np.int32), None, z.astype(bool), z]

  def _try_recv(self):
    if self.ready == 0:
      return None
    self.ready -= 1
    return self._batch()

  def _recv(self):
    self.blocking_recvs += 1
    raise AssertionError("the front-end must not block on one shard")

  def _set_recv_timeout(self, ms):
    pass


def test_sharded_async_recv_takes_whoever_is_ready():
  """A node-wide async batch is K shard batches from WHICHEVER shards have one complete
  (state_buffer_queue.h:114-150 hands out the first batch_size arrivals): a shard with nothing
  ready does not hold the batch up, a shard that is always ready does not crowd the others out,
  and with nobody ready the call honours the recv timeout instead of sitting on one shard."""
  from envpool_amd.python.sharded import ShardedRawPool
  K, n, b = 4, 64, 8
  sh = ShardedRawPool(K * n, K * b, 1, 10, K, _PollableShard)
  s = sh.shards
  # shard 2 is starved, shard 0 has plenty: the batch comes from 0, 1, 3 and 0 again
  s[0].ready, s[1].ready, s[2].ready, s[3].ready = 5, 1, 0, 1
  got = sh._recv()
  assert len(got[0]) == K * b and sorted(sh.last_owners) == [0, 0, 1, 3]
  owners_of_rows = (np.asarray(got[0]) // n).tolist()
  assert sorted(set(owners_of_rows)) == [0, 1, 3]
  # now shard 2 wakes up: it is served although shard 0 still has batches
  s[2].ready = 2
  got = sh._recv()
  assert sorted(sh.last_owners) == [0, 0, 2, 2] or sorted(sh.last_owners) == [0, 2, 0, 2]
  assert sh.last_owners.count(2) == 2
  # nobody ready and a bounded timeout: would block, and what was taken is kept for the next call
  sh._set_recv_timeout(5)
  s[0].ready, s[1].ready, s[2].ready, s[3].ready = 1, 0, 0, 0   # (shard 0 has 1 left of its 5)
  with pytest.raises(RuntimeError, match="would block"):
    sh._recv()
  s[1].ready = 3
  got = sh._recv()
  assert len(got[0]) == K * b and sorted(sh.last_owners) == [0, 1, 1, 1]
  assert all(x.blocking_recvs == 0 for x in s)
  # every row handed out exactly once
  assert s[0].handed == 5 and s[1].handed == 4 and s[2].handed == 2 and s[3].handed == 1


def test_headers_are_plain_c_and_cxx(tmp_path):
  """include/*.h is the boundary: it must compile as C99 and as C++11, pedantic, with no warnings,
  and with nothing but the standard headers."""
  import shutil
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  src = tmp_path / "hdr.c"
  src.write_text('#include "envpool_amd.h"\n#include "envpool_amd_ygo.h"\n#include "envpool_amd_ygo_consts.h"\nint main(void) { return 0; }\n')
  for cc, std, extra in (("gcc", "-std=c99", []), ("g++", "-std=c++11", ["-x", "c++"])):
    if not shutil.which(cc):
      continue
    r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include")] + extra +
                       ["-c", str(src), "-o", str(tmp_path / "hdr.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_kept_results_are_handed_out_with_new_dicts():
  """python/streamstep.py:fresh - a step result kept per ring slot is handed out with new `obs` /
  `info` dicts (nested ones too) around the SAME leaf objects, as the reference builds its dicts anew
  on every recv; non-tuple results (dm's TimeStep) pass through."""
  from envpool_amd.python.streamstep import fresh
  leaf_a, leaf_b, leaf_c = object(), object(), object()
  kept = ({"raw": leaf_a}, leaf_b, leaf_c, leaf_c, {"env_id": leaf_a, "players": {"env_id": leaf_b}})
  out = fresh(kept)
  assert out is not kept and out[1] is leaf_b and out[0] is not kept[0] and out[4] is not kept[4]
  assert out[4]["players"] is not kept[4]["players"] and out[4]["players"]["env_id"] is leaf_b and out[0]["raw"] is leaf_a
  out[4]["mine"] = out[4]["players"]["mine"] = 1
  assert "mine" not in kept[4] and "mine" not in kept[4]["players"]
  nt = ("not", "a", "result")
  assert fresh(list(nt)) == list(nt)


def test_sharded_async_send_routes_runs_of_one_owner():
  """An async actor loop sends back the ids it was handed: whole shard batches in the order the shards
  were ready. `_positions` turns runs of one owner into slices (views downstream, no gather), a shard
  with two runs into one index array, and falls back to the per-shard search for scattered ids; a sync
  pool keeps the order-preserving rule (its rows come back in shard order). End to end: every batch
  sent back at once with its own ids, against one big pool, as sets keyed by info:env_id."""
  from oracle.dummy_oracle import DummyOracle
  from envpool_amd.python.sharded import ShardedRawPool
  from tests.util import dummy_action
  K, n, b, seed = 4, 16, 4, 3
  sh = ShardedRawPool(K * n, K * b, 1, 10, K, lambda *a: _OracleShard(*a, seed=seed))
  for ids in ([40, 41, 42, 43, 3, 2, 1, 0, 20, 21, 60, 63],            # runs of shards 2, 0, 1, 3
              [40, 41, 3, 2, 44, 45, 20, 21],                            # shard 2 in two runs
              list(np.random.default_rng(0).permutation(64)[:40])):    # scattered: the search per shard
    ids = np.asarray(ids, np.int32)
    pos = sh._positions(ids)
    assert len(pos) == K
    for r, p in enumerate(pos):
      sel = ids[p]
      assert sel.tolist() == [e for e in ids.tolist() if e // n == r], (ids.tolist(), r)
  assert isinstance(sh._positions(np.asarray([40, 41, 3, 2], np.int32))[2], slice)
  sync = ShardedRawPool(K * n, 0, 1, 10, K, lambda *a: _OracleShard(*a, seed=seed))
  assert not all(isinstance(p, slice) for p in sync._positions(np.asarray([40, 41, 3, 2], np.int32)))
  one = DummyOracle(num_envs=K * n, batch_size=K * b, seed=seed)
  all_ids = np.arange(K * n, dtype=np.int32)
  sh._reset(all_ids); one._reset(all_ids)
  got, want = {}, {}
  for it in range(40):
    g, w = sh._recv(), one._recv()
    _rows_by_env(g, got); _rows_by_env(w, want)
    sh._send(dummy_action(np.asarray(g[0]))); one._send(dummy_action(np.asarray(w[0])))
  common = set(got) & set(want)
  assert len(common) == K * n
  for e in common:  # (the two pools hand the envs out in different batches: compare the steps both have reported)
    m = min(len(got[e]), len(want[e]))
    assert m >= 1 and got[e][:m] == want[e][:m], e
