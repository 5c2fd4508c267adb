"""BASELINE configs 4 and 5 in their drop-in form, on ONE GPU: K shard pools (all on device 0)
behind one front-end against ONE oracle pool of K*N envs; and two torch.distributed ranks (gloo)
each stepping the PRODUCT's pool of its env_id block."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.util import assert_batches_equal, dummy_action

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_sync_pool_matches_one_big_oracle():
  """Config 4's shape: make("Dummy-v0", num_envs=K*N, num_shards=K). Permuted global env_id,
  partial sends, subset resets; row i = position i of the SENT env_id (async_envpool.h:57-80,
  state_buffer.h:92-95); bit-exact against one K*N-env oracle."""
  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  K, n, seed = 4, 3000, 42
  env = envpool_amd.make("Dummy-v0", "raw", num_envs=K * n, seed=seed, num_shards=K, shard_devices=[0] * K)
  one = DummyOracle(num_envs=K * n, seed=seed)
  ids = np.arange(K * n, dtype=np.int32)
  rng = np.random.default_rng(1)

  def same(ctx):
    got, want = env._recv(), one._recv()
    assert_batches_equal(got, want, ctx)
    assert len(got[9]) == len(want[0])

  env._reset(ids); one._reset(ids); same("reset")
  for t in range(25):
    if t % 6 == 2:
      sel = rng.permutation(ids)[: int(rng.integers(1, K * n))]
    elif t % 6 == 4:
      sel = ids[2 * n:3 * n][::-1].copy()          # one shard's envs only
    else:
      sel = rng.permutation(ids)
    pe = sel + (rng.random(sel.size) < 0.1).astype(np.int32)
    a = dummy_action(sel, players_env_id=pe)
    env._send(a); one._send(a); same(f"t={t}")
    if t == 9:
      sub = rng.permutation(ids)[:500]
      env._reset(sub); one._reset(sub); same("subset reset")
  # the user-facing API on top: step() of the whole pool
  obs, rew, term, trunc, info = env.step({"list_action": np.full((K * n, 6), 2.0), "players.action": np.zeros(K * n, np.int32),
                                          "players.id": np.zeros(K * n, np.int32)})
  one._send(dummy_action(ids))
  want = one._recv()
  assert obs["raw"].tobytes() == want[8].tobytes() and info["env_id"].tobytes() == want[0].tobytes()
  assert info["elapsed_step"].tobytes() == want[2].tobytes() and term.tobytes() == (want[3] & ~want[7]).tobytes()
  r, l, c = env._episode_stats()
  assert r.shape == (K * n,) and (c >= 0).all()


def test_config4_full_size_eight_shards_match_one_oracle():
  """BASELINE config 4 at its own size in the drop-in form: make("Dummy-v0", num_envs=524288,
  num_shards=8) - here all eight shard pools on the one GPU of the box - against ONE 524 288-env
  oracle pool: reset, then sends of every env in ascending order, permuted, one shard's envs only
  and an ascending subset; row i = position i of the SENT env_id array (async_envpool.h:57-80,
  state_buffer.h:92-95), seeds from the GLOBAL id (env.h:94). Then the episode statistics against
  the env's closed form."""
  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  K, n = 8, 65536
  G = K * n
  seed = 3 - G // 2          # L = max(1, seed + id): half the envs have one-step episodes, the rest 4, 5, 6, ...
  env = envpool_amd.make("Dummy-v0", "raw", num_envs=G, seed=seed, num_shards=K, shard_devices=[0] * K)
  one = DummyOracle(num_envs=G, seed=seed)
  ids = np.arange(G, dtype=np.int32)
  rng = np.random.default_rng(4)
  sends = np.zeros(G, dtype=np.int64)   # per env: sends since the reset

  def same(ctx):
    got, want = env._recv(), one._recv()
    assert_batches_equal(got, want, ctx)

  env._reset(ids); one._reset(ids); same("reset")
  plan = ["ascending", "permuted", "ascending", "one shard", "subset", "ascending", "permuted", "ascending"]
  for t, kind in enumerate(plan):
    if kind == "ascending":
      sel = ids
    elif kind == "permuted":
      sel = rng.permutation(ids)
    elif kind == "one shard":
      sel = ids[5 * n:6 * n]
    else:
      sel = np.sort(rng.choice(G, size=G // 3, replace=False)).astype(np.int32)
    pe = sel + (rng.random(sel.size) < 0.05).astype(np.int32)   # a few rows name another env: obs:raw[:,1] = 0
    a = dummy_action(sel, players_env_id=pe)
    env._send(a); one._send(a); same(f"send {t} ({kind})")
    sends[sel] += 1
  ret, length, count = env._episode_stats()
  L = np.maximum(1, seed + ids.astype(np.int64))
  want_count = (sends + 1) // (L + 1)
  assert count.tolist() == want_count.tolist()
  assert length.tolist() == np.where(want_count > 0, L, 0).tolist()
  assert not ret.any()
  env.close() if hasattr(env, "close") else None


def test_sharded_async_pool_config5_shape():
  """Config 5's shape scaled down: async, batch_size = num_envs / 2, K shards each with
  batch_size / K. Per env the rows equal those of one big pool (sets keyed by info:env_id)."""
  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  K, n, b, seed = 4, 1024, 512, 7
  env = envpool_amd.make("Dummy-v0", "raw", num_envs=K * n, batch_size=K * b, seed=seed, num_shards=K,
                         shard_devices=[0] * K)
  one = DummyOracle(num_envs=K * n, batch_size=K * b, seed=seed)
  ids = np.arange(K * n, dtype=np.int32)
  env._reset(ids); one._reset(ids)

  def rows(batch, into):
    for j, e in enumerate(np.asarray(batch[0]).tolist()):
      into.setdefault(e, []).append(tuple(np.asarray(batch[k])[j].tobytes() for k in (1, 2, 3, 4, 5, 6, 7, 8, 10, 11)))

  got, want = {}, {}
  for it in range(10):
    for _ in range(n // b):
      g, w = env._recv(), one._recv()
      assert len(g[0]) == K * b == len(w[0])
      rows(g, got); rows(w, want)
    a = dummy_action(ids)
    env._send(a); one._send(a)
  assert got == want


_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["EP_ROOT"])
import numpy as np, torch, torch.distributed as dist
from envpool_amd.shard import shard_range, all_gather_episode_stats
from envpool_amd.capi import DummyPoolC
from oracle.dummy_oracle import DummyOracle
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
G, seed, steps = 2048, 2, 60
off, n = shard_range(rank, world, G)
g = DummyPoolC(num_envs=n, seed=seed, env_id_offset=off, device_id=0)   # the PRODUCT's pool of this rank's block
o = DummyOracle(num_envs=n, seed=seed, env_id_offset=off)
ids = np.arange(off, off + n, dtype=np.int32)
la = np.full((n, 6), 1.5); z = np.zeros(n, np.int32)
g._reset(ids); o._reset(ids)
assert [x.tobytes() for k, x in enumerate(g._recv()) if k != 9] == [x.tobytes() for k, x in enumerate(o._recv()) if k != 9]
for t in range(steps):
  g._send([ids, ids, la, z, z]); o._send([ids, ids, la, z, z])
  a, b = g._recv(), o._recv()
  assert [x.tobytes() for k, x in enumerate(a) if k != 9] == [x.tobytes() for k, x in enumerate(b) if k != 9], (rank, t)
fin_ret, fin_len, _ = g.episode_stats()
g_ret, g_len = all_gather_episode_stats(torch.from_numpy(fin_ret), torch.from_numpy(fin_len))
L = np.maximum(1, seed + np.arange(G))
want = np.where(L <= steps, L, 0)
assert g_len.numpy().tolist() == want.tolist()
assert g_ret.numpy().tolist() == [0.0] * G
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_ranks_step_product_pools_and_gather_stats(tmp_path):
  """The N>1 path with the product in it: two ranks (gloo; both on cuda:0 of this one-GPU box) own
  env_id blocks, step their pools against the oracle and all-gather the episode stats."""
  script = tmp_path / "worker.py"
  script.write_text(_WORKER)
  env = dict(os.environ, EP_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
  out = subprocess.run(
    [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
     "--master-port", "29537", str(script)], env=env, capture_output=True, text=True, timeout=240)
  assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
  assert out.stdout.count("ok") == 2


_YGO_WORKER = r"""
import os, sys, ctypes as C
sys.path.insert(0, os.environ["EP_ROOT"])
import numpy as np, torch, torch.distributed as dist
from envpool_amd import capi, capi_ygo
from envpool_amd.shard import shard_range
from oracle import ygo_oracle
from tools import ygo_synth
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
G, B = 512, 128                               # config 5's shape scaled down: G envs over `world` pools, async batches
off, n = shard_range(rank, world, G)
cfg = capi_ygo.default_config(max_batch=B, ring_depth=0, device_id=0)
table = ygo_synth.make_table(1000, seed=0)
eng = ygo_synth.SynthEngine(cfg, table, n, seed=3 + rank)
eng.set_env_id_offset(off)
model = ygo_synth.EnginePoolModel(cfg, table, n, seed=3 + rank)
pool = capi_ygo.YgoPoolC(cfg, eng.engine(), n, B, table, len(table), num_threads=3)
pool.set_env_id_offset(off)
pool.set_recv_timeout(30000)
ids = np.arange(off, off + n, dtype=np.int32)
pool.reset(ids)
expect = {int(e): model.advance(int(e) - off) for e in ids}
seen = []
checked = 0
for it in range(6):
  out = pool.recv(allow=(capi.EP_ERR_ACTION_CHECK,))
  got = out[0]
  assert ((got >= off) & (got < off + n)).all()
  raws = [expect.pop(int(e)) for e in got]
  arr = (capi_ygo.RawState * len(raws))()
  for i, (r, e) in enumerate(zip(raws, got)):
    C.memmove(C.byref(arr[i]), C.byref(r), C.sizeof(capi_ygo.RawState))
    arr[i].env_id = int(e)
  want, status = ygo_oracle.write_batch(cfg, table, arr, len(raws))
  for j in range(len(raws)):
    if status[j] == 0:
      assert all(np.asarray(out[k][j]).tobytes() == np.asarray(want[k][j]).tobytes() for k in range(18)), (rank, it, j)
      checked += 1
  seen.append(got.copy())
  act = (got % 7).astype(np.int32)
  pool.send(got, act)
  for e, a in zip(got, act):
    expect[int(e)] = model.advance(int(e) - off, int(a))
assert checked > 0.9 * 6 * B
# the ids every rank received, gathered: the union is the global id range, each shard its own block
mine = torch.from_numpy(np.unique(np.concatenate(seen)).astype(np.int64))
cnt = torch.tensor([mine.numel()])
cnts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(cnts, cnt)
pad = torch.full((max(int(c) for c in cnts),), -1, dtype=torch.int64); pad[:mine.numel()] = mine
parts = [torch.zeros_like(pad) for _ in range(world)]
dist.all_gather(parts, pad)
allids = np.concatenate([p.numpy()[p.numpy() >= 0] for p in parts])
assert len(set(allids.tolist())) == len(allids) and allids.min() >= 0 and allids.max() < G
pool.close(); eng.close()
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_ranks_ygopro_pools_over_sharded_ids(tmp_path):
  """Config 5's multi-GPU shape with the product in it: two ranks (gloo; both on cuda:0 of this
  one-GPU box) each run an async ep_ygo_pool over their block of GLOBAL env ids
  (ep_ygo_pool_set_env_id_offset); every row is checked against the oracle, and the ids the ranks
  received partition the global range."""
  script = tmp_path / "worker.py"
  script.write_text(_YGO_WORKER)
  env = dict(os.environ, EP_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
  out = subprocess.run(
    [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
     "--master-port", "29539", str(script)], env=env, capture_output=True, text=True, timeout=240)
  assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
  assert out.stdout.count("ok") == 2


def _np(t):
  return t.detach().cpu().numpy()


@pytest.mark.parametrize("servers", [True, False, "launch"])
def test_sharded_device_path_sync_matches_one_oracle(servers, monkeypatch):
  """reset_torch / step_torch / send_torch + recv_torch of a sharded sync pool (K = 4 pools on one
  GPU): the rows never leave the device; per shard the result is that shard's block of ONE big
  oracle pool, and .cat() is the whole batch in sent (= ascending id) order. servers = True: the
  shards' resident servers are rung together (ep_pool_step_begin / _end), which pools that share a
  GPU only do when told to (EP_SERVER_SHARED_DEVICE: one pool per GPU is the deployment the
  server is for); False: what this one-GPU configuration gets by default - every shard steps in
  stream order (ep_pool_step_on_stream), the host waits for nothing and the rows are valid for work
  on torch's current stream; "launch": the blocking form without servers, one launch + wait per shard."""
  import torch
  kw = {} if servers is False else {"stream_ordered": False}
  if servers is True:
    monkeypatch.setenv("EP_SERVER_SHARED_DEVICE", "1")
    monkeypatch.setenv("EP_SERVER_MAX_PER_DEVICE", "4")

  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  K, n, seed = 4, 4096, 9
  G = K * n
  env = envpool_amd.make("Dummy-v0", "raw", num_envs=G, seed=seed, num_shards=K, shard_devices=[0] * K)
  one = DummyOracle(num_envs=G, seed=seed)
  ids = np.arange(G, dtype=np.int32)
  dev = torch.device("cuda", 0)

  def check(res, want, ctx, reset=False):
    assert len(res) == K and res.owners == list(range(K))
    for r, part in enumerate(res):
      sl = slice(r * n, (r + 1) * n)
      obs, info = (part[0], part[-1])
      assert _np(obs["raw"]).tobytes() == want[8][sl].tobytes(), (ctx, r)
      assert _np(info["env_id"]).tobytes() == want[0][sl].tobytes(), (ctx, r)
      assert _np(info["elapsed_step"]).tobytes() == want[2][sl].tobytes(), (ctx, r)
      if not reset:
        assert _np(part[1]).tobytes() == want[4][sl].tobytes() and _np(part[3]).tobytes() == want[7][sl].tobytes(), (ctx, r)
        assert _np(part[2]).tobytes() == (want[3] & ~want[7])[sl].tobytes(), (ctx, r)
    whole = res.cat()
    assert _np(whole[0]["raw"]).tobytes() == want[8].tobytes() and _np(whole[-1]["players"]["env_id"]).tobytes() == want[1].tobytes(), ctx

  res = env.reset_torch()
  one._reset(ids)
  check(res, one._recv(), "reset", reset=True)
  la = torch.full((G, 6), 1.5, dtype=torch.float64, device=dev)
  penv = torch.from_numpy(ids).to(dev)
  act = {"list_action": la, "players.action": penv, "players.id": penv, "players.env_id": penv}
  rng = np.random.default_rng(0)
  for t in range(30):
    pe = ids + (rng.random(G) < 0.1).astype(np.int32)
    penv.copy_(torch.from_numpy(pe).to(dev))
    res = env.step_torch(act, **kw)
    one._send(dummy_action(ids, players_env_id=pe))
    check(res, one._recv(), f"step {t}")
  # actions that need a conversion (float32 lanes) are copied into staging tensors the front-end keeps
  la32 = la.to(torch.float32)
  for t in range(3):
    res = env.step_torch(dict(act, list_action=la32), **kw)
    one._send(dummy_action(ids, players_env_id=pe))
    check(res, one._recv(), f"float32 lanes, step {t}")
  # the servers were rung (unless the idle watchdog, which this slow loop invites, moved a shard to the launch path)
  served = sum(s._server_stats()[2] for s in env.shards)
  assert served >= K if servers is True else served == 0
  if servers is False:
    # many steps with nothing read in between: the host runs ahead of the device, the ring's slots
    # come round (views and result templates are kept per slot), and the last result is still right
    for t in range(200):
      res = env.step_torch(act)
      one._send(dummy_action(ids, players_env_id=pe))
      want = one._recv()
    check(res, want, "after 200 unread steps")
    # a step whose list_action lanes differ is flagged by its kernel; the flag surfaces at a later
    # step_torch (or _sync_all), not at the step itself
    la[n + 5, 3] = 2.5
    env.step_torch(act)
    one._send(dummy_action(ids, players_env_id=pe))
    one._recv()
    torch.cuda.synchronize()
    la[n + 5, 3] = 1.5
    with pytest.raises(RuntimeError, match="self-check"):
      env.step_torch(act)
      env._sync_all()
    # the raising call stepped nothing (every shard is polled before any steps): carry on
    res = env.step_torch(act)
    one._send(dummy_action(ids, players_env_id=pe))
    check(res, one._recv(), "step after the flagged one")
  # send_torch + recv_torch with per-part lists: one shard only, then two shards
  for who in ([2], [0, 3]):
    parts_ids = [torch.from_numpy(ids[r * n:(r + 1) * n].copy()).to(dev) for r in who]
    parts_act = [{"list_action": la[r * n:(r + 1) * n], "players.action": p, "players.id": p} for r, p in zip(who, parts_ids)]
    env.send_torch(parts_act, parts_ids, owners=who)
    res = env.recv_torch()
    sel = np.concatenate([ids[r * n:(r + 1) * n] for r in who])
    one._send(dummy_action(sel))
    want = one._recv()
    assert res.owners == who
    got = res.cat()
    assert _np(got[0]["raw"]).tobytes() == want[8].tobytes() and _np(got[-1]["env_id"]).tobytes() == want[0].tobytes()
  # and the numpy boundary still works on the same pool afterwards
  a = dummy_action(rng.permutation(ids))
  env._send(a); one._send(a)
  assert_batches_equal(env._recv(), one._recv(), "numpy after torch")
  env.close()


def test_sharded_rollout_torch_matches_one_oracle():
  """rollout_torch of a sharded pool (K = 4 pools on one GPU, ring_depth 8): one fused launch per
  shard reading its columns of the caller's [T, G, 6] / [T, G] tensors in place; every step's parts
  are that shard's block of ONE big oracle pool stepped T times. Twice, so the shards' rings wrap."""
  import torch
  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  K, n, seed, T = 4, 4096, 6, 6
  G = K * n
  env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=G, seed=seed, num_shards=K, shard_devices=[0] * K, ring_depth=8)
  one = DummyOracle(num_envs=G, seed=seed)
  ids = np.arange(G, dtype=np.int32)
  dev = torch.device("cuda", 0)
  env.reset_torch()
  one._reset(ids); one._recv()
  rng = np.random.default_rng(2)
  for rnd in range(2):
    pe = ids[None, :] + (rng.random((T, G)) < 0.1).astype(np.int32)
    lane = rng.integers(0, 4, size=(T, G)).astype(np.float64)
    la = torch.from_numpy(np.repeat(lane[:, :, None], 6, axis=2)).to(dev)
    out = env.rollout_torch({"list_action": la, "players.env_id": torch.from_numpy(pe).to(dev)})
    assert len(out) == T
    for t in range(T):
      one._send(dummy_action(ids, players_env_id=pe[t]))
      want = one._recv()
      assert len(out[t]) == K
      for r, (obs, rew, term, trunc, info) in enumerate(out[t]):
        sl = slice(r * n, (r + 1) * n)
        ctx = (rnd, t, r)
        assert _np(obs["raw"]).tobytes() == want[8][sl].tobytes(), ctx
        assert _np(info["env_id"]).tobytes() == want[0][sl].tobytes() and _np(info["elapsed_step"]).tobytes() == want[2][sl].tobytes(), ctx
        assert _np(trunc).tobytes() == want[7][sl].tobytes() and _np(term).tobytes() == (want[3] & ~want[7])[sl].tobytes(), ctx
  # the same rows every step, no players.env_id; then an ordinary step on the same pools
  la1 = torch.full((G, 6), 2.0, dtype=torch.float64, device=dev)
  out = env.rollout_torch(la1, T=3)
  for t in range(3):
    one._send(dummy_action(ids))
    want = one._recv()
    assert _np(out[t].cat()[0]["raw"]).tobytes() == want[8].tobytes(), t
  res = env.step_torch({"list_action": la1})
  one._send(dummy_action(ids))
  assert _np(res.cat()[0]["raw"]).tobytes() == one._recv()[8].tobytes()
  env.close()


def test_sharded_device_path_async_sets_by_env():
  """Async sharded pool on the device path: recv_torch hands out K shard batches from whichever
  shards are ready (`owners`), send_torch routes per-part actions back; per env the rows equal one
  big pool's (sets keyed by info:env_id)."""
  import torch

  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  K, n, b, seed = 4, 1024, 256, 11
  G = K * n
  env = envpool_amd.make("Dummy-v0", "raw", num_envs=G, batch_size=K * b, seed=seed, num_shards=K, shard_devices=[0] * K)
  env._set_recv_timeout(2000)
  one = DummyOracle(num_envs=G, batch_size=K * b, seed=seed)
  ids = np.arange(G, dtype=np.int32)
  dev = torch.device("cuda", 0)
  one._reset(ids)
  res = env.reset_torch()
  got, want = {}, {}

  def note(into, env_ids, raw, elapsed):
    for e, r, s in zip(env_ids.tolist(), raw, elapsed.tolist()):
      into.setdefault(e, []).append((r.tobytes(), s))

  outstanding = G
  for it in range(24):
    assert len(res) == K and all(0 <= r < K for r in res.owners)
    ids_parts, act_parts = [], []
    for part in res:
      e = part[-1]["env_id"]
      assert e.shape[0] == b
      note(got, _np(e), _np(part[0]["raw"]), _np(part[-1]["elapsed_step"]))
      ids_parts.append(e)
      act_parts.append({"list_action": torch.full((b, 6), 0.5, dtype=torch.float64, device=dev), "players.action": e, "players.id": e})
    w = one._recv()
    note(want, w[0], w[8], w[2])
    # send every received env back; the oracle gets the ids IT handed out
    env.send_torch(act_parts, ids_parts, owners=res.owners)
    one._send(dummy_action(w[0], value=0.5))
    res = env.recv_torch()
  # both sides handed out every env equally often only over whole rounds; compare the common prefix per env
  checked = 0
  for e in range(G):
    g_rows, w_rows = got.get(e, []), want.get(e, [])
    m = min(len(g_rows), len(w_rows))
    assert g_rows[:m] == w_rows[:m], e
    checked += m
  assert checked > 0.8 * 24 * K * b
  env.close()


def test_sharded_ygopro_pools_behind_one_front_end():
  """make("YGOPro-v0", num_envs=K*n, num_shards=K): K ep_ygo_pools (global ids in blocks,
  ep_ygo_pool_set_env_id_offset), one duel engine per shard from the init_module factory. Sync:
  rows in sent order; async (config 5's shape scaled down): parts from whoever is ready. Every row
  against the oracle's encoding of the engine model's state."""
  import ctypes as C

  import envpool_amd
  from envpool_amd import capi_ygo, ygopro
  from oracle import ygo_oracle
  from tools import ygo_synth
  K, n = 4, 64
  G = K * n
  cfg = capi_ygo.default_config(max_batch=n, ring_depth=0)
  table = ygo_synth.make_table(1000, seed=0)
  engines, models = {}, {}

  def factory(spec, env_id_offset=0, shard=0):
    eng = ygo_synth.SynthEngine(cfg, table, n, seed=20 + shard)
    eng.set_env_id_offset(env_id_offset)
    engines[shard] = eng
    models[shard] = ygo_synth.EnginePoolModel(cfg, table, n, seed=20 + shard)
    return eng.ctx, eng.reset_addr, eng.step_addr, eng

  def expect(env_ids, actions):
    raws = []
    for e, a in zip(env_ids, actions):
      r, loc = divmod(int(e), n)
      raw = models[r].advance(loc) if a is None else models[r].advance(loc, int(a))
      raw.env_id = int(e)
      raws.append(raw)
    arr = (capi_ygo.RawState * len(raws))()
    for i, r in enumerate(raws):
      C.memmove(C.byref(arr[i]), C.byref(r), C.sizeof(capi_ygo.RawState))
    return ygo_oracle.write_batch(cfg, table, arr, len(raws))

  ygopro.init_module(factory, table)
  try:
    # ---- sync ----
    env = envpool_amd.make("YGOPro-v0", "raw", num_envs=G, num_threads=2, num_shards=K, shard_devices=[0] * K)
    env._set_recv_timeout(30000)
    ids = np.arange(G, dtype=np.int32)
    rng = np.random.default_rng(2)
    env._reset(ids)
    out = env._recv()
    want, status = expect(ids, [None] * G)
    ok = np.flatnonzero(np.asarray(status) == 0)
    for k in range(16):
      assert np.asarray(out[k])[ok].tobytes() == np.asarray(want[k])[ok].tobytes(), ("reset", k)
    assert env.last_env_ids.tolist() == ids.tolist() and env.last_action_mask.shape == (G, 16)
    for it in range(6):
      sel = ids if it % 2 == 0 else np.sort(rng.choice(G, size=int(rng.integers(1, G)), replace=False)).astype(np.int32)
      act = rng.integers(0, 16, size=sel.size).astype(np.int32)
      env._send([sel, sel, act])
      out = env._recv()
      assert out[0].tolist() == sel.tolist()
      want, status = expect(sel, act)
      ok = np.flatnonzero(np.asarray(status) == 0)
      for k in range(16):
        assert np.asarray(out[k])[ok].tobytes() == np.asarray(want[k])[ok].tobytes(), (it, k)
      assert np.asarray(env.last_h_card_ids)[ok].tobytes() == np.asarray(want[16])[ok].tobytes()
    # the user-facing API and the device path: torch views per shard, mask included
    obs, rew, term, trunc, info = env.step(np.ones(G, dtype=np.int32))
    assert obs["cards_"].shape == (G, 150, 39) and info["env_id"].tolist() == ids.tolist()
    expect(ids, np.ones(G, np.int32))
    env.send(np.full(G, 2, np.int32))
    res = env.recv_torch()
    expect(ids, np.full(G, 2, np.int32))
    assert len(res) == K and all(p[0]["cards_"].is_cuda and tuple(p[0]["cards_"].shape) == (n, 150, 39) for p in res)
    assert all(tuple(p[-1]["action_mask"].shape) == (n, 16) for p in res)
    assert np.concatenate([p[-1]["env_id"].cpu().numpy() for p in res]).tolist() == ids.tolist()
    env.close()
    for e in engines.values():
      e.close()
    engines.clear(); models.clear()

    # ---- async: 4 x 64 envs, node-wide batches of 128 = 4 shard batches of 32 ----
    env = envpool_amd.make("YGOPro-v0", "raw", num_envs=G, batch_size=G // 2, num_threads=2, num_shards=K, shard_devices=[0] * K)
    env._set_recv_timeout(30000)
    env._reset(ids)
    pending = {}
    want0, status0 = expect(ids, [None] * G)
    for j, e in enumerate(ids.tolist()):
      pending[e] = (tuple(np.asarray(want0[k][j]).tobytes() for k in range(16)), int(status0[j]))
    checked = 0
    for it in range(10):
      out = env._recv()
      got_ids = np.asarray(out[0])
      assert got_ids.shape[0] == G // 2 and len(set(got_ids.tolist())) == G // 2
      assert env.last_env_ids.tolist() == got_ids.tolist()
      for j, e in enumerate(got_ids.tolist()):
        row, st = pending.pop(e)
        if st == 0:
          assert tuple(np.asarray(out[k][j]).tobytes() for k in range(16)) == row, (it, j, e)
          checked += 1
      act = rng.integers(0, 16, size=got_ids.size).astype(np.int32)
      env._send([got_ids, got_ids, act])
      w, stt = expect(got_ids, act)
      for j, e in enumerate(got_ids.tolist()):
        pending[e] = (tuple(np.asarray(w[k][j]).tobytes() for k in range(16)), int(stt[j]))
    assert checked > 0.85 * 10 * (G // 2)
    env.close()
  finally:
    ygopro.init_module(None, None)
    for e in engines.values():
      e.close()


_SLOT_KEYS = [("<i4", 1, 0), ("<i4", 1, 1), ("<i4", 1, 2), ("|b1", 1, 3), ("<f4", 1, 4), ("<f4", 1, 5), ("<i4", 1, 6),
              ("|b1", 1, 7), ("<i4", 10, 8), ("|b1", 1, 10), ("<i4", 1, 11)]   # (dtype, elements per row, state key)


def _target_layout(rows):
  offs, at = [], 0
  for dt, elems, _ in _SLOT_KEYS:
    offs.append(at)
    at = (at + np.dtype(dt).itemsize * elems * rows + 255) // 256 * 256
  return offs, at


def _target_batch(blk, offs, rows, row0=0):
  out = [None] * 12
  for off, (dt, elems, k) in zip(offs, _SLOT_KEYS):
    a = blk.array(off, (row0 + rows,) + ((elems,) if elems > 1 else ()), dt)
    out[k] = a[row0:]
  return out


@pytest.mark.parametrize("direct", ["1", "0"])
def test_host_target_delivers_a_batch_into_the_callers_block(direct, monkeypatch):
  """ep_pool_host_target: the next batch a pool starts lands in the caller's pinned block, key by
  key at the offsets given - written there by the step kernels (host-memory sends, direct = "1") or
  copied there by recv (EP_DIRECT_HOST=0) - bit-exact against the oracle; two pools fill two row
  ranges of ONE array per key (what python/sharded.py builds on). What is refused; a cancelled
  target; a batch received as device views after all; a block is not recycled while an array over it
  lives."""
  import torch
  from envpool_amd.dummy.dummy_envpool import HostBlocks, _DummyEnvPool, _DummyEnvSpec
  from oracle.dummy_oracle import DummyOracle
  monkeypatch.setenv("EP_DIRECT_HOST", direct)
  n, seed = 5000, 11
  keys = _DummyEnvSpec._config_keys
  dflt = dict(zip(keys, _DummyEnvSpec._default_config_values))

  def pool(**kw):
    return _DummyEnvPool(_DummyEnvSpec(tuple(dict(dflt, **kw)[k] for k in keys)))

  lo, hi = pool(num_envs=n, batch_size=n, seed=seed), pool(num_envs=n, batch_size=n, seed=seed, env_id_offset=n)
  one = DummyOracle(num_envs=2 * n, seed=seed)
  ids = np.arange(2 * n, dtype=np.int32)
  offs, nbytes = _target_layout(2 * n)
  hb = HostBlocks(nbytes)
  rng = np.random.default_rng(0)
  kept = []
  for t in range(12):
    blk = hb.take()
    lo._host_target(blk, offs)
    hi._host_target(blk, offs, n)
    if t == 0:
      lo._reset(ids[:n]); hi._reset(ids[n:]); one._reset(ids)
    else:
      pe = ids + (rng.random(2 * n) < 0.1).astype(np.int32)
      a = dummy_action(ids, players_env_id=pe, value=float(t))
      lo._send([x[:n] for x in a]); hi._send([x[n:] for x in a]); one._send(a)
    assert lo._recv_target() == n and hi._recv_target() == n
    got, want = _target_batch(blk, offs, 2 * n), one._recv()
    assert_batches_equal(got, want, f"t={t}")
    kept.append((got, want, blk.ptr))
    del blk
    if t >= 3:   # three batches stay alive: their blocks must not come round again
      kept.pop(0)
    assert len({p for *_, p in kept}) == len(kept)
    for g, w, _ in kept:
      assert_batches_equal(g, w, f"kept at t={t}")
  ptrs_seen = {p for *_, p in kept}
  kept.clear(); del got
  assert hb.take().ptr in ptrs_seen                      # ... and do once nothing refers to them
  # refusals (nothing is launched for any of them)
  blk = hb.take()
  with pytest.raises((RuntimeError, ValueError), match="needs .* bytes at offset"):
    lo._host_target(blk, [o + nbytes for o in offs])
  with pytest.raises((RuntimeError, ValueError), match="11 key offsets"):
    lo._host_target(blk, offs[:10])
  mp = pool(num_envs=64, batch_size=64, max_num_players=3, seed=seed)
  with pytest.raises((RuntimeError, ValueError), match="max_num_players == 1"):
    mp._host_target(blk, offs)
  # a target that is cancelled, and one whose batch is received as device views
  from envpool_amd import interop
  p2, o2 = pool(num_envs=n, batch_size=n, seed=seed + 1), DummyOracle(num_envs=n, seed=seed + 1)
  p2._reset(ids[:n]); o2._reset(ids[:n])
  assert_batches_equal(p2._recv(), o2._recv(), "reset of the second pool")
  sentinel = _target_batch(blk, offs, 2 * n)
  for k in (0, 2, 8):
    sentinel[k][...] = -7
  a = dummy_action(ids[:n], value=3.0)
  p2._host_target(blk, offs)
  p2._host_target_cancel()
  p2._send(a); o2._send(a)
  assert_batches_equal(p2._recv(), o2._recv(), "after a cancelled target")
  p2._host_target(blk, offs)
  p2._send(a); o2._send(a)
  descs, want = p2._recv_device()[0], o2._recv()
  dev = torch.device("cuda", 0)
  for k in (0, 2, 3, 4, 8):
    assert interop.as_torch(descs[k], dev).cpu().numpy().tobytes() == np.asarray(want[k]).tobytes(), k
  if direct == "0":   # (in direct mode the kernels had written the block before the batch was asked for as device views)
    for k in (0, 2, 8):
      assert (sentinel[k] == -7).all()
  for q in (lo, hi, mp, p2):
    q.close()


def test_host_target_c_abi_refusals():
  """The C entry points themselves: a block of another set, a block already given back, null
  arguments; and a block given back while a pool still holds it as a target stays mapped until the
  pool is done with it (the pool's batch is received into it afterwards)."""
  import ctypes as C
  from envpool_amd import capi
  from tests.test_dummy_gpu_r2 import _pair
  L = capi.lib()
  n = 4096
  g, o = _pair(num_envs=n, seed=5)
  offs, nbytes = _target_layout(n)
  c_offs = (C.c_size_t * 11)(*offs)
  A, B = C.c_void_p(), C.c_void_p()
  assert L.ep_host_blocks_create(nbytes + 256, C.byref(A)) == 0 and L.ep_host_blocks_create(nbytes + 256, C.byref(B)) == 0
  none = C.c_void_p()
  assert L.ep_host_blocks_create(0, C.byref(none)) == capi.EP_ERR_INVALID and not none.value
  blk = C.c_void_p()
  assert L.ep_host_blocks_take(A, C.byref(blk)) == 0 and blk.value
  assert L.ep_pool_host_target(g.handle, B, blk, c_offs, 0) == capi.EP_ERR_INVALID
  assert b"not a block that is out of this set" in L.ep_last_error()
  assert L.ep_pool_host_target(g.handle, A, blk, None, 0) == capi.EP_ERR_INVALID
  assert L.ep_pool_host_target(None, A, blk, c_offs, 0) == capi.EP_ERR_INVALID
  assert L.ep_pool_host_target(g.handle, A, blk, c_offs, 100) == capi.EP_ERR_INVALID    # the last key would run past the block
  assert L.ep_pool_host_target(g.handle, A, blk, c_offs, -1) == capi.EP_ERR_INVALID
  odd = (C.c_size_t * 11)(*[o + 4 for o in offs])
  assert L.ep_pool_host_target(g.handle, A, blk, odd, 0) == capi.EP_ERR_INVALID and b"16-byte aligned" in L.ep_last_error()
  assert L.ep_host_blocks_give(B, blk) == capi.EP_ERR_INVALID
  # target set, then the caller's own hold dropped: the pool's hold keeps the block
  assert L.ep_pool_host_target(g.handle, A, blk, c_offs, 0) == 0, L.ep_last_error()
  assert L.ep_host_blocks_give(A, blk) == 0
  ids = np.arange(n, dtype=np.int32)
  g._reset(ids); o._reset(ids)
  b = capi.Batch()
  assert L.ep_pool_recv(g.handle, capi.EP_MEM_HOST, C.byref(b)) == 0, L.ep_last_error()
  assert b.owner is None and b.arrays[0].data == blk.value + offs[0] and b.arrays[8].data == blk.value + offs[8]
  got, want = capi.batch_to_numpy(b), o._recv()
  assert L.ep_batch_release(C.byref(b)) == 0
  assert_batches_equal(got, want, "reset into a block the caller has let go of")
  a = dummy_action(ids, value=2.0)
  g._send(a); o._send(a)
  assert_batches_equal(g._recv(), o._recv(), "the next batch goes into the pool's own block again")
  assert L.ep_host_blocks_destroy(A) == 0 and L.ep_host_blocks_destroy(B) == 0


def test_stitch_parts_copies_every_key_of_every_part():
  """stitch_parts (the sharded front-end's concatenate): parts of different row counts, a key without an
  array, a non-contiguous input, sizes below and above the point where the copy is handed to the thread
  crew; what does not fit is refused before anything is copied."""
  from envpool_amd.dummy.dummy_envpool import HostBlocks, stitch_parts
  rng = np.random.default_rng(3)
  for rows_per_part in ([5, 1, 9], [3000, 4096, 1, 2500, 4096, 4096]):
    shapes = [((), np.int32), ((160, 41), np.uint8), None, ((7,), np.float32), ((), np.bool_)]
    total = sum(rows_per_part)
    offs, at = [], 0
    for sh in shapes:
      offs.append(at)
      if sh is not None:
        at = (at + np.dtype(sh[1]).itemsize * int(np.prod(sh[0], dtype=np.int64)) * total + 255) // 256 * 256
    hb = HostBlocks(at)
    parts = []
    for j, r in enumerate(rows_per_part):
      part = []
      for sh in shapes:
        if sh is None:
          part.append(None)
          continue
        a = rng.integers(0, 2 if sh[1] is np.bool_ else 100, size=(2 * r,) + sh[0]).astype(sh[1])
        part.append(a[::2] if j % 2 else np.ascontiguousarray(a[::2]))    # odd parts: a strided view
      parts.append(part)
    blk = hb.take()
    rows = stitch_parts(blk, offs, parts)
    assert list(rows) == [total, total, 0, total, total]
    for k, sh in enumerate(shapes):
      if sh is None:
        continue
      got = blk.array(offs[k], (total,) + sh[0], np.dtype(sh[1]).str)
      want = np.concatenate([p[k] for p in parts])
      assert got.dtype == want.dtype and got.tobytes() == want.tobytes(), (rows_per_part, k)
    with pytest.raises(ValueError, match="do not fit"):
      stitch_parts(blk, [o + 256 for o in offs], parts + parts)
    with pytest.raises(ValueError, match="one entry per key"):
      stitch_parts(blk, offs, [parts[0][:3]])
