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
