"""Soak of the sharded numpy front-end (python/sharded.py) against ONE oracle pool: random ascending
subsets (the path whose shards write one pinned array per key between them, ep_pool_host_target),
permuted sends (the gather / scatter path), one-shard sends and subset resets, mixed; a random
selection of earlier results is KEPT and compared again later (a block must not come round while an
array over it lives, and must once they are gone). SOAK_SECONDS (default 60), SOAK_SHARDS (4)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  from tests.util import assert_batches_equal, dummy_action
  K = int(os.environ.get("SOAK_SHARDS", "4"))
  seconds = float(os.environ.get("SOAK_SECONDS", "60"))
  rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
  total = 0
  t_end = time.time() + seconds
  rounds = 0
  while time.time() < t_end:
    n = int(rng.choice([64, 1000, 4096, 16384]))
    G, seed = K * n, int(rng.integers(1, 40))
    env = envpool_amd.make("Dummy-v0", "raw", num_envs=G, seed=seed, num_shards=K, shard_devices=[0] * K)
    one = DummyOracle(num_envs=G, seed=seed)
    ids = np.arange(G, dtype=np.int32)
    env._reset(ids); one._reset(ids)
    kept = [(env._recv(), one._recv())]
    assert_batches_equal(*kept[0], "reset")
    for t in range(int(rng.integers(40, 120))):
      kind = int(rng.integers(0, 6))
      if kind <= 1:
        sel = ids
      elif kind == 2:
        sel = np.sort(rng.choice(G, size=int(rng.integers(1, G)), replace=False)).astype(np.int32)
      elif kind == 3:
        sel = rng.permutation(ids)[: int(rng.integers(1, G + 1))]
      elif kind == 4:
        r = int(rng.integers(0, K))
        sel = ids[r * n:(r + 1) * n]
      else:
        sel = np.sort(rng.choice(G, size=int(rng.integers(1, G)), replace=False)).astype(np.int32)
        env._reset(sel); one._reset(sel)
        got, want = env._recv(), one._recv()
        assert_batches_equal(got, want, f"round {rounds} t={t} subset reset")
        total += sel.size
        continue
      pe = sel + (rng.random(sel.size) < 0.1).astype(np.int32)
      a = dummy_action(sel, players_env_id=pe, value=float(t % 5))
      env._send(a); one._send(a)
      got, want = env._recv(), one._recv()
      assert_batches_equal(got, want, f"round {rounds} t={t} kind {kind}")
      total += sel.size
      if rng.random() < 0.3:
        kept.append((got, want))
      if len(kept) > 12 or (kept and rng.random() < 0.1):
        kept.pop(int(rng.integers(0, len(kept))))
      if t % 16 == 7:
        for g, w in kept:
          assert_batches_equal(g, w, f"round {rounds} t={t}: a result kept from earlier")
    for g, w in kept:
      assert_batches_equal(g, w, f"round {rounds}: kept results at the end")
    env.close()
    for g, w in kept:   # the pools are gone; the arrays (and their blocks) are not
      assert_batches_equal(g, w, f"round {rounds}: kept results after close()")
    rounds += 1
    print(f"round {rounds}: {K} x {n} envs ok, {total} env-steps compared so far", flush=True)
  print(f"sharded soak passed: {rounds} pools, {total} env-steps compared with the oracle")


if __name__ == "__main__":
  main()
