#!/usr/bin/env python3
"""GPU box: YGOPro-v0 behind the sharded front-end (make(..., num_shards=K), all shards on this one
GPU; stand-in engine in replay mode so that the engine is free): numpy `env.step` of a sync pool and
`recv` + `send` of an async pool (config 5's shape: 8192 envs, batches of 4096, 8 shards), with the
share of the per-key concatenate, against the unsharded pool of the same size."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def med(ts):
  ts = sorted(ts)
  return ts[len(ts) // 2] * 1e6


def main():
  import envpool_amd
  from envpool_amd import capi_ygo, ygopro
  from envpool_amd.python import sharded
  from tools import ygo_synth
  table = ygo_synth.make_table(1000, seed=0)
  cat = [0.0]
  orig = sharded.ShardedRawPool._cat_parts

  def timed(self, parts):
    t = time.perf_counter()
    out = orig(self, parts)
    cat[0] += time.perf_counter() - t
    return out
  sharded.ShardedRawPool._cat_parts = timed

  cases = ((1024, 1024, 1), (1024, 1024, 4), (8192, 4096, 1), (8192, 4096, 8))
  if os.environ.get("CASES"):
    cases = [cases[int(c)] for c in os.environ["CASES"].split(",")]
  for N, bs, K in cases:
    n = N // K
    cfg = capi_ygo.default_config(max_batch=min(bs, max(1, bs // K) if bs < N else n), ring_depth=0)
    engines = []

    def factory(spec, env_id_offset=0, shard=0):
      eng = ygo_synth.SynthEngine(cfg, table, n, seed=2 + shard)
      eng.set_replay()
      eng.set_env_id_offset(env_id_offset)
      engines.append(eng)
      return eng.ctx, eng.reset_addr, eng.step_addr, eng
    ygopro.init_module(factory, table)
    try:
      kw = dict(num_shards=K, shard_devices=[0] * K) if K > 1 else {}
      env = envpool_amd.make("YGOPro-v0", "gymnasium", num_envs=N, batch_size=bs, num_threads=4, **kw)
      act = np.zeros(bs, np.int32)
      ts = []
      if bs == N:
        env.reset()
        for i in range(120):
          if i == 20:
            cat[0] = 0.0
          t0 = time.perf_counter()
          env.step(act)
          ts.append(time.perf_counter() - t0)
        print(f"sync  {N} envs, {K} shard(s): env.step (numpy out) {med(ts[20:]):8.1f} us per step, "
              f"of which concatenate {cat[0] / 100 * 1e6:6.1f}", flush=True)
      else:
        env.async_reset()
        for i in range(120):
          if i == 20:
            cat[0] = 0.0
          t0 = time.perf_counter()
          obs, rew, term, trunc, info = env.recv()
          env.send(act, info["env_id"])
          ts.append(time.perf_counter() - t0)
        print(f"async {N} envs / {bs}, {K} shard(s): recv + send (numpy) {med(ts[20:]):8.1f} us per batch, "
              f"of which concatenate {cat[0] / 100 * 1e6:6.1f}", flush=True)
      env.close()
    finally:
      ygopro.init_module(None, None)
      for e in engines:
        e.close()


if __name__ == "__main__":
  main()
