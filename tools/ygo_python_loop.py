#!/usr/bin/env python3
"""GPU box: what the Python front-end of YGOPro-v0 costs per step (make("YGOPro-v0") after
init_module, stand-in engine in replay mode so that the engine is free): a sync pool of 1024 envs
through env.step (numpy out) and through send + recv_torch (device views out), and an async pool
(1024 envs, batches of 512) through recv_torch + send(actions, env.last_env_ids)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def med(ts):
  ts = sorted(ts)
  return ts[len(ts) // 2] * 1e6


def main():
  import torch

  import envpool_amd
  from envpool_amd import capi_ygo, ygopro
  from tools import ygo_synth
  N = int(os.environ.get("ENVS", "1024"))
  B = max(1, N // 2)
  table = ygo_synth.make_table(1000, seed=0)
  for mode, bs in (("sync", N), ("async", B)):
    cfg = capi_ygo.default_config(max_batch=bs, ring_depth=0)
    eng = ygo_synth.SynthEngine(cfg, table, N, seed=2)
    eng.set_replay()
    ygopro.init_module(eng.factory, table)
    try:
      env = envpool_amd.make("YGOPro-v0", "gymnasium", num_envs=N, batch_size=bs)
      act = np.zeros(bs, np.int32)
      if mode == "sync":
        env.reset()
        ts = []
        for i in range(220):
          t0 = time.perf_counter()
          env.step(act)
          ts.append(time.perf_counter() - t0)
        print(f"sync  {N} envs  env.step (numpy arrays out):            {med(ts[20:]):7.1f} us per step", flush=True)
        ts = []
        for i in range(220):
          t0 = time.perf_counter()
          env.send(act)
          obs, rew, term, trunc, info = env.recv_torch()
          ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        print(f"sync  {N} envs  send + recv_torch (device views out):    {med(ts[20:]):7.1f} us per step", flush=True)
      else:
        env.async_reset()
        ts = []
        for i in range(420):
          t0 = time.perf_counter()
          obs, rew, term, trunc, info = env.recv_torch()
          env.send(act, env.last_env_ids)
          ts.append(time.perf_counter() - t0)
        print(f"async {N} envs / {bs}  recv_torch + send(act, last_env_ids): {med(ts[20:]):7.1f} us per batch "
              f"= {bs / med(ts[20:]):.2f}e6 env-steps/s", flush=True)
        ts = []
        for i in range(420):
          t0 = time.perf_counter()
          obs, rew, term, trunc, info = env.recv_torch()
          env.send(act, info["env_id"].cpu().numpy())
          ts.append(time.perf_counter() - t0)
        print(f"async {N} envs / {bs}  recv_torch + send(act, info['env_id'].cpu()): {med(ts[20:]):7.1f} us per batch", flush=True)
      env.close()
    finally:
      ygopro.init_module(None, None)
      eng.close()


if __name__ == "__main__":
  main()
