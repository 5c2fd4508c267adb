#!/usr/bin/env python3
"""GPU box: per-step cost of Dummy-v0 through the Python front-end (numpy in, numpy out): sync
step() at a few sizes, and an async pool's recv() + send() loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def med(ts):
  ts = sorted(ts)
  return ts[len(ts) // 2] * 1e6


def main():
  import envpool_amd
  for n in (8, 256, 4096, 65536):
    env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=42)
    env.reset()
    act = {"list_action": np.full((n, 6), 1.5), "players.action": np.zeros(n, np.int32), "players.id": np.zeros(n, np.int32)}
    ts = []
    for i in range(700):
      t0 = time.perf_counter()
      env.step(act)
      ts.append(time.perf_counter() - t0)
    print(f"sync  {n:6d} envs: env.step {med(ts[100:]):7.1f} us", flush=True)
    env.close()
  for n, b in ((64, 16), (1024, 256), (8192, 2048)):
    env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, batch_size=b, seed=42)
    env.async_reset()
    act = {"list_action": np.full((b, 6), 1.5), "players.action": np.zeros(b, np.int32), "players.id": np.zeros(b, np.int32)}
    ts = []
    for i in range(1200):
      t0 = time.perf_counter()
      obs, rew, term, trunc, info = env.recv()
      env.send(act, info["env_id"])
      ts.append(time.perf_counter() - t0)
    print(f"async {n:6d} envs / {b}: recv + send {med(ts[200:]):7.1f} us per batch", flush=True)
    env.close()


if __name__ == "__main__":
  main()
