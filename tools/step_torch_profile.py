#!/usr/bin/env python3
"""GPU box: where the Python time of env.step_torch goes (cProfile over 2000 steps of a 65 536-env
pool), blocking and stream-ordered."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch

import envpool_amd

n = 65536
env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=42)
env.reset_torch()
la = torch.full((n, 6), 1.5, dtype=torch.float64, device="cuda")
act = {"list_action": la}
import gc
gc.collect(); gc.freeze()  # (a full collection of a torch process is a ~40 ms pause; see tools/sharded_step_time.py)
for so in (True, False, True):
  for i in range(4000):  # (long enough for the clocks to come up: a cold GPU runs the first milliseconds slowly)
    env.step_torch(act, stream_ordered=so)
  torch.cuda.current_stream().synchronize()
  t0 = time.perf_counter()
  for i in range(2000):
    env.step_torch(act, stream_ordered=so)
  torch.cuda.current_stream().synchronize()
  print(f"stream_ordered={so}: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per step", flush=True)
  pr = cProfile.Profile()
  pr.enable()
  for i in range(2000):
    env.step_torch(act, stream_ordered=so)
  pr.disable()
  torch.cuda.current_stream().synchronize()
  pstats.Stats(pr).sort_stats("tottime").print_stats(12)
