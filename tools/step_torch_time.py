#!/usr/bin/env python3
"""GPU box: env.step_torch (CUDA tensors in, views of the device ring out) over pool sizes, including
ones the resident step server does not serve (num_envs % 256 != 0: send + recv underneath)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  import torch

  import envpool_amd
  for n in (8, 100, 256, 1000, 4096, 65536, 65537):
    env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=42)
    env.reset_torch()
    la = torch.full((n, 6), 1.5, dtype=torch.float64, device="cuda")
    ts = []
    for i in range(1500):
      t0 = time.perf_counter()
      obs, rew, term, trunc, info = env.step_torch({"list_action": la})
      ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    assert term.dtype == torch.bool and int(info["elapsed_step"].max()) >= 0
    ts = sorted(ts[300:])
    print(f"step_torch {n:6d} envs: median {ts[len(ts) // 2] * 1e6:6.1f} us, p90 {ts[int(len(ts) * 0.9)] * 1e6:6.1f}", flush=True)
    env.close()


if __name__ == "__main__":
  main()
