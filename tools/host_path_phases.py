"""Phases of one host-memory (numpy in, numpy out) lockstep step at N=65536: the send call (copy of
the action arrays into pinned staging + launch) and the recv call (wait for the kernel, which reads
the staging area and writes the pinned result block over PCIe)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from envpool_amd import capi


def main(N=65536, steps=200):
  pool = capi.DummyPoolC(num_envs=N, seed=42, device_id=0, ring_depth=4)
  ids = np.arange(N, dtype=np.int32)
  la = np.full((N, 6), 1.5)
  z = np.zeros(N, dtype=np.int32)
  pool._reset(ids)
  pool._recv()
  ts, tr = [], []
  for i in range(steps + 20):
    t0 = time.perf_counter()
    pool._send([ids, ids, la, z, z])
    t1 = time.perf_counter()
    out = pool._recv()
    t2 = time.perf_counter()
    if i >= 20:
      ts.append(t1 - t0)
      tr.append(t2 - t1)
  ts.sort(); tr.sort()
  print(f"host path N={N}: send {ts[len(ts) // 2] * 1e6:.1f} us, recv {tr[len(tr) // 2] * 1e6:.1f} us (medians)", flush=True)
  pool.close()


if __name__ == "__main__":
  main()


def python_api(N=65536, steps=100):
  """The drop-in user's loop: envpool_amd.make(...).step(action) with numpy in and out."""
  import envpool_amd
  env = envpool_amd.make("Dummy-v0", "gym", num_envs=N, seed=42)
  env.reset()
  act = {"list_action": np.full((N, 6), 1.5), "players.action": np.zeros(N, np.int32), "players.id": np.zeros(N, np.int32)}
  for _ in range(10):
    env.step(act)
  t0 = time.perf_counter()
  for _ in range(steps):
    env.step(act)
  dt = (time.perf_counter() - t0) / steps
  t0 = time.perf_counter()
  for _ in range(steps):
    arrays = env._from(act, None)
  tf = (time.perf_counter() - t0) / steps
  raw = env._recv if False else None
  t0 = time.perf_counter()
  for _ in range(steps):
    env._send(arrays)
    vals = env._recv()
  tr = (time.perf_counter() - t0) / steps
  t0 = time.perf_counter()
  for _ in range(steps):
    env._to(vals, False, True)
  tt = (time.perf_counter() - t0) / steps
  print(f"python API N={N}: step {dt * 1e6:.1f} us = _from {tf * 1e6:.1f} + raw _send/_recv {tr * 1e6:.1f} + _to {tt * 1e6:.1f}", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "api":
  python_api(N=int(sys.argv[2]) if len(sys.argv) > 2 else 65536, steps=int(sys.argv[3]) if len(sys.argv) > 3 else 100)
