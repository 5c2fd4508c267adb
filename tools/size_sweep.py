#!/usr/bin/env python3
"""GPU box: blocking sync step (numpy in, numpy out, C ABI through ctypes) over pool sizes from 1 to
131072, powers of two and their neighbours: looks for cliffs at the thresholds of the host path
(pieces, direct host mode, staging ring), not for a headline number."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  from envpool_amd import capi
  sizes = sorted({max(1, n + d) for k in range(0, 18) for n in [1 << k] for d in (-1, 0, 1)})
  for P in (1, 3):
    prev = None
    for N in sizes:
      if P == 3 and N > 32768:
        continue
      pool = capi.DummyPoolC(num_envs=N, batch_size=N, seed=42, max_num_players=P)
      ids = np.arange(N, dtype=np.int32)
      pool._reset(ids)
      out = pool._recv()
      ts = []
      for i in range(160):
        penv = np.asarray(out[1])
        act = [ids, penv, np.full((N, 6), 1.5), np.zeros(penv.size, np.int32), np.zeros(penv.size, np.int32)] if P > 1 else \
              [ids, ids, np.full((N, 6), 1.5) if i == 0 else act[2], ids, ids]
        t0 = time.perf_counter()
        pool._send(act)
        out = pool._recv()
        ts.append(time.perf_counter() - t0)
      ts = sorted(ts[40:])
      us = ts[len(ts) // 2] * 1e6
      flag = "  <-- more than 1.6x the size before" if prev and us > 1.6 * prev and us > 25 else ""
      print(f"P={P} N={N:7d}: {us:8.1f} us per step{flag}", flush=True)
      prev = us
      pool.close()


if __name__ == "__main__":
  main()
