#!/usr/bin/env python3
"""GPU box: ONE Dummy pool driven for tens of millions of lockstep steps (the soaks cover variety,
this covers duration: ring wrap-arounds, error-word generations, launch stamps and the resident step
server over millions of uses), checked against the env's closed form every so often: with
L = max(1, seed + env_id) and p = t mod (L + 1), elapsed_step = obs:raw[:, 0] = p, done = (p == L)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  from envpool_amd import capi
  N, T, seed = 4096, 64, 5
  rollouts = int(os.environ.get("LONG_ROLLOUTS", "300000"))
  server_steps = int(os.environ.get("LONG_SERVER_STEPS", "1500000"))
  L = capi.lib()
  L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
  L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
  pool = capi.DummyPoolC(num_envs=N, batch_size=N, seed=seed, ring_depth=2 * T)
  h = pool.handle
  ids = np.arange(N, dtype=np.int32)
  la = np.full((N, 6), 1.5)
  zeros = np.zeros(N, np.int32)
  dptr = []
  for a in (ids, ids, la, zeros, zeros):
    d = C.c_void_p()
    assert L.hipMalloc(C.byref(d), a.nbytes) == 0 and L.hipMemcpy(d, a.ctypes.data, a.nbytes, 1) == 0
    dptr.append(d.value)
  ptrs = (C.c_void_p * 5)(*dptr)
  dims = (C.c_int64 * 5)(*([N] * 5))
  strides = (C.c_int64 * 5)(*([0] * 5))
  ep_len = np.maximum(1, seed + ids.astype(np.int64))

  def check(batch, t, what):
    p = t % (ep_len + 1)
    assert np.array_equal(batch[0], ids), (what, t, "env_id")
    assert np.array_equal(batch[2], p), (what, t, "elapsed_step")
    assert np.array_equal(np.asarray(batch[8])[:, 0], p), (what, t, "obs:raw")
    assert np.array_equal(batch[3], p == ep_len), (what, t, "done")

  pool._reset(ids)
  check(pool._recv(), 0, "reset")
  t = 0
  t0 = time.time()
  many = (capi.Batch * T)()
  for r in range(rollouts):
    capi._check(L.ep_pool_rollout(h, T, 5, ptrs, dims, strides))
    if r % 5000 == 0:
      for k in range(T):
        t += 1
        check(pool._recv(), t, "rollout")
      print(f"rollout {r}: step {t} checked, {time.time() - t0:.0f}s", flush=True)
    else:
      capi._check(L.ep_pool_recv_many(h, capi.EP_MEM_DEVICE, T, many))
      t += T
  print(f"{t} lockstep steps ({t * N:.3e} env-steps) through ep_pool_rollout", flush=True)
  b = capi.Batch()
  for s in range(server_steps):
    capi._check(L.ep_pool_step(h, 5, ptrs, dims, C.byref(b)))
    t += 1
    if s % 250000 == 0:
      capi._check(L.ep_pool_send(h, 5, ptrs, dims, capi.EP_MEM_DEVICE))   # leaves the server, takes the launch path
      t += 1
      check(pool._recv(), t, "send + recv between server steps")
      print(f"server step {s}: step {t} checked, {time.time() - t0:.0f}s", flush=True)
  capi._check(L.ep_pool_send(h, 5, ptrs, dims, capi.EP_MEM_DEVICE))
  t += 1
  check(pool._recv(), t, "final")
  pool.close()
  print(f"dummy long run passed: {t} lockstep steps of {N} envs = {t * N:.3e} env-steps, {time.time() - t0:.0f}s")


if __name__ == "__main__":
  main()
