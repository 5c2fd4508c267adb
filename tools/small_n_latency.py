"""Blocking host-memory step latency (C ABI through ctypes, numpy in / host block out) at small N."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from envpool_amd import capi


def main():
  L = capi.lib()
  for N in (8, 64, 256, 1024, 4096):
    pool = capi.DummyPoolC(num_envs=N, seed=42, device_id=0)
    h = pool.handle
    ids = np.arange(N, dtype=np.int32)
    la = np.full((N, 6), 1.5)
    z = np.zeros(N, dtype=np.int32)
    ph = (C.c_void_p * 5)(ids.ctypes.data, ids.ctypes.data, la.ctypes.data, z.ctypes.data, z.ctypes.data)
    dims = (C.c_int64 * 5)(N, N, N, N, N)
    batch = capi.Batch()
    assert L.ep_pool_reset(h, ids.ctypes.data, N, capi.EP_MEM_HOST) == 0
    assert L.ep_pool_recv(h, capi.EP_MEM_HOST, C.byref(batch)) == 0
    L.ep_batch_release(C.byref(batch))
    ts = []
    for i in range(1200):
      t0 = time.perf_counter()
      assert L.ep_pool_send(h, 5, ph, dims, capi.EP_MEM_HOST) == 0
      assert L.ep_pool_recv(h, capi.EP_MEM_HOST, C.byref(batch)) == 0
      L.ep_batch_release(C.byref(batch))
      if i >= 200:
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"N={N}: {ts[len(ts) // 2] * 1e6:.1f} us per blocking host step (EP_DIRECT_MIN_ROWS={os.environ.get('EP_DIRECT_MIN_ROWS', 'default')})", flush=True)
    pool.close()


if __name__ == "__main__" and len(sys.argv) == 1:
  main()


def async_small():
  """An async pool's loop through host memory: recv a batch, send its envs back."""
  L = capi.lib()
  for N, B in ((64, 16), (1024, 256), (8192, 2048)):
    pool = capi.DummyPoolC(num_envs=N, batch_size=B, seed=42, device_id=0)
    h = pool.handle
    ids = np.arange(N, dtype=np.int32)
    la = np.full((B, 6), 1.5)
    z = np.zeros(B, dtype=np.int32)
    batch = capi.Batch()
    assert L.ep_pool_reset(h, ids.ctypes.data, N, capi.EP_MEM_HOST) == 0
    ts = []
    got = np.zeros(B, dtype=np.int32)
    for i in range(1200):
      t0 = time.perf_counter()
      assert L.ep_pool_recv(h, capi.EP_MEM_HOST, C.byref(batch)) == 0
      C.memmove(got.ctypes.data, batch.arrays[0].data, 4 * B)
      L.ep_batch_release(C.byref(batch))
      ph = (C.c_void_p * 5)(got.ctypes.data, got.ctypes.data, la.ctypes.data, z.ctypes.data, z.ctypes.data)
      dims = (C.c_int64 * 5)(B, B, B, B, B)
      assert L.ep_pool_send(h, 5, ph, dims, capi.EP_MEM_HOST) == 0
      if i >= 200:
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"async N={N} B={B}: {ts[len(ts) // 2] * 1e6:.1f} us per recv + send through host memory", flush=True)
    pool.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "async":
  async_small()


def mp_small():
  """Multi-player pools through host memory (numpy in and out), blocking step."""
  from envpool_amd.capi import DummyPoolC
  for N in (64, 1024, 8192):
    pool = DummyPoolC(num_envs=N, seed=42, device_id=0, max_num_players=3)
    ids = np.arange(N, dtype=np.int32)
    pool._reset(ids)
    out = pool._recv()
    L = capi.lib()
    h = pool.handle
    batch = capi.Batch()
    ts = []
    penv = np.ascontiguousarray(out[1])
    la = np.full((N, 6), 1.5)
    for i in range(700):
      z = np.zeros(len(penv), dtype=np.int32)
      ph = (C.c_void_p * 5)(ids.ctypes.data, penv.ctypes.data, la.ctypes.data, z.ctypes.data, z.ctypes.data)
      dims = (C.c_int64 * 5)(N, len(penv), N, len(penv), len(penv))
      t0 = time.perf_counter()
      assert L.ep_pool_send(h, 5, ph, dims, capi.EP_MEM_HOST) == 0
      assert L.ep_pool_recv(h, capi.EP_MEM_HOST, C.byref(batch)) == 0
      t1 = time.perf_counter()
      penv = np.ctypeslib.as_array(C.cast(batch.arrays[1].data, C.POINTER(C.c_int32)), shape=(batch.num_player_rows,)).copy()
      L.ep_batch_release(C.byref(batch))
      if i >= 100:
        ts.append(t1 - t0)
    ts.sort()
    print(f"multi-player P=3 N={N}: {ts[len(ts) // 2] * 1e6:.1f} us per blocking host step", flush=True)
    pool.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "mp":
  mp_small()
