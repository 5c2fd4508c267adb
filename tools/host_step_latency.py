"""Host-buffer path (numpy in, numpy out) of one 65536-env lockstep step: where the time goes (GPU box)."""
import ctypes as C, time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from envpool_amd import capi
L = capi.lib(); N = 65536
pool = capi.DummyPoolC(num_envs=N, seed=42, ring_depth=4); h = pool.handle
ids = np.arange(N, dtype=np.int32); la = np.full((N, 6), 1.5); z = np.zeros(N, np.int32)
p = (C.c_void_p * 5)(ids.ctypes.data, ids.ctypes.data, la.ctypes.data, z.ctypes.data, z.ctypes.data)
dims = (C.c_int64 * 5)(N, N, N, N, N)
b = capi.Batch()
L.ep_pool_reset(h, ids.ctypes.data, N, capi.EP_MEM_HOST); L.ep_pool_recv(h, capi.EP_MEM_HOST, C.byref(b)); L.ep_batch_release(C.byref(b))
for _ in range(20):
  L.ep_pool_send(h, 5, p, dims, capi.EP_MEM_HOST); L.ep_pool_recv(h, capi.EP_MEM_HOST, C.byref(b)); L.ep_batch_release(C.byref(b))
ts = tr = tf = 0.0; K = 200
for _ in range(K):
  t0 = time.perf_counter(); L.ep_pool_send(h, 5, p, dims, capi.EP_MEM_HOST)
  t1 = time.perf_counter(); L.ep_pool_recv(h, capi.EP_MEM_HOST, C.byref(b)); t2 = time.perf_counter()
  L.ep_batch_release(C.byref(b)); t3 = time.perf_counter()
  ts += t1 - t0; tr += t2 - t1; tf += t3 - t2
print("host path: send us", ts / K * 1e6, "recv us", tr / K * 1e6, "release us", tf / K * 1e6)
src = np.empty(4 << 20, np.uint8); dst = np.empty(4 << 20, np.uint8)
t0 = time.perf_counter()
for _ in range(50): np.copyto(dst, src)
print("4 MiB memcpy us", (time.perf_counter() - t0) / 50 * 1e6)
