"""Host-side cost of one send + blocking recv per lockstep step (GPU box): where the ~15 us go."""
import ctypes as C, time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from envpool_amd import capi
L = capi.lib(); N = 65536
pool = capi.DummyPoolC(num_envs=N, seed=42, ring_depth=4); h = pool.handle
dev = torch.device("cuda:0")
env_id = torch.arange(N, dtype=torch.int32, device=dev)
la = torch.full((N, 6), 1.5, dtype=torch.float64, device=dev)
z = torch.zeros(N, dtype=torch.int32, device=dev)
p = (C.c_void_p * 5)(env_id.data_ptr(), env_id.data_ptr(), la.data_ptr(), z.data_ptr(), z.data_ptr())
dims = (C.c_int64 * 5)(N, N, N, N, N)
b = capi.Batch()
L.ep_pool_reset(h, env_id.data_ptr(), N, capi.EP_MEM_DEVICE); L.ep_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(b))
for _ in range(200):
  L.ep_pool_send(h, 5, p, dims, capi.EP_MEM_DEVICE); L.ep_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(b))
ts = tr = 0.0; K = 2000
for _ in range(K):
  t0 = time.perf_counter(); L.ep_pool_send(h, 5, p, dims, capi.EP_MEM_DEVICE)
  t1 = time.perf_counter(); L.ep_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(b)); t2 = time.perf_counter()
  ts += t1 - t0; tr += t2 - t1
print("send us", ts / K * 1e6, "recv us", tr / K * 1e6)
# small pool for the floor
pool2 = capi.DummyPoolC(num_envs=256, seed=42, ring_depth=4); h2 = pool2.handle
d2 = (C.c_int64 * 5)(256, 256, 256, 256, 256)
L.ep_pool_reset(h2, env_id.data_ptr(), 256, capi.EP_MEM_DEVICE); L.ep_pool_recv(h2, capi.EP_MEM_DEVICE, C.byref(b))
for _ in range(200):
  L.ep_pool_send(h2, 5, p, d2, capi.EP_MEM_DEVICE); L.ep_pool_recv(h2, capi.EP_MEM_DEVICE, C.byref(b))
ts = tr = 0.0
for _ in range(K):
  t0 = time.perf_counter(); L.ep_pool_send(h2, 5, p, d2, capi.EP_MEM_DEVICE)
  t1 = time.perf_counter(); L.ep_pool_recv(h2, capi.EP_MEM_DEVICE, C.byref(b)); t2 = time.perf_counter()
  ts += t1 - t0; tr += t2 - t1
print("N=256: send us", ts / K * 1e6, "recv us", tr / K * 1e6)
