"""Kernel time of the one-launch-per-send step at N=65536 (HIP events of the pool's timing mode),
stream-ordered loop. Run with ENVPOOL_AMD_LIB pointing at another build to compare builds."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from envpool_amd import capi


def main(N=65536, steps=3000):
  L = capi.lib()
  dev = torch.device("cuda", 0)
  env_id = torch.arange(N, dtype=torch.int32, device=dev)
  la = torch.full((N, 6), 1.5, dtype=torch.float64, device=dev)
  unused = torch.zeros(N, dtype=torch.int32, device=dev)
  pool = capi.DummyPoolC(num_envs=N, seed=42, device_id=0, ring_depth=8)
  h = pool.handle
  batch = capi.Batch()
  p1 = (C.c_void_p * 5)(env_id.data_ptr(), env_id.data_ptr(), la.data_ptr(), unused.data_ptr(), unused.data_ptr())
  dims = (C.c_int64 * 5)(N, N, N, N, N)

  def check(rc):
    if rc != 0:
      raise RuntimeError(L.ep_last_error().decode())
  check(L.ep_pool_reset(h, env_id.data_ptr(), N, capi.EP_MEM_DEVICE))
  check(L.ep_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(batch)))
  for _ in range(3000):
    check(L.ep_pool_send(h, 5, p1, dims, capi.EP_MEM_DEVICE))
    check(L.ep_pool_recv_on_stream(h, None, C.byref(batch)))
  check(L.ep_pool_sync(h))
  t0 = time.perf_counter()
  for _ in range(steps):
    check(L.ep_pool_send(h, 5, p1, dims, capi.EP_MEM_DEVICE))
    check(L.ep_pool_recv_on_stream(h, None, C.byref(batch)))
  check(L.ep_pool_sync(h))
  dt = (time.perf_counter() - t0) / steps
  tot, cnt = C.c_double(), C.c_int64()
  check(L.ep_pool_set_timing(h, 1))
  check(L.ep_pool_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1))
  for _ in range(500):
    check(L.ep_pool_send(h, 5, p1, dims, capi.EP_MEM_DEVICE))
    check(L.ep_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(batch)))
  check(L.ep_pool_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1))
  print(f"{os.environ.get('ENVPOOL_AMD_LIB', 'in-tree')}: stream-ordered {dt * 1e6:.2f} us/step, step kernel {tot.value / max(1, cnt.value):.2f} us", flush=True)
  pool.close()


if __name__ == "__main__":
  main()
