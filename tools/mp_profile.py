"""Multi-player step loop for profiling: max_num_players=3, N=65536, every env sent back each step
with one action row per player row it reported. Run under rocprofv3 --kernel-trace --stats."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from envpool_amd import capi


def main(steps=300, N=65536, P=3):
  L = capi.lib()
  dev = torch.device("cuda", 0)
  env_id = torch.arange(N, dtype=torch.int32, device=dev)
  la = torch.full((N, 6), 1.5, dtype=torch.float64, device=dev)
  unused = torch.zeros(4 * N, dtype=torch.int32, device=dev)
  mp = capi.DummyPoolC(num_envs=N, seed=42, device_id=0, max_num_players=P, ring_depth=4)
  h = mp.handle
  batch = capi.Batch()

  def check(rc):
    if rc != 0:
      raise RuntimeError(L.ep_last_error().decode())
  check(L.ep_pool_reset(h, env_id.data_ptr(), N, capi.EP_MEM_DEVICE))
  check(L.ep_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(batch)))

  def step():
    rows = batch.num_player_rows
    pm = (C.c_void_p * 5)(env_id.data_ptr(), batch.arrays[1].data, la.data_ptr(), unused.data_ptr(), unused.data_ptr())
    dm = (C.c_int64 * 5)(N, rows, N, rows, rows)
    check(L.ep_pool_send(h, 5, pm, dm, capi.EP_MEM_DEVICE))
    check(L.ep_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(batch)))
  for _ in range(200):
    step()
  t0 = time.perf_counter()
  for _ in range(steps):
    step()
  dt = (time.perf_counter() - t0) / steps
  tot, cnt = C.c_double(), C.c_int64()
  check(L.ep_pool_set_timing(h, 1))
  check(L.ep_pool_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1))
  for _ in range(100):
    step()
  check(L.ep_pool_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1))
  print(f"MP P={P} N={N}: {dt * 1e6:.1f} us per step, fused kernel {tot.value / max(1, cnt.value):.2f} us "
        f"(EP_MP_SKIP_COUNT={os.environ.get('EP_MP_SKIP_COUNT', '0')})", flush=True)
  mp.close()


if __name__ == "__main__":
  main()
