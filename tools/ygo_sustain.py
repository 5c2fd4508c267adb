import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from envpool_amd import capi, capi_ygo
from tools import ygo_synth
L = capi_ygo.lib(); n_big = 65536
cfg = capi_ygo.default_config(max_batch=n_big, ring_depth=2)
table = ygo_synth.make_table(1000, seed=0)
sb = ygo_synth.SynthBatch(cfg, table, 8192, seed=0)
recs, est = capi_ygo.emit_records(cfg, sb.raw, 8192)
enc = capi_ygo.YgoEncoderC(cfg, table, len(table)); h = enc.handle
big = torch.from_numpy(np.tile(recs, (8, 1))).to("cuda:0"); torch.cuda.synchronize()
b = capi.Batch(); ok = (0, capi.EP_ERR_ACTION_CHECK)
def pipelined(iters):
  assert L.ep_ygo_submit_device(h, big.data_ptr(), n_big) == 0
  for _ in range(iters - 1):
    assert L.ep_ygo_submit_device(h, big.data_ptr(), n_big) == 0
    assert L.ep_ygo_collect(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok
  assert L.ep_ygo_collect(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok
for it in (20, 20, 100, 400, 400, 20):
  t0 = time.perf_counter(); pipelined(it); t1 = time.perf_counter()
  print(it, "us/batch", (t1 - t0) / it * 1e6, flush=True)
time.sleep(2)
for it in (20, 20):
  t0 = time.perf_counter(); pipelined(it); t1 = time.perf_counter()
  print("after sleep", it, "us/batch", (t1 - t0) / it * 1e6, flush=True)
