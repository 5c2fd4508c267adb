"""Blocking YGOPro encode latency for small batches of host records (ep_ygo_featurize)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from envpool_amd import capi, capi_ygo
from tools import ygo_synth


def main():
  L = capi_ygo.lib()
  table = ygo_synth.make_table(1000, seed=0)
  for n in (1, 16, 256, 1024):
    cfg = capi_ygo.default_config(max_batch=n, ring_depth=4)
    sb = ygo_synth.SynthBatch(cfg, table, n, seed=1)
    recs, _ = capi_ygo.emit_records(cfg, sb.raw, n)
    enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
    b = capi.Batch()
    for mem_out, name in ((capi.EP_MEM_DEVICE, "device views"), (capi.EP_MEM_HOST, "host arrays")):
      ts = []
      for i in range(600):
        t0 = time.perf_counter()
        rc = L.ep_ygo_featurize(enc.handle, recs.ctypes.data, n, capi.EP_MEM_HOST, mem_out, C.byref(b))
        assert rc in (0, capi.EP_ERR_ACTION_CHECK)
        if mem_out == capi.EP_MEM_HOST:
          L.ep_batch_release(C.byref(b))
        if i >= 100:
          ts.append(time.perf_counter() - t0)
      ts.sort()
      print(f"ygo featurize n={n}, host records -> {name}: {ts[len(ts) // 2] * 1e6:.1f} us", flush=True)
    enc.close()


if __name__ == "__main__":
  main()
