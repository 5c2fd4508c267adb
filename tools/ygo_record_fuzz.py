#!/usr/bin/env python3
"""GPU box: records are caller-supplied bytes (ep_ygo_submit / ep_ygo_featurize take them as they
are), so the kernel must survive any bytes: valid records with random bytes overwritten (header
counts, card words, option words), and records of pure noise, host and device-resident. Nothing to
compare with (the oracle consumes duel states, not records): the run must end without a fault, every
output array must come back, and untouched records of a batch must still equal the clean encoding."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rounds=int(os.environ.get("FUZZ_ROUNDS", "40"))):
  import torch

  from envpool_amd import capi, capi_ygo
  from tools import ygo_synth

  rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
  allow = (capi.EP_ERR_ACTION_CHECK,)
  n = 512
  total = 0
  for k in range(rounds):
    kw = {} if k % 2 == 0 else dict(max_cards=int(rng.integers(4, 100)), max_options=int(rng.integers(1, 48)),
                                     n_history_actions=int(rng.integers(1, 33)), max_multi_select=int(rng.integers(1, 8)))
    cfg = capi_ygo.default_config(max_batch=n, **kw)
    table = ygo_synth.make_table(1000, seed=k)
    try:
      enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
    except capi.EpError as ex:
      print("skip", kw, ex)
      continue
    sb = ygo_synth.SynthBatch(cfg, table, n, seed=100 + k)
    recs, est = capi_ygo.emit_records(cfg, sb.raw, n)
    clean = [np.asarray(a).copy() for a in enc.featurize(recs, allow=allow)]
    bad = recs.copy()
    hit = rng.random(n) < 0.5
    stride = recs.shape[1]
    for i in np.flatnonzero(hit):
      mode = int(rng.integers(0, 4))
      if mode == 0:    # header bytes (counts, num_options, ha_p, msg ...)
        pos = rng.integers(0, 64, size=int(rng.integers(1, 12)))
      elif mode == 1:  # anywhere
        pos = rng.integers(0, stride, size=int(rng.integers(1, 200)))
      elif mode == 2:  # the whole record is noise
        pos = np.arange(stride)
      else:            # counts at their maximum
        pos = np.arange(8, 40)
        bad[i, pos] = 0xff
        continue
      bad[i, pos] = rng.integers(0, 256, size=len(pos), dtype=np.uint8)
    for where in ("host", "device"):
      if where == "host":
        got = enc.featurize(bad, allow=allow)
      else:
        t = torch.from_numpy(bad.copy()).to("cuda:0")
        torch.cuda.synchronize()
        got = enc.featurize_device(t.data_ptr(), n, allow=allow)
      assert len(got) == len(clean)
      keep = np.flatnonzero(~hit)
      for key in range(len(clean)):
        a, b = np.asarray(got[key])[keep], clean[key][keep]
        assert a.tobytes() == b.tobytes(), ("an untouched record changed", where, key, kw)
    enc.close()
    total += n
    print(f"ok {k}: {kw or 'default'} corrupted={int(hit.sum())} flagged={enc.last_status == capi.EP_ERR_ACTION_CHECK}", flush=True)
  print("record fuzz passed:", total, "records")


if __name__ == "__main__":
  main()
