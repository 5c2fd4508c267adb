#!/usr/bin/env python3
"""GPU box: randomised parity soak of the YGOPro encoder against the oracle - many seeds, the
default config and a spread of other configs, host records (in-place PCIe reads) and
device-resident records. Prints one line per config and a total; exits 1 on the first mismatch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(total=int(os.environ.get("SOAK_RECORDS", "600000"))):
  import torch

  from envpool_amd import capi, capi_ygo
  from oracle import ygo_oracle
  from tools import ygo_synth

  rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
  done = 0
  t0 = time.time()
  k = 0
  stats = {"flagged": 0}
  while done < total:
    k += 1
    if k % 3 == 1:
      kw = {}
    else:
      mms = int(rng.integers(1, 9))
      kw = dict(max_cards=int(rng.integers(4, 128)), max_options=int(rng.integers(1, 65)),
                n_history_actions=int(rng.integers(1, 41)), max_multi_select=mms)
    n = int(rng.integers(1, 20000))
    seed = int(rng.integers(1, 1 << 30))
    cfg = capi_ygo.default_config(max_batch=n, **kw)
    try:
      enc_probe = None
      table = ygo_synth.make_table(int(rng.integers(10, 3000)), seed=seed)
      enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
    except capi.EpError as ex:  # e.g. a config whose LDS arena does not fit
      print("skip", kw, ex)
      continue
    sb = ygo_synth.SynthBatch(cfg, table, n, seed=seed)
    want, status = ygo_oracle.write_batch(cfg, table, sb.raw, n)
    recs, est = capi_ygo.emit_records(cfg, sb.raw, n)
    # (a state the emitter refuses comes back as its empty observation: it raises no flag of its own)
    ok = np.nonzero((status == 0) & (est == 0))[0]
    got = enc.featurize(recs, allow=(capi.EP_ERR_ACTION_CHECK,))
    flagged = enc.last_status == capi.EP_ERR_ACTION_CHECK
    assert flagged == bool(((status != 0) & (est == 0)).any()), ("flag", kw, n, seed)
    t = torch.from_numpy(recs.copy()).to("cuda:0")
    torch.cuda.synchronize()
    got_dev = enc.featurize_device(t.data_ptr(), n, allow=(capi.EP_ERR_ACTION_CHECK,))
    for name, g in (("host", got), ("device", got_dev)):
      for key in range(len(want)):
        a, b = np.asarray(g[key])[ok], np.asarray(want[key])[ok]
        if a.tobytes() != b.tobytes():
          bad = [i for i in range(len(ok)) if a[i].tobytes() != b[i].tobytes()]
          print("MISMATCH", name, "key", key, "config", kw, "n", n, "seed", seed, "first record", int(ok[bad[0]]), "count", len(bad))
          sys.exit(1)
    enc.close()
    stats["flagged"] += int(flagged)
    done += n
    print(f"ok {k}: n={n} seed={seed} {kw or 'default'} usable={len(ok)} total={done} t={time.time() - t0:.0f}s", flush=True)
  print("soak passed:", done, "records in", k, "batches;", stats)


if __name__ == "__main__":
  main()
