"""Tiny YGOPro-encode invocation for __graft_entry__.smoke(): 40 synthetic states through the
C ABI on cuda:0, checked against the oracle."""
import numpy as np


def ygo_smoke():
  from envpool_amd import capi, capi_ygo
  from oracle import ygo_oracle
  from tools import ygo_synth
  cfg = capi_ygo.default_config(max_batch=64)
  table = ygo_synth.make_table(200, seed=1)
  sb = ygo_synth.SynthBatch(cfg, table, 40, seed=1)
  want, status = ygo_oracle.write_batch(cfg, table, sb.raw, 40)
  recs, est = capi_ygo.emit_records(cfg, sb.raw, 40)
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  got = enc.featurize(recs, allow=(capi.EP_ERR_ACTION_CHECK,))
  ok = np.nonzero((status == 0) & (est == 0))[0]
  assert len(ok) >= 30
  for k in range(len(want)):
    assert np.asarray(got[k])[ok].tobytes() == np.asarray(want[k])[ok].tobytes(), f"ygo smoke: key {k} differs"
  enc.close()
