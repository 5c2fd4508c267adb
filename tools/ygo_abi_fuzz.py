#!/usr/bin/env python3
"""GPU box: the YGOPro C ABI (encoder, record ring, pool) under calls it must refuse - null
arguments, counts outside the batch, unknown memory kinds, calls of the wrong mode, configs outside
the limits - each followed by a valid encode of the same batch, which must still give the clean
result. Pointers that are passed are valid; what is fuzzed is everything the library can check."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rounds=int(os.environ.get("FUZZ_ROUNDS", "10"))):
  from envpool_amd import capi, capi_ygo
  from tools import ygo_synth

  rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
  L = capi_ygo.lib()
  H, D = capi.EP_MEM_HOST, capi.EP_MEM_DEVICE
  allow = (capi.EP_ERR_ACTION_CHECK,)
  refused = 0
  for r in range(rounds):
    B = int(rng.choice([1, 16, 256]))
    n = int(rng.integers(1, B + 1))
    cfg = capi_ygo.default_config(max_batch=B, ring_depth=3)
    table = ygo_synth.make_table(500, seed=r)
    enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
    e = enc.handle
    sb = ygo_synth.SynthBatch(cfg, table, n, seed=200 + r)
    recs, _ = capi_ygo.emit_records(cfg, sb.raw, n)
    clean = [np.array(a) for a in enc.featurize(recs, allow=allow)]
    stride = recs.shape[1]
    b = capi.Batch()
    slot, tick, done = C.c_void_p(), C.c_uint64(0), C.c_int32(0)
    offs = np.zeros(B, np.uint32)
    bad_cfg = capi_ygo.default_config(max_batch=B)
    bad_cfg.max_cards = 0
    huge_cfg = capi_ygo.default_config(max_batch=B)
    huge_cfg.max_options = 100000
    ids_p, ids_n = C.POINTER(C.c_int32)(), C.c_int32(0)
    hp = C.c_int32(0)
    ring = np.zeros(16 * 19, np.uint8)
    menu = [
      lambda: L.ep_ygo_featurize(e, recs.ctypes.data, -1, H, H, C.byref(b)),
      lambda: L.ep_ygo_featurize(e, recs.ctypes.data, B + 1, H, H, C.byref(b)),
      lambda: L.ep_ygo_featurize(e, None, n, H, H, C.byref(b)),
      lambda: L.ep_ygo_featurize(e, recs.ctypes.data, n, H, H, None),
      lambda: L.ep_ygo_featurize(e, recs.ctypes.data, n, int(rng.choice([2, -1, 77])), H, C.byref(b)),
      lambda: L.ep_ygo_featurize(e, recs.ctypes.data, n, H, int(rng.choice([2, -1, 77])), C.byref(b)),
      lambda: L.ep_ygo_featurize(None, recs.ctypes.data, n, H, H, C.byref(b)),
      lambda: L.ep_ygo_submit(e, recs.ctypes.data, B + 7),
      lambda: L.ep_ygo_submit(e, None, n),
      lambda: L.ep_ygo_submit_device(e, None, n),
      lambda: L.ep_ygo_commit_slot(e, -2),
      lambda: L.ep_ygo_commit_slot(e, B + 1),
      lambda: L.ep_ygo_commit_slot_packed(e, n, None, 64 * n),
      lambda: L.ep_ygo_commit_slot_packed(e, n, offs.ctypes.data, -5),
      lambda: L.ep_ygo_commit_slot_packed(e, n, offs.ctypes.data, stride * B + 4096),
      lambda: L.ep_ygo_commit_slot_rows(e, -1, 1, n, C.byref(done)),
      lambda: L.ep_ygo_commit_slot_rows(e, 0, 0, n, C.byref(done)),
      lambda: L.ep_ygo_commit_slot_rows(e, 0, n + 1, n, C.byref(done)),
      lambda: L.ep_ygo_commit_slot_rows(e, 0, 1, B + 1, C.byref(done)),
      lambda: L.ep_ygo_commit_slot_rows(e, 0, 1, n, None),
      lambda: L.ep_ygo_collect(e, H, C.byref(b)),           # nothing submitted
      lambda: L.ep_ygo_collect(e, H, None),
      lambda: L.ep_ygo_recv(e, H, C.byref(b)),              # not in arrival-ordered mode
      lambda: L.ep_ygo_alloc_record(e, C.byref(slot), C.byref(tick)),
      lambda: L.ep_ygo_record_done(e, 0),
      lambda: L.ep_ygo_put_arrival(e, recs.ctypes.data),
      lambda: L.ep_ygo_emit_arrival(e, sb.raw),
      lambda: L.ep_ygo_last_env_ids(e, C.byref(ids_p), C.byref(ids_n)),
      lambda: L.ep_ygo_set_packed(e, 1),
      lambda: L.ep_ygo_async_begin(e, 0),
      lambda: L.ep_ygo_async_begin(e, B + 1),
      lambda: L.ep_ygo_set_card_table(e, None, 5),
      lambda: L.ep_ygo_set_card_table(e, table, -1),
      lambda: L.ep_ygo_acquire_slot(e, None),
      lambda: L.ep_ygo_emit_record(C.byref(cfg), None, recs.ctypes.data),
      lambda: L.ep_ygo_emit_record(C.byref(cfg), sb.raw, None),
      lambda: L.ep_ygo_emit_record(None, sb.raw, recs.ctypes.data),
      lambda: L.ep_ygo_emit_record(C.byref(bad_cfg), sb.raw, recs.ctypes.data),
      lambda: L.ep_ygo_emit_record(C.byref(huge_cfg), sb.raw, recs.ctypes.data),
      lambda: 1 if L.ep_ygo_record_stride(None) < 0 else 0,
      lambda: 1 if L.ep_ygo_record_stride(C.byref(bad_cfg)) < 0 else 0,
      lambda: 1 if L.ep_ygo_record_used_bytes(C.byref(cfg), None) < 0 else 0,
      lambda: L.ep_ygo_history_push(C.byref(cfg), None, C.byref(hp), 11, b"b", None, 0),
      lambda: L.ep_ygo_history_push(C.byref(cfg), ring.ctypes.data, C.byref(hp), 11, None, None, 0),
      lambda: L.ep_ygo_history_push(C.byref(cfg), ring.ctypes.data, C.byref(hp), 999, b"b", None, 0),
      lambda: L.ep_ygo_encoder_create(C.byref(bad_cfg), C.byref(slot)),
      lambda: L.ep_ygo_encoder_create(None, C.byref(slot)),
      lambda: L.ep_ygo_ring_stats(None, None, None, None),
    ]
    recs_keep = recs.copy()
    for it in range(60):
      which = int(rng.integers(0, len(menu)))
      rc = menu[which]()
      assert rc != 0, ("an invalid call was accepted: menu item", which, "round", r, it)
      refused += 1
      if it % 6 == 0:
        assert recs.tobytes() == recs_keep.tobytes(), ("a refused call wrote into the caller's records", which)
        got = enc.featurize(recs, allow=allow)
        for k in range(len(clean)):
          assert np.asarray(got[k]).tobytes() == clean[k].tobytes(), ("after refused call: key", k, "menu item", which)
    enc.close()

    # the pool
    N = int(rng.choice([4, 64]))
    Bp = N if rng.integers(0, 2) else max(1, N // 2)
    pcfg = capi_ygo.default_config(max_batch=Bp, ring_depth=0)
    eng = ygo_synth.SynthEngine(pcfg, table, N, seed=r)
    pool = capi_ygo.YgoPoolC(pcfg, eng.engine(), N, Bp, table, len(table), num_threads=2)
    pool.set_recv_timeout(20000)
    ph = pool.handle
    ids = np.arange(N, dtype=np.int32)
    act = np.zeros(N, np.int32)
    far = ids + 10 * N
    dup = np.array([0, 1, 0][: min(3, N)] if N >= 2 else [0], np.int32)
    engine = eng.engine()
    null_engine = capi_ygo.Engine()
    out_pool = C.c_void_p()
    pmenu = [
      lambda: L.ep_ygo_pool_send(ph, N + 1, ids.ctypes.data, act.ctypes.data),
      lambda: L.ep_ygo_pool_send(ph, -1, ids.ctypes.data, act.ctypes.data),
      lambda: L.ep_ygo_pool_send(ph, N, None, act.ctypes.data),
      lambda: L.ep_ygo_pool_send(ph, N, ids.ctypes.data, None),
      lambda: L.ep_ygo_pool_send(ph, N, far.ctypes.data, act.ctypes.data),
      lambda: L.ep_ygo_pool_send(ph, dup.size, dup.ctypes.data, act.ctypes.data) if N >= 2 else 1,
      lambda: L.ep_ygo_pool_reset(ph, N + 1, ids.ctypes.data),
      lambda: L.ep_ygo_pool_reset(ph, 1, None),
      lambda: L.ep_ygo_pool_reset(ph, 1, far.ctypes.data),
      lambda: L.ep_ygo_pool_recv(ph, H, None),
      lambda: L.ep_ygo_pool_recv(ph, 9, C.byref(b)),
      lambda: L.ep_ygo_pool_last_env_ids(ph, None, C.byref(ids_n)),
      lambda: L.ep_ygo_pool_create(C.byref(pcfg), C.byref(null_engine), N, Bp, 1, -1, table, len(table), C.byref(out_pool)),
      lambda: L.ep_ygo_pool_create(C.byref(pcfg), C.byref(engine), 0, 0, 1, -1, table, len(table), C.byref(out_pool)),
      lambda: L.ep_ygo_pool_create(C.byref(pcfg), C.byref(engine), N, N + 1, 1, -1, table, len(table), C.byref(out_pool)),
      lambda: L.ep_ygo_pool_create(C.byref(pcfg), None, N, Bp, 1, -1, table, len(table), C.byref(out_pool)),
      lambda: L.ep_ygo_pool_create(C.byref(pcfg), C.byref(engine), N, Bp, 1, -1, table, len(table), None),
    ]
    pool.reset(ids)
    in_ring = N
    for it in range(30):
      which = int(rng.integers(0, len(pmenu)))
      rc = pmenu[which]()
      assert rc != 0, ("an invalid pool call was accepted: menu item", which, "round", r, it)
      refused += 1
      if in_ring >= Bp:
        out = pool.recv(allow=(capi.EP_ERR_ACTION_CHECK,))
        got = np.array(out[0])
        assert got.size == Bp and len(set(got.tolist())) == Bp
        in_ring -= Bp
        pool.send(got, act[: got.size])
        in_ring += got.size
    assert L.ep_ygo_pool_set_env_id_offset(ph, 5) != 0   # too late
    refused += 1
    pool.close()
    eng.close()
    print(f"ok {r}: refused so far {refused}", flush=True)
  print("ygo ABI fuzz passed:", refused, "calls refused")


if __name__ == "__main__":
  main()
