#!/usr/bin/env python3
"""GPU box: randomised parity soak of ep_ygo_pool (sync and async pools, 1-16 workers, default and
random configs, partial sends, forced resets) over the deterministic stand-in engine of
tools/ygo_synth.c: every received row against the oracle's encoding of the state that engine
produced for that env at that point. Prints a line per pool; exits 1 on the first mismatch."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NK = 18


def _rows(out, j):
  return tuple(np.asarray(out[k][j]).tobytes() for k in range(NK))


def main(total=int(os.environ.get("SOAK_ROWS", "400000"))):
  from envpool_amd import capi, capi_ygo
  from oracle import ygo_oracle
  from tools import ygo_synth

  rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
  ok_rc = (capi.EP_ERR_ACTION_CHECK,)

  def expect(cfg, table, raws):
    arr = (capi_ygo.RawState * len(raws))()
    for i, r in enumerate(raws):
      C.memmove(C.byref(arr[i]), C.byref(r), C.sizeof(capi_ygo.RawState))
    return ygo_oracle.write_batch(cfg, table, arr, len(raws))

  def compare(tag, cfg, table, out, raws, ids):
    want, status = expect(cfg, table, raws)
    assert out[0].tolist() == [int(e) for e in ids], (tag, "ids")
    n_ok = 0
    for j in range(len(raws)):
      if status[j] == 0:
        if _rows(out, j) != _rows(want, j):
          bad = [k for k in range(NK) if np.asarray(out[k][j]).tobytes() != np.asarray(want[k][j]).tobytes()]
          print("MISMATCH", tag, "row", j, "env", int(ids[j]), "keys", bad)
          sys.exit(1)
        n_ok += 1
    return n_ok

  done, k, t0 = 0, 0, time.time()
  while done < total:
    k += 1
    kw = {} if k % 3 == 1 else dict(max_cards=int(rng.integers(20, 100)), max_options=int(rng.integers(2, 40)),
                                     n_history_actions=int(rng.integers(1, 33)), max_multi_select=int(rng.integers(1, 7)))
    N = int(rng.integers(1, 700))
    sync = bool(rng.integers(0, 2))
    B = N if sync else int(rng.integers(1, N + 1))
    threads = int(rng.integers(1, 17))
    seed = int(rng.integers(1, 1 << 30))
    cfg = capi_ygo.default_config(max_batch=B, ring_depth=0, **kw)
    table = ygo_synth.make_table(int(rng.integers(50, 2000)), seed=seed)
    try:
      eng = ygo_synth.SynthEngine(cfg, table, N, seed=seed)
      model = ygo_synth.EnginePoolModel(cfg, table, N, seed=seed)
      pool = capi_ygo.YgoPoolC(cfg, eng.engine(), N, B, table, len(table), num_threads=threads)
    except capi.EpError as ex:
      print("skip", kw, ex)
      continue
    pool.set_recv_timeout(60000)
    ids = np.arange(N, dtype=np.int32)
    rows = 0
    tag = (k, "sync" if sync else "async", N, B, threads, seed, kw)
    if sync:
      for it in range(int(rng.integers(3, 12))):
        sel = ids if it == 0 or rng.integers(0, 3) == 0 else rng.permutation(N)[: int(rng.integers(1, N + 1))].astype(np.int32)
        if it == 0 or rng.integers(0, 5) == 0:
          pool.reset(sel)
          raws = [model.advance(int(e)) for e in sel]
        else:
          act = rng.integers(0, 16, size=sel.size).astype(np.int32)
          pool.send(sel, act)
          raws = [model.advance(int(e), int(a)) for e, a in zip(sel, act)]
        out = pool.recv(allow=ok_rc)
        assert pool.last_env_ids().tolist() == sel.tolist(), (tag, "last_env_ids")
        rows += compare(tag, cfg, table, out, raws, sel)
    else:
      pending = {}
      pool.reset(ids)
      for e in ids:
        pending[int(e)] = model.advance(int(e))
      in_ring = N
      for it in range(int(rng.integers(3, 12))):
        if in_ring < B:
          break
        out = pool.recv(allow=ok_rc)
        got = out[0]
        in_ring -= B
        assert got.shape[0] == B and len(set(got.tolist())) == B, (tag, "batch shape")
        assert pool.last_env_ids().tolist() == got.tolist(), (tag, "last_env_ids")
        rows += compare(tag, cfg, table, out, [pending.pop(int(e)) for e in got], got)
        # send back a random part of what arrived (the rest of those envs idle from here on)
        back = got[: int(rng.integers(0, B + 1))] if rng.integers(0, 3) == 0 else got
        if back.size:
          act = rng.integers(0, 16, size=back.size).astype(np.int32)
          pool.send(back, act)
          for e, a in zip(back, act):
            pending[int(e)] = model.advance(int(e), int(a))
          in_ring += back.size
    pool.close()
    eng.close()
    done += rows
    print(f"ok {k}: {'sync ' if sync else 'async'} N={N} B={B} threads={threads} {kw or 'default'} rows={rows} total={done} t={time.time() - t0:.0f}s", flush=True)
  print("ygo pool soak passed:", done, "rows in", k, "pools")


if __name__ == "__main__":
  main()
