#!/usr/bin/env python3
"""GPU box: randomised parity soak of the Dummy pool against the oracle - random pool shapes (sync
and async, one to three players per env), random send sizes, host arrays through the C ABI.
Exits 1 on the first differing byte."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rounds=int(os.environ.get("SOAK_POOLS", "60"))):
  from envpool_amd import capi
  from envpool_amd.capi import DummyPoolC
  from oracle.dummy_oracle import DummyOracle
  from tests.util import assert_batches_equal, dummy_action

  rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
  t0 = time.time()
  steps_total = 0
  user = None  # (SOAK_STREAM) a stream of the caller's, kept for the whole run
  for r in range(rounds):
    n = int(rng.choice([1, 7, 64, 257, 4096, 5000, 20000, 65536]))
    P = int(rng.choice([1, 1, 1, 2, 3]))
    sync = bool(rng.random() < 0.5) or n == 1
    B = n if sync else int(rng.integers(1, n + 1))
    seed = int(rng.integers(1, 2000))
    kw = dict(num_envs=n, batch_size=B, seed=seed, max_num_players=P)  # default ring depth: the reference's queue size
    g = DummyPoolC(**kw)
    g.set_recv_timeout(0)
    o = DummyOracle(**{k: v for k, v in kw.items() if k != "ring_depth"})
    ids = np.arange(n, dtype=np.int32)
    # SOAK_MIXED=1: sends and recvs alternate between host and device memory (a device-memory send
    # may not repeat an env: it would be reported, not stepped twice)
    mixed = os.environ.get("SOAK_MIXED") == "1"

    def send(a):
      if mixed and len(set(np.asarray(a[0]).tolist())) == len(a[0]) and rng.random() < 0.3:
        g._send_via_device(a)
      else:
        g._send(a)
    g._reset(ids); o._reset(ids)
    outstanding = n
    steps = int(rng.integers(5, 40))
    for it in range(steps):
      while outstanding >= B:
        # (a fifth of the batches are received as device views: a batch that host-memory sends
        # put into a pinned block has to come back through the ring then)
        got, want = (g._recv_via_device() if mixed and rng.random() < 0.2 else g._recv()), o._recv()
        assert_batches_equal(got, want, f"pool {kw} it {it}")
        outstanding -= B
        last_ids = np.asarray(want[0])
      if P > 1:
        # multi-player: one action row per player the env reported in the last batch, the envs in
        # random order and a few of them twice (their items see the same row count, env.h:128-132)
        sel = rng.permutation(np.unique(last_ids)).astype(np.int32)
        if rng.random() < 0.5 and len(sel) < n:  # (a send carries at most num_envs items here)
          sel = np.concatenate([sel, rng.choice(sel, size=min(3, len(sel), n - len(sel)))]).astype(np.int32)
        penv = np.asarray(want[1]).astype(np.int32)
        a = dummy_action(sel, penv)
        send(a); o._send(a)
        outstanding += len(sel)
        steps_total += len(sel)
        continue
      elif sync:
        sel = ids if rng.random() < 0.8 else rng.permutation(n).astype(np.int32)
      else:
        # with repeats: an env named twice in one send is stepped twice, in order
        sel = rng.choice(last_ids, size=int(rng.integers(1, len(last_ids) + 1)), replace=True).astype(np.int32)
      penv = sel + (rng.random(len(sel)) < 0.1).astype(np.int32)
      a = dummy_action(sel, penv)
      if os.environ.get("SOAK_STREAM") == "1" and sync and P == 1 and rng.random() < 0.6:
        # SOAK_STREAM=1: a whole-pool step in stream order (ep_pool_step_on_stream): 0 - from a stream
        # of the caller's, ordered against the pool's own by events; 1 - with that stream, or the legacy
        # default stream, adopted by the pool; 2 - on the pool's own stream
        mode = int(rng.integers(0, 3))
        if user is None:
          user = C.c_void_p()
          g._L.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
          assert g._L.hipStreamCreate(C.byref(user)) == 0
        if mode != 1:
          assert g._L.ep_pool_set_stream(g.handle, None, 1) == 0
        st = user if mode == 0 or (mode == 1 and rng.random() < 0.5) else None
        got = g._step_on_stream_via_device(a, st, adopt=(mode == 1))
        o._send(a)
        want = o._recv()
        assert_batches_equal(got, want, f"pool {kw} it {it} stream-ordered mode {mode}")
        assert g.last_terminated is not None and g.last_terminated.tobytes() == (np.asarray(want[3]) & ~np.asarray(want[7])).astype(np.uint8).tobytes(), \
            f"pool {kw} it {it}: terminated beside the slot"
        last_ids = np.asarray(want[0])
        steps_total += len(sel)
        continue
      if mixed and not sync and len(sel) > 1 and len(set(sel.tolist())) == len(sel) and rng.random() < 0.3:
        # one send as two, the halves through different memories: both write rows of the same batch
        h = len(sel) // 2
        for part, dev_mem in ((slice(0, h), False), (slice(h, None), True)):
          pa = [a[0][part], a[1][part], a[2][part], a[3][part], a[4][part]]
          (g._send_via_device if dev_mem else g._send)(pa); o._send(pa)
      else:
        send(a); o._send(a)
      outstanding += len(sel)
      steps_total += len(sel)
    del g
    print(f"ok pool {r}: {kw} steps={steps} env-steps so far {steps_total} t={time.time() - t0:.0f}s", flush=True)
  print("dummy soak passed:", rounds, "pools,", steps_total, "env-steps")


if __name__ == "__main__":
  main()
