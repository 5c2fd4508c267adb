#!/usr/bin/env python3
"""GPU box: the Dummy C ABI under calls it must refuse - wrong array counts, null pointers, dims that
disagree / are negative / exceed the pool, unknown memory kinds, rollouts that do not fit, resets of
envs the pool does not own - interleaved with valid lockstep steps that are compared with the
oracle: a refused call must leave the pool exactly as it was. Pointers that are passed are always
valid (a wild pointer is the caller's); what is fuzzed is everything the library can check."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rounds=int(os.environ.get("FUZZ_ROUNDS", "30"))):
  from envpool_amd import capi
  from oracle.dummy_oracle import DummyOracle

  rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
  L = capi.lib()
  L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
  L.hipFree.argtypes = [C.c_void_p]
  refused = accepted = steps = 0
  for r in range(rounds):
    N = int(rng.choice([1, 7, 64, 256, 1000, 4096]))
    sync = bool(rng.integers(0, 2)) or N == 1
    B = N if sync else int(rng.integers(1, N))
    seed = int(rng.integers(1, 50))
    pool = capi.DummyPoolC(num_envs=N, batch_size=B, seed=seed)
    pool.set_recv_timeout(20000)
    orc = DummyOracle(num_envs=N, batch_size=B, seed=seed)
    h = pool.handle
    ids = np.arange(N, dtype=np.int32)
    la = np.full((N, 6), 1.5)
    zeros = np.zeros(N, np.int32)
    big = np.zeros(4 * N + 8, np.int32)
    dev = C.c_void_p()
    assert L.hipMalloc(C.byref(dev), 64 * (N + 8)) == 0  # a valid device buffer for the device-memory refusals

    def valid_step(first=False):
      # one batch received and compared, then sent back (a batch is always there: everything that
      # was received has been sent again, and num_envs >= batch_size)
      nonlocal steps
      if first:
        pool._reset(ids); orc._reset(ids)
      got, want = pool._recv(), orc._recv()
      for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11):
        assert np.asarray(got[k]).tobytes() == np.asarray(want[k]).tobytes(), ("after a refused call", r, k)
      e = np.array(got[0])
      act = [e, e, la[: e.size], zeros[: e.size], zeros[: e.size]]
      pool._send(act); orc._send(act)
      steps += e.size

    def call_send(n_arrays, arrays, dims, mem):
      ptrs = (C.c_void_p * max(1, len(arrays)))(*arrays) if arrays is not None else None
      dd = (C.c_int64 * max(1, len(dims)))(*dims) if dims is not None else None
      return L.ep_pool_send(h, n_arrays, ptrs, dd, mem)

    host = [ids.ctypes.data, ids.ctypes.data, la.ctypes.data, zeros.ctypes.data, zeros.ctypes.data]
    devp = [dev.value] * 5
    good = [N] * 5 if sync else [B] * 5
    menu = [
      lambda: call_send(int(rng.choice([0, 1, 3, 4, 6, -1, 100])), host, good, capi.EP_MEM_HOST),
      lambda: call_send(5, None, good, capi.EP_MEM_HOST),
      lambda: call_send(5, host, None, capi.EP_MEM_HOST),
      lambda: call_send(5, [0] + host[1:], good, capi.EP_MEM_HOST),
      lambda: call_send(5, host[:2] + [0] + host[3:], good, capi.EP_MEM_HOST),
      lambda: call_send(5, host, [good[0], good[0] - 1] + good[2:], capi.EP_MEM_HOST) if good[0] > 1 else 1,  # fewer player rows than envs
      lambda: call_send(5, host, [good[0], good[1], good[2] - 1 if good[2] > 1 else 2] + good[3:], capi.EP_MEM_HOST),
      lambda: call_send(5, host, [-1] * 5, capi.EP_MEM_HOST),
      lambda: call_send(5, [big.ctypes.data] * 2 + [la.ctypes.data] + [big.ctypes.data] * 2, [2 * N + 3] * 5, capi.EP_MEM_HOST),
      lambda: call_send(5, host, good, int(rng.choice([2, 3, -1, 99]))),
      lambda: call_send(5, devp, [-5] * 5, capi.EP_MEM_DEVICE),
      lambda: call_send(5, devp, [2 * N + 1] * 5, capi.EP_MEM_DEVICE),
      lambda: call_send(5, [0] * 5, good, capi.EP_MEM_DEVICE),
      lambda: call_send(5, [dev.value, dev.value, dev.value + 4, dev.value, dev.value], good, capi.EP_MEM_DEVICE),  # list_action off its alignment
      lambda: L.ep_pool_reset(h, ids.ctypes.data, N + 1, capi.EP_MEM_HOST),
      lambda: L.ep_pool_reset(h, ids.ctypes.data, -3, capi.EP_MEM_HOST),
      lambda: L.ep_pool_reset(h, None, 1, capi.EP_MEM_HOST),
      lambda: L.ep_pool_reset(h, (big + N).ctypes.data, 1, capi.EP_MEM_HOST),   # an env id outside the pool
      lambda: L.ep_pool_reset(h, ids.ctypes.data, 1, 7),
      lambda: L.ep_pool_recv(h, 5, C.byref(capi.Batch())),
      lambda: L.ep_pool_recv(h, capi.EP_MEM_HOST, None),
      lambda: L.ep_pool_checkpoint_restore(h, None),
      lambda: L.ep_pool_checkpoint_save(h, None),
    ]
    def call_rollout(T, n_arrays, arrays, dims, strides):
      ptrs = (C.c_void_p * 5)(*arrays)
      return L.ep_pool_rollout(h, T, n_arrays, ptrs, (C.c_int64 * 5)(*dims), (C.c_int64 * 5)(*strides))

    Q = L.ep_pool_ring_depth(h)
    full = [N] * 5
    menu += [
      lambda: call_rollout(0, 5, devp, full, [0] * 5),
      lambda: call_rollout(Q + 1, 5, devp, full, [0] * 5),
      lambda: call_rollout(1, 4, devp, full, [0] * 5),
      lambda: call_rollout(1, 5, devp, [N - 1 if N > 1 else 2] * 5, [0] * 5),
      lambda: call_rollout(1, 5, devp, full, [N, 0, 0, 0, 0]),            # env_id must not move between steps
      lambda: call_rollout(1, 5, [0] + devp[1:], full, [0] * 5),
      lambda: call_rollout(1, 5, [dev.value, dev.value + 2, dev.value, dev.value, dev.value], full, [0] * 5),
    ]
    # round 3's entry points: the stream-ordered step and its halves, the error poll, the adopted stream,
    # the terminated helper, the blocking step
    def call_sos(n_arrays, arrays, dims, out=True, half=False):
      ptrs = (C.c_void_p * max(1, len(arrays)))(*arrays) if arrays is not None else None
      dd = (C.c_int64 * max(1, len(dims)))(*dims) if dims is not None else None
      if half:
        return L.ep_pool_send_on_stream(h, None, n_arrays, ptrs, dd)
      return L.ep_pool_step_on_stream(h, None, n_arrays, ptrs, dd, C.byref(capi.Batch()) if out else None)

    one = (C.c_void_p * 1)(dev.value)
    menu += [
      lambda: call_sos(int(rng.choice([0, 4, 6, -1])), devp, good),
      lambda: call_sos(5, devp, good, out=False),
      lambda: call_sos(5, [0] * 5, good),
      lambda: call_sos(5, devp, [2 * N + 1] * 5),
      lambda: call_sos(5, devp, [-2] * 5, half=True),
      lambda: call_sos(5, None, good, half=True),
      lambda: L.ep_pool_step(h, 3, (C.c_void_p * 5)(*devp), (C.c_int64 * 5)(*good), C.byref(capi.Batch())),
      lambda: L.ep_pool_step(h, 5, (C.c_void_p * 5)(*devp), (C.c_int64 * 5)(*good), None),
      lambda: L.ep_pool_step_end(h, C.byref(capi.Batch())),                      # no begin before it
      lambda: L.ep_pool_poll_errors(None),
      lambda: L.ep_pool_set_stream(None, None, 0),
      lambda: L.ep_terminated_many(-1, one, one, one, (C.c_int64 * 1)(4), -1, None),
      lambda: L.ep_terminated_many(1, None, one, one, (C.c_int64 * 1)(4), -1, None),
      lambda: L.ep_terminated_many(1, one, one, one, (C.c_int64 * 1)(-4), -1, None),
      lambda: L.ep_terminated_many(1, one, (C.c_void_p * 1)(None), one, (C.c_int64 * 1)(4), -1, None),
    ]
    # caller-laid-out result memory (ep_host_blocks_*, ep_pool_host_target): what is refused, and valid
    # targets now and then - the batches that land in them are compared like any other
    rowb = [4, 4, 4, 1, 4, 4, 4, 1, 40, 1, 4]
    t_offs, at = [], 0
    for b_ in rowb:
      t_offs.append(at)
      at = (at + b_ * (B + 16) + 255) // 256 * 256   # (room for a row offset of 16 in every key)
    set_a, set_b, blk = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert L.ep_host_blocks_create(at + 4096, C.byref(set_a)) == 0 and L.ep_host_blocks_create(at + 4096, C.byref(set_b)) == 0
    assert L.ep_host_blocks_take(set_a, C.byref(blk)) == 0
    offs_ok = (C.c_size_t * 11)(*t_offs)
    menu += [
      lambda: L.ep_pool_host_target(h, set_b, blk, offs_ok, 0),                                  # a block of another set
      lambda: L.ep_pool_host_target(h, set_a, blk, None, 0),
      lambda: L.ep_pool_host_target(h, set_a, None, offs_ok, 0),
      lambda: L.ep_pool_host_target(None, set_a, blk, offs_ok, 0),
      lambda: L.ep_pool_host_target(h, set_a, blk, (C.c_size_t * 11)(*[o + 8 for o in t_offs]), 0),  # not 16-byte aligned
      lambda: L.ep_pool_host_target(h, set_a, blk, offs_ok, -1),
      lambda: L.ep_pool_host_target(h, set_a, blk, offs_ok, 4096 + 1),                            # past the end of the block
      lambda: L.ep_pool_host_target(h, set_a, blk, (C.c_size_t * 11)(*[o + (1 << 40) for o in t_offs]), 0),
      lambda: L.ep_host_blocks_give(set_b, blk),
      lambda: L.ep_host_blocks_give(set_a, None),
      lambda: L.ep_host_blocks_take(None, C.byref(C.c_void_p())),
      lambda: L.ep_host_blocks_create(0, C.byref(C.c_void_p())),
    ]
    if not sync:
      menu.append(lambda: call_rollout(1, 5, devp, full, [0] * 5))          # rollouts are for sync pools
    valid_step(first=True)
    for it in range(40):
      if rng.integers(0, 2):
        which = int(rng.integers(0, len(menu)))
        rc = menu[which]()
        assert rc != 0, ("an invalid call was accepted: menu item", which, "pool", N, B, "round", r, it)
        refused += 1
      if it % 13 == 5:  # valid calls that must change nothing: adopt the legacy stream and come back, poll with nothing flagged
        assert L.ep_pool_set_stream(h, None, 0) == 0 and L.ep_pool_poll_errors(h) == 0
        if it % 2:
          assert L.ep_pool_set_stream(h, None, 1) == 0
      if it % 7 == 3:  # the next batch to START goes into the block (row offset 0 or 16), or the target is cancelled again
        # a block of its own per target (several batches may be in flight), let go of at once: the pool's hold keeps it
        b2 = C.c_void_p()
        assert L.ep_host_blocks_take(set_a, C.byref(b2)) == 0
        assert L.ep_pool_host_target(h, set_a, b2, offs_ok, int(rng.choice([0, 16]))) == 0, L.ep_last_error()
        assert L.ep_host_blocks_give(set_a, b2) == 0
        if it % 2:
          assert L.ep_pool_host_target(h, None, None, None, 0) == 0
      valid_step()
    pool.close()
    assert L.ep_host_blocks_give(set_a, blk) == 0
    assert L.ep_host_blocks_destroy(set_a) == 0 and L.ep_host_blocks_destroy(set_b) == 0
    L.hipFree(dev)
    print(f"ok {r}: N={N} B={B} refused so far {refused}, env-steps compared {steps}", flush=True)
  print("dummy ABI fuzz passed:", refused, "calls refused,", steps, "env-steps compared with the oracle")


if __name__ == "__main__":
  main()
