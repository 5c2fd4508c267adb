#!/usr/bin/env python3
"""YGOPro observation-encode throughput (GPU box). Prints one JSON object.

Workloads: BASELINE config 3 shape (n=1024 records per call, host records through the pinned
submit/collect pipeline) and a large device-resident batch for the kernel's roofline figure."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n_big=65536, n_cfg3=1024, iters=200, cpu_seconds=0.0, device=0):
  import numpy as np
  import torch

  from envpool_amd import capi, capi_ygo
  from tools import ygo_synth

  L = capi_ygo.lib()
  dev = f"cuda:{device}"
  cfg = capi_ygo.default_config(max_batch=n_big, ring_depth=2, device_id=device)
  table = ygo_synth.make_table(1000, seed=0)
  n_gen = 8192
  sb = ygo_synth.SynthBatch(cfg, table, n_gen, seed=0)
  t0 = time.perf_counter()
  recs, est = capi_ygo.emit_records(cfg, sb.raw, n_gen)
  emit_s = time.perf_counter() - t0
  stride = recs.shape[1]
  alg_per_rec = alg_bytes_per_record(cfg, recs)

  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  h = enc.handle
  reps = n_big // n_gen
  big = torch.from_numpy(np.tile(recs, (reps, 1))).to(dev)
  torch.cuda.synchronize()
  b = capi.Batch()
  ok = (0, capi.EP_ERR_ACTION_CHECK)
  for _ in range(3):
    assert L.ep_ygo_featurize(h, big.data_ptr(), n_big, capi.EP_MEM_DEVICE, capi.EP_MEM_DEVICE, C.byref(b)) in ok
  tot, cnt = C.c_double(), C.c_int64()
  # two batches in flight (ep_ygo_submit_device / ep_ygo_collect): the launch of one runs under
  # the kernel of the other
  def pipelined(iters=iters):
    assert L.ep_ygo_submit_device(h, big.data_ptr(), n_big) == 0
    for _ in range(iters - 1):
      assert L.ep_ygo_submit_device(h, big.data_ptr(), n_big) == 0
      assert L.ep_ygo_collect(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok
    assert L.ep_ygo_collect(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok

  # this kernel is bound by instruction issue, so it follows the shader clock, and that takes
  # ~15 ms of load to come up from idle (155 -> 112 -> 104 -> 97.5 us per batch over the first
  # 150 batches): warm up that long
  pipelined(150)
  w0 = time.perf_counter()
  pipelined()
  w1 = time.perf_counter()
  # kernel duration from HIP events in a pass of its own (the events sit between the kernels), the
  # same back-to-back schedule: kernels launched one by one after host waits run ~8 % slower
  L.ep_ygo_set_timing(h, 1)
  L.ep_ygo_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1)
  pipelined()
  L.ep_ygo_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1)
  L.ep_ygo_set_timing(h, 0)
  k_us = tot.value / max(1, cnt.value)
  big_res = {"records_per_launch": n_big, "kernel_us": k_us, "encodes_per_s_kernel": n_big / (k_us * 1e-6),
             "encodes_per_s_wall": n_big * iters / (w1 - w0), "alg_bytes_per_record": alg_per_rec,
             "achieved_GBps": alg_per_rec * n_big / (k_us * 1e-6) / 1e9, "frac_of_8TBps": alg_per_rec * n_big / (k_us * 1e-6) / 8e12}
  enc.close()

  # host records through the pinned pipeline (copy of batch k+1 under the kernel of batch k):
  # config 3 (1024 records per step) and the per-GPU share of config 5 (512-record batches)
  zero_copy = os.environ.get("EP_YGO_ZERO_COPY", "1") != "0"  # the library's default (ygo.hip)

  def pipeline(n_rec, steps=400):
    cfgp = capi_ygo.default_config(max_batch=n_rec, ring_depth=4, device_id=device)
    encp = capi_ygo.YgoEncoderC(cfgp, table, len(table))
    hp = encp.handle
    recp = np.ascontiguousarray(recs[:n_rec])
    for _ in range(8):
      L.ep_ygo_submit(hp, recp.ctypes.data, n_rec)
      L.ep_ygo_collect(hp, capi.EP_MEM_DEVICE, C.byref(b))
    L.ep_ygo_set_timing(hp, 1)
    L.ep_ygo_kernel_time_us(hp, C.byref(tot), C.byref(cnt), 1)
    w0 = time.perf_counter()
    inflight = 0
    for s in range(steps):
      assert L.ep_ygo_submit(hp, recp.ctypes.data, n_rec) == 0
      inflight += 1
      if inflight == 3:
        assert L.ep_ygo_collect(hp, capi.EP_MEM_DEVICE, C.byref(b)) in ok
        inflight -= 1
    while inflight:
      assert L.ep_ygo_collect(hp, capi.EP_MEM_DEVICE, C.byref(b)) in ok
      inflight -= 1
    w1 = time.perf_counter()
    L.ep_ygo_kernel_time_us(hp, C.byref(tot), C.byref(cnt), 1)
    # in-place form: the producers write records straight into the pinned slot
    # (ep_ygo_acquire_slot / ep_ygo_commit_slot), so the staging memcpy of submit disappears
    slot = C.c_void_p()
    for _ in range(4):  # fill every slot of the ring once
      assert L.ep_ygo_acquire_slot(hp, C.byref(slot)) == 0
      C.memmove(slot, recp.ctypes.data, recp.nbytes)
      assert L.ep_ygo_commit_slot(hp, n_rec) == 0
      assert L.ep_ygo_collect(hp, capi.EP_MEM_DEVICE, C.byref(b)) in ok
    i0 = time.perf_counter()
    inflight = 0
    for s in range(steps):
      assert L.ep_ygo_acquire_slot(hp, C.byref(slot)) == 0
      assert L.ep_ygo_commit_slot(hp, n_rec) == 0
      inflight += 1
      if inflight == 3:
        assert L.ep_ygo_collect(hp, capi.EP_MEM_DEVICE, C.byref(b)) in ok
        inflight -= 1
    while inflight:
      assert L.ep_ygo_collect(hp, capi.EP_MEM_DEVICE, C.byref(b)) in ok
      inflight -= 1
    i1 = time.perf_counter()
    inplace_us = (i1 - i0) / steps * 1e6
    encp.close()
    # the H2D copy alone, same bytes, pinned source: what the pipeline cannot go below
    src = torch.from_numpy(recp).pin_memory()
    dst = torch.empty_like(src, device=dev)
    for _ in range(4):
      dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(100):
      dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    copy_us = (time.perf_counter() - c0) / 100 * 1e6
    step_us = (w1 - w0) / steps * 1e6
    k_us = tot.value / max(1, cnt.value)
    return {"records_per_step": n_rec, "steps": steps, "encodes_per_s_wall_pcie_inclusive": n_rec * steps / (w1 - w0),
            "us_per_step_wall": step_us, "kernel_us": k_us, "h2d_copy_us": copy_us,
            "us_per_step_wall_in_place": inplace_us, "encodes_per_s_in_place_pcie_inclusive": n_rec / (inplace_us * 1e-6),
            "record_stride_bytes": int(stride), "zero_copy": zero_copy,
            # staged mode: how much of a step is the longer of its two overlapped legs. Zero-copy
            # mode has no copy leg; the staging copy of the same fixed-stride records is what a
            # staged pipeline cannot go below, so the ratio says what reading only the used bytes buys
            **({"step_vs_staging_copy": inplace_us / copy_us} if zero_copy else
               {"overlap_efficiency": max(copy_us, k_us) / inplace_us})}

  cfg3_res = pipeline(n_cfg3)
  cfg5_res = pipeline(512)

  res = {"large_batch": big_res, "config3_host_pipeline": cfg3_res, "config5_per_gpu_host_pipeline": cfg5_res,
         "host_emit_records_per_s_1core": n_gen / emit_s}
  try:
    res["config3_packed"] = run_config3_packed(device)
  except Exception as ex:  # noqa: BLE001
    res["config3_packed"] = {"error": repr(ex)}
  try:
    res["config5_async_ring"] = run_config5(device)
  except Exception as ex:  # noqa: BLE001
    res["config5_async_ring"] = {"error": repr(ex)}
  try:
    res["config3_pool"] = run_config3_pool(device)
  except Exception as ex:  # noqa: BLE001
    res["config3_pool"] = {"error": repr(ex)}
  try:
    res["config5_pool"] = run_config5_pool(device)
  except Exception as ex:  # noqa: BLE001
    res["config5_pool"] = {"error": repr(ex)}
  if cpu_seconds > 0:
    from oracle import ygo_oracle
    n_cpu = 2048
    c0 = time.perf_counter()
    done = 0
    while time.perf_counter() - c0 < cpu_seconds:
      ygo_oracle.write_batch(cfg, table, sb.raw, n_cpu)
      done += n_cpu
    res["cpu_oracle_encodes_per_s_1core"] = done / (time.perf_counter() - c0)
  return res


def run_config5(device=0, threads_list=(1, 4, 12, 16), n_envs=1024, batch=512, steps_per_env=40, engine_us=None):
  """BASELINE config 5 per GPU: async mode, 1024 envs, 512-record batches. Host producer threads (a
  stand-in duel engine spending `engine_us` per env-step, tools/ygo_synth.c) emit records in place
  into the encoder's arrival-ordered pinned ring; the batch is encoded (zero-copy PCIe reads) when
  its 512th record lands; one consumer receives the batches as device views. Per thread count:
  batches/s, and how much of the GPU side (kernel incl. its PCIe reads) is hidden behind emission.
  The duel threads are the critical path; the GPU side is exposed only where it holds them up (the
  time they wait for a free ring slot) and in the tail after the last record (its batch's kernel):
  hidden = 1 - (producer wait + tail) / GPU time. The producers-alone run (records into a private
  scratch record) gives the cost of the ring itself per record."""
  import threading

  from envpool_amd import capi, capi_ygo
  from tools import ygo_synth

  if engine_us is None:
    engine_us = float(os.environ.get("EP_YGO_ENGINE_US", "20"))
  L = capi_ygo.lib()
  cfg = capi_ygo.default_config(max_batch=batch, ring_depth=4, device_id=device)
  table = ygo_synth.make_table(1000, seed=0)
  sb = ygo_synth.SynthBatch(cfg, table, n_envs, seed=5)
  n_batches = n_envs * steps_per_env // batch
  tot, cnt = C.c_double(), C.c_int64()
  b = capi.Batch()
  ok = (0, capi.EP_ERR_ACTION_CHECK)
  out = {"envs": n_envs, "batch_size": batch, "engine_us_per_env_step": engine_us, "records": n_envs * steps_per_env,
         "note": "the box gives one GPU's job 16 host cores: 16 producer threads plus the consumer oversubscribe them",
         "by_threads": {}}
  for nt in threads_list:
    enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
    h = enc.handle
    assert L.ep_ygo_async_begin(h, batch) == 0
    # producers alone (records into a private scratch record, no ring, no GPU)
    s0 = time.perf_counter()
    rc, _ = ygo_synth.run_producers(h, cfg, sb.raw, n_envs, nt, steps_per_env, engine_us, sink_only=True)
    sink_s = time.perf_counter() - s0
    assert rc == 0
    L.ep_ygo_set_timing(h, 1)
    L.ep_ygo_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1)
    res = {}

    def produce():
      r = ygo_synth.run_producers(h, cfg, sb.raw, n_envs, nt, steps_per_env, engine_us)
      res.update(r=r, done_at=time.perf_counter())
    th = threading.Thread(target=produce)
    w0 = time.perf_counter()
    th.start()
    for _ in range(n_batches):
      assert L.ep_ygo_recv(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok
    last_recv = time.perf_counter()
    th.join()
    wall = last_recv - w0
    assert res["r"][0] == 0
    L.ep_ygo_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1)
    gpu_s = tot.value * 1e-6
    wait_s = C.c_double()
    L.ep_ygo_ring_stats(h, C.byref(wait_s), None, None)
    tail_s = max(0.0, last_recv - res["done_at"])
    hidden = max(0.0, min(1.0, 1.0 - (wait_s.value + tail_s) / gpu_s)) if gpu_s > 0 else None
    recs_n = n_envs * steps_per_env
    out["by_threads"][str(nt)] = {
      "batches_per_s": n_batches / wall, "encodes_per_s_pcie_inclusive": recs_n / wall,
      "wall_s": wall, "producers_alone_s": sink_s, "ring_overhead_us_per_record": (wall - sink_s) / recs_n * 1e6,
      "gpu_kernel_s": gpu_s, "kernel_us_per_batch": tot.value / max(1, cnt.value),
      "emit_us_per_record": res["r"][1] / recs_n * 1e6, "producer_wait_for_slot_s": wait_s.value, "tail_s": tail_s,
      "gpu_time_hidden_behind_emission": hidden}
    enc.close()
  return out

def run_config5_pool(device=0, threads_list=(4, 16), n_envs=1024, batch=512, batches=120, engine_us=None):
  """BASELINE config 5 per GPU as a POOL (ep_ygo_pool, envpool_amd/csrc/ygo_pool.hip): async mode,
  1024 envs, batches of 512, the stand-in duel engine behind the ep_ygo_engine callbacks. The loop
  is the user's: recv a batch (device views), send an action for every env in it (whose ids the
  pool hands over from host memory, ep_ygo_pool_last_env_ids; EP_YGO_IDS_FROM_DEVICE=1: read back
  from the batch's info:env_id instead, 2 KiB). Reports batches/s per worker-thread count."""
  import numpy as np

  from envpool_amd import capi, capi_ygo
  from tools import ygo_synth

  if engine_us is None:
    engine_us = float(os.environ.get("EP_YGO_ENGINE_US", "20"))
  L = capi_ygo.lib()
  hip = L  # dlsym on the library's handle also finds its dependency libamdhip64's symbols
  hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
  table = ygo_synth.make_table(1000, seed=0)
  out = {"envs": n_envs, "batch_size": batch, "engine_us_per_env_step": engine_us, "by_threads": {}}
  ok = (0, capi.EP_ERR_ACTION_CHECK)
  for nt in threads_list:
    cfg = capi_ygo.default_config(max_batch=batch, ring_depth=0, device_id=device)
    eng = ygo_synth.SynthEngine(cfg, table, n_envs, seed=5, engine_us=engine_us)
    eng.set_replay()  # generating a synthetic state costs more than the 20 us "engine": keep it out
    pool = capi_ygo.YgoPoolC(cfg, eng.engine(), n_envs, batch, table, len(table), num_threads=nt)
    h = pool.handle
    pool.set_recv_timeout(60000)
    ids = np.arange(n_envs, dtype=np.int32)
    act = np.zeros(batch, dtype=np.int32)
    got = np.zeros(batch, dtype=np.int32)
    b = capi.Batch()
    pool.reset(ids)

    from_device = os.environ.get("EP_YGO_IDS_FROM_DEVICE") == "1"
    idp, idn = C.POINTER(C.c_int32)(), C.c_int32(0)

    def one():
      assert L.ep_ygo_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok
      if from_device:
        assert hip.hipMemcpy(got.ctypes.data, b.arrays[0].data, 4 * batch, 2) == 0  # 2 = device to host
        assert L.ep_ygo_pool_send(h, batch, got.ctypes.data, act.ctypes.data) == 0
      else:
        assert L.ep_ygo_pool_last_env_ids(h, C.byref(idp), C.byref(idn)) == 0 and idn.value == batch
        assert L.ep_ygo_pool_send(h, batch, idp, act.ctypes.data) == 0
    for _ in range(10):
      one()
    t0 = time.perf_counter()
    for _ in range(batches):
      one()
    dt = time.perf_counter() - t0
    for _ in range(n_envs // batch):  # drain
      assert L.ep_ygo_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok
    out["by_threads"][str(nt)] = {"batches_per_s": batches / dt, "env_steps_per_s": batches * batch / dt,
                                  "us_per_batch": dt / batches * 1e6}
    pool.close()
    eng.close()
  return out


def run_config3_pool(device=0, threads_list=(4, 0), n_envs=1024, steps=150, engine_us=0.0):
  """BASELINE config 3 as a POOL (ep_ygo_pool, sync: batch_size = num_envs = 1024): one step =
  send 1024 actions, the workers run the (stand-in, here free) engine and emit the records, the last
  one commits the packed slot, recv waits for the encode. PCIe-inclusive, emission included."""
  import numpy as np

  from envpool_amd import capi, capi_ygo
  from tools import ygo_synth

  L = capi_ygo.lib()
  table = ygo_synth.make_table(1000, seed=0)
  out = {"envs": n_envs, "engine_us_per_env_step": engine_us, "by_threads": {}}
  ok = (0, capi.EP_ERR_ACTION_CHECK)
  for nt in threads_list:
    if nt == 0:  # the pool's default for one GPU's share of a node (16 CPUs), also where the job sees the whole machine
      nt = 0 if len(os.sched_getaffinity(0)) <= 16 else 15
    cfg = capi_ygo.default_config(max_batch=n_envs, ring_depth=0, device_id=device)
    eng = ygo_synth.SynthEngine(cfg, table, n_envs, seed=5, engine_us=engine_us)
    eng.set_replay()
    pool = capi_ygo.YgoPoolC(cfg, eng.engine(), n_envs, n_envs, table, len(table), num_threads=nt)
    h = pool.handle
    pool.set_recv_timeout(60000)
    ids = np.arange(n_envs, dtype=np.int32)
    act = np.zeros(n_envs, dtype=np.int32)
    b = capi.Batch()
    pool.reset(ids)
    assert L.ep_ygo_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok
    ts = []
    mem = capi.EP_MEM_HOST if os.environ.get("EP_YGO_HOST_OUT") == "1" else capi.EP_MEM_DEVICE  # host arrays out: 6.9 MB per step over PCIe
    for i in range(steps + 20):
      t0 = time.perf_counter()
      assert L.ep_ygo_pool_send(h, n_envs, ids.ctypes.data, act.ctypes.data) == 0
      assert L.ep_ygo_pool_recv(h, mem, C.byref(b)) in ok
      if mem == capi.EP_MEM_HOST:
        L.ep_batch_release(C.byref(b))
      if i >= 20:
        ts.append(time.perf_counter() - t0)
    ts.sort()
    # (0 asks for the pool's default: one worker per usable CPU but one, the receiving thread steps envs too)
    out["by_threads"][str(pool.num_threads)] = {"us_per_step_median": ts[len(ts) // 2] * 1e6, "env_steps_per_s": n_envs / ts[len(ts) // 2]}
    pool.close()
    eng.close()
  return out


def run_config3_packed(device=0, n_envs=1024, steps_per_env=100, threads=4):
  """BASELINE config 3 (1024 envs, sync: batch = 1024) with host records in the PACKED form:
  producer threads hand finished records (emitted beforehand: the duel engine and the emitter are
  not what is measured here) to the arrival-ordered ring, each batch reaches the device as one
  copy-engine transfer of the bytes in use, the encode kernel reads HBM. PCIe-inclusive."""
  import threading

  from envpool_amd import capi, capi_ygo
  from tools import ygo_synth

  L = capi_ygo.lib()
  cfg = capi_ygo.default_config(max_batch=n_envs, ring_depth=4, device_id=device)
  table = ygo_synth.make_table(1000, seed=0)
  sb = ygo_synth.SynthBatch(cfg, table, n_envs, seed=0)
  recs, _ = capi_ygo.emit_records(cfg, sb.raw, n_envs)
  used = sum(L.ep_ygo_record_used_bytes(C.byref(cfg), recs[i].ctypes.data) for i in range(n_envs))
  tot, cnt = C.c_double(), C.c_int64()
  b = capi.Batch()
  ok = (0, capi.EP_ERR_ACTION_CHECK)
  out = {"envs": n_envs, "batch_size": n_envs, "producer_threads": threads, "bytes_per_record_packed": used / n_envs,
         "bytes_per_record_fixed_stride": int(recs.shape[1])}
  # (1) the pipeline by itself: slots already filled in the packed form (as config3_host_pipeline's
  # in-place figure has them filled at the fixed stride), commit + collect with 3 batches in flight
  import numpy as np
  lens = np.array([L.ep_ygo_record_used_bytes(C.byref(cfg), recs[i].ctypes.data) for i in range(n_envs)], np.int64)
  offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint32)
  packed_buf = np.concatenate([recs[i, :lens[i]] for i in range(n_envs)])
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  h = enc.handle
  slot = C.c_void_p()
  for _ in range(4):
    assert L.ep_ygo_acquire_slot(h, C.byref(slot)) == 0
    C.memmove(slot, packed_buf.ctypes.data, packed_buf.nbytes)
    assert L.ep_ygo_commit_slot_packed(h, n_envs, offs.ctypes.data, int(used)) == 0
    assert L.ep_ygo_collect(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok
  L.ep_ygo_set_timing(h, 1)
  L.ep_ygo_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1)
  steps = 400
  i0 = time.perf_counter()
  inflight = 0
  for _ in range(steps):
    assert L.ep_ygo_acquire_slot(h, C.byref(slot)) == 0
    assert L.ep_ygo_commit_slot_packed(h, n_envs, offs.ctypes.data, int(used)) == 0
    inflight += 1
    if inflight == 3:
      assert L.ep_ygo_collect(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok
      inflight -= 1
  while inflight:
    assert L.ep_ygo_collect(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok
    inflight -= 1
  i1 = time.perf_counter()
  L.ep_ygo_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1)
  out["pipeline_slots_prefilled"] = {"us_per_step_wall": (i1 - i0) / steps * 1e6,
                                     "encodes_per_s_pcie_inclusive": n_envs * steps / (i1 - i0),
                                     "kernel_us": tot.value / max(1, cnt.value),
                                     "copy_GBps": (used + 4096) / ((i1 - i0) / steps) / 1e9}
  enc.close()
  # (2) with producer threads handing finished records over
  for name, packed, zero_copy, mode in (("packed_copy_engine", 1, 0, ygo_synth.PACKED_PUT),):
    enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
    h = enc.handle
    assert L.ep_ygo_set_zero_copy(h, zero_copy) == 0
    assert L.ep_ygo_async_begin(h, n_envs) == 0
    assert L.ep_ygo_set_packed(h, packed) == 0
    L.ep_ygo_set_timing(h, 1)
    L.ep_ygo_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1)
    res = {}
    th = threading.Thread(target=lambda: res.update(r=ygo_synth.run_producers(h, cfg, sb.raw, n_envs, threads, steps_per_env, 0.0,
                                                                             mode=mode, pre_records=recs)))
    w0 = time.perf_counter()
    th.start()
    for _ in range(steps_per_env):
      assert L.ep_ygo_recv(h, capi.EP_MEM_DEVICE, C.byref(b)) in ok
    wall = time.perf_counter() - w0
    th.join()
    assert res["r"][0] == 0
    L.ep_ygo_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1)
    wait_s = C.c_double()
    L.ep_ygo_ring_stats(h, C.byref(wait_s), None, None)
    out[name] = {"us_per_step_wall": wall / steps_per_env * 1e6, "encodes_per_s_pcie_inclusive": n_envs * steps_per_env / wall,
                 "kernel_us": tot.value / max(1, cnt.value), "producer_put_us_per_record": res["r"][1] / (n_envs * steps_per_env) * 1e6,
                 "producer_wait_for_slot_s": wait_s.value}
    enc.close()
  return out


def alg_bytes_per_record(cfg, recs):
  """Algorithmic bytes per record: what the kernel must read (header + history + listed cards +
  options actually present) and write (every output key), averaged over `recs`."""
  import numpy as np
  hdr = recs[:, :64]
  loc = hdr[:, 48:62].reshape(-1, 2, 7).astype(np.int64)
  revealed_any = hdr[:, 5].astype(bool)
  listed = loc[:, 0, :].sum(1) + loc[:, 1, 2:6].sum(1) + np.where(revealed_any, loc[:, 1, 1], 0)
  n_opt = np.minimum(hdr[:, 3].astype(np.int64), cfg.max_options)
  nf = 9 + 2 * cfg.max_multi_select
  read_b = 64 + cfg.n_history_actions * nf + listed * 16 + n_opt * (12 + 4 * cfg.max_multi_select)
  read_b = np.where(hdr[:, 3] == 0, 64, read_b)
  write_b = 2 * cfg.max_cards * 39 + 8 + cfg.max_options * nf + cfg.n_history_actions * nf + 4 * 4 + (4 + 4 + 4 + 1 + 4 + 4 + 4 + 1) \
      + cfg.max_options * cfg.max_multi_select * 2 + cfg.max_options  # (+ aux:h_card_ids, aux:action_mask)
  return float(read_b.mean() + write_b)


def run_shard(device, barrier, n_big=65536, n_step=512, iters=200, steps=400):
  """One rank's share of a multi-GPU run (envs shard by env_id, no exchange): the device-resident
  large batch and the per-GPU shape of BASELINE config 5 (512 host records per step through the
  pinned pipeline), each timed between barriers. Returns the two wall times; the caller takes the
  maximum over ranks."""
  import numpy as np
  import torch

  from envpool_amd import capi, capi_ygo
  from tools import ygo_synth

  L = capi_ygo.lib()
  dev = f"cuda:{device}"
  cfg = capi_ygo.default_config(max_batch=n_big, ring_depth=2, device_id=device)
  table = ygo_synth.make_table(1000, seed=0)
  n_gen = 8192
  sb = ygo_synth.SynthBatch(cfg, table, n_gen, seed=0)
  recs, _ = capi_ygo.emit_records(cfg, sb.raw, n_gen)
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  big = torch.from_numpy(np.tile(recs, (n_big // n_gen, 1))).to(dev)
  torch.cuda.synchronize()
  b = capi.Batch()
  ok = (0, capi.EP_ERR_ACTION_CHECK)
  for _ in range(150):  # bring the shader clock up (see run())
    assert L.ep_ygo_featurize(enc.handle, big.data_ptr(), n_big, capi.EP_MEM_DEVICE, capi.EP_MEM_DEVICE, C.byref(b)) in ok
  barrier()
  t0 = time.perf_counter()
  assert L.ep_ygo_submit_device(enc.handle, big.data_ptr(), n_big) == 0
  for _ in range(iters - 1):
    assert L.ep_ygo_submit_device(enc.handle, big.data_ptr(), n_big) == 0
    assert L.ep_ygo_collect(enc.handle, capi.EP_MEM_DEVICE, C.byref(b)) in ok
  assert L.ep_ygo_collect(enc.handle, capi.EP_MEM_DEVICE, C.byref(b)) in ok
  barrier()
  t_big = time.perf_counter() - t0
  # this rank's kernel duration (HIP events, a pass of its own outside the timed region): the roofline figure
  tot, cnt = C.c_double(), C.c_int64()
  L.ep_ygo_set_timing(enc.handle, 1)
  L.ep_ygo_kernel_time_us(enc.handle, C.byref(tot), C.byref(cnt), 1)
  assert L.ep_ygo_submit_device(enc.handle, big.data_ptr(), n_big) == 0
  for _ in range(99):
    assert L.ep_ygo_submit_device(enc.handle, big.data_ptr(), n_big) == 0
    assert L.ep_ygo_collect(enc.handle, capi.EP_MEM_DEVICE, C.byref(b)) in ok
  assert L.ep_ygo_collect(enc.handle, capi.EP_MEM_DEVICE, C.byref(b)) in ok
  L.ep_ygo_kernel_time_us(enc.handle, C.byref(tot), C.byref(cnt), 1)
  L.ep_ygo_set_timing(enc.handle, 0)
  big_kernel_us = tot.value / max(1, cnt.value)
  alg_per_rec = alg_bytes_per_record(cfg, recs)
  enc.close()
  del big

  cfgp = capi_ygo.default_config(max_batch=n_step, ring_depth=4, device_id=device)
  encp = capi_ygo.YgoEncoderC(cfgp, table, len(table))
  hp = encp.handle
  recp = np.ascontiguousarray(recs[:n_step])
  slot = C.c_void_p()
  for _ in range(4):
    assert L.ep_ygo_acquire_slot(hp, C.byref(slot)) == 0
    C.memmove(slot, recp.ctypes.data, recp.nbytes)
    assert L.ep_ygo_commit_slot(hp, n_step) == 0
    assert L.ep_ygo_collect(hp, capi.EP_MEM_DEVICE, C.byref(b)) in ok
  barrier()
  t0 = time.perf_counter()
  inflight = 0
  for _ in range(steps):
    assert L.ep_ygo_acquire_slot(hp, C.byref(slot)) == 0
    assert L.ep_ygo_commit_slot(hp, n_step) == 0
    inflight += 1
    if inflight == 3:
      assert L.ep_ygo_collect(hp, capi.EP_MEM_DEVICE, C.byref(b)) in ok
      inflight -= 1
  while inflight:
    assert L.ep_ygo_collect(hp, capi.EP_MEM_DEVICE, C.byref(b)) in ok
    inflight -= 1
  barrier()
  t_step = time.perf_counter() - t0
  encp.close()
  # config 5 as a whole on this rank: 4 stand-in duel threads feeding the arrival-ordered ring
  ring = run_config5(device, threads_list=(4,), steps_per_env=20)["by_threads"]["4"]
  return {"large_wall_s": t_big, "large_records": n_big * iters, "pipeline_wall_s": t_step, "pipeline_records": n_step * steps,
          "records_per_launch": n_big, "records_per_step": n_step, "ring_4_threads": ring,
          "large_kernel_us": big_kernel_us, "alg_bytes_per_record": alg_per_rec}

if __name__ == "__main__":
  print(json.dumps(run(cpu_seconds=float(os.environ.get("EP_YGO_CPU_SECONDS", "3")))))
