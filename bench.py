#!/usr/bin/env python3
"""bench.py — env-steps/s of the Dummy-v0 lockstep reset/step + StateBuffer path.

Workload at N=1 (BASELINE.json configs[1]): Dummy-v0, num_envs=65536, seed=42, sync mode,
everything resident on one MI355X. A "step" is one lockstep step of ALL envs of a rank:
65 536 action rows read (env_id + players.env_id + 6-lane list_action), 65 536 env states
advanced with auto-reset, one full StateBuffer (11 fixed-shape keys) written into the ring.

Timed mode: the fused lockstep loop (`ep_pool_rollout`, T steps per launch, a distinct action
block per step already resident in HBM, results into a 2T-deep StateBuffer ring and handed out
by `ep_pool_recv` as device views). The one-launch-per-step mode (`ep_pool_send` +
`ep_pool_recv`) is measured next to it and reported under "per_step_launch".

Multi-GPU (--gpus N, launched by torch.distributed.run): envs shard by env_id, rank r owns
global ids [r*65536, (r+1)*65536) (seed_ = seed + GLOBAL id); no data-path collective; every
rollout the per-episode (return, length) stats are all-gathered over RCCL ("weak" scaling).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NUM_ENVS = 65536
SEED = 42
T_FUSED = int(os.environ.get("EP_BENCH_T", "128"))  # lockstep steps per rollout launch
# Algorithmic bytes. One-launch-per-step kernel, per env-step: 56 B action read
# (env_id 4 + players.env_id 4 + list_action 48; players.action / players.id are never read by
# DummyEnv::Step) + 8 B env state read + 8 B write + 71 B StateBuffer row write = 143 B.
# Fused rollout kernel: env_id and the env state stay in registers for the T steps of a launch,
# so per env-step only 52 B (players.env_id + list_action) + 71 B = 123 B move, plus 20 B per
# env per launch (env_id 4 + state 8 read, state 8 write).
ALG_BYTES_PER_ENV_STEP = 143
ALG_ROLLOUT_PER_ENV_STEP = 123
ALG_ROLLOUT_PER_ENV_LAUNCH = 20
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


class NodeBarrier:
  """Barrier for the ranks of ONE node: an 8-byte generation counter per rank, each on its own
  cache line of a shared-memory file; a rank publishes its generation and spins until every
  rank's has reached it. Set up (file name, creation) over the process group's host backend."""

  def __init__(self, rank, world, dist):
    import mmap
    import tempfile
    import numpy as np
    self.rank, self.world, self.dist, self.gen = rank, world, dist, 0
    names = [None]
    if rank == 0:
      d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
      fd, path = tempfile.mkstemp(prefix="envpool_amd_bench_", dir=d)
      os.write(fd, b"\0" * (64 * world))
      os.close(fd)
      names[0] = path
    dist.broadcast_object_list(names, src=0)
    self.path = names[0]
    self.f = open(self.path, "r+b")
    self.mm = mmap.mmap(self.f.fileno(), 64 * world)
    self.slots = np.frombuffer(self.mm, dtype=np.int64)[::8]
    self._host_barrier()  # every rank has the file mapped
    if rank == 0:
      os.unlink(self.path)

  def _host_barrier(self):
    import torch
    self.dist.all_reduce(torch.zeros(1, dtype=torch.int32))  # a CPU tensor: the host backend

  def __call__(self, timeout_s=300.0):
    self.gen += 1
    g = self.gen
    self.slots[self.rank] = g
    slots = self.slots
    spins, t0 = 0, None
    while int(slots.min()) < g:
      spins += 1
      if spins & 0xffff == 0:
        now = time.monotonic()
        t0 = t0 or now
        if now - t0 > timeout_s:
          raise RuntimeError("NodeBarrier: a rank did not arrive")

  def close(self):
    self._host_barrier()
    self.slots = None
    try:
      self.mm.close()
    except BufferError:
      pass
    self.f.close()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=2048)
  ap.add_argument("--warmup", type=int, default=256)
  ap.add_argument("--repeats", type=int, default=int(os.environ.get("EP_BENCH_REPEATS", "9")),
                  help="the timed K-step region is run this many times; the median is reported")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-ygo", action="store_true", help="skip the YGOPro observation-encode section")
  args = ap.parse_args()

  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      raise SystemExit("launch multi-GPU runs with torch.distributed.run --nproc-per-node N")
    args.gpus = world

  import torch
  import torch.distributed as dist

  from envpool_amd import capi

  # EP_BENCH_REHEARSE=1: run the N>1 code path on a box with ONE GPU (every rank on cuda:0, gloo
  # instead of RCCL, stats gathered through host memory). For checking the logic only.
  # EP_BENCH_REHEARSE=2: the same, but with the real process-group set-up (gloo + RCCL): RCCL refuses
  # two ranks on one GPU, which exercises the fall-back of the stat exchange.
  rehearse = os.environ.get("EP_BENCH_REHEARSE") in ("1", "2")
  rehearse_real_init = os.environ.get("EP_BENCH_REHEARSE") == "2"
  # (a launcher that shows every rank only its own GPU leaves one visible device: index 0 there)
  dev_index = 0 if rehearse else local_rank % max(1, torch.cuda.device_count())
  torch.cuda.set_device(dev_index)
  dev = torch.device("cuda", dev_index)
  # EP_BENCH_FORCE_DIST=1: take the N>1 code path (process group, RCCL stat exchange on a side
  # stream, node barrier, shard legs) with whatever world size there is - with ONE rank it runs real
  # RCCL on a one-GPU box. For checking the code path; the figures of such a run are not a bench line.
  dist_on = world > 1 or os.environ.get("EP_BENCH_FORCE_DIST") == "1"
  node_barrier = None
  if dist_on:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:  # (EP_BENCH_FORCE_DIST without a launcher: a one-rank group of our own)
      os.environ.setdefault("RANK", "0")
      os.environ.setdefault("WORLD_SIZE", "1")
      os.environ.setdefault("MASTER_PORT", "29533")
    # gloo carries everything that lives on the host (timings, the barrier's set-up); RCCL carries
    # the one device-side exchange there is, the episode-stat all-gather
    dist.init_process_group("gloo" if rehearse and not rehearse_real_init else "cpu:gloo,cuda:nccl")
    node_barrier = NodeBarrier(rank, world, dist)

  def barrier():
    # "barrier + torch.cuda.synchronize()" of the bench contract. All ranks are on ONE node by that
    # contract, so the barrier is a generation counter per rank in a shared-memory file (about a
    # microsecond); an RCCL or gloo barrier costs more than the 20-step region (47 us) it brackets.
    # EP_BENCH_BARRIER=dist restores dist.barrier(). Device first, then ranks: when the barrier
    # falls, every rank's GPU is idle (opening side) / has finished its K steps (closing side).
    torch.cuda.synchronize()
    if dist_on:
      if os.environ.get("EP_BENCH_BARRIER") == "dist":
        dist.barrier()
      else:
        node_barrier()

  L = capi.lib()
  N, T = NUM_ENVS, T_FUSED
  K = max(1, args.steps)   # timed lockstep steps: exactly what was asked for
  W = max(0, args.warmup)
  pool = capi.DummyPoolC(num_envs=N, seed=SEED, device_id=dev_index, env_id_offset=rank * N,
                         ring_depth=2 * T)
  h = pool.handle

  # synthetic actions, resident in HBM: a distinct block per step of a rollout
  env_id = torch.arange(rank * N, (rank + 1) * N, dtype=torch.int32, device=dev)
  penv = env_id.repeat(T, 1).contiguous()                       # [T, N]
  la = (torch.arange(T * N, dtype=torch.float64, device=dev) % 997.0).reshape(T, N, 1).repeat(1, 1, 6).contiguous()
  unused = torch.zeros(N, dtype=torch.int32, device=dev)
  torch.cuda.synchronize()
  ptrs = (C.c_void_p * 5)(env_id.data_ptr(), penv.data_ptr(), la.data_ptr(), unused.data_ptr(), unused.data_ptr())
  dims = (C.c_int64 * 5)(N, N, N, N, N)
  strides = (C.c_int64 * 5)(0, N, N, 0, 0)
  batch = capi.Batch()
  batches = (capi.Batch * T)()

  # episode-stat exchange buffers (multi-GPU only)
  if dist_on:
    st_ret = torch.zeros(N, dtype=torch.float32, device=dev)
    st_len = torch.zeros(N, dtype=torch.int32, device=dev)
    g_ret = torch.zeros(world * N, dtype=torch.float32, device=dev)
    g_len = torch.zeros(world * N, dtype=torch.int32, device=dev)

  def check(rc):
    if rc != 0:
      raise RuntimeError(L.ep_last_error().decode())

  # How the batches of a rollout are handed out inside the timed region. "stream" (default): as
  # device views ordered on the pool's stream (ep_pool_recv_many_on_stream: what a consumer that
  # lives on the GPU gets; the host waits for nothing until the closing synchronize, which waits for
  # every kernel of the region). "blocking": ep_pool_recv_many, the host waits for each rollout's
  # completion event first (what round 1 and 2 timed; measured beside it below).
  recv_mode = {"v": os.environ.get("EP_BENCH_RECV", "stream")}

  def recv_n(n):
    if recv_mode["v"] == "stream":
      check(L.ep_pool_recv_many_on_stream(h, None, n, batches))
    else:
      check(L.ep_pool_recv_many(h, capi.EP_MEM_DEVICE, n, batches))

  # BASELINE config 4 / SURVEY.md 8(d): "every K=64 steps RCCL all-gather of (return, length) per
  # rank on a side stream". The pool snapshots its statistics on its own stream; the side stream
  # waits for that snapshot and runs the all-gather; the host waits for neither.
  STATS_EVERY = 64
  exch = {"since": 0, "count": 0,
          "via": "host+gloo (rehearsal)" if rehearse and not rehearse_real_init else "RCCL all_gather_into_tensor",
          "in_run": 0}
  side = torch.cuda.Stream(device=dev) if dist_on else None

  def exchange_stats_now():
    with torch.cuda.stream(side):
      check(L.ep_pool_episode_stats_async(h, st_ret.data_ptr(), st_len.data_ptr(), None, side.cuda_stream))
      if exch["via"].startswith("RCCL"):
        dist.all_gather_into_tensor(g_ret, st_ret)
        dist.all_gather_into_tensor(g_len, st_len)
      else:
        side.synchronize()
        parts = [torch.zeros(N, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, st_ret.cpu())
        g_ret.copy_(torch.cat(parts))
        parts = [torch.zeros(N, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(parts, st_len.cpu())
        g_len.copy_(torch.cat(parts))
    exch["count"] += 1
    exch["in_run"] += 1

  def exchange_stats(steps_done):
    if dist_on:
      exch["since"] += steps_done
      if exch["since"] >= STATS_EVERY:
        exch["since"] = 0
        exchange_stats_now()

  if dist_on:
    # first exchange, untimed: RCCL connects here. Should it fail on this machine, the gather falls
    # back to host memory + gloo on every rank alike and the line says so.
    failed = 0
    try:
      exchange_stats_now()
      torch.cuda.synchronize()
    except Exception as ex:  # noqa: BLE001
      failed = 1
      print(f"[bench] rank {rank}: RCCL all-gather failed ({ex!r}); using host + gloo", file=sys.stderr, flush=True)
    flag = torch.tensor([failed], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()) and exch["via"].startswith("RCCL"):
      exch["via"] = "host+gloo (RCCL all-gather failed on this machine)"
      exchange_stats_now()

  breakdown = os.environ.get("EP_BENCH_BREAKDOWN") == "1"
  phases = {}

  _chunks = {}
  _marks = [0.0, 0.0, 0.0]  # perf_counter at: region's first call, its return, the last batch handed out
  _rollout, _recv_stream, _recv_block = L.ep_pool_rollout, L.ep_pool_recv_many_on_stream, L.ep_pool_recv_many
  _pc = time.perf_counter

  def run_fused(n_steps):
    # whole launches of T steps, then one shorter launch for the remainder (n_steps % T). Nothing in
    # here but the calls themselves and three clock reads: the driver's 20-step region is 44 us long,
    # and every microsecond of Python bookkeeping inside it is 2 % of the figure.
    chunks = _chunks.get(n_steps)
    if chunks is None:
      chunks = _chunks[n_steps] = [T] * (n_steps // T) + ([n_steps % T] if n_steps % T else [])
    stream_mode = recv_mode["v"] == "stream"
    _marks[0] = _pc()
    rc = _rollout(h, chunks[0], 5, ptrs, dims, strides)
    _marks[1] = _pc()
    for i in range(1, len(chunks)):
      rc |= _rollout(h, chunks[i], 5, ptrs, dims, strides)  # next rollout queued ...
      rc |= _recv_stream(h, None, chunks[i - 1], batches) if stream_mode else _recv_block(h, capi.EP_MEM_DEVICE, chunks[i - 1], batches)
      if dist_on:
        exchange_stats(chunks[i - 1])                       # ... while this one is received
    # stream mode: nothing is waited for here; blocking mode: the last launch's completion event
    rc |= _recv_stream(h, None, chunks[-1], batches) if stream_mode else _recv_block(h, capi.EP_MEM_DEVICE, chunks[-1], batches)
    _marks[2] = _pc()
    if dist_on:
      exchange_stats(chunks[-1])
    if rc:
      check(rc)
    # (round 2 called ep_pool_sync here; the closing barrier synchronises the device, and the error
    # words are looked at after it: the call only cost host time inside the region)

  # reset all envs (forced reset of every env; sync-mode reset())
  check(L.ep_pool_reset(h, env_id.data_ptr(), N, capi.EP_MEM_DEVICE))
  check(L.ep_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(batch)))

  if W:
    run_fused(W)
  # Clock warm-up, independent of --warmup: after idling the shader clock needs some tens of
  # milliseconds of load to come up (the first ~150 launches after idle run 10-60 % slower), and a
  # short timed region would otherwise measure the ramp, not the kernel. Not part of the timed steps.
  # Few long launches (T steps each, queued back to back), not many short ones: a burst of a
  # thousand 20-step launches leaves the runtime with a backlog of completed commands whose clean-up
  # makes the next launches 15-25 us slower for the first millisecond (measured: 70 instead of 48 us
  # per 20-step region). A kernel trace of a --steps 20 run therefore holds launches of two shapes
  # under one kernel name; profiles/ splits them by duration.
  CLOCK_WARM_MS = float(os.environ.get("EP_BENCH_CLOCK_WARM_MS", "30"))
  warm_chunk = min(K, T) if os.environ.get("EP_BENCH_WARM_STYLE") == "same" else T
  CLOCK_WARM_LAUNCHES = max(8, min(6000, int(CLOCK_WARM_MS * 1e-3 / (warm_chunk * 1.45e-6)) + 1))
  check(L.ep_pool_rollout(h, warm_chunk, 5, ptrs, dims, strides))
  for _ in range(CLOCK_WARM_LAUNCHES - 1):
    check(L.ep_pool_rollout(h, warm_chunk, 5, ptrs, dims, strides))
    recv_n(warm_chunk)
  recv_n(warm_chunk)
  check(L.ep_pool_sync(h))
  tot = C.c_double()
  cnt = C.c_int64()
  R = max(1, args.repeats)
  # The K-step region, R times; each run is bracketed by barrier + synchronize on both sides and
  # times exactly K lockstep steps. The line reports the MEDIAN run (max over ranks per run).
  walls, kern, exch_in_run, host_phases = [], [], [], []
  REHEARSALS = 4  # untimed runs of the same region first: the pool creates its timing events in them
  for r in range(-REHEARSALS, R):
    check(L.ep_pool_set_timing(h, 1))
    check(L.ep_pool_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1))
    exch["in_run"] = 0
    barrier()
    t0 = time.perf_counter()
    run_fused(K)
    barrier()
    t1 = time.perf_counter()
    check(L.ep_pool_sync(h))  # outside the region: deferred error words, if any, surface here
    check(L.ep_pool_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1))
    check(L.ep_pool_set_timing(h, 0))
    if r < 0:
      continue
    walls.append(t1 - t0)
    host_phases.append({"enter_to_first_rollout_call_returned_us": (_marks[1] - t0) * 1e6,
                        "until_last_batch_received_us": (_marks[2] - _marks[1]) * 1e6,
                        "closing_synchronize_and_barrier_us": (t1 - _marks[2]) * 1e6})
    exch_in_run.append(exch["in_run"])
    kern.append((tot.value, max(1, cnt.value)))
    if breakdown and rank == 0:
      print(f"[breakdown] run {r}: wall {1e6 * (t1 - t0):.1f} us, kernel {tot.value:.1f} us in {cnt.value} launches, "
            + ", ".join(f"{k} {v:.1f}" for k, v in host_phases[-1].items()), file=sys.stderr, flush=True)
  if dist_on:
    tmax = torch.tensor(walls, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    walls = [float(x) for x in tmax.tolist()]
  order = sorted(range(R), key=lambda i: walls[i])
  med = order[R // 2]
  elapsed = walls[med]
  kern_total_us, n_launch = kern[med]
  kern_us = kern_total_us / n_launch  # average rollout-kernel duration of the median run (HIP events)

  # the same region with the host waiting for every rollout's completion before the closing
  # synchronize (rounds 1-2's way of receiving), for comparison; rank 0's own clock
  blocking_walls = []
  if recv_mode["v"] == "stream" and not dist_on:
    recv_mode["v"] = "blocking"
    for r in range(-2, R):
      barrier()
      t0 = time.perf_counter()
      run_fused(K)
      barrier()
      t1 = time.perf_counter()
      check(L.ep_pool_sync(h))
      if r >= 0:
        blocking_walls.append(t1 - t0)
    recv_mode["v"] = "stream"

  # one-launch-per-step mode, measured on rank 0's pool the same way (reported, not `value`)
  per_step = None
  if True:
    ks = max(16, min(K, 512))
    la1 = la[0].contiguous()
    p1 = (C.c_void_p * 5)(env_id.data_ptr(), env_id.data_ptr(), la1.data_ptr(), unused.data_ptr(), unused.data_ptr())
    torch.cuda.synchronize()
    for _ in range(16):
      check(L.ep_pool_send(h, 5, p1, dims, capi.EP_MEM_DEVICE))
      check(L.ep_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(batch)))
    check(L.ep_pool_sync(h))
    s0 = time.perf_counter()
    for _ in range(ks):
      check(L.ep_pool_send(h, 5, p1, dims, capi.EP_MEM_DEVICE))
      check(L.ep_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(batch)))
    check(L.ep_pool_sync(h))
    s1 = time.perf_counter()
    # kernel duration from HIP events in a separate short pass (the events cost host time)
    check(L.ep_pool_set_timing(h, 1))
    check(L.ep_pool_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1))
    for _ in range(64):
      check(L.ep_pool_send(h, 5, p1, dims, capi.EP_MEM_DEVICE))
      check(L.ep_pool_recv(h, capi.EP_MEM_DEVICE, C.byref(batch)))
    check(L.ep_pool_sync(h))
    check(L.ep_pool_kernel_time_us(h, C.byref(tot), C.byref(cnt), 1))
    check(L.ep_pool_set_timing(h, 0))
    k1_us = tot.value / max(1, cnt.value)
    per_step = {
      "value": N * ks / (s1 - s0), "unit": "env-steps/s (this rank)", "ms_per_step": (s1 - s0) / ks * 1e3,
      "kernel": "dummy_step_sp<10, true>", "kernel_us": k1_us, "kernel_GBps": ALG_BYTES_PER_ENV_STEP * N / (k1_us * 1e-6) / 1e9,
      "kernel_frac_of_hbm_peak": ALG_BYTES_PER_ENV_STEP * N / (k1_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
      "note": "ep_pool_send + blocking ep_pool_recv per lockstep step, device-resident actions",
    }
    # the same blocking step without any launch: ep_pool_step's resident server kernel
    for _ in range(64):
      check(L.ep_pool_step(h, 5, p1, dims, C.byref(batch)))
    kss = 8 * ks
    s0 = time.perf_counter()
    for _ in range(kss):
      check(L.ep_pool_step(h, 5, p1, dims, C.byref(batch)))
    s1 = time.perf_counter()
    per_step["resident_server"] = {
      "value": N * kss / (s1 - s0), "unit": "env-steps/s (this rank)", "ms_per_step": (s1 - s0) / kss * 1e3,
      "GBps_incl_doorbell_round_trip": ALG_BYTES_PER_ENV_STEP * N * kss / (s1 - s0) / 1e9,
      "kernel_frac_of_hbm_peak_incl_doorbell_round_trip": ALG_BYTES_PER_ENV_STEP * N * kss / (s1 - s0) / 1e9 / HBM_PEAK_GBS,
      "note": "ep_pool_step: blocking step, no kernel launch (doorbell word in device memory written through the PCIe BAR, "
              "resident kernel, sc1 loads/stores, per-shard answer words in host memory)"}
    check(L.ep_pool_sync(h))
    # the same loop with the stream-ordered recv: batches are handed out as device views on the
    # pool's own stream (where a GPU-resident policy would run), the host never waits per step
    own = L.ep_pool_stream(h)
    for _ in range(16):
      check(L.ep_pool_step_on_stream(h, own, 5, p1, dims, C.byref(batch)))
    check(L.ep_pool_sync(h))
    kso = max(4 * ks, 2000)
    s0 = time.perf_counter()
    for _ in range(kso):
      check(L.ep_pool_step_on_stream(h, own, 5, p1, dims, C.byref(batch)))
    check(L.ep_pool_sync(h))
    s1 = time.perf_counter()
    per_step["stream_ordered"] = {
      "value": N * kso / (s1 - s0), "unit": "env-steps/s (this rank)", "ms_per_step": (s1 - s0) / kso * 1e3,
      "note": "ep_pool_step_on_stream on the pool's step stream per lockstep step (send + stream-ordered recv in one call: "
              "no event, no host wait inside the loop; a consumer adopts or is adopted by that stream, ep_pool_set_stream)"}

  # multi-player pools (ragged player dimension): 3 players max, every env sent back each step with
  # one action row per player it reported (the previous batch's info:players.env_id, a device view)
  if rank == 0 and not dist_on:
    try:
      mp = capi.DummyPoolC(num_envs=N, seed=SEED, device_id=dev_index, max_num_players=3, ring_depth=4)
      hm = mp.handle
      check(L.ep_pool_reset(hm, env_id.data_ptr(), N, capi.EP_MEM_DEVICE))
      check(L.ep_pool_recv(hm, capi.EP_MEM_DEVICE, C.byref(batch)))
      la_mp = la[0].contiguous()

      def mp_step():
        rows = batch.num_player_rows
        pm = (C.c_void_p * 5)(env_id.data_ptr(), batch.arrays[1].data, la_mp.data_ptr(), unused.data_ptr(), unused.data_ptr())
        dm = (C.c_int64 * 5)(N, rows, N, rows, rows)
        check(L.ep_pool_send(hm, 5, pm, dm, capi.EP_MEM_DEVICE))
        check(L.ep_pool_recv(hm, capi.EP_MEM_DEVICE, C.byref(batch)))
        return rows
      for _ in range(32):
        mp_step()
      km = 256
      s0 = time.perf_counter()
      prow = 0
      for _ in range(km):
        prow += mp_step()
      s1 = time.perf_counter()
      check(L.ep_pool_set_timing(hm, 1))
      check(L.ep_pool_kernel_time_us(hm, C.byref(tot), C.byref(cnt), 1))
      for _ in range(64):
        mp_step()
      check(L.ep_pool_kernel_time_us(hm, C.byref(tot), C.byref(cnt), 1))
      per_step["multiplayer_P3"] = {
        "value": N * km / (s1 - s0), "unit": "env-steps/s (this rank)", "ms_per_step": (s1 - s0) / km * 1e3,
        "player_rows_per_step": prow / km, "fused_kernel_us": tot.value / max(1, cnt.value),
        "note": "max_num_players=3: histogram kernel + fused step / look-back scan / player-row kernel per send, "
                "player total through a host-mapped word (no blocking copy in recv)"}
      mp.close()
    except Exception as ex:  # noqa: BLE001
      per_step["multiplayer_P3"] = {"error": repr(ex)}

  # the boundary also takes host buffers (the reference's numpy path): PCIe-inclusive rate,
  # host actions in, host StateBuffer out, one launch per step. Reported, never `value`.
  if rank == 0 and not dist_on:
    import numpy as np
    hk = 64
    h_env = np.arange(N, dtype=np.int32)
    h_la = np.full((N, 6), 1.5, dtype=np.float64)
    h_z = np.zeros(N, dtype=np.int32)
    ph = (C.c_void_p * 5)(h_env.ctypes.data, h_env.ctypes.data, h_la.ctypes.data, h_z.ctypes.data, h_z.ctypes.data)
    for _ in range(4):
      check(L.ep_pool_send(h, 5, ph, dims, capi.EP_MEM_HOST))
      check(L.ep_pool_recv(h, capi.EP_MEM_HOST, C.byref(batch)))
      L.ep_batch_release(C.byref(batch))
    s0 = time.perf_counter()
    for _ in range(hk):
      check(L.ep_pool_send(h, 5, ph, dims, capi.EP_MEM_HOST))
      check(L.ep_pool_recv(h, capi.EP_MEM_HOST, C.byref(batch)))
      L.ep_batch_release(C.byref(batch))
    s1 = time.perf_counter()
    per_step["host_buffers_pcie_inclusive"] = {
      "value": N * hk / (s1 - s0), "unit": "env-steps/s", "ms_per_step": (s1 - s0) / hk * 1e3,
      "note": "numpy actions in (4 MiB/step), numpy StateBuffer out (4.4 MiB/step), pageable host memory; "
              "kernels read the pinned staging area and write the pinned result block over PCIe themselves"}

  cpu = None
  # (N > 1 too, on rank 0 only, with a shorter sample: the other ranks wait in a host-side barrier)
  cpu_scale = 0.4 if dist_on else 1.0
  if rank == 0 and not args.no_cpu_baseline:
    from oracle.dummy_oracle import DummyOracle, bench_sync_mt, bench_sync_threadpool
    o = DummyOracle(num_envs=N, seed=SEED)
    c0 = time.perf_counter()
    o.bench_sync(40)
    rate1 = 40 / (time.perf_counter() - c0)
    n1 = int(max(100, min(20000, rate1 * 5.0 * cpu_scale)))  # ~5 s single-threaded
    o = DummyOracle(num_envs=N, seed=SEED)
    c0 = time.perf_counter()
    o.bench_sync(n1)
    single = N * n1 / (time.perf_counter() - c0)
    try:
      cores = len(os.sched_getaffinity(0))
    except AttributeError:
      cores = os.cpu_count() or 1
    cores = max(1, min(16, cores))  # the GPU box gives one GPU's job 16 host cores
    c0 = time.perf_counter()
    bench_sync_mt(N, SEED, 40, cores)
    rate_mt = 40 / (time.perf_counter() - c0)
    n_cpu = int(max(100, min(200000, rate_mt * 10.0 * cpu_scale)))  # ~10 s on all cores
    c0 = time.perf_counter()
    bench_sync_mt(N, SEED, n_cpu, cores)
    c1 = time.perf_counter()
    # the same workload through the reference's ARCHITECTURE (oracle/dummy_pool_mt.c: action queue
    # -> worker threads -> state-buffer queue, fresh zeroed buffer per batch): ~3 s per thread count
    arch = {}
    for th in (() if dist_on else (1, cores)):
      c2 = time.perf_counter()
      bench_sync_threadpool(N, SEED, 3, th)
      arch_step_s = (time.perf_counter() - c2) / 4
      n_arch = int(max(4, min(2000, 3.0 / arch_step_s)))
      c2 = time.perf_counter()
      bench_sync_threadpool(N, SEED, n_arch, th)
      arch[str(th)] = N * (n_arch + 1) / (time.perf_counter() - c2)
    cpu = {
      "value": N * n_cpu / (c1 - c0), "unit": "env-steps/s", "cores": cores, "kind": "port",
      "single_thread_value": single,
      "reference_architecture": {
        "value_by_worker_threads": arch, "unit": "env-steps/s",
        "note": "oracle/dummy_pool_mt.c: the reference's ActionBufferQueue -> worker threads -> "
                "StateBufferQueue restated with pthreads and a spin-then-futex semaphore; every "
                "env-step crosses the queue's two semaphores (action_buffer_queue.h:71-80), so more "
                "workers do not help; pool construction is inside the sample. `value` above is the "
                "idealised port (independent shards, no queue) and is the figure the GPU is compared to"},
      "sample": f"{n_cpu} sync lockstep steps of the same {N}-env workload (reset + send/recv) on "
                f"oracle/dummy_oracle.c, {cores} threads each owning a contiguous env_id shard (no "
                "barrier between shards); obs:dyn kept lazy (the reference allocates O(N^2) bytes "
                "per step there and cannot run this N at all)",
    }

  if dist_on and not args.no_cpu_baseline:
    node_barrier._host_barrier()  # ranks other than 0 wait here (blocking, gloo) while rank 0 times the CPU leg

  # second path of the north star: the YGOPro observation-encode tail (BASELINE.json configs[2]
  # shape: 1024 records per step through the pinned pipeline; plus a 65536-record device-resident
  # batch for the encode kernel's own roofline figure). Reported next to the headline value.
  ygo = None
  if rank == 0 and not dist_on and not args.no_ygo:
    try:
      from tools import bench_ygo
      r = bench_ygo.run(cpu_seconds=0.0 if args.no_cpu_baseline else 5.0)
      big, c3 = r["large_batch"], r["config3_host_pipeline"]
      ygo = {
        "metric": "ygopro obs-encodes/s", "unit": "encodes/s",
        "value_device_resident": big["encodes_per_s_wall"],
        "value_config3_host_records_pcie_inclusive": c3["encodes_per_s_wall_pcie_inclusive"],
        "config": {"workload": "YGOPro-v0 observation-encode tail, default config (max_cards 75, max_options 16, "
                               "16 history actions), synthetic duel states; 65536 records per launch device-resident, "
                               "and 1024 host records per step through the pinned submit/collect pipeline"},
        "roofline": {"bound": "hbm", "achieved": big["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": big["achieved_GBps"] / HBM_PEAK_GBS, "traffic": None, "kernel": "ygo_encode",
                     "kernel_us": big["kernel_us"], "alg_bytes_per_record": big["alg_bytes_per_record"]},
        "value_config3_in_place_pcie_inclusive": c3["encodes_per_s_in_place_pcie_inclusive"],
        "config3_us_per_step": c3["us_per_step_wall"], "config3_us_per_step_in_place": c3["us_per_step_wall_in_place"],
        "config3_kernel_us": c3["kernel_us"], "config3_zero_copy": c3.get("zero_copy"),
        "config3_staging_copy_us": c3["h2d_copy_us"],
        "host_emit_records_per_s_1core": r["host_emit_records_per_s_1core"],
        "config3_packed": r.get("config3_packed"),
        "config5_async_ring": r.get("config5_async_ring"),
        "config5_pool": r.get("config5_pool"),
        "config3_pool": r.get("config3_pool"),
      }
      if "cpu_oracle_encodes_per_s_1core" in r:
        ygo["cpu_baseline"] = {"value": r["cpu_oracle_encodes_per_s_1core"], "unit": "encodes/s", "cores": 1, "kind": "port",
                               "sample": "about 5 s of oracle/ygo_oracle.c WriteState restatement on the same synthetic states"}
      pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
      if os.path.exists(pmc_path):
        t = json.load(open(pmc_path)).get("ygo_hbm_bytes_per_record")
        if t:
          ygo["roofline"]["traffic"] = t * big["records_per_launch"]
    except Exception as ex:  # the headline line must still be printed
      ygo = {"error": repr(ex)}
  elif dist_on and not args.no_ygo:
    # N > 1: every rank encodes its own shard of envs (no exchange on this path either); the
    # aggregate is all ranks' records over the slowest rank's time
    try:
      from tools import bench_ygo
      r = bench_ygo.run_shard(dev_index, barrier)
      tm = torch.tensor([r["large_wall_s"], r["pipeline_wall_s"]], dtype=torch.float64)
      dist.all_reduce(tm, op=dist.ReduceOp.MAX)
      ring = torch.tensor([r["ring_4_threads"]["batches_per_s"], r["ring_4_threads"]["gpu_time_hidden_behind_emission"] or 0.0],
                          dtype=torch.float64)
      dist.all_reduce(ring, op=dist.ReduceOp.SUM)
      if rank == 0:
        ygo = {
          "metric": "ygopro obs-encodes/s", "unit": "encodes/s", "scaling": "weak",
          "value_device_resident": world * r["large_records"] / float(tm[0].item()),
          "value_config5_host_records_pcie_inclusive": world * r["pipeline_records"] / float(tm[1].item()),
          "roofline": {"bound": "hbm", "achieved": r["alg_bytes_per_record"] * r["records_per_launch"] / (r["large_kernel_us"] * 1e-6) / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": r["alg_bytes_per_record"] * r["records_per_launch"] / (r["large_kernel_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                       "traffic": None, "kernel": "ygo_encode", "kernel_us": r["large_kernel_us"],
                       "alg_bytes_per_record": r["alg_bytes_per_record"], "note": "rank 0's kernel, HIP events, outside the timed region"},
          "config5_async_ring": {"producer_threads_per_gpu": 4, "batches_per_s_all_gpus": float(ring[0].item()),
                                 "gpu_time_hidden_behind_emission_mean": float(ring[1].item()) / world,
                                 "rank0": r["ring_4_threads"]},
          "config": {"workload": f"YGOPro-v0 observation-encode tail, default config, synthetic duel states, {world} ranks: "
                                 f"{r['records_per_launch']} device-resident records per launch per GPU, and BASELINE config 5's "
                                 f"per-GPU shape, {r['records_per_step']} host records per step written in place into the pinned slots"},
        }
    except Exception as ex:
      if rank == 0:
        ygo = {"error": repr(ex)}

  if rank == 0:
    L.ep_pool_last_rollout_kernel.restype = C.c_char_p
    L.ep_pool_last_rollout_kernel.argtypes = [C.c_void_p]
    rollout_kernel = L.ep_pool_last_rollout_kernel(h).decode()
    # HBM traffic per launch: NOT measured in this run (PMC counters need rocprofv3 passes of their
    # own). What the line carries is the per-launch figure of the committed profile whose launch
    # shape matches the timed region's (every launch T steps, or one launch of K steps), labelled as
    # such; null when no profiled shape matches.
    traffic, traffic_source = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
      try:
        pmc = json.load(open(pmc_path))
        shape = T if K % T == 0 else (K if K < T else None)
        key_f, key_w = f"rollout_fetch_bytes_per_launch_T{shape}", f"rollout_write_bytes_per_launch_T{shape}"
        if shape is not None and key_f in pmc and key_w in pmc:
          traffic = pmc[key_f] + pmc[key_w]
          traffic_source = (f"profiled constant, not measured in this run: {pmc.get('source', 'profiles/pmc_latest.json')}; "
                            f"FETCH_SIZE (x2, gfx950) + WRITE_SIZE per {shape}-step launch of {N} envs")
      except Exception:
        traffic = None
    # K steps in n_launch launches (T steps each, the last one possibly shorter): per-launch figures
    # are averages over the timed region
    alg_launch = N * (K * ALG_ROLLOUT_PER_ENV_STEP + n_launch * ALG_ROLLOUT_PER_ENV_LAUNCH) / n_launch
    achieved = alg_launch / (kern_us * 1e-6) / 1e9
    line = {
      "metric": "env-steps/sec (whole node) Dummy-v0", "value": world * N * K / elapsed, "unit": "env-steps/s",
      "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
      "repeats": R, "clock_warm_launches": CLOCK_WARM_LAUNCHES, "untimed_rehearsals": REHEARSALS,
      "ms_per_step_all_runs": [w / K * 1e3 for w in walls],
      "host_phases_of_median_run_us": host_phases[med],
      "batches_handed_out": ("stream-ordered device views (ep_pool_recv_many_on_stream); the closing synchronize is the region's only host wait"
                             if recv_mode["v"] == "stream" else "ep_pool_recv_many: the host waits for each rollout's completion event"),
      **({"value_with_blocking_recv": world * N * K / sorted(blocking_walls)[len(blocking_walls) // 2],
          "ms_per_step_with_blocking_recv": sorted(blocking_walls)[len(blocking_walls) // 2] / K * 1e3} if blocking_walls else {}),
      **({"stats_exchanges_in_each_run": exch_in_run, "stats_exchanges_in_median_run": exch_in_run[med],
          "stats_exchanges_total": exch["count"],
          "barrier": os.environ.get("EP_BENCH_BARRIER", "shared-memory generation barrier (one node)")} if dist_on else {}),
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
      "data": "synthetic",
      "config": {
        "workload": f"Dummy-v0 num_envs={N} per GPU, seed={SEED}, sync lockstep reset/step, "
                    f"fused rollout T={T} steps/launch, device-resident actions, StateBuffer ring depth {2 * T}",
        "global_num_envs": world * N, "sharding": "env_id blocks, one independent pool per GPU",
        "collective": (f"all-gather of episode (return,length), {N * 8} B per rank, every {STATS_EVERY} steps on a side "
                       f"stream, via {exch['via']}") if dist_on else "none",
      },
      "roofline": {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
        "kernel": rollout_kernel, "kernel_us": kern_us, "launches": n_launch,
        "alg_bytes_per_launch": alg_launch,
      },
      "per_step_launch": per_step,
    }
    if cpu is not None:
      line["cpu_baseline"] = cpu
    if ygo is not None:
      line["ygo_encode"] = ygo
    print(json.dumps(line), flush=True)

  pool.close()
  if dist_on:
    node_barrier.close()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
