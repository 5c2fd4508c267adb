import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import envpool_amd
def med(ts):
    ts = sorted(ts); return ts[len(ts)//2]*1e6
NUMPY = os.environ.get('ONLY_TORCH') != '1'
for n, k in ((4096, 1), (4096, 4), (65536, 1), (65536, 4), (65536, 8)) if NUMPY else ():
    env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=42, num_shards=k) if k > 1 else envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=42)
    env.reset()
    act = {"list_action": np.full((n, 6), 1.5), "players.action": np.zeros(n, np.int32), "players.id": np.zeros(n, np.int32)}
    ts = []
    for i in range(150):
        t0 = time.perf_counter(); env.step(act); ts.append(time.perf_counter() - t0)
    print(f"n={n} shards={k}: env.step {med(ts[30:]):8.1f} us", flush=True)

# async: config 5's shape in one process (8192 envs, batches of 4096, 8 shards), recv + send
for n, b, k in ((8192, 4096, 1), (8192, 4096, 8)) if NUMPY else ():
    kw = dict(num_shards=k) if k > 1 else {}
    env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, batch_size=b, seed=42, **kw)
    env.async_reset()
    act = {"list_action": np.full((b, 6), 1.5), "players.action": np.zeros(b, np.int32), "players.id": np.zeros(b, np.int32)}
    ts = []
    for i in range(300):
        t0 = time.perf_counter()
        obs, rew, term, trunc, info = env.recv()
        env.send(act, info["env_id"])
        ts.append(time.perf_counter() - t0)
    print(f"async n={n} batch={b} shards={k}: recv + send {med(ts[50:]):8.1f} us per batch", flush=True)

if os.environ.get('ONLY_NUMPY') == '1':
    sys.exit(0)

# the device path (round 3): CUDA tensors in, per-shard views of the rings out, nothing crosses PCIe;
# a whole-pool step rings every shard's resident server and then waits for each
import gc
import torch
# (a full garbage collection of a process that has imported torch takes ~40 ms, and the per-step result
# tuples are enough garbage to trigger one now and then: keep what exists now out of the collector's way)
gc.collect(); gc.freeze()
for n, k in ((65536, 1), (65536, 4), (65536, 8), (524288, 8)) if os.environ.get('ONLY_K4') != '1' else ((65536, 4),):
    kw = dict(num_shards=k, shard_devices=[0] * k) if k > 1 else {}
    env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=42, **kw)
    env.reset_torch()
    la = torch.full((n, 6), 1.5, dtype=torch.float64, device="cuda")
    ts, tc = [], []
    main = {"stream_ordered": False} if k == 1 else {}  # (unsharded: the blocking form first, the default in brackets)
    for i in range(200):
        env.step_torch({"list_action": la}, **main)
    torch.cuda.current_stream().synchronize()
    t_all = time.perf_counter()
    for i in range(1000):
        t0 = time.perf_counter(); res = env.step_torch({"list_action": la}, **main); ts.append(time.perf_counter() - t0)
    torch.cuda.current_stream().synchronize()
    t_all = (time.perf_counter() - t_all) / 1000 * 1e6
    # (median of the calls, and 1000 calls + one closing stream synchronise over 1000: the same for the
    # blocking forms; for the stream-ordered form the second one is the figure that cannot hide work)
    line = f"n={n} shards={k}: env.step_torch{'(stream_ordered=False)' if k == 1 else ''} {med(ts):8.1f} us  ({t_all:6.1f} us per step over 1000 steps incl. the closing synchronise)"
    if k == 1:
        for i in range(200):
            env.step_torch({"list_action": la}, stream_ordered=True)
        torch.cuda.current_stream().synchronize()
        ts2 = []
        t2 = time.perf_counter()
        for i in range(1000):
            t0 = time.perf_counter(); env.step_torch({"list_action": la}, stream_ordered=True); ts2.append(time.perf_counter() - t0)
        torch.cuda.current_stream().synchronize()
        t2 = (time.perf_counter() - t2) / 1000 * 1e6
        line += f"   [the default, in stream order: {med(ts2):6.1f} us per call, {t2:6.1f} us per step over 1000 steps incl. the closing synchronise]"
    if k > 1:
        ts2 = []
        for i in range(600):
            t0 = time.perf_counter(); env.step_torch({"list_action": la}, stream_ordered=False); ts2.append(time.perf_counter() - t0)
        line += f"   [blocking form, one launch + wait per shard: {med(ts2[100:]):6.1f} us]"
    if k > 1:
        for i in range(200):
            res = env.step_torch({"list_action": la})
            t0 = time.perf_counter(); whole = res.cat(); torch.cuda.current_stream().synchronize(); tc.append(time.perf_counter() - t0)
        line += f"   (+ .cat() of the {k} parts into one tensor per key, synchronised: {med(tc[50:]):6.1f} us)"
        line += f"   servers (launches, idle stops, steps served): {[s._server_stats()[:3] for s in env.shards]}"
    print(line, flush=True)
    env.close()

# the fused rollout from the Python front-end: T = 128 steps per call
if os.environ.get('ONLY_K4') != '1':
    n, T = 65536, 128
    env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=42, ring_depth=2 * T)
    env.reset_torch()
    la = torch.full((n, 6), 1.5, dtype=torch.float64, device="cuda")
    for i in range(10):
        env.rollout_torch(la, T=T)
    torch.cuda.current_stream().synchronize()
    ts = []
    t_all = time.perf_counter()
    for i in range(40):
        t0 = time.perf_counter(); res = env.rollout_torch(la, T=T); ts.append(time.perf_counter() - t0)
    torch.cuda.current_stream().synchronize()
    t_all = (time.perf_counter() - t_all) / (40 * T) * 1e6
    print(f"n={n}: env.rollout_torch(T={T}) {med(ts):8.1f} us per call, {t_all:6.2f} us per step over 40 calls incl. the closing synchronise "
          f"= {n / t_all * 1e6:.3g} env-steps/s from Python", flush=True)
    env.close()

if os.environ.get("PROFILE_K4") == "1":
    import cProfile, pstats
    env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=65536, seed=42, num_shards=4, shard_devices=[0] * 4)
    env.reset_torch()
    la = torch.full((65536, 6), 1.5, dtype=torch.float64, device="cuda")
    for i in range(300):
        env.step_torch({"list_action": la})
    pr = cProfile.Profile()
    pr.enable()
    for i in range(1000):
        env.step_torch({"list_action": la})
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
