import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import envpool_amd
def med(ts):
    ts = sorted(ts); return ts[len(ts)//2]*1e6
for n, k in ((4096, 1), (4096, 4), (65536, 1), (65536, 4), (65536, 8)):
    env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=42, num_shards=k) if k > 1 else envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=42)
    env.reset()
    act = {"list_action": np.full((n, 6), 1.5), "players.action": np.zeros(n, np.int32), "players.id": np.zeros(n, np.int32)}
    ts = []
    for i in range(150):
        t0 = time.perf_counter(); env.step(act); ts.append(time.perf_counter() - t0)
    print(f"n={n} shards={k}: env.step {med(ts[30:]):8.1f} us", flush=True)

# async: config 5's shape in one process (8192 envs, batches of 4096, 8 shards), recv + send
for n, b, k in ((8192, 4096, 1), (8192, 4096, 8)):
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
