import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import envpool_amd
def med(ts):
    ts = sorted(ts); return ts[len(ts)//2]*1e6
rng = np.random.default_rng(0)
for n in (4096, 65536):
    env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=42)
    env.reset()
    for name, ids in (("ascending", np.arange(n, dtype=np.int32)), ("permuted", rng.permutation(n).astype(np.int32)), ("reversed", np.arange(n, dtype=np.int32)[::-1].copy())):
        act = {"list_action": np.full((n, 6), 1.5), "players.action": np.zeros(n, np.int32), "players.id": np.zeros(n, np.int32)}
        ts = []
        for i in range(200):
            t0 = time.perf_counter(); env.step(act, ids); ts.append(time.perf_counter() - t0)
        print(f"n={n:6d} {name:10s} env.step {med(ts[50:]):8.1f} us", flush=True)
    env.close()
