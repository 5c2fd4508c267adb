import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import envpool_amd
def med(ts):
    ts = sorted(ts); return ts[len(ts)//2]*1e6
for P in (1, 3):
  for k in range(0, 17):
    n = 1 << k
    if P == 3 and n > 16384: continue
    env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=42, max_num_players=P)
    obs, info = env.reset()
    ts = []
    for i in range(200):
        m = info["players"]["env_id"].shape[0] if P > 1 else n
        act = {"list_action": np.full((n, 6), 1.5), "players.action": np.zeros(m, np.int32), "players.id": np.zeros(m, np.int32)}
        t0 = time.perf_counter()
        obs, rew, term, trunc, info = env.step(act)
        ts.append(time.perf_counter() - t0)
    print(f"P={P} n={n:6d} env.step {med(ts[50:]):8.1f} us", flush=True)
    env.close()
