import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import envpool_amd
def med(ts):
    ts = sorted(ts); return ts[len(ts)//2]*1e6
for fl in ("raw", "gymnasium", "gym", "dm"):
  for n in (8, 65536):
    try:
      env = envpool_amd.make("Dummy-v0", fl, num_envs=n, seed=42)
    except Exception as ex:
      print(fl, n, "unavailable:", type(ex).__name__, str(ex)[:80]); continue
    env.reset()
    act = {"list_action": np.full((n, 6), 1.5), "players.action": np.zeros(n, np.int32), "players.id": np.zeros(n, np.int32)}
    ts = []
    for i in range(300):
        t0 = time.perf_counter(); env.step(act); ts.append(time.perf_counter() - t0)
    print(f"{fl:10s} n={n:6d} env.step {med(ts[50:]):8.1f} us", flush=True)
