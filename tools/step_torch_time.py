import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import envpool_amd
N = 65536
env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=N, seed=42)
obs, info = env.reset_torch()
la = torch.full((N, 6), 1.5, dtype=torch.float64, device="cuda")
for name in ("step_torch",):
    ts = []
    for i in range(3000):
        t0 = time.perf_counter()
        obs, rew, term, trunc, info = env.step_torch({"list_action": la})
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    ts = sorted(ts[500:])
    print(name, "median us", ts[len(ts)//2]*1e6, "p90", ts[int(len(ts)*0.9)]*1e6)
assert term.dtype == torch.bool and bool((term == (info["elapsed_step"] >= 0) & term).all())
