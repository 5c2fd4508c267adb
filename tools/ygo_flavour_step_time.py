import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import envpool_amd
from envpool_amd import capi_ygo, ygopro
from tools import ygo_synth
def med(ts):
    ts = sorted(ts); return ts[len(ts)//2]*1e6
N = 64
table = ygo_synth.make_table(1000, seed=0)
for fl in ("raw", "gymnasium", "dm"):
    cfg = capi_ygo.default_config(max_batch=N, ring_depth=0)
    eng = ygo_synth.SynthEngine(cfg, table, N, seed=2); eng.set_replay()
    ygopro.init_module(eng.factory, table)
    try:
        env = envpool_amd.make("YGOPro-v0", fl, num_envs=N)
        env.reset()
        act = np.zeros(N, np.int32)
        ts = []
        for i in range(400):
            t0 = time.perf_counter(); env.step(act); ts.append(time.perf_counter() - t0)
        print(f"YGOPro {fl:10s} n={N} env.step {med(ts[100:]):7.1f} us", flush=True)
        env.close()
    finally:
        ygopro.init_module(None, None); eng.close()
