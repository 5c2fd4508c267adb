"""Config 3 as a sync ep_ygo_pool over worker-thread counts, beside the cgroup's CPU quota and its
throttle counters (a quota-limited box stalls every thread of the job once spinning workers use the
quota up): python -m tools.ygo_pool_threads [threads ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _read(path):
  try:
    with open(path) as f:
      return f.read().strip()
  except OSError:
    return None


def _throttle():
  s = _read("/sys/fs/cgroup/cpu.stat") or ""
  d = dict(l.split() for l in s.splitlines() if len(l.split()) == 2)
  return int(d.get("nr_throttled", -1)), int(d.get("throttled_usec", -1))


def main():
  from tools import bench_ygo
  threads = [int(a) for a in sys.argv[1:]] or [2, 4, 8, 12, 15, 16]
  print("cpu.max:", _read("/sys/fs/cgroup/cpu.max"), " affinity:", len(os.sched_getaffinity(0)), " cpu_count:", os.cpu_count(), flush=True)
  for nt in threads:
   for spin in [int(x) for x in os.environ.get("SPINS", "4000").split(",")]:
    os.environ["EP_YGO_POOL_SPIN"] = str(spin)
    t0 = _throttle()
    if os.environ.get("ASYNC"):  # config 5's shape (1024 envs, batches of 512) with a free engine
      r = bench_ygo.run_config5_pool(threads_list=(nt,), batches=int(os.environ.get("BATCHES", "400")), engine_us=float(os.environ.get("ENGINE_US", "0")))
      v = r["by_threads"][str(nt)]
      print(f"threads {nt:3d}: {v['batches_per_s']:8.0f} batches/s  {v['env_steps_per_s']:.3e} env-steps/s", flush=True)
      continue
    r = bench_ygo.run_config3_pool(threads_list=(nt,), steps=300, n_envs=int(os.environ.get("ENVS", "1024")))
    t1 = _throttle()
    v = list(r["by_threads"].values())[0]
    print(f"threads {nt:3d} spin {spin:7d}: median {v['us_per_step_median']:7.1f} us/step  throttled periods +{t1[0] - t0[0]}  (+{(t1[1] - t0[1]) / 1e3:.1f} ms)", flush=True)


if __name__ == "__main__":
  main()
