"""GPU: the drop-in Python surface — envpool_amd.make(...) -> reset/step/send/recv — against the
oracle, plus the raw pybind `_DummyEnvPool` boundary (py_envpool.h:279-287)."""
import numpy as np
import pytest

from tests.util import assert_batches_equal, dummy_action

pytestmark = pytest.mark.gpu


def test_raw_pybind_boundary_config1():
  """BASELINE config 1 shape (sync, N=8, B=8, seed 42) through _reset/_send/_recv, including the
  obs:dyn object array of arrays."""
  from envpool_amd.dummy.dummy_envpool import _DummyEnvPool, _DummyEnvSpec
  from oracle.dummy_oracle import DummyOracle
  spec = _DummyEnvSpec((8, 8, 0, 1, -1, "envpool2", 42, False, 2**31 - 1, 10, 6))
  pool, o = _DummyEnvPool(spec), DummyOracle(num_envs=8, seed=42)
  ids = np.arange(8, dtype=np.int32)
  pool._reset(ids); o._reset(ids)
  for t in range(60):
    got, want = pool._recv(), o._recv(dyn="eager")
    assert len(got) == 12
    assert_batches_equal(got, want, f"t={t}")
    dyn = got[9]
    assert dyn.dtype == object and dyn.shape == (8,)
    for e in range(8):
      assert dyn[e].dtype == np.int32 and np.array_equal(dyn[e], want[9][e])
    a = dummy_action(ids, value=1.5)
    # force-cast like the reference (py_envpool.h:81,90): float32 list_action, int64 ids
    pool._send([a[0].astype(np.int64), a[1], a[2].astype(np.float32), a[3], a[4]])
    o._send(a)
  kept = pool._recv()  # returned buffers stay valid after the pool is gone
  snapshot = [np.array(x) for k, x in enumerate(kept) if k != 9]
  del pool
  for x, y in zip([x for k, x in enumerate(kept) if k != 9], snapshot):
    assert np.array_equal(x, y)


def test_make_step_reset():
  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=32, seed=3)
  o = DummyOracle(num_envs=32, seed=3)
  ids = np.arange(32, dtype=np.int32)
  obs, info = env.reset()
  o._reset(ids); w = o._recv()
  assert np.array_equal(obs["raw"], w[8]) and np.array_equal(info["env_id"], w[0])
  act = {"list_action": np.full((32, 6), 0.5), "players.action": np.zeros(32, np.int32), "players.id": np.zeros(32, np.int32)}
  for t in range(45):
    obs, rew, term, trunc, info = env.step(act)
    o._send(dummy_action(ids)); w = o._recv()
    assert np.array_equal(obs["raw"], w[8]) and np.array_equal(term, w[3] & ~w[7]) and np.array_equal(trunc, w[7])
    assert np.array_equal(info["elapsed_step"], w[2]) and np.array_equal(info["players"]["done"], w[10])
    assert rew.tobytes() == w[4].tobytes()
  # step a subset by env_id
  sub = np.array([9, 3, 20], np.int32)
  a2 = {k: v[:3] for k, v in act.items()}
  obs, rew, term, trunc, info = env.step(a2, env_id=sub)
  o._send(dummy_action(sub)); w = o._recv()
  assert info["env_id"].tolist() == [9, 3, 20] and np.array_equal(obs["raw"], w[8])
  assert "Dummy" in repr(env) and len(env) == 32


def test_make_async_and_lazy_dyn():
  import envpool_amd
  from envpool_amd.python.lazy import LazyDyn
  env = envpool_amd.make("Dummy-v0", "raw", num_envs=2048, batch_size=512, seed=1)
  assert env.is_async
  env.async_reset()
  seen = []
  for _ in range(4):
    obs, info = env.recv(reset=True)
    assert obs["raw"].shape == (512, 10)
    assert isinstance(obs["dyn"], LazyDyn) and obs["dyn"][5].shape == (int(info["env_id"][5]) + 1, 10)
    seen.append(info["env_id"])
  assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(2048))


def test_device_views_and_torch_interop():
  import torch
  from envpool_amd.dummy.dummy_envpool import _DummyEnvPool, _DummyEnvSpec
  from envpool_amd.interop import as_torch
  from oracle.dummy_oracle import DummyOracle
  n = 512
  spec = _DummyEnvSpec((n, 0, 0, 1, -1, "envpool2", 7, False, 2**31 - 1, 10, 6))
  pool, o = _DummyEnvPool(spec), DummyOracle(num_envs=n, seed=7)
  ids = np.arange(n, dtype=np.int32)
  dev = torch.device("cuda", 0)
  t_ids = torch.from_numpy(ids).to(dev)
  t_la = torch.full((n, 6), 2.0, dtype=torch.float64, device=dev)
  torch.cuda.synchronize()
  pool._reset(ids); o._reset(ids)
  for t in range(10):
    descs, rows, prows = pool._recv_device()
    w = o._recv()
    assert rows == n and prows == n
    got = [None if d is None else as_torch(d, dev).cpu().numpy() for d in descs]
    assert_batches_equal(got, w, f"t={t}")
    pool._send_device([t_ids.data_ptr(), t_ids.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr()], [n] * 5)
    o._send(dummy_action(ids))

  # the stream-ordered form: torch works on the pool's step stream, nothing waits on the host
  ext = torch.cuda.ExternalStream(pool._stream())
  kept = []
  with torch.cuda.stream(ext):
    for t in range(10):
      descs, rows, prows = pool._recv_on_stream()
      kept.append([None if d is None else as_torch(d, dev).clone() for d in descs])
      pool._send_device([t_ids.data_ptr(), t_ids.data_ptr(), t_la.data_ptr(), t_ids.data_ptr(), t_ids.data_ptr()], [n] * 5)
  pool._sync()
  for t in range(10):
    assert_batches_equal([None if x is None else x.cpu().numpy() for x in kept[t]], o._recv(), f"on stream t={t}")
    o._send(dummy_action(ids))


@pytest.mark.parametrize("n,blocking", [(512, True), (300, True), (512, False), (300, False)])
def test_step_torch_front_end(n, blocking):
  """step_torch / reset_torch: a policy on the GPU steps the pool with torch tensors and gets torch
  tensors back; same structure as step(), values equal to the oracle's. blocking: stream_ordered=False
  (n = 512 takes the resident-kernel path of ep_pool_step, n = 300 the launch path); otherwise the
  default, the step in stream order on torch's current stream."""
  import torch
  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  env = envpool_amd.make("Dummy-v0", "gym", num_envs=n, seed=5)
  o = DummyOracle(num_envs=n, seed=5)
  ids = np.arange(n, dtype=np.int32)
  dev = torch.device("cuda", 0)
  obs, info = env.reset_torch()
  o._reset(ids); w = o._recv()
  assert obs["raw"].is_cuda and obs["raw"].cpu().numpy().tobytes() == w[8].tobytes()
  la = torch.full((n, 6), 1.5, dtype=torch.float64, device=dev)
  for t in range(12):
    obs, rew, term, trunc, info = env.step_torch({"list_action": la}, **({"stream_ordered": False} if blocking else {}))
    o._send(dummy_action(ids)); w = o._recv()
    assert info["env_id"].cpu().numpy().tobytes() == w[0].tobytes()
    assert info["elapsed_step"].cpu().numpy().tobytes() == w[2].tobytes(), t
    assert obs["raw"].cpu().numpy().tobytes() == w[8].tobytes(), t
    assert rew.cpu().numpy().tobytes() == w[4].tobytes()
    assert (term.cpu().numpy() == (w[3] & ~w[7])).all() and trunc.cpu().numpy().tobytes() == w[7].tobytes()
  # a subset, in another order
  sel = torch.tensor([7, 3, 200], dtype=torch.int32, device=dev)
  obs, rew, term, trunc, info = env.step_torch({"list_action": la[:3]}, env_id=sel)
  o._send(dummy_action(np.array([7, 3, 200], np.int32))); w = o._recv()
  assert info["env_id"].cpu().numpy().tolist() == [7, 3, 200] and obs["raw"].cpu().numpy().tobytes() == w[8].tobytes()


@pytest.mark.parametrize("n", [4096, 300])
def test_rollout_torch_front_end(n):
  """rollout_torch: T lockstep steps in one launch from the Python front-end (n = 4096: the wave
  kernel; n = 300: the generic one), every step's result equal to the oracle's stepwise - with
  per-step action rows, then with the same rows every step, twice, so that the 16-slot ring wraps
  and the cached per-slot results are handed out again."""
  import torch
  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=n, seed=5, ring_depth=16)
  o = DummyOracle(num_envs=n, seed=5)
  ids = np.arange(n, dtype=np.int32)
  dev = torch.device("cuda", 0)
  env.reset_torch()
  o._reset(ids); o._recv()
  T = 12
  rng = np.random.default_rng(3)
  pe = ids[None, :] + (rng.random((T, n)) < 0.1).astype(np.int32)
  lane = rng.integers(0, 4, size=(T, n)).astype(np.float64)
  la = torch.from_numpy(np.repeat(lane[:, :, None], 6, axis=2)).to(dev)

  def check(res, want, ctx):
    obs, rew, term, trunc, info = res
    assert obs["raw"].cpu().numpy().tobytes() == want[8].tobytes(), ctx
    assert info["env_id"].cpu().numpy().tobytes() == want[0].tobytes() and info["elapsed_step"].cpu().numpy().tobytes() == want[2].tobytes(), ctx
    assert rew.cpu().numpy().tobytes() == want[4].tobytes() and trunc.cpu().numpy().tobytes() == want[7].tobytes(), ctx
    assert term.cpu().numpy().tobytes() == (want[3] & ~want[7]).tobytes(), ctx

  out = env.rollout_torch({"list_action": la, "players.env_id": torch.from_numpy(pe).to(dev)})
  assert len(out) == T
  for t in range(T):
    o._send(dummy_action(ids, players_env_id=pe[t]))
    check(out[t], o._recv(), f"per-step rows, t={t}")
  for rnd in range(2):
    out = env.rollout_torch(la[0], T=T)
    for t in range(T):
      o._send(dummy_action(ids))
      check(out[t], o._recv(), f"same rows, round {rnd}, t={t}")
  # a step in stream order and a blocking step on the same pool afterwards
  check(env.step_torch({"list_action": la[0]}, stream_ordered=True), (o._send(dummy_action(ids)), o._recv())[1], "stream-ordered step after")
  check(env.step_torch({"list_action": la[0]}, stream_ordered=False), (o._send(dummy_action(ids)), o._recv())[1], "blocking step after")
  with pytest.raises(Exception):
    env.rollout_torch(la[0], T=17)  # more steps than ring slots
  env.close()


@pytest.mark.parametrize("kw", [dict(num_envs=512), dict(num_envs=96, batch_size=32)])
def test_recv_torch_front_end(kw):
  """send (host arrays) + recv_torch of a Dummy pool, sync and async: the batch as torch views of
  the ring slot, results kept per slot, `terminated` from the step kernels' array beside the slot
  (no torch launch); equal to the oracle batch by batch while the ring's slots come round."""
  import torch
  import envpool_amd
  from oracle.dummy_oracle import DummyOracle
  env = envpool_amd.make("Dummy-v0", "gymnasium", seed=3, **kw)
  o = DummyOracle(seed=3, **kw)
  n, bsz = kw["num_envs"], kw.get("batch_size", kw["num_envs"])
  ids = np.arange(n, dtype=np.int32)
  env.async_reset(); o._reset(ids)
  seen_term = False
  for it in range(60):
    res = env.recv_torch()
    want = o._recv()
    obs, rew, term, trunc, info = res
    assert obs["raw"].is_cuda and obs["raw"].cpu().numpy().tobytes() == want[8].tobytes(), it
    assert info["env_id"].cpu().numpy().tobytes() == want[0].tobytes() and info["elapsed_step"].cpu().numpy().tobytes() == want[2].tobytes(), it
    assert trunc.cpu().numpy().tobytes() == want[7].tobytes() and term.cpu().numpy().tobytes() == (want[3] & ~want[7]).tobytes(), it
    seen_term = seen_term or bool((want[3] & ~want[7]).any())
    sel = np.asarray(want[0])
    a = dummy_action(sel)
    env.send({"list_action": a[2], "players.action": a[3], "players.id": a[4]}, sel)
    o._send(a)
  assert seen_term
  env.close()


def test_close_releases_the_pool():
  """Additive close() (gym's convention): the device memory goes at once, later calls are refused."""
  import envpool_amd
  env = envpool_amd.make("Dummy-v0", "gymnasium", num_envs=16, seed=1)
  env.reset()
  env.close()
  env.close()   # idempotent
  with pytest.raises(ValueError):
    env.reset()
