"""ctypes front-end of tools/ygo_synth.c: synthetic duel states in the reference's data forms."""
import ctypes as C
import os
import subprocess

from envpool_amd.capi_ygo import CardData, RawState, YgoConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
  global _LIB
  if _LIB is None:
    path = os.path.join(_HERE, "libygosynth.so")
    if not os.path.exists(path):
      subprocess.check_call(["make", "-s", "-C", os.path.dirname(_HERE), "synth"])
    L = C.CDLL(path)
    L.ys_make_table.argtypes = [C.c_int32, C.c_uint64, C.POINTER(CardData)]
    L.ys_gen_batch.restype = C.c_void_p
    L.ys_gen_batch.argtypes = [C.c_uint64, C.c_int64, C.c_int32, C.POINTER(YgoConfig), C.POINTER(CardData), C.c_int32,
                               C.POINTER(RawState)]
    L.ys_free_batch.argtypes = [C.c_void_p, C.c_int32]
    L.ys_run_producers.argtypes = [C.c_void_p, C.POINTER(RingApi), C.POINTER(YgoConfig), C.POINTER(RawState), C.c_int32,
                                   C.c_int32, C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_double), C.c_void_p]
    L.ys_engine_create.restype = C.c_void_p
    L.ys_engine_create.argtypes = [C.POINTER(YgoConfig), C.POINTER(CardData), C.c_int32, C.c_int32, C.c_uint64, C.c_double]
    L.ys_engine_destroy.argtypes = [C.c_void_p]
    L.ys_engine_set_fail_env.argtypes = [C.c_void_p, C.c_int32]
    L.ys_engine_set_replay.argtypes = [C.c_void_p, C.c_int32]
    L.ys_engine_set_env_id_offset.argtypes = [C.c_void_p, C.c_int32]
    _LIB = L
  return _LIB


class RingApi(C.Structure):
  """Entry points of the encoder's arrival-ordered ring, handed to the stand-in engine as pointers."""
  _fields_ = [("alloc_record", C.c_void_p), ("emit_record", C.c_void_p), ("record_done", C.c_void_p),
              ("record_stride", C.c_void_p), ("emit_arrival", C.c_void_p), ("put_arrival", C.c_void_p)]


RING_IN_PLACE, SINK, PACKED_EMIT, PACKED_PUT = 0, 1, 2, 3  # how a finished env-step reaches the encoder


def ring_api():
  from envpool_amd import capi_ygo
  L = capi_ygo.lib()
  addr = lambda f: C.cast(f, C.c_void_p).value
  return RingApi(addr(L.ep_ygo_alloc_record), addr(L.ep_ygo_emit_record), addr(L.ep_ygo_record_done),
                 addr(L.ep_ygo_record_stride), addr(L.ep_ygo_emit_arrival), addr(L.ep_ygo_put_arrival))


def run_producers(enc_handle, cfg, raw_states, n_envs, n_threads, steps_per_env, engine_us=0.0, sink_only=False,
                  mode=None, pre_records=None):
  """Blocks until the producer threads of the stand-in engine are done (call it from a thread of
  its own next to the consumer; ctypes drops the GIL). `mode`: RING_IN_PLACE (default), SINK,
  PACKED_EMIT or PACKED_PUT (`pre_records`: uint8 [n_envs, stride] emitted beforehand).
  Returns (status, seconds spent emitting)."""
  api = ring_api()
  emit = C.c_double()
  if mode is None:
    mode = SINK if sink_only else RING_IN_PLACE
  pre = pre_records.ctypes.data if pre_records is not None else None
  rc = lib().ys_run_producers(enc_handle, C.byref(api), C.byref(cfg), raw_states, n_envs, n_threads, steps_per_env,
                              engine_us, mode, C.byref(emit), pre)
  return rc, emit.value


def make_table(n=1000, seed=0):
  t = (CardData * n)()
  lib().ys_make_table(n, seed, t)
  return t


class SynthBatch:
  """n synthetic states; `.raw` is a ctypes array of ygo_raw_state valid while this object lives."""

  def __init__(self, cfg, table, n, seed=0, first=0):
    self.n = n
    self.raw = (RawState * max(n, 1))()
    self._h = lib().ys_gen_batch(seed, first, n, C.byref(cfg), table, len(table), self.raw)

  def __del__(self):
    if getattr(self, "_h", None):
      lib().ys_free_batch(self._h, self.n)
      self._h = None


class SynthEngine:
  """The stand-in duel engine of ygo_synth.c behind the `ep_ygo_engine` callbacks: `.factory` is
  what `envpool_amd.ygopro.init_module` takes, `.engine()` the ctypes struct for `YgoPoolC`.
  `expected_state(env, k, ...)` rebuilds the state its k-th callback for `env` produced."""

  def __init__(self, cfg, table, num_envs, seed=0, engine_us=0.0):
    L = lib()
    self.cfg, self.table, self.num_envs, self.seed = cfg, table, num_envs, seed
    self.ctx = L.ys_engine_create(C.byref(cfg), table, len(table), num_envs, seed, engine_us)
    self.reset_addr = C.cast(L.ys_engine_reset, C.c_void_p).value
    self.step_addr = C.cast(L.ys_engine_step, C.c_void_p).value

  def engine(self):
    from envpool_amd import capi_ygo
    return capi_ygo.Engine(self.ctx, self.reset_addr, self.step_addr)

  def factory(self, spec):
    return self.ctx, self.reset_addr, self.step_addr, self

  def set_replay(self, on=True):
    """Every callback of an env hands out the same pre-generated state (benchmarks)."""
    lib().ys_engine_set_replay(self.ctx, int(on))

  def set_env_id_offset(self, off):
    lib().ys_engine_set_env_id_offset(self.ctx, int(off))

  def set_fail_env(self, env):
    lib().ys_engine_set_fail_env(self.ctx, env)

  def close(self):
    if getattr(self, "ctx", None):
      lib().ys_engine_destroy(self.ctx)
      self.ctx = None

  __del__ = close


class EnginePoolModel:
  """What an ep_ygo_pool over SynthEngine must deliver, env by env: the framework half of Env
  (auto-reset one step after done, current_step_, env.h:181-189 / async_envpool.h:125) around the
  engine's deterministic states. `send` / `reset` return, per env, the reference-form state the
  oracle has to encode for the row that will come back."""

  def __init__(self, cfg, table, num_envs, seed=0):
    self.cfg, self.table, self.seed = cfg, table, seed
    self.calls = [0] * num_envs
    self.t_ep = [0] * num_envs
    self.done = [True] * num_envs
    self.cur = [-1] * num_envs
    self._keep = {}  # env -> the generator batch its current state points into

  def advance(self, env, action=None):
    """-> RawState expected for env's next row (action None = forced reset)."""
    reset = action is None or self.done[env]
    self.cur[env] = 0 if reset else self.cur[env] + 1
    sb = SynthBatch(self.cfg, self.table, 1, seed=self.seed, first=env * 1000003 + self.calls[env])
    self._keep[env] = sb
    self.calls[env] += 1
    self.t_ep[env] = 0 if reset else self.t_ep[env] + 1
    raw = sb.raw[0]
    raw.done = int(self.t_ep[env] >= 3 + env % 5)
    raw.win_reason = 0 if reset else int(action)
    raw.reward = 0.0 if reset else float(int(action) % 3)
    raw.env_id = env
    raw.elapsed_step = self.cur[env]
    self.done[env] = bool(raw.done)
    return raw
