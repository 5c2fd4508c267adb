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
