"""ctypes front-end of oracle/dummy_oracle.c — TEST INFRASTRUCTURE ONLY.

Mirrors the raw boundary of the reference (`_reset`, `_send`, `_recv` of
`_DummyEnvPool`, envpool2/core/py_envpool.h:279-287) so parity tests read like
calls on the reference's own pybind class.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STATE_KEYS = [
  "info:env_id", "info:players.env_id", "elapsed_step", "done", "reward", "discount",
  "step_type", "trunc", "obs:raw", "obs:dyn", "info:players.done", "info:players.id",
]


class _Batch(C.Structure):
  _fields_ = [
    ("num_rows", C.c_int32), ("num_player_rows", C.c_int32), ("state_num", C.c_int32),
    ("info_env_id", C.POINTER(C.c_int32)), ("info_players_env_id", C.POINTER(C.c_int32)),
    ("elapsed_step", C.POINTER(C.c_int32)), ("done", C.POINTER(C.c_uint8)),
    ("reward", C.POINTER(C.c_float)), ("discount", C.POINTER(C.c_float)),
    ("step_type", C.POINTER(C.c_int32)), ("trunc", C.POINTER(C.c_uint8)),
    ("obs_raw", C.POINTER(C.c_int32)), ("dyn_rows", C.POINTER(C.c_int32)),
    ("dyn_fill", C.POINTER(C.c_int32)), ("players_done", C.POINTER(C.c_uint8)),
    ("players_id", C.POINTER(C.c_int32)),
  ]


def lib():
  global _LIB
  if _LIB is None:
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
      subprocess.check_call(["make", "-s", "-C", _HERE])
    L = C.CDLL(path)
    L.od_create.restype = C.c_void_p
    L.od_create.argtypes = [C.c_int32] * 7
    L.od_destroy.argtypes = [C.c_void_p]
    L.od_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.od_send.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.od_recv.argtypes = [C.c_void_p, C.POINTER(_Batch)]
    L.od_batch_free.argtypes = [C.POINTER(_Batch)]
    L.od_bench_sync.restype = C.c_uint64
    L.od_bench_sync.argtypes = [C.c_void_p, C.c_int32]
    L.od_bench_sync_mt.restype = C.c_uint64
    L.od_bench_sync_mt.argtypes = [C.c_int32] * 4
    L.od_queue_size.argtypes = [C.c_void_p]
    L.od_is_sync.argtypes = [C.c_void_p]
    # dummy_pool_mt.c: the reference's queue + worker-thread architecture
    L.rp_create.restype = C.c_void_p
    L.rp_create.argtypes = [C.c_int32] * 7
    L.rp_destroy.argtypes = [C.c_void_p]
    L.rp_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.rp_send.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.rp_send_copy.argtypes = L.rp_send.argtypes
    L.rp_recv.argtypes = [C.c_void_p, C.POINTER(_Batch)]
    L.rp_batch_free.argtypes = [C.POINTER(_Batch)]
    L.rp_num_threads.argtypes = [C.c_void_p]
    L.rp_action_check_failed.argtypes = [C.c_void_p]
    L.rp_bench_sync.restype = C.c_uint64
    L.rp_bench_sync.argtypes = [C.c_int32] * 4
    _LIB = L
  return _LIB


def _arr(ptr, n, dtype, shape=None):
  if n == 0:
    a = np.zeros(0, dtype=dtype)
  else:
    a = np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).copy()
  return a.reshape(shape) if shape is not None else a


class OracleError(RuntimeError):
  def __init__(self, code):
    super().__init__(f"oracle status {code}")
    self.code = code


class DummyOracle:
  """Sequential restatement of `_DummyEnvPool` (one worker thread, FIFO)."""

  def __init__(self, num_envs=1, batch_size=0, max_num_players=1, seed=42,
               max_episode_steps=2**31 - 1, state_num=10, env_id_offset=0):
    self._L = lib()
    self._p = self._L.od_create(num_envs, batch_size, max_num_players, seed, max_episode_steps,
                                state_num, env_id_offset)
    if not self._p:
      raise ValueError("invalid oracle config")
    self.num_envs, self.state_num = num_envs, state_num

  def __del__(self):
    if getattr(self, "_p", None):
      self._L.od_destroy(self._p)
      self._p = None

  @property
  def queue_size(self):
    return self._L.od_queue_size(self._p)

  @property
  def is_sync(self):
    return bool(self._L.od_is_sync(self._p))

  def _reset(self, env_ids):
    ids = np.ascontiguousarray(env_ids, dtype=np.int32)
    return self._L.od_reset(self._p, ids.ctypes.data, ids.shape[0])

  def _send(self, action):
    """action = [env_id, players.env_id, list_action, players.action, players.id]."""
    env_id = np.ascontiguousarray(action[0], dtype=np.int32)
    penv = np.ascontiguousarray(action[1], dtype=np.int32)
    la = np.ascontiguousarray(action[2], dtype=np.float64)
    assert la.shape == (env_id.shape[0], 6)
    return self._L.od_send(self._p, env_id.shape[0], env_id.ctypes.data, penv.shape[0],
                           penv.ctypes.data, la.ctypes.data)

  def _recv(self, dyn="lazy"):
    b = _Batch()
    rc = self._L.od_recv(self._p, C.byref(b))
    if rc != 0:
      raise OracleError(rc)
    n, m, s = b.num_rows, b.num_player_rows, b.state_num
    dyn_rows = _arr(b.dyn_rows, m, np.int32)
    dyn_fill = _arr(b.dyn_fill, m, np.int32)
    if dyn == "eager":
      dyn_arr = np.empty(m, dtype=object)
      for i in range(m):
        dyn_arr[i] = None if dyn_rows[i] == 0 else np.full((dyn_rows[i], s), dyn_fill[i], np.int32)
    else:
      dyn_arr = (dyn_rows, dyn_fill)
    out = [
      _arr(b.info_env_id, n, np.int32), _arr(b.info_players_env_id, m, np.int32),
      _arr(b.elapsed_step, n, np.int32), _arr(b.done, n, np.bool_),
      _arr(b.reward, m, np.float32), _arr(b.discount, m, np.float32),
      _arr(b.step_type, n, np.int32), _arr(b.trunc, n, np.bool_),
      _arr(b.obs_raw, m * s, np.int32, (m, s)), dyn_arr,
      _arr(b.players_done, m, np.bool_), _arr(b.players_id, m, np.int32),
    ]
    self._L.od_batch_free(C.byref(b))
    return out

  def bench_sync(self, steps):
    return self._L.od_bench_sync(self._p, steps)


def _unpack(b, free, dyn="lazy"):
  n, m, s = b.num_rows, b.num_player_rows, b.state_num
  dyn_rows = _arr(b.dyn_rows, m, np.int32)
  dyn_fill = _arr(b.dyn_fill, m, np.int32)
  out = [
    _arr(b.info_env_id, n, np.int32), _arr(b.info_players_env_id, m, np.int32),
    _arr(b.elapsed_step, n, np.int32), _arr(b.done, n, np.bool_),
    _arr(b.reward, m, np.float32), _arr(b.discount, m, np.float32),
    _arr(b.step_type, n, np.int32), _arr(b.trunc, n, np.bool_),
    _arr(b.obs_raw, m * s, np.int32, (m, s)), (dyn_rows, dyn_fill),
    _arr(b.players_done, m, np.bool_), _arr(b.players_id, m, np.int32),
  ]
  free(C.byref(b))
  return out


class ThreadedDummyOracle:
  """dummy_pool_mt.c: ActionBufferQueue -> worker threads -> StateBufferQueue, as the reference
  runs it. `_recv` blocks like the reference's. `_send` hands the pool its own copy of the action
  arrays, released with the batch's last reference (envpool2/core/env.h:106-110 and
  py_envpool.h:88-100 hold and release the numpy arrays the same way); keeping them alive on the
  Python side "for long enough" is what a descheduled worker can outlast."""

  def __init__(self, num_envs=1, batch_size=0, max_num_players=1, seed=42,
               max_episode_steps=2**31 - 1, state_num=10, num_threads=0):
    self._L = lib()
    self._p = self._L.rp_create(num_envs, batch_size, max_num_players, seed, max_episode_steps,
                                state_num, num_threads)
    if not self._p:
      raise ValueError("invalid oracle config")
    self.num_envs, self.state_num = num_envs, state_num

  def close(self):
    if getattr(self, "_p", None):
      self._L.rp_destroy(self._p)
      self._p = None

  __del__ = close

  @property
  def num_threads(self):
    return self._L.rp_num_threads(self._p)

  @property
  def action_check_failed(self):
    return bool(self._L.rp_action_check_failed(self._p))

  def _reset(self, env_ids):
    ids = np.ascontiguousarray(env_ids, dtype=np.int32)
    self._L.rp_reset(self._p, ids.ctypes.data, ids.shape[0])

  def _send(self, action):
    env_id = np.ascontiguousarray(action[0], dtype=np.int32)
    penv = np.ascontiguousarray(action[1], dtype=np.int32)
    la = np.ascontiguousarray(action[2], dtype=np.float64)
    assert la.shape == (env_id.shape[0], 6)
    self._L.rp_send_copy(self._p, env_id.shape[0], env_id.ctypes.data, penv.shape[0], penv.ctypes.data,
                    la.ctypes.data)

  def _recv(self):
    b = _Batch()
    self._L.rp_recv(self._p, C.byref(b))
    return _unpack(b, self._L.rp_batch_free)


def bench_sync_threadpool(num_envs, seed, steps, threads):
  """bench.py's second cpu_baseline figure: the reference's architecture (action queue, worker
  threads, state-buffer queue with fresh zeroed buffers) at `threads` workers."""
  return lib().rp_bench_sync(num_envs, seed, steps, threads)


def bench_sync_mt(num_envs, seed, steps, threads):
  """bench.py's multi-threaded cpu_baseline leg: `threads` independent shards of one sync pool."""
  return lib().od_bench_sync_mt(num_envs, seed, steps, threads)
