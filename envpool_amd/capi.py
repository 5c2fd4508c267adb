"""ctypes binding of include/envpool_amd.h — the C ABI exactly as a foreign caller sees it.

Used by the parity tests and bench.py so that they exercise the drop-in boundary itself,
not a Python convenience layer.
"""
import ctypes as C

import numpy as np

from ._loader import load_library

EP_OK, EP_ERR_INVALID, EP_ERR_HIP, EP_ERR_NO_DEVICE = 0, 1, 2, 3
EP_ERR_WOULD_BLOCK, EP_ERR_RING_FULL, EP_ERR_ACTION_CHECK, EP_ERR_TOO_LARGE, EP_ERR_DUP_ENV = 4, 5, 6, 7, 8
EP_MEM_HOST, EP_MEM_DEVICE = 0, 1
EP_I32, EP_F32, EP_F64, EP_BOOL, EP_U8, EP_CONTAINER_I32 = range(6)
EP_MAX_KEYS, EP_MAX_NDIM = 20, 4

_NP = {EP_I32: np.int32, EP_F32: np.float32, EP_F64: np.float64, EP_BOOL: np.bool_, EP_U8: np.uint8}


class DummyConfig(C.Structure):
  _fields_ = [(k, C.c_int32) for k in (
    "num_envs", "batch_size", "num_threads", "max_num_players", "thread_affinity_offset", "seed",
    "max_episode_steps", "state_num", "action_num", "device_id", "env_id_offset", "ring_depth")]


class Array(C.Structure):
  _fields_ = [("data", C.c_void_p), ("shape", C.c_int64 * EP_MAX_NDIM), ("ndim", C.c_int32),
              ("dtype", C.c_int32)]


class Batch(C.Structure):
  _fields_ = [("num_keys", C.c_int32), ("mem", C.c_int32), ("num_rows", C.c_int32),
              ("num_player_rows", C.c_int32), ("seq", C.c_uint64), ("owner", C.c_void_p),
              ("arrays", Array * EP_MAX_KEYS)]


SYMBOLS = [
  "ep_dummy_pool_create", "ep_pool_destroy", "ep_pool_reset", "ep_pool_send", "ep_pool_recv",
  "ep_pool_recv_many", "ep_pool_recv_on_stream", "ep_pool_set_recv_timeout", "ep_pool_episode_stats_async",
  "ep_batch_release", "ep_pool_rollout", "ep_pool_sync", "ep_pool_episode_stats",
  "ep_pool_checkpoint_bytes", "ep_pool_checkpoint_save", "ep_pool_checkpoint_restore",
  "ep_pool_num_envs", "ep_pool_batch_size", "ep_pool_ring_depth", "ep_pool_is_sync",
  "ep_pool_stream", "ep_pool_set_timing", "ep_pool_kernel_time_us", "ep_device_count",
  "ep_last_error", "ep_version", "ep_dummy_send_noalloc", "ep_pool_step", "ep_pool_last_rollout_kernel",
  "ep_pool_step_begin", "ep_pool_step_end", "ep_pool_server_stats", "ep_pool_server_trace", "ep_pool_recv_many_on_stream",
  "ep_pool_step_on_stream", "ep_pool_send_on_stream", "ep_pool_poll_errors", "ep_pool_set_stream", "ep_terminated_many", "ep_stream_synchronize", "ep_pool_terminated_of",
  "ep_host_blocks_create", "ep_host_blocks_take", "ep_host_blocks_give", "ep_host_blocks_destroy", "ep_pool_host_target",
]

_bound = None


def lib():
  global _bound
  if _bound is None:
    L = load_library()
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.ep_dummy_pool_create.argtypes = [C.POINTER(DummyConfig), C.POINTER(vp)]
    L.ep_pool_destroy.argtypes = [vp]
    L.ep_pool_reset.argtypes = [vp, vp, i32, i32]
    L.ep_pool_send.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i64), i32]
    L.ep_pool_recv.argtypes = [vp, i32, C.POINTER(Batch)]
    L.ep_pool_recv_many.argtypes = [vp, i32, i32, C.POINTER(Batch)]
    L.ep_pool_recv_on_stream.argtypes = [vp, vp, C.POINTER(Batch)]
    L.ep_pool_recv_many_on_stream.argtypes = [vp, vp, i32, C.POINTER(Batch)]
    L.ep_pool_set_recv_timeout.argtypes = [vp, i64]
    L.ep_pool_step.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(Batch)]
    L.ep_pool_step_begin.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i64)]
    L.ep_pool_step_end.argtypes = [vp, C.POINTER(Batch)]
    L.ep_pool_step_on_stream.argtypes = [vp, vp, i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(Batch)]
    L.ep_pool_poll_errors.argtypes = [vp]
    L.ep_pool_set_stream.argtypes = [vp, vp, i32]
    L.ep_stream_synchronize.argtypes = [i32, vp]
    L.ep_pool_terminated_of.argtypes = [vp, C.POINTER(Batch), C.POINTER(vp)]
    L.ep_terminated_many.argtypes = [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), i32, vp]
    L.ep_pool_send_on_stream.argtypes = [vp, vp, i32, C.POINTER(vp), C.POINTER(i64)]
    L.ep_host_blocks_create.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.ep_host_blocks_take.argtypes = [vp, C.POINTER(vp)]
    L.ep_host_blocks_give.argtypes = [vp, vp]
    L.ep_host_blocks_destroy.argtypes = [vp]
    L.ep_pool_host_target.argtypes = [vp, vp, vp, C.POINTER(C.c_size_t), i64]
    L.ep_pool_server_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(i64)]
    L.ep_pool_last_rollout_kernel.argtypes = [vp]
    L.ep_pool_last_rollout_kernel.restype = C.c_char_p
    L.ep_pool_episode_stats_async.argtypes = [vp, vp, vp, vp, vp]
    L.ep_batch_release.argtypes = [C.POINTER(Batch)]
    L.ep_pool_rollout.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]
    L.ep_pool_sync.argtypes = [vp]
    L.ep_pool_episode_stats.argtypes = [vp, vp, vp, vp, i32]
    L.ep_pool_checkpoint_bytes.argtypes = [vp]
    L.ep_pool_checkpoint_bytes.restype = i64
    L.ep_pool_checkpoint_save.argtypes = [vp, vp]
    L.ep_pool_checkpoint_restore.argtypes = [vp, vp]
    for f in ("ep_pool_num_envs", "ep_pool_batch_size", "ep_pool_ring_depth", "ep_pool_is_sync"):
      getattr(L, f).argtypes = [vp]
    L.ep_pool_stream.argtypes = [vp]
    L.ep_pool_stream.restype = vp
    L.ep_pool_set_timing.argtypes = [vp, i32]
    L.ep_pool_kernel_time_us.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64), i32]
    L.ep_dummy_send_noalloc.argtypes = [C.POINTER(DummyConfig), vp, i32, vp, i32, vp, vp, i32, C.POINTER(Array), i32, vp, vp]
    L.ep_last_error.restype = C.c_char_p
    L.ep_version.restype = C.c_char_p
    _bound = L
  return _bound


class EpError(RuntimeError):
  def __init__(self, code, msg):
    super().__init__(f"[ep_status {code}] {msg}")
    self.code = code


def _check(rc, allow=()):
  if rc != EP_OK and rc not in allow:
    raise EpError(rc, lib().ep_last_error().decode())
  return rc


def batch_to_numpy(b: Batch):
  """Host batch -> list of numpy copies in _state_keys order (obs:dyn -> None)."""
  out = []
  for k in range(b.num_keys):
    a = b.arrays[k]
    if a.dtype == EP_CONTAINER_I32:
      out.append(None)
      continue
    shape = tuple(a.shape[i] for i in range(a.ndim))
    n = int(np.prod(shape))
    dt = np.dtype(_NP[a.dtype])
    if n == 0:
      out.append(np.zeros(shape, dt))
    else:
      buf = (C.c_char * (n * dt.itemsize)).from_address(a.data)
      out.append(np.frombuffer(buf, dtype=dt).reshape(shape).copy())
  return out


_TYPESTR = None


def batch_to_descs(b: Batch):
  """Device batch -> list of (ptr, shape, typestr) descriptors in key order (obs:dyn -> None), the
  form envpool_amd.interop.as_torch takes."""
  global _TYPESTR
  if _TYPESTR is None:
    _TYPESTR = {k: np.dtype(v).str for k, v in _NP.items()}
  out = []
  for k in range(b.num_keys):
    a = b.arrays[k]
    if a.dtype == EP_CONTAINER_I32:
      out.append(None)
      continue
    out.append((a.data or 0, tuple(a.shape[i] for i in range(a.ndim)), _TYPESTR[a.dtype]))
  return out


class DummyPoolC:
  """Thin object wrapper over the raw C entry points (host-memory flavour)."""

  def __init__(self, **kw):
    cfg = DummyConfig(num_envs=1, batch_size=0, num_threads=0, max_num_players=1,
                      thread_affinity_offset=-1, seed=42, max_episode_steps=2**31 - 1, state_num=10,
                      action_num=6, device_id=0, env_id_offset=0, ring_depth=0)
    for k, v in kw.items():
      setattr(cfg, k, v)
    self.cfg = cfg
    self._L = lib()
    h = C.c_void_p()
    _check(self._L.ep_dummy_pool_create(C.byref(cfg), C.byref(h)))
    self._h = h

  def set_recv_timeout(self, ms):
    _check(self._L.ep_pool_set_recv_timeout(self._h, ms))

  def close(self):
    if getattr(self, "_h", None):
      self._L.ep_pool_destroy(self._h)
      self._h = None

  __del__ = close

  @property
  def handle(self):
    return self._h

  def _reset(self, env_ids):
    ids = np.ascontiguousarray(env_ids, dtype=np.int32)
    _check(self._L.ep_pool_reset(self._h, ids.ctypes.data, ids.shape[0], EP_MEM_HOST))

  def _send(self, action):
    arrs = [np.ascontiguousarray(action[0], np.int32), np.ascontiguousarray(action[1], np.int32),
            np.ascontiguousarray(action[2], np.float64), np.ascontiguousarray(action[3], np.int32),
            np.ascontiguousarray(action[4], np.int32)]
    ptrs = (C.c_void_p * 5)(*[a.ctypes.data for a in arrs])
    dims = (C.c_int64 * 5)(*[a.shape[0] for a in arrs])
    _check(self._L.ep_pool_send(self._h, 5, ptrs, dims, EP_MEM_HOST))

  def _recv(self, allow=()):
    b = Batch()
    rc = _check(self._L.ep_pool_recv(self._h, EP_MEM_HOST, C.byref(b)), allow=allow)
    out = batch_to_numpy(b)
    self._L.ep_batch_release(C.byref(b))
    self.last_status = rc
    return out

  # device-memory flavours of the two calls above, for tests and soaks: the arrays travel through
  # plain hipMalloc'd memory (no torch needed), results come back as numpy copies of the views
  def _send_via_device(self, action):
    arrs = [np.ascontiguousarray(action[0], np.int32), np.ascontiguousarray(action[1], np.int32),
            np.ascontiguousarray(action[2], np.float64), np.ascontiguousarray(action[3], np.int32),
            np.ascontiguousarray(action[4], np.int32)]
    L = self._L
    L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.hipFree.argtypes = [C.c_void_p]
    dptr = []
    for a in arrs:
      d = C.c_void_p()
      assert L.hipMalloc(C.byref(d), max(256, a.nbytes)) == 0
      if a.nbytes:
        assert L.hipMemcpy(d, a.ctypes.data, a.nbytes, 1) == 0  # host to device (synchronous)
      dptr.append(d)
    ptrs = (C.c_void_p * 5)(*[d.value for d in dptr])
    dims = (C.c_int64 * 5)(*[a.shape[0] for a in arrs])
    try:
      _check(L.ep_pool_send(self._h, 5, ptrs, dims, EP_MEM_DEVICE))
    finally:
      _check(L.ep_pool_sync(self._h))  # the kernels have read the arrays: they can go
      for d in dptr:
        L.hipFree(d)

  def _recv_via_device(self, allow=()):
    L = self._L
    L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    b = Batch()
    rc = _check(L.ep_pool_recv(self._h, EP_MEM_DEVICE, C.byref(b)), allow=allow)
    self.last_status = rc
    out = []
    for k in range(b.num_keys):
      a = b.arrays[k]
      if a.dtype == EP_CONTAINER_I32:
        out.append(None)
        continue
      shape = tuple(a.shape[i] for i in range(a.ndim))
      h = np.zeros(shape, np.dtype(_NP[a.dtype]))
      if h.nbytes:
        assert L.hipMemcpy(h.ctypes.data, a.data, h.nbytes, 2) == 0  # device to host
      out.append(h)
    return out

  def _step_on_stream_via_device(self, action, stream=None, adopt=False):
    """ep_pool_step_on_stream with host arrays brought to the device first (tests, soaks): the
    batch it hands out, copied back after `stream` (None: the pool's step stream) has been waited
    for. adopt=True makes `stream` (None: the legacy default stream) the pool's step stream first
    (ep_pool_set_stream)."""
    arrs = [np.ascontiguousarray(action[0], np.int32), np.ascontiguousarray(action[1], np.int32),
            np.ascontiguousarray(action[2], np.float64), np.ascontiguousarray(action[3], np.int32),
            np.ascontiguousarray(action[4], np.int32)]
    L = self._L
    L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.hipFree.argtypes = [C.c_void_p]
    dptr = []
    for a in arrs:
      d = C.c_void_p()
      assert L.hipMalloc(C.byref(d), max(256, a.nbytes)) == 0
      if a.nbytes:
        assert L.hipMemcpy(d, a.ctypes.data, a.nbytes, 1) == 0
      dptr.append(d)
    ptrs = (C.c_void_p * 5)(*[d.value for d in dptr])
    dims = (C.c_int64 * 5)(*[a.shape[0] for a in arrs])
    b = Batch()
    out = []
    try:
      if adopt:
        _check(L.ep_pool_set_stream(self._h, stream, 0))
      _check(L.ep_pool_step_on_stream(self._h, stream, 5, ptrs, dims, C.byref(b)))
      _check(L.ep_stream_synchronize(-1, stream if stream is not None else L.ep_pool_stream(self._h)))
      for k in range(b.num_keys):
        a = b.arrays[k]
        if a.dtype == EP_CONTAINER_I32:
          out.append(None)
          continue
        h = np.zeros(tuple(a.shape[i] for i in range(a.ndim)), np.dtype(_NP[a.dtype]))
        if h.nbytes:
          assert L.hipMemcpy(h.ctypes.data, a.data, h.nbytes, 2) == 0
        out.append(h)
      # the step kernel's `terminated` array beside the slot (ep_pool_terminated_of), kept for callers that check it
      term = C.c_void_p()
      self.last_terminated = None
      if L.ep_pool_terminated_of(self._h, C.byref(b), C.byref(term)) == EP_OK:
        t = np.zeros(b.num_rows, np.uint8)
        if t.nbytes:
          assert L.hipMemcpy(t.ctypes.data, term, t.nbytes, 2) == 0
        self.last_terminated = t
    finally:
      _check(L.ep_pool_sync(self._h))
      for d in dptr:
        L.hipFree(d)
    return out

  def sync(self):
    _check(self._L.ep_pool_sync(self._h))

  def server_stats(self):
    """(kernels launched, idle stops, steps served, idle limit in us) of the resident step server."""
    a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int64()
    _check(self._L.ep_pool_server_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
    return a.value, b.value, c.value, d.value

  def episode_stats(self):
    n = self._L.ep_pool_num_envs(self._h)
    r, l, c = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    _check(self._L.ep_pool_episode_stats(self._h, r.ctypes.data, l.ctypes.data, c.ctypes.data, EP_MEM_HOST))
    return r, l, c
