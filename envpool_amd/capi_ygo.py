"""ctypes binding of include/envpool_amd_ygo.h (the YGOPro observation-encode C ABI)."""
import ctypes as C

import numpy as np

from . import capi
from ._loader import load_library

EP_U16 = 6
YGO_STATE_KEYS = [
  "info:env_id", "info:players.env_id", "elapsed_step", "done", "reward", "discount", "step_type", "trunc",
  "obs:cards_", "obs:global_", "obs:actions_", "obs:h_actions_", "info:num_options", "info:to_play",
  "info:is_selfplay", "info:win_reason", "aux:h_card_ids", "aux:action_mask",
]

capi._NP[EP_U16] = np.uint16


class YgoConfig(C.Structure):
  _fields_ = [(k, C.c_int32) for k in ("max_cards", "max_options", "n_history_actions", "max_multi_select",
                                       "max_episode_steps", "device_id", "ring_depth", "max_batch")]


def default_config(**kw):
  cfg = YgoConfig(max_cards=75, max_options=16, n_history_actions=16, max_multi_select=5,
                  max_episode_steps=2**31 - 1, device_id=0, ring_depth=4, max_batch=1024)
  for k, v in kw.items():
    setattr(cfg, k, v)
  return cfg


class CardData(C.Structure):
  _fields_ = [("code", C.c_uint32), ("type", C.c_uint32), ("level", C.c_uint32), ("attribute", C.c_uint32),
              ("race", C.c_uint32), ("attack", C.c_int32), ("defense", C.c_int32), ("card_id", C.c_uint16),
              ("pad", C.c_uint16)]


class CoreConstants(C.Structure):
  """ep_ygo_core_constants (include/envpool_amd_ygo_consts.h)."""
  _fields_ = [("location", C.c_int32 * 8), ("pos", C.c_int32 * 8), ("type", C.c_uint32 * 25), ("attribute", C.c_uint32 * 7),
              ("race", C.c_uint32 * 26), ("phase", C.c_int32 * 10), ("msg", C.c_int32 * 14), ("query_equip_card", C.c_uint32),
              ("len_empty", C.c_int32)]


class RawState(C.Structure):
  _fields_ = [
    ("to_play", C.c_int32), ("turn_player", C.c_int32), ("msg", C.c_int32), ("current_phase", C.c_int32),
    ("is_selfplay", C.c_int32), ("win_reason", C.c_int32), ("lp", C.c_int32 * 2), ("reward", C.c_float),
    ("env_id", C.c_int32), ("elapsed_step", C.c_int32), ("done", C.c_int32), ("n_options", C.c_int32),
    ("options", C.POINTER(C.c_char_p)), ("n_revealed", C.c_int32), ("revealed", C.POINTER(C.c_char_p)),
    ("field_count", (C.c_int32 * 7) * 2), ("field_buf", (C.c_void_p * 7) * 2), ("field_len", (C.c_int32 * 7) * 2),
    ("history", C.c_void_p * 2), ("ha_p", C.c_int32 * 2),
  ]


SYMBOLS = [
  "ep_ygo_encoder_create", "ep_ygo_encoder_destroy", "ep_ygo_set_card_table", "ep_ygo_record_stride",
  "ep_ygo_action_feats", "ep_ygo_featurize", "ep_ygo_submit", "ep_ygo_collect", "ep_ygo_acquire_slot",
  "ep_ygo_commit_slot", "ep_ygo_sync", "ep_ygo_stream", "ep_ygo_set_timing", "ep_ygo_kernel_time_us",
  "ep_ygo_emit_record", "ep_ygo_emit_records", "ep_ygo_history_push", "ep_ygo_set_zero_copy", "ep_ygo_submit_device",
  "ep_ygo_async_begin", "ep_ygo_async_end", "ep_ygo_alloc_record", "ep_ygo_record_done", "ep_ygo_recv", "ep_ygo_set_recv_timeout",
  "ep_ygo_ring_stats", "ep_ygo_set_packed", "ep_ygo_emit_arrival", "ep_ygo_put_arrival", "ep_ygo_record_used_bytes",
  "ep_ygo_commit_slot_packed", "ep_ygo_commit_slot_rows", "ep_ygo_emit_records_packed",
  "ep_ygo_pool_create", "ep_ygo_pool_destroy", "ep_ygo_pool_send", "ep_ygo_pool_reset", "ep_ygo_pool_recv",
  "ep_ygo_last_env_ids", "ep_ygo_pool_last_env_ids", "ep_ygo_pool_set_recv_timeout", "ep_ygo_pool_encoder", "ep_ygo_pool_num_threads", "ep_ygo_pool_set_env_id_offset",
  "ep_ygo_default_constants", "ep_ygo_check_constants",
]

# the duel-engine callbacks of ep_ygo_pool_create
ENGINE_RESET_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(RawState))
ENGINE_STEP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(RawState))


class Engine(C.Structure):
  """ep_ygo_engine. `reset` / `step` may be addresses of native functions (C.cast(fn, c_void_p).value)
  or ENGINE_*_FN objects wrapping Python callables (keep those alive as long as the pool)."""
  _fields_ = [("ctx", C.c_void_p), ("reset", C.c_void_p), ("step", C.c_void_p)]

_bound = None


def lib():
  global _bound
  if _bound is None:
    capi.lib()
    L = load_library()
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.ep_ygo_encoder_create.argtypes = [C.POINTER(YgoConfig), C.POINTER(vp)]
    L.ep_ygo_encoder_destroy.argtypes = [vp]
    L.ep_ygo_set_card_table.argtypes = [vp, C.POINTER(CardData), i32]
    L.ep_ygo_record_stride.argtypes = [C.POINTER(YgoConfig)]
    L.ep_ygo_record_stride.restype = i64
    L.ep_ygo_action_feats.argtypes = [C.POINTER(YgoConfig)]
    L.ep_ygo_featurize.argtypes = [vp, vp, i32, i32, i32, C.POINTER(capi.Batch)]
    L.ep_ygo_submit.argtypes = [vp, vp, i32]
    L.ep_ygo_submit_device.argtypes = [vp, vp, i32]
    L.ep_ygo_collect.argtypes = [vp, i32, C.POINTER(capi.Batch)]
    L.ep_ygo_acquire_slot.argtypes = [vp, C.POINTER(vp)]
    L.ep_ygo_commit_slot.argtypes = [vp, i32]
    L.ep_ygo_sync.argtypes = [vp]
    L.ep_ygo_stream.argtypes = [vp]
    L.ep_ygo_stream.restype = vp
    L.ep_ygo_set_timing.argtypes = [vp, i32]
    L.ep_ygo_set_zero_copy.argtypes = [vp, i32]
    L.ep_ygo_kernel_time_us.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64), i32]
    L.ep_ygo_emit_record.argtypes = [C.POINTER(YgoConfig), C.POINTER(RawState), vp]
    L.ep_ygo_emit_records.argtypes = [C.POINTER(YgoConfig), C.POINTER(RawState), i32, vp, vp]
    L.ep_ygo_async_begin.argtypes = [vp, i32]
    L.ep_ygo_async_end.argtypes = [vp]
    L.ep_ygo_alloc_record.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.ep_ygo_record_done.argtypes = [vp, C.c_uint64]
    L.ep_ygo_recv.argtypes = [vp, i32, C.POINTER(capi.Batch)]
    L.ep_ygo_set_recv_timeout.argtypes = [vp, i64]
    L.ep_ygo_set_packed.argtypes = [vp, i32]
    L.ep_ygo_emit_arrival.argtypes = [vp, C.POINTER(RawState)]
    L.ep_ygo_put_arrival.argtypes = [vp, vp]
    L.ep_ygo_record_used_bytes.argtypes = [C.POINTER(YgoConfig), vp]
    L.ep_ygo_record_used_bytes.restype = i64
    L.ep_ygo_commit_slot_packed.argtypes = [vp, i32, vp, i64]
    L.ep_ygo_commit_slot_rows.argtypes = [vp, i32, i32, i32, vp]
    L.ep_ygo_emit_records_packed.argtypes = [vp, C.POINTER(RawState), i32, vp]
    L.ep_ygo_ring_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ep_ygo_pool_create.argtypes = [C.POINTER(YgoConfig), C.POINTER(Engine), i32, i32, i32, i32, C.POINTER(CardData), i32,
                                     C.POINTER(vp)]
    L.ep_ygo_pool_destroy.argtypes = [vp]
    L.ep_ygo_pool_send.argtypes = [vp, i32, vp, vp]
    L.ep_ygo_pool_reset.argtypes = [vp, i32, vp]
    L.ep_ygo_pool_recv.argtypes = [vp, i32, C.POINTER(capi.Batch)]
    L.ep_ygo_pool_set_recv_timeout.argtypes = [vp, i64]
    L.ep_ygo_pool_encoder.argtypes = [vp]
    L.ep_ygo_pool_encoder.restype = vp
    L.ep_ygo_pool_num_threads.argtypes = [vp]
    L.ep_ygo_pool_last_env_ids.argtypes = [vp, vp, vp]
    L.ep_ygo_last_env_ids.argtypes = [vp, vp, vp]
    L.ep_ygo_pool_set_env_id_offset.argtypes = [vp, i32]
    L.ep_ygo_history_push.argtypes = [C.POINTER(YgoConfig), vp, C.POINTER(i32), i32, C.c_char_p, vp, i32]
    L.ep_ygo_default_constants.argtypes = [C.POINTER(CoreConstants)]
    L.ep_ygo_default_constants.restype = None
    L.ep_ygo_check_constants.argtypes = [C.POINTER(CoreConstants)]
    _bound = L
  return _bound


def emit_records(cfg, raw_states, n):
  """Reference-form states -> packed records (numpy uint8 [n, stride]) + per-record status."""
  L = lib()
  stride = L.ep_ygo_record_stride(C.byref(cfg))
  recs = np.zeros((max(n, 1), stride), dtype=np.uint8)
  status = np.zeros(max(n, 1), dtype=np.int32)
  L.ep_ygo_emit_records(C.byref(cfg), raw_states, n, recs.ctypes.data, status.ctypes.data)
  return recs[:n], status[:n]


class YgoEncoderC:
  def __init__(self, cfg, table, n_table):
    self._L = lib()
    self.cfg = cfg
    h = C.c_void_p()
    capi._check(self._L.ep_ygo_encoder_create(C.byref(cfg), C.byref(h)))
    self._h = h
    capi._check(self._L.ep_ygo_set_card_table(h, table, n_table))

  def close(self):
    if getattr(self, "_h", None):
      self._L.ep_ygo_encoder_destroy(self._h)
      self._h = None

  __del__ = close

  @property
  def handle(self):
    return self._h

  def featurize(self, records: np.ndarray, allow=()):
    """records: uint8 [n, stride] host array -> list of numpy arrays in YGO_STATE_KEYS order."""
    b = capi.Batch()
    rc = capi._check(self._L.ep_ygo_featurize(self._h, records.ctypes.data, records.shape[0], capi.EP_MEM_HOST,
                                              capi.EP_MEM_HOST, C.byref(b)), allow=allow)
    out = capi.batch_to_numpy(b)
    self._L.ep_batch_release(C.byref(b))
    self.last_status = rc
    return out

  def featurize_device(self, records_ptr: int, n: int, allow=()):
    """records already in device memory (uint8 [n, stride] at `records_ptr`) -> numpy arrays."""
    b = capi.Batch()
    rc = capi._check(self._L.ep_ygo_featurize(self._h, records_ptr, n, capi.EP_MEM_DEVICE, capi.EP_MEM_HOST,
                                              C.byref(b)), allow=allow)
    out = capi.batch_to_numpy(b)
    self._L.ep_batch_release(C.byref(b))
    self.last_status = rc
    return out

  def submit(self, records: np.ndarray):
    capi._check(self._L.ep_ygo_submit(self._h, records.ctypes.data, records.shape[0]))

  def collect(self, allow=()):
    b = capi.Batch()
    rc = capi._check(self._L.ep_ygo_collect(self._h, capi.EP_MEM_HOST, C.byref(b)), allow=allow)
    out = capi.batch_to_numpy(b)
    self._L.ep_batch_release(C.byref(b))
    self.last_status = rc
    return out


def make_engine(ctx, reset, step):
  """Engine struct from native function addresses (ints) or Python callables
  `reset(ctx, env_id, state_ptr) -> int`, `step(ctx, env_id, action, state_ptr) -> int`.
  Returns (Engine, keepalive): hold `keepalive` as long as the pool lives."""
  keep = []

  def addr(f, proto):
    if isinstance(f, int):
      return f
    cb = f if isinstance(f, proto) else proto(f)
    keep.append(cb)
    return C.cast(cb, C.c_void_p).value
  return Engine(ctx, addr(reset, ENGINE_RESET_FN), addr(step, ENGINE_STEP_FN)), keep


class YgoPoolC:
  """ep_ygo_pool: AsyncEnvPool<YGOProEnv> with the duel engine behind callbacks."""

  def __init__(self, cfg, engine, num_envs, batch_size, table, n_table, num_threads=0, thread_affinity_offset=-1):
    self._L = lib()
    self.cfg = cfg
    self._engine = engine  # (the struct is copied by the library; the callbacks it names must stay alive)
    h = C.c_void_p()
    capi._check(self._L.ep_ygo_pool_create(C.byref(cfg), C.byref(engine), num_envs, batch_size, num_threads,
                                           thread_affinity_offset, table, n_table, C.byref(h)))
    self._h = h

  def close(self):
    if getattr(self, "_h", None):
      self._L.ep_ygo_pool_destroy(self._h)
      self._h = None

  __del__ = close

  @property
  def handle(self):
    return self._h

  @property
  def encoder_handle(self):
    return C.c_void_p(self._L.ep_ygo_pool_encoder(self._h))

  def last_env_ids(self):
    ids, n = C.POINTER(C.c_int32)(), C.c_int32(0)
    capi._check(self._L.ep_ygo_pool_last_env_ids(self._h, C.byref(ids), C.byref(n)))
    return np.ctypeslib.as_array(ids, shape=(n.value,)).copy()

  @property
  def num_threads(self):
    return self._L.ep_ygo_pool_num_threads(self._h)

  def set_recv_timeout(self, ms):
    capi._check(self._L.ep_ygo_pool_set_recv_timeout(self._h, ms))

  def set_env_id_offset(self, offset):
    capi._check(self._L.ep_ygo_pool_set_env_id_offset(self._h, offset))

  def send(self, env_id, action):
    env_id = np.ascontiguousarray(env_id, dtype=np.int32)
    action = np.ascontiguousarray(action, dtype=np.int32)
    if action.shape != env_id.shape:
      raise ValueError(f"action has shape {action.shape}, env_id {env_id.shape}")
    capi._check(self._L.ep_ygo_pool_send(self._h, env_id.shape[0], env_id.ctypes.data, action.ctypes.data))

  def reset(self, env_id):
    env_id = np.ascontiguousarray(env_id, dtype=np.int32)
    capi._check(self._L.ep_ygo_pool_reset(self._h, env_id.shape[0], env_id.ctypes.data))

  def recv(self, allow=()):
    b = capi.Batch()
    rc = capi._check(self._L.ep_ygo_pool_recv(self._h, capi.EP_MEM_HOST, C.byref(b)), allow=allow)
    out = capi.batch_to_numpy(b)
    self._L.ep_batch_release(C.byref(b))
    self.last_status = rc
    return out

  def recv_device(self, allow=()):
    """The next batch as device views: (ptr, shape, typestr) per key, valid until the next recv."""
    b = capi.Batch()
    rc = capi._check(self._L.ep_ygo_pool_recv(self._h, capi.EP_MEM_DEVICE, C.byref(b)), allow=allow)
    self.last_status = rc
    return capi.batch_to_descs(b), b.num_rows
