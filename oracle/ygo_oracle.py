"""ctypes front-end of oracle/ygo_oracle.c — TEST INFRASTRUCTURE ONLY (parity unpinned, see the
header of the C file)."""
import ctypes as C

import numpy as np

from envpool_amd.capi_ygo import CardData, RawState, YgoConfig
from oracle.dummy_oracle import lib as _lib

_bound = False


def lib():
  global _bound
  L = _lib()
  if not _bound:
    vp = C.c_void_p
    L.yo_write_batch.argtypes = [C.POINTER(YgoConfig), C.POINTER(CardData), C.c_int32, C.POINTER(RawState), C.c_int32] + [vp] * 18
    L.yo_history_push.argtypes = [C.POINTER(YgoConfig), vp, C.POINTER(C.c_int32), C.c_int32, C.c_char_p, vp, C.c_int32]
    _bound = True
  return L


def write_batch(cfg, table, raw_states, n):
  """Oracle WriteState for n reference-form states -> (list of arrays in YGO_STATE_KEYS order, status[n])."""
  L = lib()
  nf = 9 + 2 * cfg.max_multi_select
  m = max(n, 1)
  i32 = lambda: np.zeros(m, np.int32)
  outs = [i32(), i32(), i32(), np.zeros(m, np.uint8), np.zeros(m, np.float32), np.zeros(m, np.float32), i32(),
          np.zeros(m, np.uint8), np.zeros((m, 2 * cfg.max_cards, 39), np.uint8), np.zeros((m, 8), np.uint8),
          np.zeros((m, cfg.max_options, nf), np.uint8), np.zeros((m, cfg.n_history_actions, nf), np.uint8),
          i32(), i32(), i32(), i32(), np.zeros((m, cfg.max_options, cfg.max_multi_select), np.uint16)]
  status = np.zeros(m, np.int32)
  L.yo_write_batch(C.byref(cfg), table, len(table), raw_states, n, *[o.ctypes.data for o in outs], status.ctypes.data)
  outs[3] = outs[3].view(np.bool_)
  outs[7] = outs[7].view(np.bool_)
  # the additive aux:action_mask key is, by definition, the mask the reference leaves implicit:
  # row i of obs:actions_ is an option iff i < info:num_options
  outs.append(np.arange(cfg.max_options, dtype=np.int32)[None, :] < outs[12][:, None])
  return [o[:n] for o in outs], status[:n]


def history_push(cfg, ring: np.ndarray, ha_p: int, msg: int, option: str, card_ids):
  L = lib()
  p = C.c_int32(ha_p)
  ids = np.ascontiguousarray(card_ids, dtype=np.uint16)
  rc = L.yo_history_push(C.byref(cfg), ring.ctypes.data, C.byref(p), msg, option.encode(), ids.ctypes.data, ids.shape[0])
  return rc, p.value
