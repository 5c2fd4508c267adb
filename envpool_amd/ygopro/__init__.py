"""YGOPro-v0: spec surface and the GPU observation-encode front-end.

The reference's `_YGOProEnvSpec` (envpool2/ygopro/ygopro.h:1177-1211, registered at
envpool2/ygopro/ygopro.cpp:4-10) fixes the config keys, the state keys/dtypes/shapes and the
action spec; `_YGOProEnvSpec` below reproduces that surface in the tuple forms of
envpool2/core/py_envpool.h:102-169. The duel engine (ygopro-core + Lua + SQLite) is NOT part of
this library, so there is no `_YGOProEnvPool` that can run duels: what replaces the reference's
`WriteState` is `YGOProEncoder`, which takes the host engine's per-env state (reference forms or
packed records) and returns the batch in `_state_keys` order, encoded on the GPU.
"""
import ctypes as C
from typing import Any, Dict, List, Sequence, Tuple

import numpy as np

_INT_MAX = 2**31 - 1
_INT_MIN = -(2**31)
_F32_MIN, _F32_MAX = float(np.finfo(np.float32).tiny), float(np.finfo(np.float32).max)

# common config (env_spec.h:25-30) ++ YGOProEnvFns::DefaultConfig (ygopro.h:1179-1186)
CONFIG_KEYS: List[str] = [
  "num_envs", "batch_size", "num_threads", "max_num_players", "thread_affinity_offset", "base_path", "seed",
  "gym_reset_return_info", "max_episode_steps", "deck1", "deck2", "player", "play_mode", "verbose",
  "max_options", "max_cards", "n_history_actions", "max_multi_select",
]
DEFAULT_CONFIG_VALUES: Tuple[Any, ...] = (
  1, 0, 0, 1, -1, "envpool2", 42, False, _INT_MAX, "OldSchool", "OldSchool", -1, "bot", False, 16, 75, 16, 5,
)
# common state keys (env_spec.h:36-42) then StateSpec (ygopro.h:1188-1201)
STATE_KEYS: List[str] = [
  "info:env_id", "info:players.env_id", "elapsed_step", "done", "reward", "discount", "step_type", "trunc",
  "obs:cards_", "obs:global_", "obs:actions_", "obs:h_actions_", "info:num_options", "info:to_play",
  "info:is_selfplay", "info:win_reason",
]
ACTION_KEYS: List[str] = ["env_id", "players.env_id", "action"]


from .._loader import load_library  # noqa: E402

load_library()  # fail loudly before the pybind module tries to resolve its symbols

from .ygopro_envpool import _YGOProEnvSpec, _YGOProNativePool  # noqa: E402

assert list(_YGOProEnvSpec._config_keys) == CONFIG_KEYS and list(_YGOProEnvSpec._state_keys) == STATE_KEYS


def _conf_of(spec) -> Dict[str, Any]:
  return dict(zip(CONFIG_KEYS, spec._config_values))


from ..python.spec import make_spec_class  # noqa: E402

YGOProEnvSpec = make_spec_class("YGOProEnvSpec", _YGOProEnvSpec)


class YGOProEncoder:
  """GPU replacement of YGOProEnv::WriteState for a batch of envs (ygopro.h:1828-1893).

  `card_table`: sequence of (code, type, level, attribute, race, attack, defense, card_id) — the
  reference's `cards_` / `card_ids_` after `init_module` (ygopro.h:1030-1064)."""

  def __init__(self, spec: _YGOProEnvSpec, card_table, device_id: int = 0, max_batch: int = 0, ring_depth: int = 4):
    from .. import capi_ygo  # needs the HIP library: fails loudly without it
    self._capi = capi_ygo
    c = _conf_of(spec)
    self.spec = spec
    self.cfg = capi_ygo.default_config(
      max_cards=c["max_cards"], max_options=c["max_options"], n_history_actions=c["n_history_actions"],
      max_multi_select=c["max_multi_select"], max_episode_steps=c["max_episode_steps"], device_id=device_id,
      ring_depth=ring_depth, max_batch=max_batch or c["batch_size"])
    rows = list(card_table)
    tab = (capi_ygo.CardData * max(1, len(rows)))()
    for i, r in enumerate(rows):
      tab[i] = capi_ygo.CardData(*[int(v) for v in r], 0)
    self._enc = capi_ygo.YgoEncoderC(self.cfg, tab, len(rows))

  def close(self):
    self._enc.close()

  def emit(self, raw_states, n: int) -> np.ndarray:
    """Reference-form states (capi_ygo.RawState array) -> packed records uint8 [n, stride]."""
    recs, status = self._capi.emit_records(self.cfg, raw_states, n)
    if (status != 0).any():
      raise ValueError(f"record {int(np.flatnonzero(status)[0])}: the reference throws while parsing this state")
    return recs

  def encode(self, records: np.ndarray, strict: bool = True) -> Dict[str, np.ndarray]:
    """Packed records -> {state key: array} in `_state_keys` order, plus `aux:h_card_ids` and the
    legal-action mask `mask` = arange(max_options) < info:num_options."""
    from .. import capi
    out = self._enc.featurize(np.ascontiguousarray(records), allow=() if strict else (capi.EP_ERR_ACTION_CHECK,))
    d = dict(zip(self._capi.YGO_STATE_KEYS, out))
    d["mask"] = d["aux:action_mask"]
    return d

  def submit(self, records: np.ndarray) -> None:
    self._enc.submit(np.ascontiguousarray(records))

  def collect(self, strict: bool = True) -> Dict[str, np.ndarray]:
    from .. import capi
    out = self._enc.collect(allow=() if strict else (capi.EP_ERR_ACTION_CHECK,))
    d = dict(zip(self._capi.YGO_STATE_KEYS, out))
    d["mask"] = d["aux:action_mask"]
    return d


_MODULE = {"engine_factory": None, "card_table": None, "device_id": 0, "env_id_offset": 0}


def init_module(engine_factory, card_table, device_id: int = 0, env_id_offset: int = 0) -> None:
  """Counterpart of the reference's `init_module(db_path, code_list_file, decks)`
  (envpool2/ygopro/ygopro.h:1030-1064, called once before any pool is made): tells this package
  where the duel engine and the card data come from, since neither is part of envpool_amd.

  `engine_factory(spec) -> (ctx, reset, step[, owner])`: `ctx` an integer / pointer handed back to
  the callbacks; `reset` / `step` addresses of native functions with the `ep_ygo_engine` signatures
  (include/envpool_amd_ygo.h) or Python callables (those hold the GIL: fine for tests, not for
  throughput); `owner` anything that must stay alive as long as the pool.
  `card_table`: sequence of (code, type, level, attribute, race, attack, defense, card_id) or a
  ctypes array of `capi_ygo.CardData` — `cards_` / `card_ids_` after the reference's init_module.
  `env_id_offset`: this process's pool owns the global env ids [offset, offset + num_envs) (one
  process per GPU, ids sharded in blocks)."""
  _MODULE.update(engine_factory=engine_factory, card_table=card_table, device_id=device_id, env_id_offset=env_id_offset)


class _YGOProEnvPool(_YGOProNativePool):
  """The raw boundary of the reference's pybind `_YGOProEnvPool` (py_envpool.h:279-287): `_spec`,
  `_state_keys`, `_action_keys`, `_send(list)`, `_recv() -> list`, `_reset(env_ids)` - the pybind
  class of envpool_amd/csrc/py_ygopro.cpp over `ep_ygo_pool` (ygo_pool.hip), constructed from a spec
  alone as the reference's is: the engine and the card table come from `init_module`. `_recv`
  returns the 16 arrays of `_state_keys` as numpy views of the pinned result block (no copy); the
  additive `aux:h_card_ids` of the last batch is kept in `last_h_card_ids`."""

  def __init__(self, spec, device_id=None, env_id_offset=None, shard=None):
    """`device_id`, `env_id_offset`, `shard`: set by the sharded front-end (make(..., num_shards=K)):
    this pool is shard `shard`, lives on GPU `device_id` and owns the global env ids from
    `env_id_offset` on. The engine factory is then called as
    `engine_factory(spec, env_id_offset=..., shard=..., device_id=...)` when it takes those keywords."""
    if _MODULE["engine_factory"] is None:
      raise RuntimeError(
        "YGOPro-v0: the duel engine (ygopro-core, Lua, SQLite card database) is not part of envpool_amd; "
        "call envpool_amd.ygopro.init_module(engine_factory, card_table) first (the reference's init_module "
        "plays the same role), or encode states of your own engine with envpool_amd.ygopro.YGOProEncoder")
    from .. import capi_ygo
    factory = _MODULE["engine_factory"]
    dev = _MODULE["device_id"] if device_id is None else device_id
    off = _MODULE["env_id_offset"] if env_id_offset is None else env_id_offset
    if shard is None:
      made = tuple(factory(spec))
    else:
      import inspect
      try:
        params = inspect.signature(factory).parameters
        takes = lambda k: k in params or any(p.kind is p.VAR_KEYWORD for p in params.values())  # noqa: E731
      except (TypeError, ValueError):
        takes = lambda k: False  # noqa: E731
      kw = {k: v for k, v in (("env_id_offset", off), ("shard", shard), ("device_id", dev)) if takes(k)}
      made = tuple(factory(spec, **kw))
    ctx, reset, step = made[:3]
    engine, keep = capi_ygo.make_engine(ctx, reset, step)
    table = _MODULE["card_table"]
    if not isinstance(table, C.Array):
      rows = list(table)
      tab = (capi_ygo.CardData * max(1, len(rows)))()
      for i, r in enumerate(rows):
        tab[i] = capi_ygo.CardData(*[int(v) for v in r], 0)
      table = tab
    super().__init__(spec, (engine.ctx or 0, engine.reset, engine.step), table, dev, off, (keep, made[3:], table))
    self.last_h_card_ids = self.last_action_mask = None

  def _recv(self) -> List[np.ndarray]:
    arrays, _err = self._recv_all()  # (_err: a state the reference throws on; its row is an empty observation)
    self.last_h_card_ids, self.last_action_mask = arrays[16], arrays[17]
    return arrays[:16]

  def _try_recv(self):
    """Additive: `_recv` that never waits for a batch that is not complete yet (None then)."""
    got = self._try_recv_all()
    if got is None:
      return None
    arrays, _err = got
    self.last_h_card_ids, self.last_action_mask = arrays[16], arrays[17]
    return arrays[:16]

  def _try_recv_device(self):
    got = self._try_recv_device_all()
    if got is None:
      return None
    descs, rows, prows = got
    self.last_h_card_ids, self.last_action_mask = descs[16], descs[17]
    return descs[:16], rows, prows

  def _recv_device(self):
    """Additive: the next batch as device views (descriptors for envpool_amd.interop.as_torch): the
    policy that consumes the observations lives on the GPU, the actions it picks go back through
    `_send` as host arrays (the duel engine is on the host)."""
    descs, rows, prows = self._recv_device_all()
    self.last_h_card_ids, self.last_action_mask = descs[16], descs[17]
    return descs[:16], rows, prows

  _fixed_rows = True  # (player rows == rows: a slot's views have the same shapes whenever it comes round)

  def _last_descs(self):
    """Descriptors of the batch `_recv_device_slot` handed out last (the 16 `_state_keys`); the two
    additive keys are noted for `last_h_card_ids` / `last_action_mask` as `_recv_device` does."""
    descs, rows, prows = self._last_descs_all()
    self.last_h_card_ids, self.last_action_mask = descs[16], descs[17]
    return descs[:16], rows, prows

  @property
  def last_env_ids(self) -> np.ndarray:
    """Additive: `info:env_id` of the batch received last as a host array, from host memory (the
    pool noted the ids as the rows arrived): `env.send(actions, env.last_env_ids)` after a
    `recv_torch` needs no read-back of the batch's own `info:env_id` from the device."""
    return self._last_env_ids()


from ..python.api import py_env, raw_env  # noqa: E402

_, YGOProDMEnvPool, YGOProGymEnvPool, YGOProGymnasiumEnvPool = py_env(_YGOProEnvSpec, _YGOProEnvPool)
YGOProRawEnvPool = raw_env(_YGOProEnvPool)

__all__ = ["YGOProEnvSpec", "YGOProEncoder", "init_module", "YGOProDMEnvPool", "YGOProGymEnvPool",
           "YGOProGymnasiumEnvPool", "YGOProRawEnvPool", "CONFIG_KEYS", "DEFAULT_CONFIG_VALUES", "STATE_KEYS", "ACTION_KEYS"]
