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


def _spec(dtype, shape, lo, hi):
  # (dtype, shape, (lo, hi), (elementwise lo, elementwise hi)) — py_envpool.h:102-125
  return (np.dtype(dtype), list(shape), (lo, hi), ([], []))


class _YGOProEnvSpec:
  """Same attributes as the pybind `_YGOProEnvSpec` (py_envpool.h:269-278)."""

  _config_keys = CONFIG_KEYS
  _default_config_values = DEFAULT_CONFIG_VALUES
  _state_keys = STATE_KEYS
  _action_keys = ACTION_KEYS

  def __init__(self, config_values: Sequence[Any] = DEFAULT_CONFIG_VALUES):
    if len(config_values) != len(CONFIG_KEYS):
      raise ValueError(f"_YGOProEnvSpec expects {len(CONFIG_KEYS)} config values, got {len(config_values)}")
    conf = dict(zip(CONFIG_KEYS, config_values))
    if conf["batch_size"] > conf["num_envs"]:  # env_spec.h:74-79
      raise ValueError(f"It is required that batch_size <= num_envs, got num_envs = {conf['num_envs']}, "
                       f"batch_size = {conf['batch_size']}")
    if conf["batch_size"] == 0:
      conf["batch_size"] = conf["num_envs"]
    for k in ("max_options", "max_cards", "n_history_actions", "max_multi_select"):
      if conf[k] < 1:
        raise ValueError(f"{k} must be >= 1")
    self._conf = conf
    self._config_values = tuple(conf[k] for k in CONFIG_KEYS)
    nf = 9 + 2 * conf["max_multi_select"]  # ygopro.h:1189
    self.n_action_feats = nf
    self._state_spec = (
      _spec(np.int32, [], _INT_MIN, _INT_MAX), _spec(np.int32, [-1], _INT_MIN, _INT_MAX),
      _spec(np.int32, [], _INT_MIN, _INT_MAX), _spec(np.bool_, [], False, True),
      _spec(np.float32, [-1], _F32_MIN, _F32_MAX), _spec(np.float32, [-1], 0.0, 1.0),
      _spec(np.int32, [], _INT_MIN, _INT_MAX), _spec(np.bool_, [], False, True),
      _spec(np.uint8, [conf["max_cards"] * 2, 39], 0, 255), _spec(np.uint8, [8], 0, 255),
      _spec(np.uint8, [conf["max_options"], nf], 0, 255), _spec(np.uint8, [conf["n_history_actions"], nf], 0, 255),
      _spec(np.int32, [], 0, conf["max_options"] - 1), _spec(np.int32, [], 0, 1), _spec(np.int32, [], 0, 1),
      _spec(np.int32, [], -1, 1),
    )
    self._action_spec = (
      _spec(np.int32, [], _INT_MIN, _INT_MAX), _spec(np.int32, [-1], _INT_MIN, _INT_MAX),  # env_spec.h:33-34
      _spec(np.int32, [], 0, conf["max_options"] - 1),  # ygopro.h:1205-1206
    )


from ..python.spec import make_spec_class  # noqa: E402

YGOProEnvSpec = make_spec_class("YGOProEnvSpec", _YGOProEnvSpec)


class YGOProEncoder:
  """GPU replacement of YGOProEnv::WriteState for a batch of envs (ygopro.h:1828-1893).

  `card_table`: sequence of (code, type, level, attribute, race, attack, defense, card_id) — the
  reference's `cards_` / `card_ids_` after `init_module` (ygopro.h:1030-1064)."""

  def __init__(self, spec: _YGOProEnvSpec, card_table, device_id: int = 0, max_batch: int = 0, ring_depth: int = 4):
    from .. import capi_ygo  # needs the HIP library: fails loudly without it
    self._capi = capi_ygo
    c = spec._conf
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
    d["mask"] = np.arange(self.cfg.max_options, dtype=np.int32)[None, :] < d["info:num_options"][:, None]
    return d

  def submit(self, records: np.ndarray) -> None:
    self._enc.submit(np.ascontiguousarray(records))

  def collect(self, strict: bool = True) -> Dict[str, np.ndarray]:
    from .. import capi
    out = self._enc.collect(allow=() if strict else (capi.EP_ERR_ACTION_CHECK,))
    d = dict(zip(self._capi.YGO_STATE_KEYS, out))
    d["mask"] = np.arange(self.cfg.max_options, dtype=np.int32)[None, :] < d["info:num_options"][:, None]
    return d


_MODULE = {"engine_factory": None, "card_table": None, "device_id": 0}


def init_module(engine_factory, card_table, device_id: int = 0) -> None:
  """Counterpart of the reference's `init_module(db_path, code_list_file, decks)`
  (envpool2/ygopro/ygopro.h:1030-1064, called once before any pool is made): tells this package
  where the duel engine and the card data come from, since neither is part of envpool_amd.

  `engine_factory(spec) -> (ctx, reset, step[, owner])`: `ctx` an integer / pointer handed back to
  the callbacks; `reset` / `step` addresses of native functions with the `ep_ygo_engine` signatures
  (include/envpool_amd_ygo.h) or Python callables (those hold the GIL: fine for tests, not for
  throughput); `owner` anything that must stay alive as long as the pool.
  `card_table`: sequence of (code, type, level, attribute, race, attack, defense, card_id) or a
  ctypes array of `capi_ygo.CardData` — `cards_` / `card_ids_` after the reference's init_module."""
  _MODULE.update(engine_factory=engine_factory, card_table=card_table, device_id=device_id)


class _YGOProEnvPool:
  """The raw boundary of the reference's pybind `_YGOProEnvPool` (py_envpool.h:279-287): `_spec`,
  `_state_keys`, `_action_keys`, `_send(list)`, `_recv() -> list`, `_reset(env_ids)`, over
  `ep_ygo_pool` (envpool_amd/csrc/ygo_pool.hip). `_recv` returns the 16 arrays of `_state_keys`;
  the additive `aux:h_card_ids` of the last batch is kept in `last_h_card_ids`."""

  _state_keys = STATE_KEYS
  _action_keys = ACTION_KEYS

  def __init__(self, spec: _YGOProEnvSpec):
    if _MODULE["engine_factory"] is None:
      raise RuntimeError(
        "YGOPro-v0: the duel engine (ygopro-core, Lua, SQLite card database) is not part of envpool_amd; "
        "call envpool_amd.ygopro.init_module(engine_factory, card_table) first (the reference's init_module "
        "plays the same role), or encode states of your own engine with envpool_amd.ygopro.YGOProEncoder")
    from .. import capi_ygo
    self._spec = spec
    c = dict(zip(CONFIG_KEYS, spec._config_values))
    made = tuple(_MODULE["engine_factory"](spec))
    ctx, reset, step = made[:3]
    self._owner = made[3:] or None
    engine, self._keep = capi_ygo.make_engine(ctx, reset, step)
    table = _MODULE["card_table"]
    if not isinstance(table, C.Array):
      rows = list(table)
      tab = (capi_ygo.CardData * max(1, len(rows)))()
      for i, r in enumerate(rows):
        tab[i] = capi_ygo.CardData(*[int(v) for v in r], 0)
      table = tab
    cfg = capi_ygo.default_config(
      max_cards=c["max_cards"], max_options=c["max_options"], n_history_actions=c["n_history_actions"],
      max_multi_select=c["max_multi_select"], max_episode_steps=c["max_episode_steps"],
      device_id=_MODULE["device_id"], ring_depth=0, max_batch=c["batch_size"])
    self._pool = capi_ygo.YgoPoolC(cfg, engine, c["num_envs"], c["batch_size"], table, len(table),
                                   num_threads=c["num_threads"], thread_affinity_offset=c["thread_affinity_offset"])
    self.last_h_card_ids = None

  def close(self) -> None:
    self._pool.close()

  def _send(self, action: List[np.ndarray]) -> None:
    # _action_keys order: env_id, players.env_id, action (env_spec.h:31-35, ygopro.h:1203-1207)
    self._pool.send(action[0], action[2])

  def _reset(self, env_ids: np.ndarray) -> None:
    self._pool.reset(env_ids)

  def _recv(self) -> List[np.ndarray]:
    from .. import capi
    out = self._pool.recv(allow=(capi.EP_ERR_ACTION_CHECK,))
    self.last_h_card_ids = out[16]
    return out[:16]

  def _recv_device(self):
    """Additive: the next batch as device views (descriptors for envpool_amd.interop.as_torch): the
    policy that consumes the observations lives on the GPU, the actions it picks go back through
    `_send` as host arrays (the duel engine is on the host)."""
    from .. import capi
    descs, rows = self._pool.recv_device(allow=(capi.EP_ERR_ACTION_CHECK,))
    self.last_h_card_ids = descs[16]
    return descs[:16], rows, rows


from ..python.api import py_env, raw_env  # noqa: E402

_, YGOProDMEnvPool, YGOProGymEnvPool, YGOProGymnasiumEnvPool = py_env(_YGOProEnvSpec, _YGOProEnvPool)
YGOProRawEnvPool = raw_env(_YGOProEnvPool)

__all__ = ["YGOProEnvSpec", "YGOProEncoder", "init_module", "YGOProDMEnvPool", "YGOProGymEnvPool",
           "YGOProGymnasiumEnvPool", "YGOProRawEnvPool", "CONFIG_KEYS", "DEFAULT_CONFIG_VALUES", "STATE_KEYS", "ACTION_KEYS"]
