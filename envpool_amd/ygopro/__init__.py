"""YGOPro-v0: spec surface and the GPU observation-encode front-end.

The reference's `_YGOProEnvSpec` (envpool2/ygopro/ygopro.h:1177-1211, registered at
envpool2/ygopro/ygopro.cpp:4-10) fixes the config keys, the state keys/dtypes/shapes and the
action spec; `_YGOProEnvSpec` below reproduces that surface in the tuple forms of
envpool2/core/py_envpool.h:102-169. The duel engine (ygopro-core + Lua + SQLite) is NOT part of
this library, so there is no `_YGOProEnvPool` that can run duels: what replaces the reference's
`WriteState` is `YGOProEncoder`, which takes the host engine's per-env state (reference forms or
packed records) and returns the batch in `_state_keys` order, encoded on the GPU.
"""
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


def _no_engine(*_a, **_k):
  raise RuntimeError(
    "YGOPro-v0: the duel engine (ygopro-core, Lua, SQLite card database) is not part of envpool_amd; "
    "run the engine on host threads and encode its states with envpool_amd.ygopro.YGOProEncoder")


class YGOProRawEnvPool:
  """Placeholder so that `make("YGOPro-v0", ...)` explains itself instead of failing obscurely."""

  def __init__(self, *a, **k):
    _no_engine()


YGOProDMEnvPool = YGOProGymEnvPool = YGOProGymnasiumEnvPool = YGOProRawEnvPool

__all__ = ["YGOProEnvSpec", "YGOProEncoder", "CONFIG_KEYS", "DEFAULT_CONFIG_VALUES", "STATE_KEYS", "ACTION_KEYS"]
