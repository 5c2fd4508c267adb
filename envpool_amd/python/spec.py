"""Python front-end of an EnvSpec (counterpart of envpool2/python/env_spec.py:36-244 and
protocol.py:100-124). gym / gymnasium / dm_env are optional: the space conversions import
them on first use and raise ImportError with a clear message when absent."""
import importlib
import pprint
from collections import namedtuple
from typing import Any, Dict, List, Optional

import numpy as np

from .tree import as_namedtuple, nest

ACTION_THRESHOLD = 2**20  # data.py:28


class ArraySpec:
  """(dtype, shape, minimum, maximum) of one state / action array."""

  def __init__(self, dtype, shape, bounds, element_wise_bounds):
    self.dtype = dtype
    self.shape = shape
    lo, hi = element_wise_bounds
    self.minimum = np.array(lo) if len(lo) else bounds[0]
    self.maximum = np.array(hi) if len(hi) else bounds[1]

  def __repr__(self) -> str:
    return (f"ArraySpec(shape={self.shape}, dtype={self.dtype}, "
            f"minimum={self.minimum}, maximum={self.maximum})")


def _optional(name: str):
  try:
    return importlib.import_module(name)
  except ImportError as e:
    raise ImportError(
      f"`{name}` is not installed; envpool_amd only needs it for the {name} space/spec "
      "conversions. The raw send/recv/step/reset API works without it."
    ) from e


def _is_discrete(spec: ArraySpec) -> bool:
  # data.py:74-75 / :93-94 / :113-114
  return (np.prod(np.abs(spec.shape)) == 1 and np.isscalar(spec.minimum) and
          np.isclose(spec.minimum, 0) and spec.maximum < ACTION_THRESHOLD)


def _space(mod, spec: ArraySpec):
  if _is_discrete(spec):
    n = int(spec.maximum - spec.minimum + 1)
    try:
      return mod.spaces.Discrete(n=n, start=int(spec.minimum))
    except TypeError:  # very old gym
      return mod.spaces.Discrete(n=n)
  return mod.spaces.Box(shape=[s for s in spec.shape if s != -1], dtype=spec.dtype,
                        low=spec.minimum, high=spec.maximum)


def _dm_array(name: str, spec: ArraySpec):
  dm_env = _optional("dm_env")
  if _is_discrete(spec):
    return dm_env.specs.DiscreteArray(name=name, dtype=spec.dtype,
                                      num_values=int(spec.maximum - spec.minimum + 1))
  return dm_env.specs.BoundedArray(name=name, shape=[s for s in spec.shape if s != -1],
                                   dtype=spec.dtype, minimum=spec.minimum, maximum=spec.maximum)


class EnvSpecFrontEnd:
  """Mixed into the pybind `_XxxEnvSpec` class by api.py_env."""

  gen_config: Any

  @property
  def config(self):
    return self.gen_config(*self._config_values)

  @property
  def reward_threshold(self) -> Optional[float]:
    return getattr(self.config, "reward_threshold", None)

  @property
  def state_array_spec(self) -> Dict[str, ArraySpec]:
    return dict(zip(self._state_keys, [ArraySpec(*s) for s in self._state_spec]))

  @property
  def action_array_spec(self) -> Dict[str, ArraySpec]:
    return dict(zip(self._action_keys, [ArraySpec(*s) for s in self._action_spec]))

  # ---- gym / gymnasium ----
  def _obs_space(self, mod):
    spaces = {k.replace("obs:", ""): _space(mod, v)
              for k, v in self.state_array_spec.items() if k.startswith("obs")}
    if len(spaces) == 1:
      return next(iter(spaces.values()))
    return nest(spaces, mod.spaces.Dict)

  def _act_space(self, mod):
    spec = self.action_array_spec
    if len(spec) == 3:  # env_id, players.env_id, <the action>
      return _space(mod, list(spec.values())[-1])
    return nest({k: _space(mod, v) for k, v in spec.items()}, mod.spaces.Dict)

  @property
  def observation_space(self):
    return self._obs_space(_optional("gym"))

  @property
  def action_space(self):
    return self._act_space(_optional("gym"))

  @property
  def gymnasium_observation_space(self):
    return self._obs_space(_optional("gymnasium"))

  @property
  def gymnasium_action_space(self):
    return self._act_space(_optional("gymnasium"))

  # ---- dm_env ----
  def observation_spec(self):
    flat = {k.replace("obs:", "").replace("info:", ""): _dm_array(k.replace(":", ".").split(".")[-1], v)
            for k, v in self.state_array_spec.items() if k.startswith(("obs", "info"))}
    return as_namedtuple("State", nest(flat))

  def action_spec(self):
    spec = self.action_array_spec
    if len(spec) == 3:
      name, last = list(spec.items())[-1]
      return _dm_array(name, last)
    return as_namedtuple("Action", nest({k: _dm_array(k.split(".")[-1], v) for k, v in spec.items()}))

  def __repr__(self) -> str:
    return f"{self.__class__.__name__}{pprint.pformat(self.config)[6:]}"


def check_unique(owner: str, kind: str, keys: List[str]) -> None:
  dup = sorted({k for k in keys if list(keys).count(k) > 1})
  if dup:
    raise SystemError(f"{owner}: duplicated {kind} keys {dup}")


def make_spec_class(name: str, base: type) -> type:
  keys = list(base._config_keys)
  check_unique(name, "config", keys)
  fields = [k.replace(".", "_") for k in keys]
  gen_config = namedtuple("Config", fields, defaults=tuple(base._default_config_values))
  return type(base)(name, (base, EnvSpecFrontEnd), {"gen_config": gen_config})
