"""Global env registry (counterpart of envpool2/registration.py:26-115)."""
import importlib
import os
from typing import Any, Dict, List, Tuple

base_path = os.path.abspath(os.path.dirname(__file__))

ENV_TYPES = ("dm", "gym", "gymnasium", "raw")


import functools


@functools.lru_cache(maxsize=None)
def _gym_is_new_api() -> bool:
  try:
    import gym
    from packaging import version
    return version.parse(gym.__version__) >= version.parse("0.26.0")
  except ImportError:
    return True  # no gym installed: behave like the current gym API


class EnvRegistry:
  def __init__(self) -> None:
    self.specs: Dict[str, Tuple[str, str, Dict[str, Any]]] = {}
    self.envpools: Dict[str, Dict[str, Tuple[str, str]]] = {}

  def register(self, task_id: str, import_path: str, spec_cls: str, dm_cls: str, gym_cls: str,
               gymnasium_cls: str, raw_cls: str = "", **kwargs: Any) -> None:
    assert task_id not in self.specs, f"{task_id} is already registered"
    kwargs.setdefault("base_path", base_path)
    self.specs[task_id] = (import_path, spec_cls, kwargs)
    self.envpools[task_id] = {"dm": (import_path, dm_cls), "gym": (import_path, gym_cls),
                              "gymnasium": (import_path, gymnasium_cls), "raw": (import_path, raw_cls)}

  def make(self, task_id: str, env_type: str, **kwargs: Any) -> Any:
    new_api = _gym_is_new_api()
    kwargs.setdefault("gym_reset_return_info", new_api)
    if new_api and not kwargs["gym_reset_return_info"]:
      raise ValueError(
        "You are using gym>=0.26.0 but passed `gym_reset_return_info=False`. The new gym API "
        "requires environments to return an info dictionary after resets.")
    assert task_id in self.specs, f"{task_id} is not supported, `list_all_envs()` may help."
    assert env_type in ENV_TYPES
    # build-only: one pool over several GPUs of this process (python/sharded.py). `num_envs` and
    # `batch_size` stay the GLOBAL figures; shard r owns a contiguous block of env ids on
    # shard_devices[r] (default: device_id + r, round robin over the visible GPUs)
    num_shards = int(kwargs.pop("num_shards", 1))
    shard_devices = kwargs.pop("shard_devices", None)
    spec = self.make_spec(task_id, **kwargs)
    if num_shards > 1:
      from .python import sharded
      if task_id == "Dummy-v0":
        return sharded.sharded_dummy_classes()[env_type](spec, num_shards, shard_devices)
      if task_id == "YGOPro-v0":
        return sharded.sharded_ygopro_classes()[env_type](spec, num_shards, shard_devices)
      raise ValueError(f"num_shards is implemented for Dummy-v0 and YGOPro-v0, not {task_id}")
    import_path, cls_name = self.envpools[task_id][env_type]
    return getattr(importlib.import_module(import_path), cls_name)(spec)

  def make_dm(self, task_id: str, **kw: Any) -> Any:
    return self.make(task_id, "dm", **kw)

  def make_gym(self, task_id: str, **kw: Any) -> Any:
    return self.make(task_id, "gym", **kw)

  def make_gymnasium(self, task_id: str, **kw: Any) -> Any:
    return self.make(task_id, "gymnasium", **kw)

  def make_raw(self, task_id: str, **kw: Any) -> Any:
    return self.make(task_id, "raw", **kw)

  def make_spec(self, task_id: str, **make_kwargs: Any) -> Any:
    import_path, spec_cls, kwargs = self.specs[task_id]
    kwargs = {**kwargs, **make_kwargs}
    if "seed" in kwargs:
      assert -2**31 <= kwargs["seed"] < 2**31, f"Seed should be in range of int32, got {kwargs['seed']}"
    if "num_envs" in kwargs:
      assert kwargs["num_envs"] >= 1
    if "batch_size" in kwargs:
      assert 0 <= kwargs["batch_size"] <= kwargs["num_envs"]
    if "max_num_players" in kwargs:
      assert 1 <= kwargs["max_num_players"]
    cls = getattr(importlib.import_module(import_path), spec_cls)
    return cls(cls.gen_config(**kwargs))

  def list_all_envs(self) -> List[str]:
    return list(self.specs.keys())


registry = EnvRegistry()
register = registry.register
make = registry.make
make_dm = registry.make_dm
make_gym = registry.make_gym
make_gymnasium = registry.make_gymnasium
make_raw = registry.make_raw
make_spec = registry.make_spec
list_all_envs = registry.list_all_envs
