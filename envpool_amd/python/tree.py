"""Flat "a:b.c" keys <-> nested containers (the job optree does for envpool2/python/data.py)."""
from collections import namedtuple
from typing import Any, Callable, Dict, List, Sequence, Tuple


def split_key(key: str) -> List[str]:
  return key.replace(":", ".").split(".")


def nest(flat: Dict[str, Any], factory: Callable[[], Any] = dict) -> Any:
  """{"a.b": 1, "a.c": 2} -> {"a": {"b": 1, "c": 2}} (insertion order kept)."""
  root = factory()
  for key, value in flat.items():
    parts = key.split(".")
    node = root
    for part in parts[:-1]:
      if part not in node:
        node[part] = factory()
      node = node[part]
    node[parts[-1]] = value
  return root


_NT_CLASSES: Dict[Tuple, Any] = {}


def as_namedtuple(name: str, tree: Dict[str, Any]) -> Tuple:
  # (one class per (name, fields), made once: building a namedtuple class costs ~15 us, and the dm
  # flavour needs four of them per step)
  key = (name, tuple(tree.keys()))
  cls = _NT_CLASSES.get(key)
  if cls is None:
    cls = _NT_CLASSES[key] = namedtuple(name, list(tree.keys()))
  return cls(*[as_namedtuple(k, v) if isinstance(v, dict) else v for k, v in tree.items()])


class Layout:
  """Pre-computed mapping from the flat state list to a nested dict / namedtuple."""

  def __init__(self, keys: Sequence[str], root: str = "", merge_info_into_obs: bool = False):
    self.paths = []
    for key in keys:
      if merge_info_into_obs:  # dm_env flavour (data.py:126-141)
        if key in ("obs", "info"):
          key = f"obs:{key}"
        key = key.replace("info:", "obs:").replace("obs:", f"{root}:")
      self.paths.append(split_key(key))

  def build_dict(self, values: Sequence[Any]) -> Dict[str, Any]:
    out: Dict[str, Any] = {}
    for path, value in zip(self.paths, values):
      node = out
      for part in path[:-1]:
        node = node.setdefault(part, {})
      node[path[-1]] = value
    return out

  def build_namedtuple(self, name: str, values: Sequence[Any]) -> Tuple:
    return as_namedtuple(name, self.build_dict(values))
