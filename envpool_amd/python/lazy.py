"""Lazy stand-in for the Dummy env's ragged `obs:dyn` output.

In the reference every env-step allocates `full((env_id + 1, state_num), env_id)`
(envpool2/dummy/dummy_envpool.h:154-162, 205-208), which is O(num_envs^2) bytes per
batch (86 GB per step at 65 536 envs) and is a pure function of `info:players.env_id`.
Large pools therefore hand out this sequence instead of an object array; item access
builds exactly the array the reference would have returned.
"""
import numpy as np


class LazyDyn:
  """Sequence of `len(players_env_id)` int32 arrays, materialised on access."""

  dtype = np.dtype(object)

  def __init__(self, players_env_id: np.ndarray, state_num: int):
    self._ids = np.asarray(players_env_id, dtype=np.int32)
    self._s = int(state_num)

  @property
  def shape(self):
    return self._ids.shape

  def __len__(self) -> int:
    return self._ids.shape[0]

  def __getitem__(self, i):
    if isinstance(i, slice):
      return LazyDyn(self._ids[i], self._s)
    e = int(self._ids[i])
    return np.full((e + 1, self._s), e, dtype=np.int32)

  def __iter__(self):
    for i in range(len(self)):
      yield self[i]

  def materialize(self) -> np.ndarray:
    """The exact object array of arrays the reference returns."""
    out = np.empty(len(self), dtype=object)
    for i in range(len(self)):
      out[i] = self[i]
    return out

  def __repr__(self) -> str:
    return f"LazyDyn(n={len(self)}, state_num={self._s})"
