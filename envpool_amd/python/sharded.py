"""One pool over several GPUs in ONE process: K independent shard pools behind the raw
`_send / _recv / _reset` surface of `_DummyEnvPool` (BASELINE configs 4 and 5).

Envs are independent (the only state the reference shares between them is the StateBuffer slot
counters, envpool2/core/state_buffer.h:85-88), so the node-wide pool is K pools: shard r owns the
contiguous global ids `[off + r*n, off + (r+1)*n)`, seeds them with the GLOBAL id (env.h:94) and
lives on its own GPU with its own ring and stream. This class is the host-side index in front of
them; there is no device exchange.

* sync mode (batch_size == num_envs, one player): `_send(action)` splits the sent batch by owner,
  `_recv` puts the shards' rows back at the positions their envs had in the SENT `env_id` array -
  row i of the result is the env at position i of the send, exactly the single-pool contract
  (async_envpool.h:69 + state_buffer.h:92-95). One send (or reset) per recv.
* async mode: every shard runs with batch_size / K and `_recv` concatenates one batch of each, so
  a batch still holds batch_size env-steps (BASELINE config 5: 8192 / 4096 over 8 GPUs is 1024 /
  512 per GPU). Rows come in shard order, then arrival order inside a shard; like the reference's
  arrival order this is to be compared as a set keyed by `info:env_id`.

The shard pools come from a factory, so the fan-out / stitch logic is testable without a GPU
(tests/test_host_cpu.py drives it with CPU oracle pools).
"""
from abc import ABCMeta
from collections import deque
from typing import Callable, List, Optional, Sequence

import numpy as np

from .lazy import LazyDyn

_PLAYER_KEYS = (1, 4, 5, 8, 9, 10, 11)  # state keys whose leading dim counts players, not envs
_DYN = 9


def _cat_dyn(parts: List, players_env_id: np.ndarray, state_num: int):
  """obs:dyn of several batches: shards of low env ids hand out the reference's object arrays,
  the others the lazy sequence; a mix becomes the lazy sequence (same items on access)."""
  if all(isinstance(p, np.ndarray) for p in parts):
    return np.concatenate(parts)
  if all(p is None for p in parts):
    return None
  return LazyDyn(players_env_id, state_num)


class ShardedRawPool(metaclass=ABCMeta):
  """K shard pools with the raw boundary of one pool. `make_shard(rank, num_envs, batch_size,
  env_id_offset, device)` returns an object with `_send(list)`, `_recv() -> list`, `_reset(ids)`."""

  def __init__(self, num_envs: int, batch_size: int, max_num_players: int, state_num: int,
               num_shards: int, make_shard: Callable, devices: Optional[Sequence[int]] = None,
               env_id_offset: int = 0):
    if num_shards < 1 or num_envs % num_shards != 0:
      raise ValueError(f"num_envs = {num_envs} is not divisible by num_shards = {num_shards}")
    batch_size = batch_size or num_envs
    self._N, self._B, self._P, self._S, self._K = num_envs, batch_size, max_num_players, state_num, num_shards
    self._off = env_id_offset
    self._n_local = num_envs // num_shards
    self._sync = batch_size == num_envs and max_num_players == 1  # async_envpool.h:95
    if batch_size == num_envs:
      b_local = self._n_local
    elif batch_size % num_shards == 0:
      b_local = batch_size // num_shards
    else:
      raise ValueError(f"async sharded pool: batch_size = {batch_size} is not divisible by num_shards = {num_shards}")
    devices = list(devices) if devices is not None else list(range(num_shards))
    if len(devices) != num_shards:
      raise ValueError("one device per shard")
    self._shards = [make_shard(r, self._n_local, b_local, env_id_offset + r * self._n_local, devices[r])
                    for r in range(num_shards)]
    self._pending = deque()  # sync mode: per outstanding send, the positions of each shard's envs

  # ---- fan-out ---------------------------------------------------------------------------
  def _owner(self, env_id: np.ndarray) -> np.ndarray:
    local = env_id.astype(np.int64) - self._off
    if local.size and (local.min() < 0 or local.max() >= self._N):
      bad = int(env_id[np.argmax((local < 0) | (local >= self._N))])
      raise ValueError(f"env_id {bad} is outside this pool [{self._off}, {self._off + self._N})")
    return local // self._n_local

  def _positions(self, env_id: np.ndarray) -> List:
    """Per shard, where its envs sit in `env_id`: a slice when the ids ascend (the usual case - all
    envs in order, or any ascending subset: each shard's envs are then one contiguous stretch, found
    by K + 1 binary searches, and everything downstream works on views), else an index array."""
    n = env_id.shape[0]
    if n and (n == 1 or bool((env_id[1:] >= env_id[:-1]).all())):
      lo, hi = int(env_id[0]) - self._off, int(env_id[-1]) - self._off
      if lo < 0 or hi >= self._N:
        raise ValueError(f"env_id {int(env_id[0]) if lo < 0 else int(env_id[-1])} is outside this pool "
                         f"[{self._off}, {self._off + self._N})")
      if not hasattr(self, "_edges"):
        self._edges = (self._off + self._n_local * np.arange(self._K + 1, dtype=np.int64)).astype(np.int64)
      b = np.searchsorted(env_id, self._edges, side="left")
      return [slice(int(b[r]), int(b[r + 1])) for r in range(self._K)]
    owner = self._owner(env_id)
    if n and bool((owner[1:] >= owner[:-1]).all()):
      # grouped by shard, in any order inside a shard (the env ids of a batch this front-end handed
      # out: an async actor loop sends back exactly that): slices as well
      b = np.searchsorted(owner, np.arange(self._K + 1), side="left")
      return [slice(int(b[r]), int(b[r + 1])) for r in range(self._K)]
    return [np.nonzero(owner == r)[0] for r in range(self._K)]

  @staticmethod
  def _count(p) -> int:
    return p.stop - p.start if isinstance(p, slice) else int(p.size)

  def _fan_out(self, env_id: np.ndarray, per_shard: Callable) -> None:
    env_id = np.ascontiguousarray(env_id, dtype=np.int32)
    pos = self._positions(env_id)
    if self._sync:
      if self._pending:
        raise RuntimeError("a sharded sync pool takes one send (or reset) per recv")
      self._pending.append((env_id.shape[0], pos))
    for r, p in enumerate(pos):
      if self._count(p):
        per_shard(r, p)

  def _send(self, action: List[np.ndarray]) -> None:
    env_id = np.asarray(action[0])
    penv = np.asarray(action[1])
    rest = [np.asarray(a) for a in action[2:]]
    if self._P == 1:
      def one(r, p):  # single player: action row i belongs to item i (env.h:115-123)
        self._shards[r]._send([env_id[p].astype(np.int32, copy=False), penv[p], rest[0][p], rest[1][p], rest[2][p]])
    else:
      # players.* arrays have one row per player: a row follows the env its players.env_id names
      # (env.h:128-132); rows that name an env outside the pool match nothing and are dropped
      plocal = penv.astype(np.int64) - self._off
      powner = np.where((plocal >= 0) & (plocal < self._N), plocal // self._n_local, -1)

      def one(r, p):
        rows = np.nonzero(powner == r)[0]
        self._shards[r]._send([env_id[p].astype(np.int32, copy=False), penv[rows], rest[0][p], rest[1][rows], rest[2][rows]])
    self._fan_out(env_id, one)

  def _reset(self, env_ids: np.ndarray) -> None:
    env_ids = np.asarray(env_ids)
    self._fan_out(env_ids, lambda r, p: self._shards[r]._reset(env_ids[p].astype(np.int32, copy=False)))

  # ---- stitch ----------------------------------------------------------------------------
  def _recv(self) -> List:
    if not self._sync:
      parts = [s._recv() for s in self._shards]
      out = [None if k == _DYN else np.concatenate([p[k] for p in parts]) for k in range(len(parts[0]))]
      out[_DYN] = _cat_dyn([p[_DYN] for p in parts], out[1], self._S)
      return out
    if not self._pending:
      return self._shards[0]._recv()  # nothing outstanding: the reference returns an empty batch
    n, pos = self._pending.popleft()
    if n == 0:
      return self._shards[0]._recv()
    if all(isinstance(p, slice) for p in pos):
      # ascending ids: the shards' batches, one after the other, ARE the sent order - one copy per key
      parts = []
      for r, p in enumerate(pos):
        if not self._count(p):
          continue
        part = self._shards[r]._recv()
        if len(part[0]) != self._count(p):
          raise RuntimeError(f"shard {r} returned {len(part[0])} rows for {self._count(p)} sent envs")
        parts.append(part)
      if len(parts) == 1:
        return parts[0]
      out = [None if k == _DYN else np.concatenate([q[k] for q in parts]) for k in range(len(parts[0]))]
      out[_DYN] = _cat_dyn([q[_DYN] for q in parts], out[1], self._S)
      return out
    out: List = [None] * 12
    dyn_ids = None
    for r, p in enumerate(pos):
      if not self._count(p):
        continue
      p = np.arange(p.start, p.stop) if isinstance(p, slice) else p
      part = self._shards[r]._recv()
      if len(part[0]) != p.size:
        raise RuntimeError(f"shard {r} returned {len(part[0])} rows for {p.size} sent envs")
      for k, a in enumerate(part):
        if k == _DYN:
          continue
        if out[k] is None:
          out[k] = np.zeros((n,) + a.shape[1:], a.dtype)
        out[k][p] = a  # single player: player rows == env rows
      if isinstance(part[_DYN], LazyDyn) or part[_DYN] is None:
        dyn_ids = True
      else:
        if out[_DYN] is None:
          out[_DYN] = np.empty(n, dtype=object)
        for j, q in enumerate(p):
          out[_DYN][q] = part[_DYN][j]
    if dyn_ids:
      out[_DYN] = LazyDyn(out[1], self._S)
    return out

  # ---- additive ---------------------------------------------------------------------------
  @property
  def shards(self):
    return list(self._shards)

  def _episode_stats(self):
    """(return f32[N], length i32[N], count i32[N]) ordered by global env_id."""
    parts = [s._episode_stats() for s in self._shards]
    return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))

  def _sync_all(self) -> None:
    for s in self._shards:
      s._sync()


_CLASSES = None


def sharded_dummy_classes():
  """{"dm", "gym", "gymnasium", "raw"} -> pool class of the sharded Dummy-v0 pool; built on the
  first `make(..., num_shards=K)`."""
  global _CLASSES
  if _CLASSES is None:
    _CLASSES = _build_classes()
  return _CLASSES


def _build_classes():
  from ..dummy.dummy_envpool import _DummyEnvPool, _DummyEnvSpec
  from .pool import DMFrontEnd, EnvPoolFrontEnd, GymFrontEnd, GymnasiumFrontEnd, make_pool_class

  class _ShardedDummyEnvPool(ShardedRawPool):
    _state_keys = _DummyEnvPool._state_keys
    _action_keys = _DummyEnvPool._action_keys

    def __init__(self, spec, num_shards: int, devices: Optional[Sequence[int]] = None):
      self._spec = spec
      cfg = dict(zip(spec._config_keys, spec._config_values))
      from ..dummy.dummy_envpool import device_count
      if devices is None:
        ndev = max(1, device_count())
        devices = [(cfg["device_id"] + r) % ndev for r in range(num_shards)]

      def make_shard(rank, n, b, off, dev):
        vals = dict(cfg, num_envs=n, batch_size=b, env_id_offset=off, device_id=dev)
        return _DummyEnvPool(_DummyEnvSpec(tuple(vals[k] for k in spec._config_keys)))

      ShardedRawPool.__init__(self, cfg["num_envs"], cfg["batch_size"], cfg["max_num_players"], cfg["state_num"],
                              num_shards, make_shard, devices, cfg["env_id_offset"])

  def flavour(front, tag):
    cls = make_pool_class(f"ShardedDummy{tag}EnvPool", _ShardedDummyEnvPool, front)
    init = front.__init__

    def __init__(self, spec, num_shards, devices=None):
      _ShardedDummyEnvPool.__init__(self, spec, num_shards, devices)
      # the front-end's own set-up, minus its super().__init__(spec) hop into the raw class
      self.spec = spec
      from .spec import check_unique
      from .tree import Layout
      check_unique(type(self).__name__, "state", list(self._state_keys))
      check_unique(type(self).__name__, "action", list(self._action_keys))
      self._layout = Layout(self._state_keys)
      self._dm_layout = Layout(self._state_keys, root="State", merge_info_into_obs=True)
      self._action_checked = False
    cls.__init__ = __init__
    del init
    return cls

  return {"dm": flavour(DMFrontEnd, "DM"), "gym": flavour(GymFrontEnd, "Gym"),
          "gymnasium": flavour(GymnasiumFrontEnd, "Gymnasium"), "raw": flavour(EnvPoolFrontEnd, "Raw")}
