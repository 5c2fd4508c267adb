"""One pool over several GPUs in ONE process: K independent shard pools behind the raw
`_send / _recv / _reset` surface of `_DummyEnvPool` / `_YGOProEnvPool` (BASELINE
configs 4 and 5).

Envs are independent (the only state the reference shares between them is the StateBuffer slot
counters, envpool2/core/state_buffer.h:85-88), so the node-wide pool is K pools: shard r owns the
contiguous global ids `[off + r*n, off + (r+1)*n)`, seeds them with the GLOBAL id (env.h:94) and
lives on its own GPU with its own ring and stream. This class is the host-side index in front of
them; there is no device exchange.

* sync mode (batch_size == num_envs, one player): `_send(action)` splits the sent batch by owner,
  `_recv` puts the shards' rows back at the positions their envs had in the SENT `env_id` array -
  row i of the result is the env at position i of the send, exactly the single-pool contract
  (async_envpool.h:69 + state_buffer.h:92-95). One send (or reset) per recv.
* async mode: every shard runs with batch_size / K, and a node-wide batch is K shard batches from
  WHICHEVER shards have one complete - several from one shard if that is who is ready - so a shard
  whose envs are slow does not hold up the batch (the reference hands out the first batch_size
  arrivals, state_buffer_queue.h:114-150; round 2 took one batch from every shard in turn). Shards
  are polled in rotation (`_try_recv`); when none is ready the call naps and polls again (it never
  sits on one shard while another may complete first). Rows come part by part, arrival order inside a
  part; like the reference's arrival order this is to be compared as a set keyed by `info:env_id`.
* host arrays without a stitch: an ascending sync send has every shard's kernels write their rows
  of ONE pinned array per key (`ep_pool_host_target`, a recycled reference-counted block); async
  Dummy shards copy their part into the node-wide block at the poll (`_try_recv_into`); whatever is
  still stitched (permuted ids scatter by index; YGOPro and multi-player parts) is one threaded copy
  into such a block (`stitch_parts`) instead of a numpy concatenate per key.
* device path (additive): `reset_torch / recv_torch / send_torch / step_torch` keep every byte on
  the shards' GPUs. A result is a `ShardedResult`: one ordinary result per shard (or per received
  part), each on its own device, in shard order = ascending env ids; `.cat(device)` stitches them
  for a consumer that wants one tensor per key. A whole-pool lockstep step rings every shard's
  resident server first and then waits for each (`ep_pool_step_begin / _end`).

The shard pools come from a factory, so the fan-out / stitch logic is testable without a GPU
(tests/test_host_cpu.py drives it with CPU oracle pools).
"""
from abc import ABCMeta
from collections import deque
from typing import Callable, List, Optional, Sequence

import numpy as np

from .lazy import LazyDyn

_DUMMY_STATE_KEYS = ("info:env_id", "info:players.env_id", "elapsed_step", "done", "reward", "discount", "step_type",
                     "trunc", "obs:raw", "obs:dyn", "info:players.done", "info:players.id")


def _cat_dyn(parts: List, players_env_id: np.ndarray, state_num: int):
  """obs:dyn of several batches: shards of low env ids hand out the reference's object arrays,
  the others the lazy sequence; a mix becomes the lazy sequence (same items on access)."""
  if all(isinstance(p, np.ndarray) for p in parts):
    return np.concatenate(parts)
  if all(p is None for p in parts):
    return None
  return LazyDyn(players_env_id, state_num)


class ShardedResult(list):
  """Per-shard (or per-part) results of a device-path call, each on its shard's GPU. `owners[i]`
  is the shard part i came from (what `send_torch` needs to route the next actions)."""

  owners: List[int]

  def cat(self, device=None):
    """One result with every leaf concatenated over the parts, on `device` (default: the first
    part's). Parts on other GPUs are copied over (peer copies)."""
    import torch

    def leaf_device(x):
      if isinstance(x, torch.Tensor):
        return x.device
      if isinstance(x, dict):
        for v in x.values():
          d = leaf_device(v)
          if d is not None:
            return d
      elif isinstance(x, (tuple, list)):
        for v in x:
          d = leaf_device(v)
          if d is not None:
            return d
      return None

    dev = torch.device(device) if device is not None else leaf_device(self[0])

    def go(xs):
      x0 = xs[0]
      if isinstance(x0, torch.Tensor):
        return torch.cat([x.to(dev, non_blocking=True) for x in xs]) if len(xs) > 1 else x0.to(dev)
      if isinstance(x0, dict):
        return {k: go([x[k] for x in xs]) for k in x0}
      if isinstance(x0, tuple) and hasattr(x0, "_fields"):
        return type(x0)(*[go([x[i] for x in xs]) for i in range(len(x0))])
      if isinstance(x0, (tuple, list)):
        return type(x0)(go([x[i] for x in xs]) for i in range(len(x0)))
      return x0

    return go(list(self))


class ShardedRawPool(metaclass=ABCMeta):
  """K shard pools with the raw boundary of one pool. `make_shard(rank, num_envs, batch_size,
  env_id_offset, device)` returns an object with `_send(list)`, `_recv() -> list`, `_reset(ids)`
  and, optionally, `_try_recv()` (None when no batch is complete), `_set_recv_timeout(ms)`,
  `close()` and the device-path methods of the pybind pools."""

  def __init__(self, num_envs: int, batch_size: int, max_num_players: int, state_num: int,
               num_shards: int, make_shard: Callable, devices: Optional[Sequence[int]] = None,
               env_id_offset: int = 0, state_keys: Sequence[str] = _DUMMY_STATE_KEYS):
    if num_shards < 1 or num_envs % num_shards != 0:
      raise ValueError(f"num_envs = {num_envs} is not divisible by num_shards = {num_shards}")
    batch_size = batch_size or num_envs
    self._N, self._B, self._P, self._S, self._K = num_envs, batch_size, max_num_players, state_num, num_shards
    self._off = env_id_offset
    self._n_local = num_envs // num_shards
    self._sync = batch_size == num_envs and max_num_players == 1  # async_envpool.h:95
    self._nkeys = len(state_keys)
    self._dyn = list(state_keys).index("obs:dyn") if "obs:dyn" in state_keys else None
    if batch_size == num_envs:
      b_local = self._n_local
    elif batch_size % num_shards == 0:
      b_local = batch_size // num_shards
    else:
      raise ValueError(f"async sharded pool: batch_size = {batch_size} is not divisible by num_shards = {num_shards}")
    self._b_local = b_local
    devices = list(devices) if devices is not None else list(range(num_shards))
    if len(devices) != num_shards:
      raise ValueError("one device per shard")
    self._devices = devices
    self._shards = [make_shard(r, self._n_local, b_local, env_id_offset + r * self._n_local, devices[r])
                    for r in range(num_shards)]
    self._pending = deque()  # sync mode: per outstanding send, the positions of each shard's envs
    self._tgt = None         # (HostBlocks, key offsets, row bytes, dtypes, row shapes): see _setup_target
    self._stitch = None      # (HostBlocks class, stitch function): see _setup_stitch
    self._inc = None         # async recv in progress: [block, parts copied into it so far]
    self._next = 0                         # async recv: where the rotation over the shards stands
    self._timeout_ms = -1
    self._last_owners: List[int] = []

  def close(self) -> None:
    for s in self._shards:
      if hasattr(s, "close"):
        s.close()

  def _set_recv_timeout(self, ms: int) -> None:
    self._timeout_ms = ms
    for s in self._shards:
      if hasattr(s, "_set_recv_timeout"):
        s._set_recv_timeout(ms)

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
    if not self._sync and n:
      # an async actor loop sends back the ids of the batch it was handed: whole shard batches, in
      # the order the shards were ready - a few runs of one owner each. (Only for async pools: a sync
      # batch's rows come back in shard order, which is the sent order only if the runs ascend.)
      cuts = np.flatnonzero(owner[1:] != owner[:-1]) + 1
      if cuts.size < 4 * self._K:
        edges = [0] + cuts.tolist() + [n]
        runs: List[List] = [[] for _ in range(self._K)]
        for a, b in zip(edges[:-1], edges[1:]):
          runs[int(owner[a])].append((a, b))
        return [slice(0, 0) if not rs else slice(*rs[0]) if len(rs) == 1 else np.concatenate([np.arange(a, b) for a, b in rs])
                for rs in runs]
    return [np.nonzero(owner == r)[0] for r in range(self._K)]

  @staticmethod
  def _count(p) -> int:
    return p.stop - p.start if isinstance(p, slice) else int(p.size)

  def _setup_target(self, make_blocks: Callable, slot_keys: Sequence) -> None:
    """One host array per key over ALL shards, written by the shards' kernels themselves
    (ep_pool_host_target): `slot_keys` = per key of the pools' slot order (dtype string, row shape,
    index in the state keys); `make_blocks(bytes)` returns the recycling set of pinned blocks. A sync
    batch whose ids ascend then needs no copy to be stitched (4.6 MB of them per step of 65 536 envs:
    127 us of a 314 us step over 4 shards)."""
    offs, rowb, at = [], [], 0
    cap = self._N if self._sync else self._B  # rows of a node-wide batch
    for dt, tail, _ in slot_keys:
      nb = np.dtype(dt).itemsize * int(np.prod(tail, dtype=np.int64))
      offs.append(at)
      rowb.append(nb)
      at = (at + nb * cap + 255) // 256 * 256
    self._tgt = (make_blocks(max(at, 256)), offs, rowb, list(slot_keys))

  def _fan_out(self, env_id: np.ndarray, per_shard: Callable) -> None:
    env_id = np.ascontiguousarray(env_id, dtype=np.int32)
    pos = self._positions(env_id)
    blk = None
    if self._sync:
      if self._pending:
        raise RuntimeError("a sharded sync pool takes one send (or reset) per recv")
      if self._tgt is not None and env_id.shape[0] and all(isinstance(p, slice) for p in pos):
        blk = self._tgt[0].take()
      self._pending.append((env_id.shape[0], pos, blk))
    try:
      for r, p in enumerate(pos):
        if self._count(p):
          if blk is not None:  # this shard's rows of the node-wide arrays: from row p.start of each
            self._shards[r]._host_target(blk, self._tgt[1], p.start)
          per_shard(r, p)
    except Exception:
      if blk is not None:  # a target no send took must not meet a later batch
        for s in self._shards:
          s._host_target_cancel()
      raise

  def _send(self, action: List[np.ndarray]) -> None:
    env_id = np.asarray(action[0])
    penv = np.asarray(action[1])
    rest = [np.asarray(a) for a in action[2:]]
    if self._P == 1:
      def one(r, p):  # single player: action row i belongs to item i (env.h:115-123)
        self._shards[r]._send([env_id[p].astype(np.int32, copy=False), penv[p]] + [a[p] for a in rest])
    else:
      # players.* arrays have one row per player: a row follows the env its players.env_id names
      # (env.h:128-132); rows that name an env outside the pool match nothing and are dropped.
      # (Dummy's action keys: list_action is per env, players.action / players.id per player)
      plocal = penv.astype(np.int64) - self._off
      powner = np.where((plocal >= 0) & (plocal < self._N), plocal // self._n_local, -1)

      def one(r, p):
        rows = np.nonzero(powner == r)[0]
        self._shards[r]._send([env_id[p].astype(np.int32, copy=False), penv[rows], rest[0][p], rest[1][rows], rest[2][rows]])
    self._fan_out(env_id, one)

  def _reset(self, env_ids: np.ndarray) -> None:
    env_ids = np.asarray(env_ids)
    self._fan_out(env_ids, lambda r, p: self._shards[r]._reset(env_ids[p].astype(np.int32, copy=False)))

  # ---- async: K shard batches from whoever is ready ----------------------------------------
  def _gather_parts(self, try_recv: Callable) -> List:
    """K complete shard batches, each as (shard, batch), from whichever shards have one:
    `try_recv(shard)` returns None when that shard has no complete batch. Shards are asked in
    rotation, starting behind the one served last, so that a shard that is always ready cannot
    crowd the others out; when a whole sweep finds nothing the call naps (20 us, doubling up to
    1 ms) and sweeps again - it never sits on ONE shard while another may complete first. The wait
    for rows nobody has sent follows `_set_recv_timeout` (< 0: for ever, like the reference)."""
    import time
    K = self._K
    parts: List = list(getattr(self, "_stash", []))
    self._stash = []
    for i, (r, _) in enumerate(parts):
      self._on_part(r, i)
    nap, t0 = 20e-6, None
    while len(parts) < K:
      progress = False
      for _ in range(K):
        r = self._next
        self._next = (self._next + 1) % K
        got = try_recv(r)
        if got is not None:
          parts.append((r, got))
          self._on_part(r, len(parts) - 1)
          progress = True
          if len(parts) == K:
            break
      if progress:
        nap, t0 = 20e-6, None
        continue
      if self._timeout_ms == 0:
        break
      now = time.monotonic()
      t0 = now if t0 is None else t0
      if self._timeout_ms > 0 and (now - t0) * 1e3 > self._timeout_ms:
        break
      time.sleep(nap)
      nap = min(2 * nap, 1e-3)
    if len(parts) < K:
      # (parts already taken stay taken: hand them back through the next call)
      self._stash = parts + getattr(self, "_stash", [])
      raise RuntimeError(f"recv would block: {len(parts)} of {K} shard batches are complete")
    self._last_owners = [r for r, _ in parts]
    return parts

  def _on_part(self, shard: int, index: int) -> None:
    """Called right after shard `shard` handed out part `index` of the batch being put together
    (index 0 starts a new batch): pools whose shards keep per-batch extras pick them up here."""

  def _cat_parts(self, parts: List) -> List:
    if self._stitch is not None:
      out = self._stitch_parts(parts)
    else:
      out = [None if k == self._dyn else np.concatenate([p[k] for _, p in parts]) for k in range(self._nkeys)]
    if self._dyn is not None:
      out[self._dyn] = _cat_dyn([p[self._dyn] for _, p in parts], out[1], self._S)
    return out

  def _setup_stitch(self, make_blocks: Callable, stitch: Callable) -> None:
    """The per-key concatenate of K parts as ONE call that copies into a recycled block over a few
    threads (csrc/py_dummy.cpp: stitch_parts) instead of K * keys numpy copies into fresh pages: a
    YGOPro batch of 4096 rows is 28 MB: 1230 -> 720 us. (Copying each async part as it arrives, under
    the PCIe traffic of the shards still writing theirs, measured SLOWER on one GPU - 1.56 -> 1.9 ms per
    batch, eight crew wake-ups instead of one - and was dropped.)"""
    self._stitch = (make_blocks, stitch)
    self._stitch_layout = None

  def _stitch_begin(self, first: List):
    make_blocks, stitch = self._stitch
    if self._stitch_layout is None:
      # fixed-shape keys, from the first batch seen: (offset, dtype, row shape); room for a whole batch per key
      cap = max(self._B, self._N if self._sync else 0) * self._P  # (players.* keys: up to P rows per env)
      offs, meta, at = [], [], 0
      for k in range(self._nkeys):
        a = first[k]
        if k == self._dyn or not isinstance(a, np.ndarray) or a.dtype == object:
          offs.append(0)
          meta.append(None)
          continue
        offs.append(at)
        meta.append((a.dtype.str, tuple(a.shape[1:])))
        at = (at + a.dtype.itemsize * int(np.prod(a.shape[1:], dtype=np.int64)) * cap + 255) // 256 * 256
      rowb = [0 if m is None else np.dtype(m[0]).itemsize * int(np.prod(m[1], dtype=np.int64)) for m in meta]
      self._stitch_layout = (make_blocks(max(at, 256)), offs, meta, rowb)
    return self._stitch_layout[0].take()

  def _stitch_parts(self, parts: List) -> List:
    blk = self._stitch_begin(parts[0][1])
    _, offs, meta, _ = self._stitch_layout
    rows = self._stitch[1](blk, offs, [[None if m is None else p[k] for k, m in enumerate(meta)] for _, p in parts])
    return [None if m is None else blk.array(offs[k], (rows[k],) + m[1], m[0]) for k, m in enumerate(meta)]

  # ---- stitch ----------------------------------------------------------------------------
  def _recv(self) -> List:
    if not self._sync:
      if self._tgt is not None and self._P == 1 and all(hasattr(s, "_try_recv_into") for s in self._shards):
        # every part is copied to its rows of ONE block by the shard's poll itself (no numpy object per
        # part, the shard's own block free again at once): part j of the batch -> rows j * batch_size / K
        blocks, offs, _, keys = self._tgt
        if self._inc is None:
          self._inc = [blocks.take(), 0]
        inc = self._inc

        def take(r):
          rows = self._shards[r]._try_recv_into(inc[0], offs, inc[1] * self._b_local)
          if rows is None:
            return None
          if rows != self._b_local:
            raise RuntimeError(f"shard {r} handed out {rows} rows, not its batch of {self._b_local}")
          inc[1] += 1
          return rows
        parts = self._gather_parts(take)
        self._inc = None
        out = [None] * self._nkeys
        for off, (dt, tail, k) in zip(offs, keys):
          out[k] = inc[0].array(off, (len(parts) * self._b_local,) + tuple(tail), dt)
        if self._dyn is not None:
          out[self._dyn] = self._shards[-1]._make_dyn(out[1])
        return out
      if all(hasattr(s, "_try_recv") for s in self._shards):
        parts = self._gather_parts(lambda r: self._shards[r]._try_recv())
      else:  # shard pools that cannot be polled: one batch of each, in turn
        parts = []
        for r, s in enumerate(self._shards):
          parts.append((r, s._recv()))
          self._on_part(r, r)
        self._last_owners = list(range(self._K))
      return self._cat_parts(parts)
    if not self._pending:
      return self._shards[0]._recv()  # nothing outstanding: the reference returns an empty batch
    n, pos, blk = self._pending.popleft()
    if n == 0:
      return self._shards[0]._recv()
    if blk is not None:
      # every shard's kernels wrote their rows of ONE array per key: wait for each shard, then hand
      # out views of the block (it goes back to the set when the last of them dies)
      self._last_owners = []
      for r, p in enumerate(pos):
        if not self._count(p):
          continue
        rows = self._shards[r]._recv_target()
        if rows != self._count(p):
          raise RuntimeError(f"shard {r} returned {rows} rows for {self._count(p)} sent envs")
        self._last_owners.append(r)
        self._on_part(r, len(self._last_owners) - 1)
      _, offs, _, keys = self._tgt
      out = [None] * self._nkeys
      for off, (dt, tail, k) in zip(offs, keys):
        out[k] = blk.array(off, (n,) + tuple(tail), dt)
      if self._dyn is not None:
        out[self._dyn] = self._shards[-1]._make_dyn(out[1])  # (eager only if the highest ids are: all shards then)
      return out
    if all(isinstance(p, slice) for p in pos):
      # ascending ids: the shards' batches, one after the other, ARE the sent order - one copy per key
      parts = []
      for r, p in enumerate(pos):
        if not self._count(p):
          continue
        part = self._shards[r]._recv()
        if len(part[0]) != self._count(p):
          raise RuntimeError(f"shard {r} returned {len(part[0])} rows for {self._count(p)} sent envs")
        parts.append((r, part))
        self._on_part(r, len(parts) - 1)
      self._last_owners = [r for r, _ in parts]
      if len(parts) == 1:
        return parts[0][1]
      return self._cat_parts(parts)
    out: List = [None] * self._nkeys
    dyn_ids = None
    self._last_owners = []
    for r, p in enumerate(pos):
      if not self._count(p):
        continue
      p = np.arange(p.start, p.stop) if isinstance(p, slice) else p
      part = self._shards[r]._recv()
      self._last_owners.append(r)
      self._on_part(r, len(self._last_owners) - 1)
      if len(part[0]) != p.size:
        raise RuntimeError(f"shard {r} returned {len(part[0])} rows for {p.size} sent envs")
      for k, a in enumerate(part):
        if k == self._dyn:
          continue
        if out[k] is None:
          out[k] = np.zeros((n,) + a.shape[1:], a.dtype)
        out[k][p] = a  # single player: player rows == env rows
      if self._dyn is None:
        continue
      if isinstance(part[self._dyn], LazyDyn) or part[self._dyn] is None:
        dyn_ids = True
      else:
        if out[self._dyn] is None:
          out[self._dyn] = np.empty(n, dtype=object)
        for j, q in enumerate(p):
          out[self._dyn][q] = part[self._dyn][j]
    if dyn_ids:
      out[self._dyn] = LazyDyn(out[1], self._S)
    return out

  # ---- additive ---------------------------------------------------------------------------
  @property
  def shards(self):
    return list(self._shards)

  @property
  def last_owners(self) -> List[int]:
    """Shard of each part of the batch `_recv` (or `recv_torch`) handed out last, in row order."""
    return list(self._last_owners)

  def _episode_stats(self):
    """(return f32[N], length i32[N], count i32[N]) ordered by global env_id."""
    parts = [s._episode_stats() for s in self._shards]
    return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))

  def _sync_all(self) -> None:
    for s in self._shards:
      s._sync()


class ShardedTorchMixin:
  """Device path of a sharded front-end (mixed in before the flavour's front-end class): the
  same calls as `EnvPoolFrontEnd.reset_torch / recv_torch / step_torch`, over K shards, with
  nothing crossing PCIe and no host hop for the rows. Results are `ShardedResult`s."""

  @property
  def all_env_ids(self) -> np.ndarray:
    if not hasattr(self, "_all_env_ids"):
      self._all_env_ids = np.arange(self._off, self._off + self._N, dtype=np.int32)
    return self._all_env_ids

  def _shard_dev(self, r):
    import torch
    if not hasattr(self, "_sh_devs"):
      self._sh_devs = [torch.device("cuda", d) for d in self._devices]
      self._sh_views = [dict() for _ in range(self._K)]
    return self._sh_devs[r]

  def _shard_out(self, r, descs, reset):
    from ..interop import as_torch
    dev = self._shard_dev(r)
    cache = self._sh_views[r]  # a shard's ring has ring_depth slots: the same views come round again
    vals = []
    for d in descs:
      if d is None:
        vals.append(None)
        continue
      key = (d[0], tuple(d[1]))
      t = cache.get(key)
      if t is None:
        if len(cache) > 4096:
          cache.clear()
        t = cache[key] = as_torch(d, dev)
      vals.append(t)
    return self._to(vals, reset, True)

  def _need_device_path(self):
    if not all(hasattr(s, "_recv_device") for s in self._shards):
      raise RuntimeError(f"{type(self).__name__}: the shard pools have no device-resident path")

  def _result(self, parts, owners):
    res = ShardedResult(parts)
    res.owners = list(owners)
    self._last_owners = list(owners)
    return res

  def reset_torch(self):
    """reset() of every env; per shard the batch as torch tensors aliasing that shard's ring."""
    self._need_device_path()
    ids = self.all_env_ids
    n = self._n_local
    for r, s in enumerate(self._shards):
      s._reset(ids[r * n:(r + 1) * n])
    if self._sync:
      parts = []
      for r, s in enumerate(self._shards):
        parts.append(self._shard_out(r, s._recv_device()[0], True))
        self._on_part(r, r)
      return self._result(parts, range(self._K))
    return self.recv_torch(reset=True)

  def recv_torch(self, reset: bool = False):
    """The next batch as device views: sync pools - every shard's rows of the outstanding send, in
    shard order; async pools - K shard batches from whichever shards are ready (`.owners` says
    which), see ShardedRawPool."""
    self._need_device_path()
    if self._sync:
      if self._pending:
        n, pos, _ = self._pending.popleft()
        who = [r for r, p in enumerate(pos) if self._count(p)]
        if not all(isinstance(pos[r], slice) for r in who):
          raise RuntimeError("recv_torch of a sharded sync pool needs the sent env ids grouped by shard in ascending shard order")
      else:
        who = list(range(self._K))
      parts = []
      for i, r in enumerate(who):
        parts.append(self._shard_out(r, self._shards[r]._recv_device()[0], reset))
        self._on_part(r, i)
      return self._result(parts, who)
    parts = self._gather_parts(lambda r: self._shards[r]._try_recv_device())
    return self._result([self._shard_out(r, got[0], reset) for r, got in parts], [r for r, _ in parts])

  def _shard_action(self, r, adict, n_rows):
    """(ptrs, dims, keep) of one shard's action arrays, converted only where dtype / layout / device differ."""
    import torch
    keys = self._spec._action_keys
    if not hasattr(self, "_torch_want"):
      names = {"int32": torch.int32, "float64": torch.float64, "float32": torch.float32}
      self._torch_want = [names[str(spec[0])] for spec in self._spec._action_spec]
    dev = self._shard_dev(r)
    ptrs, dims, keep = [], [], []
    self._converted = getattr(self, "_converted", False)
    for k, want in zip(keys, self._torch_want):
      t = adict.get(k)
      if t is None:  # keys the env never reads may be left out: a stand-in of the right length
        t = adict["players.env_id"] if k.startswith("players.") else adict["env_id"]
      if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f'send_torch: action "{k}" must be a CUDA tensor')
      if t.dtype != want or not t.is_contiguous() or t.device != dev:
        # a snapshot of the caller's tensor, to be taken again next time - into a staging tensor the
        # front-end keeps per shard and key, so that the addresses a shard's resident kernel is bound
        # to stay the same from step to step (pool.py:step_torch does the same)
        stage = self.__dict__.setdefault("_sh_stage", {})
        st = stage.get((r, k))
        if st is None or st.shape != t.shape:
          st = stage[(r, k)] = torch.empty(t.shape, dtype=want, device=dev)
        st.copy_(t)
        t = st
        self._converted = True
      keep.append(t)
      ptrs.append(t.data_ptr())
      dims.append(int(t.shape[0]))
    return ptrs, dims, keep

  def _split_actions(self, action, env_id, owners):
    """-> list of (shard, {key: tensor}) for a whole-pool action (env_id None: every env in order)
    or for per-part actions with their ids and owners (what recv_torch handed out)."""
    import torch
    from .pool import _flatten_action
    keys = self._spec._action_keys
    out = []
    if env_id is None:
      adict = _flatten_action(action) if isinstance(action, dict) else {keys[-1]: action}
      if not hasattr(self, "_sh_ids"):
        n = self._n_local
        self._sh_ids = [torch.from_numpy(self.all_env_ids[r * n:(r + 1) * n].copy()).to(self._shard_dev(r)) for r in range(self._K)]
      n = self._n_local
      for r in range(self._K):
        d = {k: v[r * n:(r + 1) * n] for k, v in adict.items()}
        d["env_id"] = self._sh_ids[r]
        d.setdefault("players.env_id", d["env_id"])
        out.append((r, d))
      return out
    if owners is None:
      owners = self._last_owners
    if not (isinstance(action, (list, tuple)) and isinstance(env_id, (list, tuple)) and len(action) == len(env_id) == len(owners)):
      raise RuntimeError("send_torch: pass either a whole-pool action (env_id=None) or per-part lists of actions and env ids "
                         "with the parts' owners (ShardedResult.owners)")
    for a, ids, r in zip(action, env_id, owners):
      d = _flatten_action(a) if isinstance(a, dict) else {keys[-1]: a}
      d["env_id"] = ids
      d.setdefault("players.env_id", ids)
      out.append((int(r), d))
    return out

  def send_torch(self, action, env_id=None, owners=None):
    """send() with CUDA tensors: a whole-pool action (sliced per shard; slices on another GPU than
    their shard's are copied over) or per-part lists as handed out by recv_torch."""
    import torch
    self._need_device_path()
    if not all(hasattr(s, "_send_device") for s in self._shards):
      raise RuntimeError(f"{type(self).__name__}: actions go through send() as host arrays (the shard pools step on the host)")
    parts = self._split_actions(action, env_id, owners)
    staged = [(r,) + self._shard_action(r, d, None) for r, d in parts]
    for dev in {self._shard_dev(r) for r, *_ in staged}:
      torch.cuda.current_stream(dev).synchronize()  # the actions were produced on torch's streams
    self._keep_actions = [k for *_, k in staged]
    if self._sync:
      if self._pending:
        raise RuntimeError("a sharded sync pool takes one send (or reset) per recv")
      who = [r for r, *_ in staged]
      if who != sorted(set(who)):
        raise RuntimeError("a sharded sync pool takes each shard once per send, in ascending shard order")
      pos = [slice(0, 0)] * self._K
      at = 0
      for r, _, dims, _ in staged:
        pos[r] = slice(at, at + dims[0])
        at += dims[0]
      self._pending.append((at, pos, None))
    for r, ptrs, dims, _ in staged:
      self._shards[r]._send_device(ptrs, dims)

  def _step_on_streams(self, staged):
    """The whole-pool step in stream order: every shard adopts torch's current stream of its device
    as its step stream (ep_pool_set_stream; one stream is in order by itself, and a cross-stream
    wait per step in each direction measured 113 us per step on this platform), the K step kernels are enqueued there
    (ep_pool_send_on_stream), the batches are described (ep_pool_recv_on_stream) and `terminated`
    of all parts is refreshed by one more launch; the host waits for nothing. Results are kept per
    ring slot (python/streamstep.py)."""
    import torch
    from .streamstep import current_stream_handle, fresh, refresh_terminated, slot_entry
    if not hasattr(self, "_sh_slots"):
      self._sh_slots = [dict() for _ in range(self._K)]
      self._sh_adopted = [None] * self._K
    handles = {}
    ents = []
    for shard in self._shards:
      shard._poll_errors()  # what earlier stream-ordered steps have flagged by now (before any shard steps)
    # every shard is sent to before any is received from (with a stream that is NOT the shard's own, a
    # recv makes it wait for that shard's rows, and a send after it would wait for them too)
    hs = []
    for r, ptrs, dims, _ in staged:
      dev = self._shard_dev(r)
      h = handles.get(dev)
      if h is None:
        h = handles[dev] = current_stream_handle(dev)
      hs.append(h)
      if self._sh_adopted[r] != h:
        self._shards[r]._set_stream(h)
        self._sh_adopted[r] = h
        self.__dict__.setdefault("_sh_stream_objs", {})[r] = torch.cuda.current_stream(dev)  # (kept alive while the shard launches on it)
      self._shards[r]._send_on_stream(ptrs, dims, h)
    for (r, *_), h in zip(staged, hs):
      shard = self._shards[r]
      key = shard._recv_slot_on_stream(h)
      ent = self._sh_slots[r].get(key)
      ents.append(ent if ent is not None else slot_entry(self, shard._last_descs()[0], self._shard_dev(r), self._sh_slots[r], key))
    refresh_terminated(ents, hs)  # one launch per device on torch's stream
    return self._result([fresh(e[0]) for e in ents], range(self._K))

  def rollout_torch(self, action, T=None):
    """T lockstep steps of every env, ONE launch per shard (ep_pool_rollout), for action sequences
    already on the GPU: `list_action` float64 `[T, G, 6]` (or `[G, 6]` with T: the same rows every
    step), optionally "players.env_id" int32 `[T, G]` / `[G]`, G = all envs in order. A shard on the
    tensors' device reads its columns of them in place (the kernel takes a row stride between
    steps); for a shard elsewhere they are copied over. Returns T `ShardedResult`s, in stream order
    on torch's current stream of each shard's device (see `_step_on_streams`). Pools made with
    `ring_depth >= T`."""
    import torch
    from .pool import _flatten_action
    from .streamstep import current_stream_handle, fresh, refresh_terminated, slot_entry
    self._need_device_path()
    if not (self._sync and not self._pending and all(hasattr(s, "_recv_slots_on_stream") for s in self._shards)):
      raise RuntimeError(f"{type(self).__name__}: rollout_torch needs a sync pool of Dummy shards with nothing outstanding")
    adict = _flatten_action(action) if isinstance(action, dict) else {"list_action": action}
    la = adict["list_action"]
    G, n = self._N, self._n_local
    if not (isinstance(la, torch.Tensor) and la.is_cuda and la.dtype == torch.float64 and la.is_contiguous()):
      raise RuntimeError('rollout_torch: "list_action" must be a contiguous float64 CUDA tensor')
    if la.dim() == 3:
      T = int(la.shape[0]) if T is None else T
    if T is None or T < 1 or (la.dim() == 3 and la.shape[0] != T) or tuple(la.shape[-2:]) != (G, la.shape[-1]) or la.dim() not in (2, 3):
      raise RuntimeError(f"rollout_torch: list_action must be [T, {G}, A] or [{G}, A] with T given, got {tuple(la.shape)}")
    A = int(la.shape[-1])
    penv = adict.get("players.env_id")
    if penv is not None and not (isinstance(penv, torch.Tensor) and penv.is_cuda and penv.dtype == torch.int32 and penv.is_contiguous()
                                 and tuple(penv.shape) in ((G,), (T, G))):
      raise RuntimeError(f"rollout_torch: players.env_id must be a contiguous int32 CUDA tensor [{T}, {G}] or [{G}]")
    if not hasattr(self, "_sh_slots"):
      self._sh_slots = [dict() for _ in range(self._K)]
      self._sh_adopted = [None] * self._K
    if not hasattr(self, "_sh_ids"):
      self._sh_ids = [torch.from_numpy(self.all_env_ids[r * n:(r + 1) * n].copy()).to(self._shard_dev(r)) for r in range(self._K)]
    for shard in self._shards:
      shard._poll_errors()
    keep, hs = [], []
    for r, shard in enumerate(self._shards):
      dev = self._shard_dev(r)
      h = current_stream_handle(dev)
      hs.append(h)
      if self._sh_adopted[r] != h:
        shard._set_stream(h)
        self._sh_adopted[r] = h
        self.__dict__.setdefault("_sh_stream_objs", {})[r] = torch.cuda.current_stream(dev)
      ids = self._sh_ids[r]
      if la.device == dev:  # in place: this shard's columns, G rows from one step to the next
        la_ptr, la_stride = la.data_ptr() + 8 * A * r * n, (G if la.dim() == 3 else 0)
      else:
        part = la[..., r * n:(r + 1) * n, :].to(dev).contiguous()
        keep.append(part)
        la_ptr, la_stride = part.data_ptr(), (n if la.dim() == 3 else 0)
      if penv is None:
        pe_ptr, pe_stride = ids.data_ptr(), 0
      elif penv.device == dev:
        pe_ptr, pe_stride = penv.data_ptr() + 4 * r * n, (G if penv.dim() == 2 else 0)
      else:
        part = penv[..., r * n:(r + 1) * n].to(dev).contiguous()
        keep.append(part)
        pe_ptr, pe_stride = part.data_ptr(), (n if penv.dim() == 2 else 0)
      shard._rollout(T, [ids.data_ptr(), pe_ptr, la_ptr, ids.data_ptr(), ids.data_ptr()], [n] * 5, [0, pe_stride, la_stride, 0, 0])
    self._keep_actions = [la, penv, keep]
    steps = [[] for _ in range(T)]
    ents_all, hs_all = [], []
    for r, shard in enumerate(self._shards):
      dev = self._shard_dev(r)
      for t, key in enumerate(shard._recv_slots_on_stream(hs[r], T)):
        ent = self._sh_slots[r].get(key)
        if ent is None:
          ent = slot_entry(self, shard._descs_of(t)[0], dev, self._sh_slots[r], key)
        steps[t].append(fresh(ent[0]))
        ents_all.append(ent)
        hs_all.append(hs[r])
    refresh_terminated(ents_all, hs_all)
    return [self._result(parts, range(self._K)) for parts in steps]

  def step_torch(self, action, env_id=None, owners=None, stream_ordered=None):
    """step() for a policy on the GPUs. A sync pool stepped as a whole takes one of two forms. The
    default (stream_ordered None or True) is the step IN STREAM ORDER, as for an unsharded pool's
    step_torch: every shard adopts torch's current stream of its device and its step kernel is
    enqueued there (ep_pool_step_on_stream) - the result is valid for work queued on those streams,
    the host waits for nothing, and a step's error (list_action self-check) surfaces at a later
    step_torch or `_sync_all()` instead of this one. stream_ordered=False is the BLOCKING form:
    the shards' resident servers are rung first (ep_pool_step_begin) and then waited for one by one
    (ep_pool_step_end) - side by side and without a kernel launch where every shard has a GPU to
    itself; shards that SHARE a device cannot each keep a resident kernel there
    (include/envpool_amd.h, ep_pool_step) and pay a launch and a wait each."""
    import torch
    self._need_device_path()
    if self._sync and env_id is None and not self._pending and all(hasattr(s, "_step_begin") for s in self._shards):
      stream_ordered = stream_ordered is not False and self._P == 1 and all(hasattr(s, "_send_on_stream") for s in self._shards)
      # the usual loop passes the same tensors every step: their per-shard slices (pointers, sizes) are kept
      flat = action.items() if isinstance(action, dict) else ((None, action),)
      sig = tuple((k, v.data_ptr(), tuple(v.shape), v.dtype, v.device) if isinstance(v, torch.Tensor) else (k, id(v)) for k, v in flat)
      if getattr(self, "_staged_sig", None) == sig:
        staged = self._staged
      else:
        self._converted = False
        staged = [(r,) + self._shard_action(r, d, None) for r, d in self._split_actions(action, None, None)]
        self._staged_sig, self._staged = (None, None) if self._converted else (sig, staged)
        self._staged_devs = {self._shard_dev(r) for r, *_ in staged}
      self._keep_actions = [k for *_, k in staged]
      if stream_ordered:
        return self._step_on_streams(staged)
      for dev in self._staged_devs:
        torch.cuda.current_stream(dev).synchronize()
      begun = 0
      try:
        for r, ptrs, dims, _ in staged:
          self._shards[r]._step_begin(ptrs, dims)
          begun += 1
      except Exception:
        for r in range(begun):  # the steps that were rung must be ended before anything else can run
          self._shards[r]._step_end()
        raise
      # the blocking form keeps its results per ring slot as well (python/streamstep.py); the kernels
      # that served the step wrote `terminated` beside the slot, so nothing is launched here
      from .streamstep import current_stream_handle, fresh, refresh_terminated, slot_entry
      if not hasattr(self, "_sh_slots"):
        self._sh_slots = [dict() for _ in range(self._K)]
        self._sh_adopted = [None] * self._K
      ents, hs = [], []
      first_error = None
      for r in range(self._K):
        shard = self._shards[r]
        try:
          key = shard._step_end_slot() if hasattr(shard, "_step_end_slot") else None
        except Exception as e:  # (every shard that was rung must be ended: remember the first error)
          first_error = first_error or e
          continue
        if key is None:
          ents.append([self._shard_out(r, shard._step_end()[0], False), None])
        else:
          ent = self._sh_slots[r].get(key)
          ents.append(ent if ent is not None else slot_entry(self, shard._last_descs()[0], self._shard_dev(r), self._sh_slots[r], key))
        hs.append(current_stream_handle(self._shard_dev(r)))
      if first_error is not None:
        raise first_error
      refresh_terminated(ents, hs)
      return self._result([fresh(e[0]) for e in ents], range(self._K))
    self.send_torch(action, env_id, owners)
    return self.recv_torch()


_CLASSES = {}


def sharded_dummy_classes():
  """{"dm", "gym", "gymnasium", "raw"} -> pool class of the sharded Dummy-v0 pool; built on the
  first `make(..., num_shards=K)`."""
  if "dummy" not in _CLASSES:
    _CLASSES["dummy"] = _build_dummy_classes()
  return _CLASSES["dummy"]


def sharded_ygopro_classes():
  """The same for YGOPro-v0: K `ep_ygo_pool`s, shard r's duel engine made by the factory given to
  `envpool_amd.ygopro.init_module` for that shard's block of global env ids."""
  if "ygopro" not in _CLASSES:
    _CLASSES["ygopro"] = _build_ygopro_classes()
  return _CLASSES["ygopro"]


def _flavours(raw_cls, tag):
  from .pool import DMFrontEnd, EnvPoolFrontEnd, GymFrontEnd, GymnasiumFrontEnd

  def flavour(front, name):
    meta = type(raw_cls)
    cls = meta(f"Sharded{tag}{name}EnvPool", (ShardedTorchMixin, front, raw_cls), {})

    def __init__(self, spec, num_shards, devices=None):
      raw_cls.__init__(self, spec, num_shards, devices)
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
    return cls

  return {"dm": flavour(DMFrontEnd, "DM"), "gym": flavour(GymFrontEnd, "Gym"),
          "gymnasium": flavour(GymnasiumFrontEnd, "Gymnasium"), "raw": flavour(EnvPoolFrontEnd, "Raw")}


def _build_dummy_classes():
  from ..dummy.dummy_envpool import _DummyEnvPool, _DummyEnvSpec

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
                              num_shards, make_shard, devices, cfg["env_id_offset"], list(self._state_keys))
      import os
      if os.environ.get("EP_SHARD_STITCH", "1") != "0":
        from ..dummy.dummy_envpool import HostBlocks, stitch_parts
        self._setup_stitch(HostBlocks, stitch_parts)
      if self._P == 1 and os.environ.get("EP_SHARD_TARGET", "1") != "0":
        from ..dummy.dummy_envpool import HostBlocks
        S = cfg["state_num"]
        sk = list(self._state_keys).index
        # the pools' slot order (include/envpool_amd.h, ep_pool_host_target)
        self._setup_target(HostBlocks, [
            ("<i4", (), sk("info:env_id")), ("<i4", (), sk("info:players.env_id")), ("<i4", (), sk("elapsed_step")),
            ("|b1", (), sk("done")), ("<f4", (), sk("reward")), ("<f4", (), sk("discount")), ("<i4", (), sk("step_type")),
            ("|b1", (), sk("trunc")), ("<i4", (S,), sk("obs:raw")), ("|b1", (), sk("info:players.done")),
            ("<i4", (), sk("info:players.id"))])

  return _flavours(_ShardedDummyEnvPool, "Dummy")


def _build_ygopro_classes():
  from .. import ygopro as Y
  from ..ygopro.ygopro_envpool import _YGOProEnvSpec

  class _ShardedYGOProEnvPool(ShardedRawPool):
    _state_keys = Y._YGOProEnvPool._state_keys
    _action_keys = Y._YGOProEnvPool._action_keys

    def __init__(self, spec, num_shards: int, devices: Optional[Sequence[int]] = None):
      self._spec = spec
      cfg = dict(zip(spec._config_keys, spec._config_values))
      if devices is None:
        from ..dummy.dummy_envpool import device_count
        ndev = max(1, device_count())
        devices = [(Y._MODULE["device_id"] + r) % ndev for r in range(num_shards)]
      base_off = Y._MODULE["env_id_offset"]

      def make_shard(rank, n, b, off, dev):
        vals = dict(cfg, num_envs=n, batch_size=b)
        sub = _YGOProEnvSpec(tuple(vals[k] for k in spec._config_keys))
        return Y._YGOProEnvPool(sub, device_id=dev, env_id_offset=off, shard=rank)

      ShardedRawPool.__init__(self, cfg["num_envs"], cfg["batch_size"], cfg["max_num_players"], 0, num_shards,
                              make_shard, devices, base_off, list(self._state_keys))
      self.last_h_card_ids = self.last_action_mask = None
      self._aux = []
      import os
      if os.environ.get("EP_SHARD_STITCH", "1") != "0":
        from ..dummy.dummy_envpool import HostBlocks, stitch_parts
        self._setup_stitch(HostBlocks, stitch_parts)

    def _on_part(self, shard: int, index: int) -> None:
      # aux:h_card_ids / aux:action_mask (additive keys 17 and 18) of the part just received: the shard
      # keeps them for ONE batch, so they are picked up part by part
      if index == 0:
        self._aux = []
      sh = self._shards[shard]
      self._aux.append((sh.last_h_card_ids, sh.last_action_mask, sh.last_env_ids))

    def _recv(self):
      if self._sync and self._pending and not all(isinstance(p, slice) for p in self._pending[0][1]):
        raise RuntimeError("a sharded YGOPro sync pool needs the sent env ids grouped by shard in ascending shard order")
      out = ShardedRawPool._recv(self)
      if self._aux and all(isinstance(a[0], np.ndarray) for a in self._aux):
        self.last_h_card_ids = np.concatenate([a[0] for a in self._aux])
        self.last_action_mask = np.concatenate([a[1] for a in self._aux])
      return out

    @property
    def last_env_ids(self) -> np.ndarray:
      """info:env_id of the batch received last, from host memory, part by part."""
      return np.concatenate([a[2] for a in self._aux]) if self._aux else np.zeros(0, np.int32)

    @property
    def last_env_ids_parts(self) -> List[np.ndarray]:
      """The same, one array per part (what `send(action_i, ids_i)` of an actor loop takes)."""
      return [a[2] for a in self._aux]

  classes = _flavours(_ShardedYGOProEnvPool, "YGOPro")
  for cls in classes.values():
    def recv_torch(self, reset: bool = False, _base=cls.recv_torch):
      from ..interop import as_torch
      res = _base(self, reset)
      # the legal-action mask of every part as a tensor on its shard's GPU (info["action_mask"])
      masks, hcards = [], []
      for r, (h, m, _) in zip(res.owners, self._aux):
        dev = self._shard_dev(r)
        masks.append(as_torch(m, dev))
        hcards.append(as_torch(h, dev))
      self.last_action_mask, self.last_h_card_ids = masks, hcards
      for part, m in zip(res, masks):
        if isinstance(part, tuple) and isinstance(part[-1], dict):
          part[-1]["action_mask"] = m
      return res
    cls.recv_torch = recv_torch
  return classes
