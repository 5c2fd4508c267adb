"""Multi-GPU sharding of a pool by env_id.

Envs are independent (the only shared state in the reference is the StateBuffer slot
counters, state_buffer.h:85-88), so the node-wide pool is N_gpu independent pools: rank r owns
the contiguous global ids [r*n_local, (r+1)*n_local) and seeds them with the GLOBAL id
(env.h:94), which makes every env's trajectory identical to the one it would have in a single
big pool. There is no data-path collective; the only exchange is an optional all-gather of the
per-episode (return, length) statistics, over RCCL on GPUs (backend "nccl") or gloo on CPU.
"""
from typing import Tuple


def shard_range(rank: int, world_size: int, global_num_envs: int) -> Tuple[int, int]:
  """(env_id_offset, num_envs) of `rank`. Requires an even split, like the BASELINE configs."""
  if global_num_envs % world_size != 0:
    raise ValueError(f"global num_envs {global_num_envs} is not divisible by world size {world_size}")
  n_local = global_num_envs // world_size
  return rank * n_local, n_local


def owner_of(env_id, world_size: int, global_num_envs: int):
  """(rank, local index) of global env ids (scalar or array)."""
  n_local = global_num_envs // world_size
  return env_id // n_local, env_id % n_local


def split_by_owner(env_ids, world_size: int, global_num_envs: int):
  """Host-side index for a sync-mode front-end that fans a sent batch out to the owning
  pools and stitches the results back in SENT order. Returns, per rank, the
  positions of its envs in `env_ids`."""
  import numpy as np
  env_ids = np.asarray(env_ids)
  rank, _ = owner_of(env_ids, world_size, global_num_envs)
  return [np.nonzero(rank == r)[0] for r in range(world_size)]


def all_gather_episode_stats(fin_return, fin_length, group=None):
  """All-gather of the local (return f32[n], length i32[n]) tensors; returns the global
  tensors ordered by global env_id. Works on CUDA tensors (RCCL) and CPU tensors (gloo)."""
  import torch
  import torch.distributed as dist
  world = dist.get_world_size(group)
  g_ret = torch.empty(world * fin_return.numel(), dtype=fin_return.dtype, device=fin_return.device)
  g_len = torch.empty(world * fin_length.numel(), dtype=fin_length.dtype, device=fin_length.device)
  dist.all_gather_into_tensor(g_ret, fin_return.contiguous(), group=group)
  dist.all_gather_into_tensor(g_len, fin_length.contiguous(), group=group)
  return g_ret, g_len
