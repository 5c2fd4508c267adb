"""Python front-end of an EnvPool (counterpart of envpool2/python/envpool.py:28-174 and the
`_to` converters of gym_envpool.py:74-95, gymnasium_envpool.py:73-88, dm_envpool.py:72-86)."""
import pprint
import warnings
from collections import namedtuple
from typing import Any, Dict, List, Optional

import numpy as np

from .spec import check_unique
from .streamstep import current_stream_handle, fresh, refresh_terminated, slot_entry
from .tree import Layout

try:  # dm_env is optional; the local TimeStep has the same fields and helpers
  from dm_env import TimeStep
except ImportError:

  class TimeStep(namedtuple("TimeStep", ["step_type", "reward", "discount", "observation"])):
    __slots__ = ()

    def first(self):
      return self.step_type == 0

    def mid(self):
      return self.step_type == 1

    def last(self):
      return self.step_type == 2


import functools


@functools.lru_cache(maxsize=None)
def _gym_is_new_api() -> bool:
  # (cached: a failing `import gym` walks sys.path every time, 60 us per step otherwise)
  try:
    import gym
    from packaging import version
    return version.parse(gym.__version__) >= version.parse("0.26.0")
  except ImportError:
    return True


def _flatten_action(action: Dict[str, Any], prefix: str = "") -> Dict[str, Any]:
  flat = {}
  for k, v in action.items():
    key = f"{prefix}.{k}" if prefix else k
    if isinstance(v, dict):
      flat.update(_flatten_action(v, key))
    else:
      flat[key] = v
  return flat


class EnvPoolFrontEnd:
  """send / recv / step / reset over the raw `_send/_recv/_reset` of the pybind class."""

  _flavour = "raw"

  def __init__(self, spec):
    super().__init__(spec)
    self.spec = spec
    check_unique(type(self).__name__, "state", list(self._state_keys))
    check_unique(type(self).__name__, "action", list(self._action_keys))
    self._layout = Layout(self._state_keys)
    self._dm_layout = Layout(self._state_keys, root="State", merge_info_into_obs=True)
    self._action_checked = False

  # ---- properties ----
  def __len__(self) -> int:
    return self.config["num_envs"]

  @property
  def config(self) -> Dict[str, Any]:
    return dict(zip(self._spec._config_keys, self._spec._config_values))

  @property
  def _cfg(self) -> Dict[str, Any]:
    """`config` built once, for the per-step paths (the property above makes a new dict from the
    spec's tuples on every access, several microseconds each)."""
    c = self.__dict__.get("_cfg_cached")
    if c is None:
      c = self.__dict__["_cfg_cached"] = self.config
    return c

  @property
  def all_env_ids(self) -> np.ndarray:
    if not hasattr(self, "_all_env_ids"):
      off = self.config.get("env_id_offset", 0)
      self._all_env_ids = np.arange(off, off + self.config["num_envs"], dtype=np.int32)
    return self._all_env_ids

  @property
  def is_async(self) -> bool:
    c = self.config
    return c["batch_size"] > 0 and c["num_envs"] != c["batch_size"]

  def seed(self, seed=None) -> None:
    warnings.warn("`seed` is abandoned; pass make(..., seed=seed) instead.", stacklevel=2)

  # ---- action marshalling (envpool.py:33-85) ----
  def _check_action(self, actions: List[np.ndarray]) -> None:
    if self._action_checked:
      return
    self._action_checked = True
    for a, (k, v) in zip(actions, self.spec.action_array_spec.items()):
      if v.dtype != a.dtype:
        raise RuntimeError(f'Expected dtype {v.dtype} with action "{k}", got {a.dtype}')
      shape = tuple(v.shape)
      if len(shape) > 0 and shape[0] == -1:
        if a.shape[1:] != shape[1:]:
          raise RuntimeError(f'Expected shape {shape} with action "{k}", got {a.shape}')
      elif len(a.shape) == 0 or a.shape[1:] != shape:
        raise RuntimeError(f'Expected shape {("num_env", *shape)} with action "{k}", got {a.shape}')

  def _from(self, action, env_id: Optional[np.ndarray] = None) -> List[np.ndarray]:
    keys = self._spec._action_keys
    if hasattr(action, "detach") and hasattr(action, "cpu"):
      action = action.detach().cpu().numpy()  # a torch tensor (e.g. a GPU policy's choice): the pool's inputs are host arrays
    if env_id is not None and hasattr(env_id, "detach") and hasattr(env_id, "cpu"):
      env_id = env_id.detach().cpu().numpy()
    if isinstance(action, dict):
      adict = _flatten_action(action)
    else:
      last_dtype = self._spec._action_spec[-1][0]
      if isinstance(action, np.ndarray):
        action = action.astype(last_dtype, order="C")
      adict = {keys[-1]: action}
    if env_id is None:
      adict.setdefault("env_id", self.all_env_ids)
    else:
      adict["env_id"] = env_id.astype(np.int32)
    adict.setdefault("players.env_id", adict["env_id"])
    return [adict[k] for k in keys]

  # ---- the API ----
  def send(self, action, env_id: Optional[np.ndarray] = None) -> None:
    arrays = self._from(action, env_id)
    self._check_action(arrays)
    self._send(arrays)

  def recv(self, reset: bool = False, return_info: bool = True):
    return self._with_mask(self._to(self._recv(), reset, return_info))

  def _with_mask(self, result, to_torch=None):
    """Pools whose raw class keeps an explicit legal-action mask of the last batch
    (`last_action_mask`, YGOPro: arange(max_options) < info:num_options, written by the encode
    kernel) hand it out as info["action_mask"] (additive; the dm flavour has no info dict)."""
    mask = getattr(self, "last_action_mask", None)
    if mask is not None and isinstance(result, tuple) and isinstance(result[-1], dict):
      result[-1]["action_mask"] = to_torch(mask) if to_torch else mask
    return result

  def async_reset(self) -> None:
    self._reset(self.all_env_ids)

  def step(self, action, env_id: Optional[np.ndarray] = None):
    self.send(action, env_id)
    return self.recv(reset=False, return_info=True)

  def reset(self, env_id: Optional[np.ndarray] = None):
    if env_id is None:
      env_id = self.all_env_ids
    self._reset(env_id)
    return self.recv(reset=True, return_info=self._cfg["gym_reset_return_info"])

  def xla(self):
    raise RuntimeError("XLA is disabled (jax is not part of this build).")

  # ---- additive: PyTorch tensors in, PyTorch tensors out, nothing crosses PCIe -----------------
  def _torch_out(self, descs, reset: bool):
    from ..interop import as_torch
    import torch
    if not hasattr(self, "_torch_views"):
      self._torch_views = {}
      self._torch_dev = torch.device("cuda", self.config.get("device_id", 0))
    cache = self._torch_views  # the ring has ring_depth slots: the same views come round again
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
        t = cache[key] = as_torch(d, self._torch_dev)
      vals.append(t)
    return self._to(vals, reset, True)

  def reset_torch(self):
    """reset() of every env with the batch returned as torch tensors that alias the device ring
    (valid until `ring_depth - 1` further batches have been received)."""
    if not hasattr(self, "_recv_device"):
      raise RuntimeError(f"{type(self).__name__} has no device-resident path")
    self._reset(self.all_env_ids)
    descs, _, _ = self._recv_device()
    return self._torch_out(descs, True)

  def recv_torch(self, reset: bool = False):
    """recv() with the batch as torch tensors aliasing device memory (valid until the next recv of
    a YGOPro pool, until `ring_depth - 1` further batches of a Dummy pool)."""
    if not hasattr(self, "_recv_device"):
      raise RuntimeError(f"{type(self).__name__} has no device-resident path")
    if not reset and hasattr(self, "_recv_device_slot") and (getattr(self, "_fixed_rows", False) or self._cfg["max_num_players"] == 1):
      # the result over a slot's views is kept (python/streamstep.py: a Dummy pool's ring slots and a
      # YGOPro encoder's output slots come round again); `terminated` comes from the step kernels'
      # array beside the slot where there is one, else from one small launch
      import torch
      if not hasattr(self, "_torch_dev"):
        self._torch_dev = torch.device("cuda", self._cfg.get("device_id", 0))
        self._torch_views = {}
      if not hasattr(self, "_slot_results"):
        self._slot_results = {}
      key = self._recv_device_slot()
      ent = self._slot_results.get(key)
      if ent is None:
        ent = slot_entry(self, self._last_descs()[0], self._torch_dev, self._slot_results, key)
        mask = getattr(self, "last_action_mask", None)  # (YGOPro: descriptors of the two additive keys of THIS slot)
        if mask is not None:
          ent.append(None)  # (the place of the terminated plan, streamstep.refresh_terminated)
          ent.append((self.last_h_card_ids, mask))
          if isinstance(ent[0], tuple) and isinstance(ent[0][-1], dict):
            ent[0][-1]["action_mask"] = self._torch_view(mask)
      elif len(ent) > 3:
        self.last_h_card_ids, self.last_action_mask = ent[3]
      refresh_terminated((ent,), (current_stream_handle(self._torch_dev),))
      return fresh(ent[0])
    descs, _, _ = self._recv_device()
    return self._with_mask(self._torch_out(descs, reset), self._torch_view)

  def _torch_view(self, desc):
    """One cached torch view per (address, shape): the ring's slots come round again."""
    key = (desc[0], tuple(desc[1]))
    t = self._torch_views.get(key)
    if t is None:
      from ..interop import as_torch
      t = self._torch_views[key] = as_torch(desc, self._torch_dev)
    return t

  def step_torch(self, action, env_id=None, stream_ordered: Optional[bool] = None):
    """step() for a policy that lives on the GPU: `action` is a torch CUDA tensor (the last action
    key) or a dict of them in the shapes step() takes, `env_id` an int32 CUDA tensor or None for all
    envs in order. Returns what step() returns, as torch tensors aliasing the device ring (valid
    until `ring_depth - 1` further batches have been received).

    A sync single-player pool stepped as a whole is stepped IN STREAM ORDER by default
    (stream_ordered=None or True): the step is enqueued on torch's current stream, which the pool
    adopts as its step stream (ep_pool_set_stream + ep_pool_step_on_stream), and the call returns at
    once - one kernel launch, no host wait (7 us per call at 65 536 envs). The result is valid for
    work queued on that stream afterwards, like the output of any torch op (reading it on the host
    synchronises the stream), and a step's error (the list_action self-check) surfaces at a later
    step or at `_sync()` rather than at the step itself, as CUDA errors do in torch.

    stream_ordered=False is the BLOCKING form (ep_pool_step, 10 us): the rows are complete and
    errors are raised when the call returns. It takes no kernel launch (the pool's resident kernel
    answers a doorbell) as long as the action tensors keep their addresses: pass contiguous int32 /
    float64 tensors and reuse them (a tensor that needs a dtype or layout conversion is copied into
    a staging tensor the pool keeps, so the addresses the resident kernel is bound to stay the same
    either way). While that kernel sits on the GPU a device-wide wait (torch.cuda.synchronize())
    between two steps has to wait for it to leave - the pool's watchdog bounds that to about ten
    step gaps, 200 us at least, and falls back to launching per step for callers that keep doing
    it; use torch.cuda.current_stream().synchronize() instead (include/envpool_amd.h,
    ep_pool_step). Every other case (async pools, partial sends, several players) is a send and a
    blocking recv."""
    import torch
    if not hasattr(self, "_step_device"):
      raise RuntimeError(f"{type(self).__name__} has no device-resident path")
    keys = self._spec._action_keys
    if not hasattr(self, "_torch_dev"):
      self._torch_dev = torch.device("cuda", self._cfg.get("device_id", 0))
      self._torch_views = {}
    dev = self._torch_dev
    # the usual loop passes the same tensors every step: their pointers and sizes are kept
    sig = None
    if isinstance(action, torch.Tensor):
      sig = (None, action.data_ptr(), action.shape, action.dtype)
    elif isinstance(action, dict) and all(isinstance(v, torch.Tensor) for v in action.values()):
      sig = tuple((k, v.data_ptr(), v.shape, v.dtype) for k, v in action.items())
    if sig is not None:
      sig = (sig, None if env_id is None else (env_id.data_ptr(), env_id.shape, env_id.dtype) if isinstance(env_id, torch.Tensor) else id(env_id))
    cached = getattr(self, "_step_torch_cached", None)
    if sig is not None and cached is not None and cached[0] == sig:
      ptrs, dims, keep = cached[1], cached[2], cached[3]
    else:
      adict = _flatten_action(action) if isinstance(action, dict) else {keys[-1]: action}
      if env_id is None:
        if not hasattr(self, "_all_env_ids_torch"):
          self._all_env_ids_torch = torch.from_numpy(self.all_env_ids).to(dev)
        env_id = self._all_env_ids_torch
      adict["env_id"] = env_id
      adict.setdefault("players.env_id", env_id)
      if not hasattr(self, "_torch_want"):
        names = {"int32": torch.int32, "float64": torch.float64, "float32": torch.float32}
        self._torch_want = [names[str(spec[0])] for spec in self._spec._action_spec]
      ptrs, dims, keep = [], [], []
      converted = False
      for k, want in zip(keys, self._torch_want):
        t = adict.get(k)
        if t is None:  # keys the env never reads may be left out: a stand-in of the right length
          t = adict["players.env_id"] if k.startswith("players.") else env_id
        if not (isinstance(t, torch.Tensor) and t.is_cuda):
          raise RuntimeError(f'step_torch: action "{k}" must be a CUDA tensor')
        if t.dtype != want or not t.is_contiguous():
          if not hasattr(self, "_torch_stage"):
            self._torch_stage = {}
          st = self._torch_stage.get(k)
          if st is None or st.shape != t.shape:
            st = self._torch_stage[k] = torch.empty(t.shape, dtype=want, device=dev)
          st.copy_(t)
          t = st
          converted = True  # (a copy of the caller's tensor: to be taken again on every step)
        keep.append(t)
        ptrs.append(t.data_ptr())
        dims.append(int(t.shape[0]))
      self._step_torch_cached = None if (converted or sig is None) else (sig, ptrs, dims, keep)
    if not hasattr(self, "_slot_results"):
      self._slot_results = {}
    h = current_stream_handle(dev)
    self._keep_actions = keep
    cfg = self._cfg
    if stream_ordered is not False and hasattr(self, "_send_on_stream") and dims[0] == cfg["num_envs"] == cfg["batch_size"] \
        and cfg["max_num_players"] == 1:
      self._poll_errors()  # what earlier stream-ordered steps have flagged by now
      if getattr(self, "_adopted_stream", None) != h:
        self._set_stream(h)
        self._adopted_stream = h
        self._adopted_stream_obj = torch.cuda.current_stream(dev)  # (kept alive for as long as the pool launches on it)
      self._send_on_stream(ptrs, dims, h)
      key = self._recv_slot_on_stream(h)
      ent = self._slot_results.get(key)
      if ent is None:
        ent = slot_entry(self, self._last_descs()[0], dev, self._slot_results, key)
    else:
      # the actions must be complete before the pool reads them: they were produced on torch's stream
      from ..dummy.dummy_envpool import stream_synchronize
      stream_synchronize(dev.index, h)
      if cfg["max_num_players"] > 1:  # (the player rows of a slot change from step to step: nothing to keep)
        return self._torch_out(self._step_device(ptrs, dims)[0], False)
      key = self._step_device_slot(ptrs, dims)
      ent = self._slot_results.get(key)
      if ent is None:
        ent = slot_entry(self, self._last_descs()[0], dev, self._slot_results, key)
    # the result over this ring slot's views is kept (python/streamstep.py); `terminated` is its one
    # computed element: refreshed by one small launch on torch's stream, like `done > trunc` would be
    refresh_terminated((ent,), (h,))
    return fresh(ent[0])

  def rollout_torch(self, action, T: Optional[int] = None):
    """T lockstep steps of every env in ONE kernel launch (ep_pool_rollout), for action sequences
    that are on the GPU already: `action` is the `list_action` CUDA tensor, float64 `[T, N, 6]` (or
    `[N, 6]` with `T` given: the same rows every step), or a dict that may also hold
    "players.env_id", int32 `[T, N]` or `[N]`. Sync single-player pools made with
    `ring_depth >= T`. Returns a list of T step results, what `step_torch` returns for each step,
    as views of T ring slots - enqueued on torch's current stream like
    `step_torch(..., stream_ordered=True)` (the pool adopts that stream; no host wait; errors
    surface at a later call or `_sync()`). This is the path bench.py times: 1.4 us per
    65 536-env step instead of 4.5 (one launch per step) or 6.7 (blocking)."""
    import torch
    if not hasattr(self, "_rollout") or not hasattr(self, "_recv_slots_on_stream"):
      raise RuntimeError(f"{type(self).__name__} has no fused rollout")
    if not hasattr(self, "_torch_dev"):
      self._torch_dev = torch.device("cuda", self.config.get("device_id", 0))
      self._torch_views = {}
    dev = self._torch_dev
    adict = _flatten_action(action) if isinstance(action, dict) else {"list_action": action}
    la = adict["list_action"]
    if not (isinstance(la, torch.Tensor) and la.is_cuda):
      raise RuntimeError('rollout_torch: "list_action" must be a CUDA tensor')
    n = self._cfg["num_envs"]
    if la.dim() == 3:
      T = int(la.shape[0]) if T is None else T
      if la.shape[0] != T:
        raise RuntimeError(f"rollout_torch: list_action has {la.shape[0]} steps, T = {T}")
    if T is None or T < 1:
      raise RuntimeError("rollout_torch: give T with a [N, 6] list_action")
    if tuple(la.shape[-2:]) != (n, self._cfg["action_num"]):
      raise RuntimeError(f"rollout_torch: list_action must end in [{n}, {self._cfg['action_num']}], got {tuple(la.shape)}")
    if la.dtype != torch.float64 or not la.is_contiguous():
      la = la.to(torch.float64).contiguous()
    if not hasattr(self, "_all_env_ids_torch"):
      self._all_env_ids_torch = torch.from_numpy(self.all_env_ids).to(dev)
    ids = self._all_env_ids_torch
    penv = adict.get("players.env_id", ids)
    if not (isinstance(penv, torch.Tensor) and penv.is_cuda) or tuple(penv.shape[-1:]) != (n,) or (penv.dim() == 2 and penv.shape[0] != T):
      raise RuntimeError(f"rollout_torch: players.env_id must be a CUDA tensor [{T}, {n}] or [{n}]")
    if penv.dtype != torch.int32 or not penv.is_contiguous():
      penv = penv.to(torch.int32).contiguous()
    if not hasattr(self, "_slot_results"):
      self._slot_results = {}
    h = current_stream_handle(dev)
    self._poll_errors()
    if getattr(self, "_adopted_stream", None) != h:
      self._set_stream(h)
      self._adopted_stream = h
      self._adopted_stream_obj = torch.cuda.current_stream(dev)
    self._keep_actions = [la, penv]
    ptrs = [ids.data_ptr(), penv.data_ptr(), la.data_ptr(), ids.data_ptr(), ids.data_ptr()]
    strides = [0, n if penv.dim() == 2 else 0, n if la.dim() == 3 else 0, 0, 0]
    self._rollout(T, ptrs, [n] * 5, strides)
    ents = []
    for i, key in enumerate(self._recv_slots_on_stream(h, T)):
      ent = self._slot_results.get(key)
      ents.append(ent if ent is not None else slot_entry(self, self._descs_of(i)[0], dev, self._slot_results, key))
    refresh_terminated(ents, [h] * T)
    return [fresh(e[0]) for e in ents]

  def __repr__(self) -> str:
    cfg = ", ".join(f"{k}={pprint.pformat(v)}" for k, v in self.config.items())
    return f"{type(self).__name__}({cfg})"

  __str__ = __repr__

  # ---- output shaping ----
  def _to(self, values, reset: bool, return_info: bool):
    state = self._layout.build_dict(values)
    info = state["info"]
    info["elapsed_step"] = state["elapsed_step"]
    if reset:
      return state["obs"], info
    terminated = state["done"] > state["trunc"]  # done & ~trunc on booleans, in one pass (one kernel for tensors)
    return state["obs"], state["reward"], terminated, state["trunc"], info


  def _term_index(self):
    """Where `_to(values, False, True)` puts `terminated` (the one element that is computed, not a
    view of the batch), or None when the flavour has none. See _to_slot."""
    return 2

  def _to_slot(self, values):
    """The step result for `values` (views of ONE ring slot) split into what stays true for as long as
    those views live and what has to be computed on every step: (`_to`'s result as a template,
    index of `terminated` in it or None, the slot's `done` view, its `trunc` view). A front-end
    that sees the same slots come round again (python/sharded.py) builds the template once."""
    state = self._layout.build_dict(values)
    return self._to(values, False, True), self._term_index(), state["done"], state["trunc"]


class GymnasiumFrontEnd(EnvPoolFrontEnd):
  _flavour = "gymnasium"

  @property
  def observation_space(self):
    return self.spec.gymnasium_observation_space

  @property
  def action_space(self):
    return self.spec.gymnasium_action_space


class GymFrontEnd(EnvPoolFrontEnd):
  _flavour = "gym"

  @property
  def observation_space(self):
    return self.spec.observation_space

  @property
  def action_space(self):
    return self.spec.action_space

  def _term_index(self):
    return 2 if _gym_is_new_api() else None

  def _to(self, values, reset: bool, return_info: bool):
    new_api = _gym_is_new_api()
    state = self._layout.build_dict(values)
    if reset and not (return_info or new_api):
      return state["obs"]
    info = state["info"]
    if not new_api:
      info["TimeLimit.truncated"] = state["trunc"]
    info["elapsed_step"] = state["elapsed_step"]
    if reset:
      return state["obs"], info
    if new_api:
      return state["obs"], state["reward"], state["done"] > state["trunc"], state["trunc"], info  # (done & ~trunc)
    return state["obs"], state["reward"], state["done"], info


class DMFrontEnd(EnvPoolFrontEnd):
  _flavour = "dm"

  def observation_spec(self):
    return self.spec.observation_spec()

  def action_spec(self):
    return self.spec.action_spec()

  def _term_index(self):
    return None

  def _to(self, values, reset: bool, return_info: bool):
    state = self._dm_layout.build_namedtuple("State", values)
    return TimeStep(step_type=state.step_type, reward=state.reward, discount=state.discount,
                    observation=state.State)


def make_pool_class(name: str, base: type, front: type) -> type:
  return type(base)(name, (front, base), {})
