"""Shared by step_torch of a pool (pool.py, both forms) and the stream-ordered step of a
sharded front-end (sharded.py:_step_on_streams): what is kept per ring slot, and the one launch that
refreshes `terminated`.

A pool's ring has `ring_depth` slots and the same device views come round again, so the whole step
result over a slot's views - obs / info dicts, reward, truncated - is built once per slot
(`EnvPoolFrontEnd._to_slot`) and handed out again whenever that slot is; the one element that is
computed rather than viewed, gymnasium's `terminated = done & ~truncated`
(gymnasium_envpool.py:73-88), gets a tensor of its own per slot that `ep_terminated_many` rewrites on
the caller's stream on every step, for all parts of a step in one launch per device."""


def slot_entry(front, descs, dev, cache, key):
  """-> [result over the slot's views, None | (done ptr, trunc ptr, terminated ptr, n, device index)]
  for a batch described by `descs` ((address, shape, typestr) per state key, as `_recv_device` /
  `_last_descs` return them); kept in `cache` under `key` (the batch's first address and row count)."""
  import torch
  from ..interop import as_torch
  vals = [None if d is None else as_torch(d, dev) for d in descs]
  tmpl, idx, done, trunc = front._to_slot(vals)
  ent = [tmpl, None]
  term_ptr = key[2] if len(key) > 2 else 0
  if idx is not None and term_ptr:
    # the step kernels wrote `terminated` beside this slot (ep_pool_terminated_of): a view, no launch
    lst = list(tmpl)
    lst[idx] = as_torch((term_ptr, tuple(done.shape), "|b1"), dev)
    ent = [tuple(lst), None]
  elif idx is not None:
    if not (type(tmpl) is tuple and done.dtype == torch.bool and trunc.dtype == torch.bool and done.is_contiguous()
            and trunc.is_contiguous()):
      raise RuntimeError("stream-ordered step: unexpected result layout")
    term = torch.empty_like(done)  # rewritten on every step this slot serves; valid as long as its views
    lst = list(tmpl)
    lst[idx] = term
    ent = [tuple(lst), (done.data_ptr(), trunc.data_ptr(), term.data_ptr(), done.numel(), dev.index)]
  if len(cache) > 1024:
    cache.clear()
  cache[key] = ent
  return ent


_PLANS = {}


def refresh_terminated(ents, handles):
  """terminated = done & ~trunc for every entry that has one, on stream `handles[i]` of its device:
  one launch of ep_terminated_many per (device, stream). The argument arrays of a launch are kept as
  a plan keyed by the addresses themselves (the same slots come round again)."""
  if len(ents) == 1:  # the unsharded step: the plan sits in the entry
    e = ents[0]
    if e[1] is not None:
      if len(e) < 3:
        e.append(None)
      if e[2] is None:
        from ..dummy.dummy_envpool import TermPlan
        p = e[1]
        e[2] = TermPlan([p[0]], [p[1]], [p[2]], [p[3]], p[4])
      e[2].launch(handles[0])
    return
  by_dev = {}
  for e, h in zip(ents, handles):
    if e[1] is not None:
      by_dev.setdefault((e[1][4], h), []).append(e[1])
  for (dev_index, h), ps in by_dev.items():
    key = tuple(ps)
    plan = _PLANS.get(key)
    if plan is None:
      from ..dummy.dummy_envpool import TermPlan
      if len(_PLANS) > 4096:
        _PLANS.clear()
      plan = _PLANS[key] = TermPlan([p[0] for p in ps], [p[1] for p in ps], [p[2] for p in ps], [p[3] for p in ps], dev_index)
    plan.launch(h)


def _copy_dict(d):
  return {k: (_copy_dict(v) if type(v) is dict else v) for k, v in d.items()}


def fresh(res):
  """A kept result as it is handed out: the tensors are the slot's views, the dicts around them are
  new ones every time, as in the reference (a caller that adds a key to `info` must not find it
  again when the slot comes round)."""
  if type(res) is tuple:
    return tuple(_copy_dict(x) if type(x) is dict else x for x in res)
  return res


def current_stream_handle(dev):
  """The raw hipStream_t of torch's current stream on `dev` as an integer. (torch.cuda.current_stream
  builds a Stream object, 4 us; the raw getter the compilers in torch use costs a tenth of that.)"""
  import torch
  try:
    return torch._C._cuda_getCurrentRawStream(dev.index)
  except AttributeError:
    return torch.cuda.current_stream(dev).cuda_stream
