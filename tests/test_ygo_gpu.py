"""GPU parity of the YGOPro observation-encode kernel, through the C ABI: for reference-form
states S, kernel(ep_ygo_emit_record(S)) == oracle(S), bit for bit on every key.
(The oracle is pinned by hand-derived vectors only: parity unpinned by reference data.)"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from envpool_amd import capi, capi_ygo
from tests import ygo_build

pytestmark = pytest.mark.gpu
K = capi_ygo.YGO_STATE_KEYS
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ygo_known_answers.json")))


def _compare(got, want, rows, ctx=""):
  for k, name in enumerate(K):
    g, w = np.asarray(got[k])[rows], np.asarray(want[k])[rows]
    assert g.dtype == w.dtype and g.shape == w.shape, f"{ctx} {name}: {g.dtype}{g.shape} vs {w.dtype}{w.shape}"
    if g.tobytes() != w.tobytes():
      bad = np.nonzero((g.reshape(len(rows), -1) != w.reshape(len(rows), -1)).any(axis=1))[0]
      r = bad[0]
      raise AssertionError(f"{ctx} {name}: {len(bad)} envs differ, first env {rows[r]}:\n got {g[r]}\nwant {w[r]}")


def test_known_answers_on_gpu():
  from oracle import ygo_oracle
  cfg = capi_ygo.default_config(max_batch=64)
  table = ygo_build.make_table(GOLD["table"])
  arr, keep = ygo_build.batch_of([c["state"] for c in GOLD["cases"]])
  n = len(GOLD["cases"])
  want, status = ygo_oracle.write_batch(cfg, table, arr, n)
  assert (status == 0).all()
  recs, est = capi_ygo.emit_records(cfg, arr, n)
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  got = enc.featurize(recs)
  _compare(got, want, np.arange(n), "known answers")
  # and the HIP path directly against the hand-derived rows (not through the oracle)
  def full(row, width):
    if isinstance(row, dict):
      out = [0] * width
      for c, v in row.items():
        out[int(c)] = v
      return out
    return row
  for i, case in enumerate(GOLD["cases"]):
    e = case["expect"]
    assert int(got[12][i]) == e["num_options"], case["name"]
    assert got[9][i].tolist() == e["global"], case["name"]
    for r, row in e.get("cards_rows", {}).items():
      assert got[8][i][int(r)].tolist() == full(row, 39), (case["name"], "cards row", r)
    assert np.nonzero(got[8][i].any(axis=1))[0].tolist() == e["nonzero_card_rows"], case["name"]
    for r, row in e.get("actions_rows", {}).items():
      assert got[10][i][int(r)].tolist() == full(row, 19), (case["name"], "actions row", r)
    for r, row in e.get("h_card_ids", {}).items():
      assert got[16][i][int(r)].tolist() == row, (case["name"], "h_card_ids row", r)
    for key, idx in (("win_reason", 15), ("is_selfplay", 14), ("to_play", 13), ("done", 3), ("step_type", 6)):
      if key in e:
        assert int(got[idx][i]) == e[key], (case["name"], key)
    if "reward" in e:
      assert float(got[4][i]) == e["reward"], case["name"]
  # several ring pushes between two WriteStates (Q11), then the rotated copy-out on the device
  for h in GOLD["history_push_cases"]:
    if "then_write_state" not in h:
      continue
    nh = h["n_history_actions"]
    cfg_h = capi_ygo.default_config(max_batch=8, n_history_actions=nh)
    L = capi_ygo.lib()
    ring = np.zeros((nh, 19), np.uint8)
    p = C.c_int32(0)
    for msg, opt, ids in h["script"]:
      idarr = np.asarray(ids, np.uint16)
      assert L.ep_ygo_history_push(C.byref(cfg_h), ring.ctypes.data, C.byref(p), msg, opt.encode(), idarr.ctypes.data, len(ids)) == 0
    w = h["then_write_state"]
    st = dict(w["state"], history={f"p{w['state']['to_play']}": ring}, ha_p=[p.value, p.value])
    arr_h, keep_h = ygo_build.batch_of([st], n_hist=nh)
    recs_h, _ = capi_ygo.emit_records(cfg_h, arr_h, 1)
    enc_h = capi_ygo.YgoEncoderC(cfg_h, table, len(table))
    got_h = enc_h.featurize(recs_h)
    assert got_h[11][0].tolist() == [full(w["h_actions_rows"].get(str(r), {}), 19) for r in range(nh)], h["name"]
    enc_h.close()


@pytest.mark.parametrize("n,seed,kw", [
  (1, 1, {}), (7, 2, {}), (8, 3, {}), (1000, 4, {}), (20000, 5, {}),
  (20000, 15, {}), (20000, 25, {}), (20000, 35, {}), (20000, 45, {}),  # 10^5 randomised records in all
  (513, 6, dict(max_cards=60, max_options=12, n_history_actions=8, max_multi_select=3)),
  (300, 7, dict(max_cards=100, max_options=24, n_history_actions=20, max_multi_select=8)),
  # history ring longer than 512 bytes (tail path), every limit at its maximum
  (200, 8, dict(max_cards=127, max_options=64, n_history_actions=40, max_multi_select=8)),
  # a record larger than the card image (the landing area decides the LDS carve), one spec per option
  (257, 9, dict(max_cards=40, max_options=32, n_history_actions=30, max_multi_select=1)),
])
def test_synthetic_parity(n, seed, kw):
  from oracle import ygo_oracle
  from tools import ygo_synth
  cfg = capi_ygo.default_config(max_batch=max(n, 8), **kw)
  table = ygo_synth.make_table(1000, seed=seed)
  sb = ygo_synth.SynthBatch(cfg, table, n, seed=seed)
  want, status = ygo_oracle.write_batch(cfg, table, sb.raw, n)
  recs, est = capi_ygo.emit_records(cfg, sb.raw, n)
  ok = np.nonzero((status == 0) & (est == 0))[0]
  assert len(ok) > 0.9 * n
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  got = enc.featurize(recs, allow=(capi.EP_ERR_ACTION_CHECK,))
  # the batch reports "the reference would throw" exactly when the oracle says so for some record
  assert (enc.last_status == capi.EP_ERR_ACTION_CHECK) == bool(((status != 0) & (est == 0)).any())
  _compare(got, want, ok, f"n={n} seed={seed}")
  # the same records resident in HBM (the other fetch schedule of the kernel) and, for host
  # records, the staged-copy way to the device: all three give the same bytes
  if n <= 1000:
    import torch
    t = torch.from_numpy(recs.copy()).to("cuda:0")
    torch.cuda.synchronize()
    got_dev = enc.featurize_device(t.data_ptr(), n, allow=(capi.EP_ERR_ACTION_CHECK,))
    _compare(got_dev, want, ok, f"device records n={n} seed={seed}")
    assert capi_ygo.lib().ep_ygo_set_zero_copy(enc.handle, 0) == 0
    got_staged = enc.featurize(recs, allow=(capi.EP_ERR_ACTION_CHECK,))
    _compare(got_staged, want, ok, f"staged records n={n} seed={seed}")
  enc.close()


def test_throw_conditions_are_reported():
  from oracle import ygo_oracle
  cfg = capi_ygo.default_config(max_batch=16)
  table = ygo_build.make_table(GOLD["table"])
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  good = GOLD["cases"][0]["state"]
  for st in ({"options": ["s h1"], "msg": 11}, {"options": ["y"], "msg": 13, "phase": 3},
             {"options": ["y"], "msg": 13, "field": {"p0": {"hand": [{"code": 424242, "seq": 0}]}}},
             {"options": ["y"], "msg": 13, "field": {"p0": {"mzone": [{"code": 89631139, "seq": 0, "pos": 6}]}}}):
    arr, keep = ygo_build.batch_of([good, st, good])
    _, status = ygo_oracle.write_batch(cfg, table, arr, 3)
    assert status.tolist() == [0, 6, 0]
    recs, est = capi_ygo.emit_records(cfg, arr, 3)
    assert (est == 0).all()
    enc.featurize(recs, allow=(capi.EP_ERR_ACTION_CHECK,))
    assert enc.last_status == capi.EP_ERR_ACTION_CHECK
  arr, keep = ygo_build.batch_of([good])
  recs, _ = capi_ygo.emit_records(cfg, arr, 1)
  enc.featurize(recs)
  assert enc.last_status == 0  # the flag does not stick


def test_more_rows_than_max_cards_is_flagged_and_neighbours_stay_exact():
  """The reference has no bound on rows per player (it would overrun obs:cards_); the oracle
  reports a throw, the kernel flags the batch and leaves the offending location out, and the other
  records of the batch are untouched by it."""
  from oracle import ygo_oracle
  cfg = capi_ygo.default_config(max_batch=16, max_cards=4)
  table = ygo_build.make_table(GOLD["table"])
  code = GOLD["table"][0]["code"]
  ok_state = {"options": ["y"], "msg": 13, "field": {"p0": {"grave": [{"code": code, "seq": i} for i in range(3)]}}}
  over_own = {"options": ["y"], "msg": 13, "field": {"p0": {"hand": [{"code": code, "seq": 0}],
                                                            "grave": [{"code": code, "seq": i} for i in range(5)]}}}
  over_opp = {"options": ["y"], "msg": 13, "field": {"p1": {"deck": [{"code": code, "seq": i} for i in range(6)]}}}
  arr, keep = ygo_build.batch_of([ok_state, over_own, ok_state, over_opp, ok_state], n_hist=16, nf=19)
  want, status = ygo_oracle.write_batch(cfg, table, arr, 5)
  assert status.tolist() == [0, 6, 0, 6, 0]
  recs, est = capi_ygo.emit_records(cfg, arr, 5)
  assert (est == 0).all()
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  got = enc.featurize(recs, allow=(capi.EP_ERR_ACTION_CHECK,))
  assert enc.last_status == capi.EP_ERR_ACTION_CHECK
  _compare(got, want, np.array([0, 2, 4]), "neighbours of overflowing records")
  cards = got[8]
  assert cards[1][0].any() and not cards[1][1:4].any()   # own hand row written, the 5-card grave left out
  assert not cards[3][4:].any()                          # the 6-row hidden deck left out
  enc.close()


def test_exact_spec_search_paths():
  """Specs the direct-mapped table cannot hold (sequence >= max_cards) and envs with two rows under
  one spec (never in a consistent duel; the reference's map keeps the LAST row) go through the
  wave-cooperative exact search."""
  from oracle import ygo_oracle
  cfg = capi_ygo.default_config(max_batch=16)
  table = ygo_build.make_table(GOLD["table"])
  codes = [c["code"] for c in GOLD["table"]]
  dup = {"options": ["s m3", "t m3", "c g1"], "msg": 11,
         "field": {"p0": {"mzone": [{"code": codes[0], "seq": 2}, {"code": codes[1], "seq": 2}, {"code": codes[2], "seq": 0}],
                          "grave": [{"code": codes[1], "seq": 0}]}}}
  far = {"options": ["g201", "g3 g201"], "msg": 15,
         "field": {"p0": {"grave": [{"code": codes[0], "seq": 2}, {"code": codes[1], "seq": 200}]},
                   "p1": {"grave": [{"code": codes[2], "seq": 200}]}}}
  both = {"options": ["os1", "os1 og150"], "msg": 15,
          "field": {"p1": {"szone": [{"code": codes[0], "seq": 0}, {"code": codes[1], "seq": 0}],
                           "grave": [{"code": codes[2], "seq": 149}]}}}
  plain = GOLD["cases"][0]["state"]
  arr, keep = ygo_build.batch_of([plain, dup, far, both, plain])
  want, status = ygo_oracle.write_batch(cfg, table, arr, 5)
  assert status.tolist() == [0] * 5
  recs, est = capi_ygo.emit_records(cfg, arr, 5)
  assert (est == 0).all()
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  got = enc.featurize(recs)
  _compare(got, want, np.arange(5), "exact search")
  acts = got[10]
  assert acts[1][0][1] == 2 and acts[1][1][1] == 2  # "m3" -> the LAST of the two rows (row index 2)
  assert acts[2][0][1] == 2 and acts[2][1][:4].tolist() == [0, 1, 0, 2]
  enc.close()


@pytest.mark.parametrize("zero_copy", [1, 0])
def test_submit_collect_pipeline_and_device_records(zero_copy):
  """Several batches in flight, results in FIFO order, both ways a pinned slot reaches the kernel
  (read in place over PCIe / staged by a side-stream copy); device-resident records give the same
  bytes."""
  import torch
  from oracle import ygo_oracle
  from tools import ygo_synth
  cfg = capi_ygo.default_config(max_batch=256, ring_depth=3)
  table = ygo_synth.make_table(500, seed=9)
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  L = capi_ygo.lib()
  assert L.ep_ygo_set_zero_copy(enc.handle, zero_copy) == 0
  batches = []
  for b in range(7):
    n = [256, 100, 8, 255, 1, 64, 200][b]
    sb = ygo_synth.SynthBatch(cfg, table, n, seed=100 + b)
    want, status = ygo_oracle.write_batch(cfg, table, sb.raw, n)
    recs, est = capi_ygo.emit_records(cfg, sb.raw, n)
    batches.append((recs, want, np.nonzero((status == 0) & (est == 0))[0]))
  pending = []
  for recs, want, ok in batches:
    if len(pending) == 3:
      with pytest.raises(capi.EpError) as e:
        enc.submit(recs)
      assert e.value.code == capi.EP_ERR_RING_FULL
      w, o = pending.pop(0)
      _compare(enc.collect(allow=(capi.EP_ERR_ACTION_CHECK,)), w, o, "pipelined")
    enc.submit(recs)
    if len(pending) == 0:
      assert L.ep_ygo_set_zero_copy(enc.handle, zero_copy) == capi.EP_ERR_INVALID  # a batch is in flight
    pending.append((want, ok))
  while pending:
    w, o = pending.pop(0)
    _compare(enc.collect(allow=(capi.EP_ERR_ACTION_CHECK,)), w, o, "drain")
  with pytest.raises(capi.EpError) as e:
    enc.collect()
  assert e.value.code == capi.EP_ERR_WOULD_BLOCK
  # device-resident records, device views out
  recs, want, ok = batches[0]
  t = torch.from_numpy(recs.copy()).to("cuda:0")
  torch.cuda.synchronize()
  bt = capi.Batch()
  rc = L.ep_ygo_featurize(enc.handle, t.data_ptr(), recs.shape[0], capi.EP_MEM_DEVICE, capi.EP_MEM_HOST, C.byref(bt))
  assert rc in (0, capi.EP_ERR_ACTION_CHECK)
  got = capi.batch_to_numpy(bt)
  L.ep_batch_release(C.byref(bt))
  _compare(got, want, ok, "device records")
  # the non-blocking form, two device batches in flight, interleaved with a host batch
  L = capi_ygo.lib()
  capi._check(L.ep_ygo_submit_device(enc.handle, t.data_ptr(), recs.shape[0]))
  enc.submit(batches[1][0])
  capi._check(L.ep_ygo_submit_device(enc.handle, t.data_ptr(), recs.shape[0]))
  _compare(enc.collect(allow=(capi.EP_ERR_ACTION_CHECK,)), want, ok, "submit_device 1")
  _compare(enc.collect(allow=(capi.EP_ERR_ACTION_CHECK,)), batches[1][1], batches[1][2], "host batch between")
  _compare(enc.collect(allow=(capi.EP_ERR_ACTION_CHECK,)), want, ok, "submit_device 2")


def test_python_front_end_and_action_mask():
  """envpool_amd.ygopro.YGOProEncoder: dict in `_state_keys` order + the implicit legal-action mask
  (rows >= num_options of obs:actions_ stay zero)."""
  import envpool_amd
  from envpool_amd.ygopro import YGOProEncoder
  from oracle import ygo_oracle
  from tools import ygo_synth
  n = 256
  spec = envpool_amd.make_spec("YGOPro-v0", num_envs=n)
  table = ygo_synth.make_table(1000, seed=3)
  rows = [(t.code, t.type, t.level, t.attribute, t.race, t.attack, t.defense, t.card_id) for t in table]
  enc = YGOProEncoder(spec, rows)
  sb = ygo_synth.SynthBatch(enc.cfg, table, n, seed=3)
  want, status = ygo_oracle.write_batch(enc.cfg, table, sb.raw, n)
  recs, est = capi_ygo.emit_records(enc.cfg, sb.raw, n)
  out = enc.encode(recs, strict=False)
  keys = list(out)
  assert keys[:16] == list(spec._state_keys) and keys[16:] == ["aux:h_card_ids", "aux:action_mask", "mask"]
  ok = np.nonzero((status == 0) & (est == 0))[0]
  _compare([out[k] for k in capi_ygo.YGO_STATE_KEYS], want, ok, "front end")
  acts = out["obs:actions_"][ok]
  nopt = out["info:num_options"][ok]
  mask = out["mask"][ok]
  assert mask.shape == (len(ok), 16) and (mask.sum(1) == np.minimum(nopt, 16)).all()
  assert (acts[~mask] == 0).all()  # rows past num_options are all zero
  has_opts = out["obs:global_"][ok][:, 7] == 0
  assert (acts[has_opts][:, 0, 10] != 0).all()  # a real option always carries its msg id (col 2*mms)
  enc.close()


def test_full_size_batch_properties():
  """65 536 records in one launch (the bench size): (i) the first 4096 match the oracle, (ii) the
  result does not depend on where in the batch a record sits (a permuted batch gives the permuted
  result), (iii) a checksum of the whole batch equals the sum of the checksums of 1024-record
  launches over the same records."""
  from oracle import ygo_oracle
  from tools import ygo_synth
  n_gen, n = 4096, 65536
  cfg = capi_ygo.default_config(max_batch=n, ring_depth=1)
  table = ygo_synth.make_table(1000, seed=11)
  sb = ygo_synth.SynthBatch(cfg, table, n_gen, seed=11)
  want, status = ygo_oracle.write_batch(cfg, table, sb.raw, n_gen)
  recs, est = capi_ygo.emit_records(cfg, sb.raw, n_gen)
  ok = np.nonzero((status == 0) & (est == 0))[0]
  rng = np.random.default_rng(11)
  pick = rng.integers(0, n_gen, size=n)
  pick[:n_gen] = np.arange(n_gen)
  big = np.ascontiguousarray(recs[pick])
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  got = enc.featurize(big, allow=(capi.EP_ERR_ACTION_CHECK,))
  _compare([g[:n_gen] for g in got], want, ok, "full-size head")
  perm = rng.permutation(n)
  got_p = enc.featurize(np.ascontiguousarray(big[perm]), allow=(capi.EP_ERR_ACTION_CHECK,))
  for k, (a, b) in enumerate(zip(got, got_p)):
    assert np.array_equal(a[perm], b), capi_ygo.YGO_STATE_KEYS[k]
  enc.close()

  def checksum(arrs):
    return [int(a.astype(np.uint64).sum()) if a.dtype != np.float32 else int(a.view(np.uint32).astype(np.uint64).sum()) for a in arrs]

  small = capi_ygo.YgoEncoderC(capi_ygo.default_config(max_batch=1024), table, len(table))
  acc = None
  for i in range(0, n, 1024):
    part = checksum(small.featurize(big[i:i + 1024], allow=(capi.EP_ERR_ACTION_CHECK,)))
    acc = part if acc is None else [x + y for x, y in zip(acc, part)]
  small.close()
  assert acc == checksum(got)


@pytest.mark.parametrize("threads,zero_copy,packed", [(1, 1, 0), (4, 1, 0), (7, 0, 0), (5, 1, 1), (2, 0, 1)])
def test_arrival_ordered_ring_with_producer_threads(threads, zero_copy, packed):
  """BASELINE config 5's shape on one GPU (1024 envs, 512-record batches): host producer threads
  emit records in place into rows handed out in ARRIVAL order (the YGO analogue of
  state_buffer_queue.h:114-128 + state_buffer.h:79-129), the batch is encoded when its 512th record
  lands, the consumer receives batches in order. Rows arrive in any order, so the comparison with
  the oracle is by (info:env_id, elapsed_step) - bit-exact on every key."""
  import ctypes as C
  import threading

  from envpool_amd import capi
  from oracle import ygo_oracle
  from tools import ygo_synth
  N, B, steps = 1024, 512, 3
  cfg = capi_ygo.default_config(max_batch=B, ring_depth=3)
  table = ygo_synth.make_table(1000, seed=0)
  sb = ygo_synth.SynthBatch(cfg, table, N, seed=11)
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  L = capi_ygo.lib()
  capi._check(L.ep_ygo_set_zero_copy(enc.handle, zero_copy))
  capi._check(L.ep_ygo_async_begin(enc.handle, B))
  capi._check(L.ep_ygo_set_recv_timeout(enc.handle, 30000))
  # packed: records back to back in arrival order + offset table, one copy-engine transfer per batch
  capi._check(L.ep_ygo_set_packed(enc.handle, packed))
  mode = ygo_synth.PACKED_EMIT if packed else ygo_synth.RING_IN_PLACE
  res = {}
  prod = threading.Thread(target=lambda: res.update(rc=ygo_synth.run_producers(enc.handle, cfg, sb.raw, N, threads, steps, engine_us=2.0, mode=mode)))
  prod.start()
  got = {}
  for bi in range(N * steps // B):
    b = capi.Batch()
    rc = L.ep_ygo_recv(enc.handle, capi.EP_MEM_HOST, C.byref(b))
    assert rc in (0, capi.EP_ERR_ACTION_CHECK), L.ep_last_error()
    assert b.num_rows == B and b.seq == bi
    out = capi.batch_to_numpy(b)
    L.ep_batch_release(C.byref(b))
    for j in range(B):
      key = (int(out[0][j]), int(out[2][j]))
      assert key not in got, f"row {key} handed out twice"
      got[key] = tuple(np.asarray(out[k][j]).tobytes() for k in range(len(K)))
  prod.join(timeout=60)
  assert not prod.is_alive() and res["rc"][0] == 0
  # nothing more is pending
  capi._check(L.ep_ygo_set_recv_timeout(enc.handle, 0))
  b = capi.Batch()
  assert L.ep_ygo_recv(enc.handle, capi.EP_MEM_HOST, C.byref(b)) == capi.EP_ERR_WOULD_BLOCK
  # the oracle on the same states, step by step
  raw = (capi_ygo.RawState * N)()
  n_checked = 0
  for s in range(steps):
    for i in range(N):
      C.memmove(C.byref(raw[i]), C.byref(sb.raw[i]), C.sizeof(capi_ygo.RawState))
      raw[i].elapsed_step = s
    want, status = ygo_oracle.write_batch(cfg, table, raw, N)
    for i in range(N):
      if status[i] != 0:
        continue  # the reference throws on this state; the product flags the batch
      key = (int(want[0][i]), s)
      assert got[key] == tuple(np.asarray(want[k][i]).tobytes() for k in range(len(K))), key
      n_checked += 1
  assert len(got) == N * steps and n_checked > 0.9 * N * steps
  # submit/collect is refused in this mode, and featurize refuses to run beside in-flight batches
  assert L.ep_ygo_commit_slot(enc.handle, 1) == capi.EP_ERR_INVALID
  enc.close()


def test_featurize_returns_its_own_batch():
  """ep_ygo_featurize is submit + collect as one operation: with a batch of ep_ygo_submit still in
  flight it must not hand out that older batch for the records it was given."""
  import ctypes as C

  from envpool_amd import capi
  from tools import ygo_synth
  cfg = capi_ygo.default_config(max_batch=64)
  table = ygo_synth.make_table(1000, seed=0)
  sa = ygo_synth.SynthBatch(cfg, table, 8, seed=1)
  sb = ygo_synth.SynthBatch(cfg, table, 8, seed=2, first=100)
  ra, _ = capi_ygo.emit_records(cfg, sa.raw, 8)
  rb, _ = capi_ygo.emit_records(cfg, sb.raw, 8)
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  enc.submit(ra)
  with pytest.raises(capi.EpError) as e:
    enc.featurize(rb)
  assert e.value.code == capi.EP_ERR_INVALID and "in flight" in str(e.value)
  a = enc.collect(allow=(capi.EP_ERR_ACTION_CHECK,))
  b = enc.featurize(rb, allow=(capi.EP_ERR_ACTION_CHECK,))
  assert a[0].tolist() == [int(sa.raw[i].env_id) for i in range(8)]
  assert b[0].tolist() == [int(sb.raw[i].env_id) for i in range(8)]


def test_single_producer_packed_slot():
  """Submit / collect mode with the slot filled in the packed form (records back to back + offset
  table; read in place by the kernel when nothing else is in flight, else one copy-engine transfer
  of the bytes in use): ep_ygo_emit_records_packed, and
  ep_ygo_commit_slot_packed with a caller-packed slot, against the oracle."""
  import ctypes as C

  from envpool_amd import capi
  from oracle import ygo_oracle
  from tools import ygo_synth
  n = 700
  cfg = capi_ygo.default_config(max_batch=1024, ring_depth=3)
  table = ygo_synth.make_table(1000, seed=0)
  sb = ygo_synth.SynthBatch(cfg, table, n, seed=21)
  want, status = ygo_oracle.write_batch(cfg, table, sb.raw, n)
  rows = np.nonzero(status == 0)[0]
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  L = capi_ygo.lib()
  est = np.zeros(n, np.int32)
  capi._check(L.ep_ygo_emit_records_packed(enc.handle, sb.raw, n, est.ctypes.data))
  assert (est == 0).all()
  _compare(enc.collect(allow=(capi.EP_ERR_ACTION_CHECK,)), want, rows, "emit_records_packed")
  # the caller packs: used prefixes of fixed-stride records, in REVERSE order in the slot
  recs, _ = capi_ygo.emit_records(cfg, sb.raw, n)
  lens = [L.ep_ygo_record_used_bytes(C.byref(cfg), recs[i].ctypes.data) for i in range(n)]
  assert all(x % 16 == 0 and 64 <= x <= recs.shape[1] for x in lens) and sum(lens) < 0.7 * recs.size
  slot = C.c_void_p()
  capi._check(L.ep_ygo_acquire_slot(enc.handle, C.byref(slot)))
  offs = np.zeros(n, np.uint32)
  pos = 0
  for i in reversed(range(n)):
    C.memmove(slot.value + pos, recs[i].ctypes.data, lens[i])
    offs[i] = pos
    pos += lens[i]
  capi._check(L.ep_ygo_commit_slot_packed(enc.handle, n, offs.ctypes.data, pos))
  _compare(enc.collect(allow=(capi.EP_ERR_ACTION_CHECK,)), want, rows, "commit_slot_packed")
  # with nothing else in flight the kernel reads the packed slot in place; a batch committed behind
  # another one goes through the copy engine: both, back to back
  for _ in range(2):
    capi._check(L.ep_ygo_acquire_slot(enc.handle, C.byref(slot)))
    pos = 0
    for i in reversed(range(n)):
      C.memmove(slot.value + pos, recs[i].ctypes.data, lens[i])
      offs[i] = pos
      pos += lens[i]
    capi._check(L.ep_ygo_commit_slot_packed(enc.handle, n, offs.ctypes.data, pos))
  _compare(enc.collect(allow=(capi.EP_ERR_ACTION_CHECK,)), want, rows, "commit_slot_packed, in place")
  _compare(enc.collect(allow=(capi.EP_ERR_ACTION_CHECK,)), want, rows, "commit_slot_packed, copy engine")
  offs[3] += 4
  capi._check(L.ep_ygo_acquire_slot(enc.handle, C.byref(slot)))
  assert L.ep_ygo_commit_slot_packed(enc.handle, n, offs.ctypes.data, pos) == capi.EP_ERR_INVALID
  enc.close()


def test_corrupted_records_do_not_disturb_their_neighbours(monkeypatch):
  """Records are caller-supplied bytes: valid records with random bytes overwritten, counts at their
  maximum, and pure noise, host and device-resident (tools/ygo_record_fuzz.py, a few rounds of it).
  The batch comes back whole and the untouched records equal their clean encoding."""
  from tools import ygo_record_fuzz
  monkeypatch.setenv("FUZZ_SEED", "3")
  ygo_record_fuzz.main(rounds=4)


def test_commit_slot_in_parts():
  """ep_ygo_commit_slot_rows: a batch handed to the encoder as parts of consecutive rows, in any order
  (each part's encode starts at once); the call that completes the batch says so; the result equals the
  whole-slot commit's. Zero-copy and staged."""
  import ctypes as C

  from envpool_amd import capi
  from oracle import ygo_oracle
  from tools import ygo_synth
  n = 700
  cfg = capi_ygo.default_config(max_batch=1024, ring_depth=3)
  table = ygo_synth.make_table(1000, seed=0)
  sb = ygo_synth.SynthBatch(cfg, table, n, seed=33)
  want, status = ygo_oracle.write_batch(cfg, table, sb.raw, n)
  rows = np.nonzero(status == 0)[0]
  recs, _ = capi_ygo.emit_records(cfg, sb.raw, n)
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  L = capi_ygo.lib()
  done = C.c_int32(0)
  for zero_copy in (1, 0):
    capi._check(L.ep_ygo_set_zero_copy(enc.handle, zero_copy))
    slot = C.c_void_p()
    capi._check(L.ep_ygo_acquire_slot(enc.handle, C.byref(slot)))
    C.memmove(slot.value, recs.ctypes.data, recs.nbytes)
    parts = [(300, 150), (0, 300), (650, 50), (450, 200)]   # out of order
    for i, (r0, cnt) in enumerate(parts):
      capi._check(L.ep_ygo_commit_slot_rows(enc.handle, r0, cnt, n, C.byref(done)))
      assert done.value == (1 if i == len(parts) - 1 else 0)
      if i == 1:  # a row twice, a part past the end, a whole-slot commit in between: refused
        assert L.ep_ygo_commit_slot_rows(enc.handle, 0, 300, n, C.byref(done)) == capi.EP_ERR_INVALID
        assert L.ep_ygo_commit_slot_rows(enc.handle, 690, 20, n, C.byref(done)) == capi.EP_ERR_INVALID
        assert L.ep_ygo_commit_slot(enc.handle, n) == capi.EP_ERR_INVALID
    _compare(enc.collect(allow=(capi.EP_ERR_ACTION_CHECK,)), want, rows, f"commit_slot_rows zero_copy={zero_copy}")
  enc.close()


def test_results_follow_the_consumer_host_or_device():
  """Where a batch's rows are written follows the kind of the last collect (host arrays: the kernels
  write the pinned block that is handed out; device views: the slot's device arrays), and a batch
  written for the other kind is carried across: every order of host / device collects, with several
  batches in flight, gives the oracle's rows."""
  import ctypes as C

  import torch

  from envpool_amd import capi, interop
  from oracle import ygo_oracle
  from tools import ygo_synth
  n = 300
  cfg = capi_ygo.default_config(max_batch=512, ring_depth=4)
  table = ygo_synth.make_table(1000, seed=0)
  enc = capi_ygo.YgoEncoderC(cfg, table, len(table))
  L = capi_ygo.lib()
  batches = []
  for k in range(3):
    sb = ygo_synth.SynthBatch(cfg, table, n, seed=50 + k)
    want, status = ygo_oracle.write_batch(cfg, table, sb.raw, n)
    recs, _ = capi_ygo.emit_records(cfg, sb.raw, n)
    batches.append((recs, want, np.nonzero(status == 0)[0]))

  def collect(mem):
    b = capi.Batch()
    rc = L.ep_ygo_collect(enc.handle, mem, C.byref(b))
    assert rc in (0, capi.EP_ERR_ACTION_CHECK)
    if mem == capi.EP_MEM_HOST:
      out = [np.array(a) for a in capi.batch_to_numpy(b)]
      L.ep_batch_release(C.byref(b))
      return out
    out = []
    for d in capi.batch_to_descs(b):
      ptr, shape, ts = d
      if ts == "<u2":   # torch has no uint16 views on this version: read the two bytes apart
        t = interop.as_torch((ptr, tuple(shape) + (2,), "|u1"), device="cuda:0").cpu().numpy()
        out.append(t.view(np.uint16).reshape(shape))
      else:
        out.append(interop.as_torch(d, device="cuda:0").cpu().numpy())
    return out

  H, D = capi.EP_MEM_HOST, capi.EP_MEM_DEVICE
  step = 0
  for pattern in ([H, H, D], [D, D, H], [H, D, H], [D, H, D], [H, H, H], [D, D, D]):
    for recs, _, _ in batches:   # three in flight, all launched under the mode the LAST collect left
      enc.submit(recs)
    for mem, (_, want, rows) in zip(pattern, batches):
      _compare(collect(mem), want, rows, f"pattern {pattern} step {step}")
      step += 1
  enc.close()


def test_refused_calls_leave_the_encoder_and_the_pool_as_they_were(monkeypatch):
  """tools/ygo_abi_fuzz.py, a few rounds: null arguments, counts outside the batch, unknown memory
  kinds, calls of the wrong mode, configs outside the limits - each refused, and the valid encodes and
  pool steps around them unaffected."""
  from tools import ygo_abi_fuzz
  monkeypatch.setenv("FUZZ_SEED", "6")
  ygo_abi_fuzz.main(rounds=3)
