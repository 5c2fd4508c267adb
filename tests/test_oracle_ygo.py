"""CPU: the YGOPro oracle (oracle/ygo_oracle.c) against the hand-derived known answers, the
host record emitter (product, host-only) against what the oracle implies, the history-ring
update, and oracle <-> emitter consistency on synthetic states."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from envpool_amd import capi_ygo
from oracle import ygo_oracle
from tests import ygo_build

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ygo_known_answers.json")))
K = capi_ygo.YGO_STATE_KEYS


def _cfg(**kw):
  return capi_ygo.default_config(**kw)


def _full(row, width):
  """Golden rows are full lists or sparse {column: value} objects."""
  if isinstance(row, dict):
    out = [0] * width
    for c, v in row.items():
      out[int(c)] = v
    return out
  return row


def _run_oracle(case):
  cfg = _cfg()
  table = ygo_build.make_table(GOLD["table"])
  arr, keep = ygo_build.batch_of([case["state"]])
  outs, status = ygo_oracle.write_batch(cfg, table, arr, 1)
  return dict(zip(K, outs)), status


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_known_answers(case):
  o, status = _run_oracle(case)
  assert status[0] == 0
  e = case["expect"]
  for key, name in (("num_options", "info:num_options"), ("to_play", "info:to_play"), ("is_selfplay", "info:is_selfplay"),
                    ("win_reason", "info:win_reason"), ("step_type", "step_type"), ("env_id", "info:env_id"),
                    ("elapsed_step", "elapsed_step"), ("done", "done"), ("trunc", "trunc")):
    if key in e:
      assert int(o[name][0]) == e[key], key
  for key, name in (("reward", "reward"), ("discount", "discount")):
    if key in e:
      assert float(o[name][0]) == e[key], key
  assert o["obs:global_"][0].tolist() == e["global"]
  cards = o["obs:cards_"][0]
  for r, row in e.get("cards_rows", {}).items():
    assert cards[int(r)].tolist() == _full(row, 39), f"cards row {r}"
  assert np.nonzero(cards.any(axis=1))[0].tolist() == e["nonzero_card_rows"]
  for r, row in e.get("actions_rows", {}).items():
    assert o["obs:actions_"][0][int(r)].tolist() == _full(row, 19), f"actions row {r}"
  for r, row in e.get("h_card_ids", {}).items():
    assert o["aux:h_card_ids"][0][int(r)].tolist() == row, f"h_card_ids row {r}"


def test_float_transform_and_lp():
  table = ygo_build.make_table(GOLD["table"])
  for x, want in GOLD["float_transform"].items():
    st = {"to_play": 0, "options": ["y"], "msg": 13, "lp": [int(x), 0]}
    arr, keep = ygo_build.batch_of([st])
    outs, status = ygo_oracle.write_batch(_cfg(), table, arr, 1)
    assert status[0] == 0 and outs[9][0][:2].tolist() == want, x


def test_history_rotation():
  table = ygo_build.make_table(GOLD["table"])
  ring = np.repeat(np.arange(1, 17, dtype=np.uint8)[:, None], 19, axis=1)
  for hp, want in GOLD["history_rotation"]["ha_p"].items():
    st = {"to_play": 1, "options": ["y"], "msg": 13, "history": {"p1": ring}, "ha_p": [3, int(hp)]}
    arr, keep = ygo_build.batch_of([st])
    outs, _ = ygo_oracle.write_batch(_cfg(), table, arr, 1)
    assert outs[11][0][:, 0].tolist() == want and (outs[11][0] == outs[11][0][:, :1]).all()


@pytest.mark.parametrize("st", [
  {"options": ["s h1"], "msg": 11},                                     # spec not on the field
  {"options": ["y"], "msg": 99},                                        # unsupported message
  {"options": ["y"], "msg": 13, "phase": 3},                            # phase2id.at throws
  {"options": ["y"], "msg": 13, "field": {"p0": {"hand": [{"code": 424242, "seq": 0}]}}},  # unknown code
  {"options": ["y"], "msg": 13, "field": {"p0": {"mzone": [{"code": 89631139, "seq": 0, "pos": 6}]}}},  # bad position
])
def test_conditions_where_the_reference_throws(st):
  table = ygo_build.make_table(GOLD["table"])
  arr, keep = ygo_build.batch_of([st])
  _, status = ygo_oracle.write_batch(_cfg(), table, arr, 1)
  assert status[0] == 6


def test_emitter_record_fields():
  """ep_ygo_emit_record parses the same buffers/strings into the POD record."""
  case = GOLD["cases"][2]
  cfg = _cfg()
  arr, keep = ygo_build.batch_of([case["state"]])
  recs, status = capi_ygo.emit_records(cfg, arr, 1)
  assert status[0] == 0
  rec = recs[0]
  assert recs.shape[1] == capi_ygo.lib().ep_ygo_record_stride(C.byref(cfg)) and recs.shape[1] % 256 == 0
  assert rec[0] == 1 and rec[1] == 1 and rec[2] == 15 and rec[3] == 2 and rec[5] == 1  # to_play tp msg n_opt revealed_any
  mask = int(np.frombuffer(rec[40:48].tobytes(), np.uint64)[0])
  assert mask == 1  # only "oh1" matters; "g9" is not an opponent-hand spec
  loc = rec[48:62].reshape(2, 7)
  assert loc.tolist() == [[0, 0, 3, 0, 0, 0, 0], [2, 2, 0, 0, 0, 0, 0]]
  off_cards = 64 + ((16 * 19 + 15) // 16) * 16
  off_opts = off_cards + 5 * 16  # the options follow the record's own (five) cards
  cards = np.frombuffer(rec[off_cards:off_cards + 5 * 16].tobytes(), dtype=np.dtype(
    [("code", "<u4"), ("ctrl", "u1"), ("loc", "u1"), ("seq", "u1"), ("pos", "u1"), ("level", "u1"), ("rank", "u1"),
     ("link", "u1"), ("mat", "u1"), ("atk", "<u2"), ("def", "<u2")]))
  assert cards["code"].tolist() == [46986414, 55144522, 10000, 55144522, 89631139]
  assert cards["loc"].tolist() == [0x84, 0x84, 0x04, 0x02, 0x02] and cards["mat"].tolist() == [1, 1, 0, 0, 0]
  assert cards["pos"].tolist()[:2] == [0, 1] and cards["rank"][2] == 4 and cards["atk"][2] == 0xfffe  # low 16 bits of -2
  opt0 = rec[off_opts:off_opts + 12 + 4 * 5]
  assert opt0[0] == 3 and opt0[9] == 2  # msg id, n_specs
  keys = np.frombuffer(opt0[12:20].tobytes(), "<u4").tolist()
  # spec references: "m1a" -> overlay slot 14*max_cards + (0*8 + 0)*8 + 0; "oh1" -> plain slot (1*7 + 2-1)*max_cards + 0
  assert keys == [14 * 75, 8 * 75]


@pytest.mark.parametrize("msg,opt,feat_col,val", [
  (11, "e", 3, 3), (11, "t x2", 1, 1), (11, "v g3c", 1, 9), (16, "c", 4, 1), (16, "os2b", 1, 8), (26, "f", 4, 2),
  (19, "3", 5, 4), (19, "4", 5, 6), (12, "n oh1", 2, 2), (13, "y", 2, 1), (10, "a m1", 1, 6), (10, "m", 3, 2),
  (14, "7", 6, 7), (18, "os8", 7, 30), (24, "m7", 7, 7), (141, "6", 8, 6),
])
def test_emitter_option_grammar(msg, opt, feat_col, val):
  """The nine spec-independent feature bytes, message by message (:1674-1765 with the id tables
  :660-686): e.g. 'v g3c' -> act v(7)+('c'-'a')=9; POSITION '4' -> 1<<3=8 -> position2id 6."""
  cfg = _cfg()
  arr, keep = ygo_build.batch_of([{"msg": msg, "options": [opt]}])
  recs, status = capi_ygo.emit_records(cfg, arr, 1)
  assert status[0] == 0
  off_opts = 64 + 304  # no cards in this state: the options follow the history ring directly
  feats = recs[0][off_opts:off_opts + 9].tolist()
  want = [0] * 9
  want[0] = [11, 16, 15, 20, 19, 12, 13, 10, 26, 14, 18, 23, 24, 141].index(msg) + 1
  want[feat_col] = val
  assert feats == want


def test_history_push_matches_oracle_and_keeps_stale_bytes():
  cfg = _cfg()
  L = capi_ygo.lib()
  rng = np.random.default_rng(0)
  ring_a = rng.integers(0, 30, size=(16, 19)).astype(np.uint8)
  ring_b = ring_a.copy()
  pa, pb = 2, 2
  script = [(11, "s h1", [300]), (11, "b", []), (16, "m2b", [7]), (15, "m1 oh2 g3", [1, 2, 3]), (13, "n", []),
            (12, "y s3", [9]), (10, "a m1", [1025]), (26, "f", []), (19, "2", []), (14, "3", []), (18, "om4", []),
            (141, "5", []), (11, "v x1c", [65535])]
  for msg, opt, ids in script * 2:  # wraps the 16-row ring: reused rows keep untouched columns (Q10)
    rc, pa = ygo_oracle.history_push(cfg, ring_a, pa, msg, opt, ids)
    assert rc == 0
    p = C.c_int32(pb)
    idarr = np.asarray(ids, np.uint16)
    assert L.ep_ygo_history_push(C.byref(cfg), ring_b.ctypes.data, C.byref(p), msg, opt.encode(), idarr.ctypes.data, len(ids)) == 0
    pb = p.value
    assert pa == pb and np.array_equal(ring_a, ring_b), (msg, opt)


def test_history_push_known_answer():
  """Hand-derived ring contents after three pushes into a 2-row ring (wrap + stale bytes, Q10):
  the oracle's and the product's host ring update both have to produce them."""
  h = GOLD["history_push"]
  cfg = _cfg(n_history_actions=h["n_history_actions"])
  L = capi_ygo.lib()
  ring_o = np.zeros((h["n_history_actions"], 19), np.uint8)
  ring_p = ring_o.copy()
  po = pp = 0
  for (msg, opt, ids), want_p in zip(h["script"], h["ha_p_after"]):
    rc, po = ygo_oracle.history_push(cfg, ring_o, po, msg, opt, ids)
    assert rc == 0 and po == want_p
    p = C.c_int32(pp)
    idarr = np.asarray(ids, np.uint16)
    assert L.ep_ygo_history_push(C.byref(cfg), ring_p.ctypes.data, C.byref(p), msg, opt.encode(), idarr.ctypes.data, len(ids)) == 0
    pp = p.value
    assert pp == want_p
  want = np.array([_full(h["final_ring"][str(r)], 19) for r in range(h["n_history_actions"])], np.uint8)
  assert ring_o.tolist() == want.tolist()
  assert ring_p.tolist() == want.tolist()


@pytest.mark.parametrize("h", GOLD["history_push_cases"], ids=[h["name"] for h in GOLD["history_push_cases"]])
def test_history_push_cases(h):
  """More hand-derived ring contents: the DISFIELD branch (Q3, ygopro.h:1758) and several pushes
  between two WriteState calls followed by the rotated copy-out (Q11, :1959-1962, :1886-1892) -
  oracle and the product's host ring update both."""
  n = h["n_history_actions"]
  cfg = _cfg(n_history_actions=n)
  L = capi_ygo.lib()
  ring_o = np.zeros((n, 19), np.uint8)
  ring_p = ring_o.copy()
  po = pp = 0
  for (msg, opt, ids), want_p in zip(h["script"], h["ha_p_after"]):
    rc, po = ygo_oracle.history_push(cfg, ring_o, po, msg, opt, ids)
    assert rc == 0 and po == want_p
    p = C.c_int32(pp)
    idarr = np.asarray(ids, np.uint16)
    assert L.ep_ygo_history_push(C.byref(cfg), ring_p.ctypes.data, C.byref(p), msg, opt.encode(), idarr.ctypes.data, len(ids)) == 0
    pp = p.value
    assert pp == want_p
  want = np.array([_full(h["final_ring"].get(str(r), {}), 19) for r in range(n)], np.uint8)
  assert ring_o.tolist() == want.tolist() and ring_p.tolist() == want.tolist()
  if "then_write_state" in h:
    w = h["then_write_state"]
    st = dict(w["state"], history={f"p{w['state']['to_play']}": ring_p}, ha_p=[pp, pp])
    arr, keep = ygo_build.batch_of([st], n_hist=n)
    outs, status = ygo_oracle.write_batch(cfg, ygo_build.make_table(GOLD["table"]), arr, 1)
    assert status[0] == 0
    want_h = [_full(w["h_actions_rows"].get(str(r), {}), 19) for r in range(n)]
    assert outs[11][0].tolist() == want_h


def test_check_constants_names_the_first_mismatch():
  """ep_ygo_check_constants: an integrator fills ep_ygo_core_constants from the real ygopro-core
  common.h; the library's own values pass, a differing one is named with both values."""
  L = capi_ygo.lib()
  c = capi_ygo.CoreConstants()
  L.ep_ygo_default_constants(C.byref(c))
  assert list(c.location) == [1, 2, 4, 8, 16, 32, 64, 128] and list(c.msg)[:3] == [11, 16, 15] and c.msg[13] == 141
  assert c.type[24] == 0x4000000 and c.race[25] == 1 << 25 and c.phase[9] == 0x200 and c.query_equip_card == 0x4000
  assert L.ep_ygo_check_constants(C.byref(c)) == 0
  c.location[4] = 0x20            # LOCATION_GRAVE
  assert L.ep_ygo_check_constants(C.byref(c)) == 1
  from envpool_amd import capi
  msg = capi.lib().ep_last_error().decode()
  assert "LOCATION_GRAVE" in msg and "0x10" in msg and "0x20" in msg
  L.ep_ygo_default_constants(C.byref(c))
  c.msg[12] = 25                  # MSG_SELECT_DISFIELD
  assert L.ep_ygo_check_constants(C.byref(c)) == 1 and "MSG_SELECT_DISFIELD" in capi.lib().ep_last_error().decode()


def test_synthetic_states_are_well_formed():
  """The generator feeds both sides of the GPU parity test: most states must encode cleanly, and
  the emitter must accept exactly the states the oracle accepts."""
  from tools import ygo_synth
  cfg = _cfg()
  table = ygo_synth.make_table(1000, seed=0)
  sb = ygo_synth.SynthBatch(cfg, table, 3000, seed=0)
  outs, status = ygo_oracle.write_batch(cfg, table, sb.raw, sb.n)
  recs, est = capi_ygo.emit_records(cfg, sb.raw, sb.n)
  assert (status == 0).mean() > 0.95, np.unique(status, return_counts=True)
  assert (est == 0).all()
  nopt = outs[12]
  assert nopt.min() >= 1 and nopt.max() == 16 and (outs[8].reshape(3000, -1).any(axis=1)).mean() > 0.95


def test_refused_state_leaves_an_empty_record():
  """A state the emitter cannot parse (here: more listed cards than 2 * max_cards, where the
  reference overruns obs:cards_) must not leave counts without their cards behind: the record is
  the state's empty observation, byte for byte what emitting it with n_options = 0 gives."""
  import ctypes as C
  from envpool_amd import capi
  from tools import ygo_synth
  small = capi_ygo.default_config(max_batch=64, max_cards=4)
  table = ygo_synth.make_table(1000, seed=0)
  sb = ygo_synth.SynthBatch(small, table, 64, seed=3)
  recs, est = capi_ygo.emit_records(small, sb.raw, 64)
  bad = np.flatnonzero(est != 0)
  assert bad.size > 32 and (est[bad] == capi.EP_ERR_INVALID).all()
  stride = recs.shape[1]
  for i in bad[:8]:
    empty = capi_ygo.RawState()
    C.memmove(C.byref(empty), C.byref(sb.raw[int(i)]), C.sizeof(capi_ygo.RawState))
    empty.n_options = 0
    want, e2 = capi_ygo.emit_records(small, (capi_ygo.RawState * 1)(empty), 1)
    assert e2[0] == 0 and recs[i].tobytes() == want[0].tobytes()
    assert capi_ygo.lib().ep_ygo_record_used_bytes(C.byref(small), recs[i].ctypes.data) < stride
