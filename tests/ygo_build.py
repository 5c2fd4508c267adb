"""Builds reference-form YGOPro states (raw query_field_card buffers, option strings, ...) from
plain Python dicts, for the hand-derived known-answer vectors. Test helper only."""
import ctypes as C
import struct

import numpy as np

from envpool_amd.capi_ygo import CardData, RawState

LOCS = ["deck", "hand", "mzone", "szone", "grave", "removed", "extra"]
LOC_BITS = [0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40]
QUERY_EQUIP_CARD = 0x4000


def card_record(player, loc_bit, c):
  """One record in the layout parsed at ygopro.h:2072-2133."""
  flags = 0x1 | 0x2 | 0x10 | 0x20 | 0x100 | 0x200 | 0x10000 | 0x20000 | 0x200000 | 0x400000 | 0x800000
  equip = c.get("equip") is not None
  if equip:
    flags |= QUERY_EQUIP_CARD
  xyz = c.get("xyz", [])
  counters = c.get("counters", [])
  body = struct.pack("<II", flags, c["code"])
  body += struct.pack("<BBBB", c.get("controller", player), loc_bit, c.get("seq", 0), c.get("pos", 0x1))
  body += struct.pack("<IIii", c.get("level", 0), c.get("rank", 0), c.get("atk", 0), c.get("def", 0))
  if equip:
    body += struct.pack("<I", c["equip"])
  body += struct.pack("<I", len(xyz)) + b"".join(struct.pack("<I", x) for x in xyz)
  body += struct.pack("<I", len(counters)) + b"".join(struct.pack("<I", x) for x in counters)
  body += struct.pack("<IIII", c.get("lscale", 0), c.get("rscale", 0), c.get("link", 0), c.get("link_marker", 0))
  return struct.pack("<I", len(body) + 4) + body


def make_table(cards):
  t = (CardData * len(cards))()
  for i, c in enumerate(cards):
    t[i] = CardData(code=c["code"], type=c["type"], level=c["level"], attribute=c["attribute"], race=c["race"],
                    attack=c.get("atk", 0), defense=c.get("def", 0), card_id=c["card_id"], pad=0)
  return t


class Built:
  """Owns every buffer a RawState points into."""

  def __init__(self, d, n_hist=16, nf=19):
    self.keep = []
    r = RawState()
    r.to_play, r.turn_player, r.msg = d.get("to_play", 0), d.get("tp", 0), d.get("msg", 11)
    r.current_phase, r.is_selfplay, r.win_reason = d.get("phase", 0x04), d.get("is_selfplay", 0), d.get("win_reason", 0)
    r.lp[0], r.lp[1] = d.get("lp", [8000, 8000])
    r.reward = d.get("reward", 0.0)
    r.env_id, r.elapsed_step, r.done = d.get("env_id", 0), d.get("elapsed_step", 1), d.get("done", 0)
    opts = [o.encode() for o in d.get("options", [])]
    r.n_options = len(opts)
    arr = (C.c_char_p * max(1, len(opts)))(*opts)
    self.keep.append(arr)
    r.options = C.cast(arr, C.POINTER(C.c_char_p))
    rev = [o.encode() for o in d.get("revealed", [])]
    r.n_revealed = len(rev)
    rarr = (C.c_char_p * max(1, len(rev)))(*rev)
    self.keep.append(rarr)
    r.revealed = C.cast(rarr, C.POINTER(C.c_char_p))
    field = d.get("field", {})
    for p in (0, 1):
      for li, name in enumerate(LOCS):
        cards = field.get(f"p{p}", {}).get(name, [])
        raw = b""
        n = 0
        for c in cards:
          if c == "empty":
            raw += struct.pack("<I", 4)  # LEN_EMPTY
            continue
          raw += card_record(p, LOC_BITS[li], c)
          n += 1
        r.field_count[p][li] = d.get("counts", {}).get(f"p{p}", {}).get(name, n)
        buf = C.create_string_buffer(raw, max(1, len(raw)))
        self.keep.append(buf)
        r.field_buf[p][li] = C.cast(buf, C.c_void_p).value
        r.field_len[p][li] = len(raw)
    for p in (0, 1):
      h = np.zeros(n_hist * nf, np.uint8)
      hv = d.get("history", {}).get(f"p{p}")
      if hv is not None:
        h[:] = np.asarray(hv, np.uint8).reshape(-1)
      self.keep.append(h)
      r.history[p] = h.ctypes.data
      r.ha_p[p] = d.get("ha_p", [0, 0])[p]
    self.raw = r


def batch_of(dicts, **kw):
  built = [Built(d, **kw) for d in dicts]
  arr = (RawState * len(built))(*[b.raw for b in built])
  return arr, built
