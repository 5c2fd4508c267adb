/*
 * ygo_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the data-parallel tail of the reference's YGOPro env:
 * YGOProEnv::WriteState and everything it calls (envpool2/ygopro/ygopro.h:1506-1893),
 * get_cards_in_location (:2060-2137), the id tables (:531-686), float_transform (:704-710),
 * the spec helpers (:336-420) and update_history_actions (:1439-1450). It consumes the
 * data in the REFERENCE's own forms: raw query_field_card buffers, option strings,
 * revealed_ strings, the per-player history rings. Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED. The reference's YGOPro module cannot be built or imported here
 * (ygopro-core, Lua, SQLiteCpp, unordered_dense and all card data are empty submodules /
 * absent) and the reference holds no test, fixture or golden vector for this path
 * (SURVEY.md §4, §8c). The ygopro-core constants below (LOCATION_*, POS_*, TYPE_*,
 * ATTRIBUTE_*, RACE_*, PHASE_*, MSG_*, QUERY_EQUIP_CARD, LEN_EMPTY) are the widely known
 * upstream values of Fluorohydride/ygopro-core common.h (version unpinned: the reference
 * carries no submodule SHA); the id tables depend only on their ORDER, which is fixed by
 * ygopro.h:531-686. This restatement is pinned only by the hand-derived known-answer
 * vectors in tests/golden/ygo_known_answers.json, each derived from the cited lines.
 */
#include <ctype.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/envpool_amd_ygo.h"

/* ---- ygopro-core constants: the ONE list in include/envpool_amd_ygo_consts.h (written from memory of
 * upstream common.h, see there; the product takes them from the same header, and an integrator
 * checks them against the real common.h with ep_ygo_check_constants) ---- */
#include "../include/envpool_amd_ygo_consts.h"
#define LOCATION_DECK EP_YGO_LOCATION_DECK
#define LOCATION_HAND EP_YGO_LOCATION_HAND
#define LOCATION_MZONE EP_YGO_LOCATION_MZONE
#define LOCATION_SZONE EP_YGO_LOCATION_SZONE
#define LOCATION_GRAVE EP_YGO_LOCATION_GRAVE
#define LOCATION_REMOVED EP_YGO_LOCATION_REMOVED
#define LOCATION_EXTRA EP_YGO_LOCATION_EXTRA
#define LOCATION_OVERLAY EP_YGO_LOCATION_OVERLAY
#define POS_FACEUP EP_YGO_POS_FACEUP
#define POS_FACEDOWN EP_YGO_POS_FACEDOWN
#define QUERY_EQUIP_CARD EP_YGO_QUERY_EQUIP_CARD
#define LEN_EMPTY EP_YGO_LEN_EMPTY
#define MSG_SELECT_BATTLECMD EP_YGO_MSG_SELECT_BATTLECMD
#define MSG_SELECT_IDLECMD EP_YGO_MSG_SELECT_IDLECMD
#define MSG_SELECT_EFFECTYN EP_YGO_MSG_SELECT_EFFECTYN
#define MSG_SELECT_YESNO EP_YGO_MSG_SELECT_YESNO
#define MSG_SELECT_OPTION EP_YGO_MSG_SELECT_OPTION
#define MSG_SELECT_CARD EP_YGO_MSG_SELECT_CARD
#define MSG_SELECT_CHAIN EP_YGO_MSG_SELECT_CHAIN
#define MSG_SELECT_PLACE EP_YGO_MSG_SELECT_PLACE
#define MSG_SELECT_POSITION EP_YGO_MSG_SELECT_POSITION
#define MSG_SELECT_TRIBUTE EP_YGO_MSG_SELECT_TRIBUTE
#define MSG_SELECT_SUM EP_YGO_MSG_SELECT_SUM
#define MSG_SELECT_DISFIELD EP_YGO_MSG_SELECT_DISFIELD
#define MSG_SELECT_UNSELECT_CARD EP_YGO_MSG_SELECT_UNSELECT_CARD
#define MSG_ANNOUNCE_ATTRIB EP_YGO_MSG_ANNOUNCE_ATTRIB

#define YO_OK 0
#define YO_THROW 6 /* the reference throws (unordered_dense::at / runtime_error) */

static const uint8_t kLocations[7] = {LOCATION_DECK, LOCATION_HAND, LOCATION_MZONE, LOCATION_SZONE,
                                      LOCATION_GRAVE, LOCATION_REMOVED, LOCATION_EXTRA};

/* make_ids over std::map => ids in ascending key order (:482-497) */
static int location2id(int loc, uint8_t* id) { /* :531-542, id_offset 1 */
  for (int i = 0; i < 7; ++i)
    if (kLocations[i] == loc) { *id = (uint8_t)(i + 1); return 0; }
  return -1;
}
static int position2id(int pos, uint8_t* id) { /* :546-559 */
  /* std::map iterates in ascending key order: POS_NONE 0x0, then the eight POS_* values sorted */
  static const int keys[9] = {0x0, EP_YGO_POS_FACEUP_ATTACK, EP_YGO_POS_FACEDOWN_ATTACK, EP_YGO_POS_ATTACK, EP_YGO_POS_FACEUP_DEFENSE,
                              EP_YGO_POS_FACEUP, EP_YGO_POS_FACEDOWN_DEFENSE, EP_YGO_POS_FACEDOWN, EP_YGO_POS_DEFENSE};
  for (int i = 0; i < 9; ++i)
    if (keys[i] == pos) { *id = (uint8_t)i; return 0; }
  return -1;
}
static int attribute2id(uint32_t a, uint8_t* id) { /* :563-571 */
  static const uint32_t keys[8] = {0x0, EP_YGO_ATTRIBUTE_EARTH, EP_YGO_ATTRIBUTE_WATER, EP_YGO_ATTRIBUTE_FIRE, EP_YGO_ATTRIBUTE_WIND,
                                   EP_YGO_ATTRIBUTE_LIGHT, EP_YGO_ATTRIBUTE_DARK, EP_YGO_ATTRIBUTE_DEVINE};
  for (int i = 0; i < 8; ++i)
    if (keys[i] == a) { *id = (uint8_t)i; return 0; }
  return -1;
}
static int race2id(uint32_t r, uint8_t* id) { /* :575-605: RACE_NONE, then bit 0 .. bit 25 */
  if (r == 0) { *id = 0; return 0; }
  for (int i = 0; i < EP_YGO_NUM_RACES; ++i)
    if (r == EP_YGO_RACE(i)) { *id = (uint8_t)(i + 1); return 0; }
  return -1;
}
static int phase2id(int ph, uint8_t* id) { /* :644-658: PHASE_DRAW 0x01 .. PHASE_END 0x200 -> 0..9 */
  static const int keys[EP_YGO_NUM_PHASES] = {EP_YGO_PHASE_DRAW, EP_YGO_PHASE_STANDBY, EP_YGO_PHASE_MAIN1, EP_YGO_PHASE_BATTLE_START,
                                              EP_YGO_PHASE_BATTLE_STEP, EP_YGO_PHASE_DAMAGE, EP_YGO_PHASE_DAMAGE_CAL, EP_YGO_PHASE_BATTLE,
                                              EP_YGO_PHASE_MAIN2, EP_YGO_PHASE_END};
  for (int i = 0; i < EP_YGO_NUM_PHASES; ++i)
    if (ph == keys[i]) { *id = (uint8_t)i; return 0; }
  return -1;
}
static int msg2id(int msg, uint8_t* id) { /* :660-669, id_offset 1 */
  static const int msgs[14] = {MSG_SELECT_IDLECMD, MSG_SELECT_CHAIN, MSG_SELECT_CARD, MSG_SELECT_TRIBUTE,
                               MSG_SELECT_POSITION, MSG_SELECT_EFFECTYN, MSG_SELECT_YESNO, MSG_SELECT_BATTLECMD,
                               MSG_SELECT_UNSELECT_CARD, MSG_SELECT_OPTION, MSG_SELECT_PLACE, MSG_SELECT_SUM,
                               MSG_SELECT_DISFIELD, MSG_ANNOUNCE_ATTRIB};
  for (int i = 0; i < 14; ++i)
    if (msgs[i] == msg) { *id = (uint8_t)(i + 1); return 0; }
  return -1;
}
static int list2id(const char* list, char c, uint8_t* id) { /* make_ids(vector<char>, 1) */
  const char* p = strchr(list, c);
  if (!p || c == 0) return -1;
  *id = (uint8_t)(p - list + 1);
  return 0;
}
static int place2id(const char* s, uint8_t* id) { /* :680-686 */
  static const char* names[30] = {"m1", "m2", "m3", "m4", "m5", "m6", "m7", "s1", "s2", "s3",
                                  "s4", "s5", "s6", "s7", "s8", "om1", "om2", "om3", "om4", "om5",
                                  "om6", "om7", "os1", "os2", "os3", "os4", "os5", "os6", "os7", "os8"};
  for (int i = 0; i < 30; ++i)
    if (strcmp(names[i], s) == 0) { *id = (uint8_t)(i + 1); return 0; }
  return -1;
}
/* type2str keys in ascending order (:607-633) */
static const uint32_t kTypeBits[25] = {
    EP_YGO_TYPE_MONSTER, EP_YGO_TYPE_SPELL, EP_YGO_TYPE_TRAP, EP_YGO_TYPE_NORMAL, EP_YGO_TYPE_EFFECT, EP_YGO_TYPE_FUSION, EP_YGO_TYPE_RITUAL,
    EP_YGO_TYPE_TRAPMONSTER, EP_YGO_TYPE_SPIRIT, EP_YGO_TYPE_UNION, EP_YGO_TYPE_DUAL, EP_YGO_TYPE_TUNER, EP_YGO_TYPE_SYNCHRO, EP_YGO_TYPE_TOKEN,
    EP_YGO_TYPE_QUICKPLAY, EP_YGO_TYPE_CONTINUOUS, EP_YGO_TYPE_EQUIP, EP_YGO_TYPE_FIELD, EP_YGO_TYPE_COUNTER, EP_YGO_TYPE_FLIP, EP_YGO_TYPE_TOON,
    EP_YGO_TYPE_XYZ, EP_YGO_TYPE_PENDULUM, EP_YGO_TYPE_SPSUMMON, EP_YGO_TYPE_LINK};

static void float_transform(int x, uint8_t* hi, uint8_t* lo) { /* :704-710 */
  x = x % 65536;
  *hi = (uint8_t)(x >> 8);
  *lo = (uint8_t)(x & 0xff);
}

/* ls_to_spec (:336-365) */
static void ls_to_spec(uint8_t loc, uint8_t seq, uint8_t pos, int opponent, char* out) {
  char* p = out;
  if (opponent) *p++ = 'o';
  if (loc & LOCATION_HAND) *p++ = 'h';
  else if (loc & LOCATION_MZONE) *p++ = 'm';
  else if (loc & LOCATION_SZONE) *p++ = 's';
  else if (loc & LOCATION_GRAVE) *p++ = 'g';
  else if (loc & LOCATION_REMOVED) *p++ = 'r';
  else if (loc & LOCATION_EXTRA) *p++ = 'x';
  int n = seq + 1;
  char digits[8];
  int nd = 0;
  do { digits[nd++] = (char)('0' + n % 10); n /= 10; } while (n);
  while (nd) *p++ = digits[--nd];
  if (loc & LOCATION_OVERLAY) *p++ = (char)('a' + pos);
  *p = 0;
}

typedef struct {
  uint32_t code, type, level, attribute, race;
  int32_t attack, defense;
  uint8_t controler, location, sequence, position;
  uint16_t card_id;
} Card;

static const ygo_card_data* find_card(const ygo_card_data* t, int n, uint32_t code) {
  for (int i = 0; i < n; ++i)
    if (t[i].code == code) return &t[i];
  return NULL;
}
static int c_get_card(const ygo_card_data* t, int n, uint32_t code, Card* c) { /* cards_.at(code) */
  const ygo_card_data* d = find_card(t, n, code);
  if (!d) return -1;
  memset(c, 0, sizeof(*c));
  c->code = code; c->type = d->type; c->level = d->level; c->attribute = d->attribute; c->race = d->race;
  c->attack = d->attack; c->defense = d->defense; c->card_id = d->card_id;
  return 0;
}

static uint32_t rd32(const uint8_t* b, int* p) { uint32_t v; memcpy(&v, b + *p, 4); *p += 4; return v; }

/* get_cards_in_location (:2060-2137). Returns number of cards or -1 (throw). */
static int get_cards_in_location(const ygo_card_data* t, int nt, const uint8_t* buf, int bl, Card* out, int cap) {
  int qdp = 0, n = 0;
  while (1) {
    if (qdp >= bl) break;
    uint32_t f = rd32(buf, &qdp);
    if (f == LEN_EMPTY) continue;
    f = rd32(buf, &qdp);
    uint32_t code = rd32(buf, &qdp);
    Card c;
    if (c_get_card(t, nt, code, &c)) return -1;
    uint8_t controller = buf[qdp++], location = buf[qdp++], sequence = buf[qdp++], position = buf[qdp++];
    c.controler = controller; c.location = location; c.sequence = sequence; c.position = position;
    uint32_t level = rd32(buf, &qdp);
    if ((level & 0xff) > 0) c.level = level & 0xff;
    uint32_t rank = rd32(buf, &qdp);
    if ((rank & 0xff) > 0) c.level = rank & 0xff;
    c.attack = (int32_t)rd32(buf, &qdp);
    c.defense = (int32_t)rd32(buf, &qdp);
    if (f & QUERY_EQUIP_CARD) rd32(buf, &qdp);
    uint32_t n_xyz = rd32(buf, &qdp);
    for (uint32_t i = 0; i < n_xyz; ++i) {
      uint32_t code2 = rd32(buf, &qdp);
      Card c2;
      if (c_get_card(t, nt, code2, &c2)) return -1;
      c2.controler = controller; c2.location = location | LOCATION_OVERLAY; c2.sequence = sequence;
      c2.position = (uint8_t)i;
      if (n >= cap) return -1;
      out[n++] = c2;
    }
    uint32_t n_counters = rd32(buf, &qdp);
    for (uint32_t i = 0; i < n_counters; ++i) rd32(buf, &qdp);
    rd32(buf, &qdp); /* lscale */
    rd32(buf, &qdp); /* rscale */
    uint32_t link = rd32(buf, &qdp);
    uint32_t link_marker = rd32(buf, &qdp);
    if ((link & 0xff) > 0) c.level = link & 0xff;
    if (link_marker > 0) c.defense = (int32_t)link_marker;
    if (n >= cap) return -1;
    out[n++] = c;
  }
  return n;
}

#define MAX_ROWS 512
typedef struct { char spec[16]; uint16_t index; } SpecEntry;
typedef struct { SpecEntry e[MAX_ROWS]; int n; } SpecIndex;

static void spec_set(SpecIndex* m, const char* spec, uint16_t idx) { /* operator[] : last write wins */
  for (int i = 0; i < m->n; ++i)
    if (strcmp(m->e[i].spec, spec) == 0) { m->e[i].index = idx; return; }
  strncpy(m->e[m->n].spec, spec, 15);
  m->e[m->n].spec[15] = 0;
  m->e[m->n++].index = idx;
}
static int spec_at(const SpecIndex* m, const char* spec, uint16_t* idx) {
  for (int i = 0; i < m->n; ++i)
    if (strcmp(m->e[i].spec, spec) == 0) { *idx = m->e[i].index; return 0; }
  return -1;
}

typedef struct {
  const ep_ygo_config* cfg;
  int nf; /* n_action_feats */
  int to_play;
} Ctx;

/* _set_obs_card_ (:1556-1603) */
static int set_obs_card(const Ctx* x, uint8_t* f_cards, int offset, const Card* c, int hide) {
  uint8_t* row = f_cards + (size_t)offset * 39;
  uint8_t location = c->location;
  int overlay = location & LOCATION_OVERLAY;
  if (overlay) location = location & 0x7f;
  if (overlay) hide = 0;
  if (!hide) {
    row[0] = (uint8_t)(c->card_id >> 8);
    row[1] = (uint8_t)(c->card_id & 0xff);
  }
  if (location2id(location, &row[2])) return YO_THROW;
  uint8_t seq = 0;
  if (location == LOCATION_MZONE || location == LOCATION_SZONE || location == LOCATION_GRAVE) seq = (uint8_t)(c->sequence + 1);
  row[3] = seq;
  row[4] = (c->controler != x->to_play) ? 1 : 0;
  if (overlay) {
    position2id(POS_FACEUP, &row[5]);
    row[6] = 1;
  } else {
    if (position2id(c->position, &row[5])) return YO_THROW;
  }
  if (!hide) {
    if (attribute2id(c->attribute, &row[7])) return YO_THROW;
    if (race2id(c->race, &row[8])) return YO_THROW;
    row[9] = (uint8_t)c->level;
    float_transform(c->attack, &row[10], &row[11]);
    float_transform(c->defense, &row[12], &row[13]);
    for (int j = 0; j < 25; ++j) row[14 + j] = (c->type & kTypeBits[j]) ? 1 : 0; /* type_to_ids :635-642 */
  }
  return YO_OK;
}

static int revealed_has(const ygo_raw_state* in, const char* spec) {
  for (int i = 0; i < in->n_revealed; ++i)
    if (strcmp(in->revealed[i], spec) == 0) return 1;
  return 0;
}

/* _set_obs_cards (:1508-1554) */
static int set_obs_cards(const Ctx* x, const ygo_card_data* t, int nt, const ygo_raw_state* in, uint8_t* f_cards,
                         SpecIndex* spec2index) {
  const int max_cards = x->cfg->max_cards;
  static const int hidden_cfg[7] = {1, 1, 0, 0, 0, 0, 1};
  Card* cards = malloc(sizeof(Card) * MAX_ROWS);
  int rc = YO_OK;
  for (int pi = 0; pi < 2 && rc == YO_OK; ++pi) {
    const int player = (x->to_play + pi) % 2;
    const int opponent = pi == 1;
    int offset = opponent ? max_cards : 0;
    const int limit = offset + max_cards; /* the reference has NO bound here; see DESIGN.md */
    for (int li = 0; li < 7 && rc == YO_OK; ++li) {
      const uint8_t location = kLocations[li];
      int hidden_for_opponent = hidden_cfg[li];
      if (opponent && location == LOCATION_HAND && in->n_revealed != 0) hidden_for_opponent = 0;
      if (opponent && hidden_for_opponent) {
        int n_cards = in->field_count[player][li];
        for (int i = 0; i < n_cards; ++i) {
          if (offset >= limit) { rc = YO_THROW; break; }
          location2id(location, &f_cards[(size_t)offset * 39 + 2]);
          f_cards[(size_t)offset * 39 + 4] = 1;
          offset++;
        }
      } else {
        int n = get_cards_in_location(t, nt, in->field_buf[player][li], in->field_len[player][li], cards, MAX_ROWS);
        if (n < 0) { rc = YO_THROW; break; }
        for (int i = 0; i < n; ++i) {
          const Card* c = &cards[i];
          char spec[16];
          ls_to_spec(c->location, c->sequence, c->position, opponent, spec); /* c.get_spec(opponent) */
          int hide = 0;
          if (opponent) {
            hide = (c->position & POS_FACEDOWN) != 0;
            if (location == LOCATION_HAND && revealed_has(in, spec)) hide = 0;
          }
          if (offset >= limit) { rc = YO_THROW; break; }
          rc = set_obs_card(x, f_cards, offset, c, hide);
          if (rc) break;
          offset++;
          spec_set(spec2index, spec, (uint16_t)offset);
        }
      }
    }
  }
  free(cards);
  return rc;
}

/* _set_obs_global (:1605-1620) */
static int set_obs_global(const ygo_raw_state* in, uint8_t* feat, int player) {
  int me = player, op = 1 - player;
  float_transform(in->lp[me], &feat[0], &feat[1]);
  float_transform(in->lp[op], &feat[2], &feat[3]);
  if (phase2id(in->current_phase, &feat[4])) return YO_THROW;
  feat[5] = (me == 0) ? 1 : 0;
  feat[6] = (me == in->turn_player) ? 1 : 0;
  return YO_OK;
}

/* _set_obs_action_spec (:1622-1628) */
static int set_action_spec(const Ctx* x, uint8_t* feat, int i, int j, const char* spec, const SpecIndex* s2i,
                           const uint16_t* card_ids, int n_ids) {
  uint16_t idx;
  if (s2i == NULL || s2i->n == 0) {
    if (j >= n_ids) return YO_THROW; /* the reference reads card_ids[j] out of bounds */
    idx = card_ids[j];
  } else if (spec_at(s2i, spec, &idx)) {
    return YO_THROW;
  }
  feat[(size_t)i * x->nf + 2 * j] = (uint8_t)(idx >> 8);
  feat[(size_t)i * x->nf + 2 * j + 1] = (uint8_t)(idx & 0xff);
  return YO_OK;
}

/* _set_obs_action (:1674-1765). `msg_member` is YGOProEnv::msg_ (quirk Q3, :1758). */
static int set_obs_action(const Ctx* x, uint8_t* feat, int i, int msg, int msg_member, const char* option,
                          const SpecIndex* s2i, const uint16_t* card_ids, int n_ids) {
  const int off = x->cfg->max_multi_select * 2;
  uint8_t* row = feat + (size_t)i * x->nf;
  uint8_t id;
  if (msg2id(msg, &id)) return YO_THROW;
  row[off] = id;
  const size_t n = strlen(option);
  char spec[64];
  if (msg == MSG_SELECT_IDLECMD) {
    if (strcmp(option, "b") == 0 || strcmp(option, "e") == 0) {
      if (list2id("bme", option[0], &id)) return YO_THROW;
      row[off + 3] = id;
    } else {
      char act = option[0];
      strncpy(spec, n >= 2 ? option + 2 : "", sizeof(spec) - 1); spec[sizeof(spec) - 1] = 0;
      uint8_t offset = 0;
      size_t sn = strlen(spec);
      if (act == 'v' && sn > 0 && isalpha((unsigned char)spec[sn - 1])) {
        offset = (uint8_t)(spec[sn - 1] - 'a');
        spec[sn - 1] = 0;
      }
      if (list2id("trcsmav", act, &id)) return YO_THROW;
      row[off + 1] = (uint8_t)(id + offset);
      int rc = set_action_spec(x, feat, i, 0, spec, s2i, card_ids, n_ids);
      if (rc) return rc;
    }
  } else if (msg == MSG_SELECT_CHAIN) {
    if (option[0] == 'c') {
      row[off + 4] = 1;
    } else {
      strncpy(spec, option, sizeof(spec) - 1); spec[sizeof(spec) - 1] = 0;
      uint8_t offset = 0;
      size_t sn = strlen(spec);
      if (sn > 0 && isalpha((unsigned char)spec[sn - 1])) {
        offset = (uint8_t)(spec[sn - 1] - 'a');
        spec[sn - 1] = 0;
      }
      list2id("trcsmav", 'v', &id);
      row[off + 1] = (uint8_t)(id + offset);
      int rc = set_action_spec(x, feat, i, 0, spec, s2i, card_ids, n_ids);
      if (rc) return rc;
    }
  } else if (msg == MSG_SELECT_CARD || msg == MSG_SELECT_TRIBUTE || msg == MSG_SELECT_SUM) {
    if (s2i == NULL || s2i->n == 0) {
      for (int k = 0; k < n_ids; ++k) {
        int rc = set_action_spec(x, feat, i, k, option, s2i, card_ids, n_ids);
        if (rc) return rc;
      }
    } else {
      int k = 0;
      size_t start = 0;
      while (start < n) {
        const char* sp = strchr(option + start, ' ');
        if (!sp) {
          strncpy(spec, option + start, sizeof(spec) - 1); spec[sizeof(spec) - 1] = 0;
          if (k >= x->cfg->max_multi_select) return YO_THROW; /* row overrun in the reference */
          int rc = set_action_spec(x, feat, i, k, spec, s2i, NULL, 0);
          if (rc) return rc;
          break;
        }
        size_t len = (size_t)(sp - (option + start));
        if (len >= sizeof(spec)) return YO_THROW;
        memcpy(spec, option + start, len); spec[len] = 0;
        if (k >= x->cfg->max_multi_select) return YO_THROW;
        int rc = set_action_spec(x, feat, i, k, spec, s2i, NULL, 0);
        if (rc) return rc;
        k++;
        start = start + len + 1;
      }
    }
  } else if (msg == MSG_SELECT_UNSELECT_CARD) {
    if (option[0] == 'f') {
      row[off + 4] = 2;
    } else {
      int rc = set_action_spec(x, feat, i, 0, option, s2i, card_ids, n_ids);
      if (rc) return rc;
    }
  } else if (msg == MSG_SELECT_POSITION) {
    int position = 1 << (option[0] - '1');
    if (position2id(position, &id)) return YO_THROW;
    row[off + 5] = id;
  } else if (msg == MSG_SELECT_EFFECTYN) {
    int rc = set_action_spec(x, feat, i, 0, n >= 2 ? option + 2 : "", s2i, card_ids, n_ids);
    if (rc) return rc;
    if (list2id("yn", option[0], &id)) return YO_THROW;
    row[off + 2] = id;
  } else if (msg == MSG_SELECT_YESNO) {
    if (list2id("yn", option[0], &id)) return YO_THROW;
    row[off + 2] = id;
  } else if (msg == MSG_SELECT_BATTLECMD) {
    if (strcmp(option, "m") == 0 || strcmp(option, "e") == 0) {
      if (list2id("bme", option[0], &id)) return YO_THROW;
      row[off + 3] = id;
    } else {
      if (list2id("trcsmav", option[0], &id)) return YO_THROW;
      row[off + 1] = id;
      int rc = set_action_spec(x, feat, i, 0, n >= 2 ? option + 2 : "", s2i, card_ids, n_ids);
      if (rc) return rc;
    }
  } else if (msg == MSG_SELECT_OPTION) {
    row[off + 6] = (uint8_t)(option[0] - '0');
  } else if (msg == MSG_SELECT_PLACE || msg_member == MSG_SELECT_DISFIELD) {
    if (place2id(option, &id)) return YO_THROW;
    row[off + 7] = id;
  } else if (msg == MSG_ANNOUNCE_ATTRIB) {
    if (attribute2id((uint32_t)(1 << (option[0] - '1')), &id)) return YO_THROW;
    row[off + 8] = id;
  } else {
    return YO_THROW;
  }
  return YO_OK;
}

/* Output of one env. All arrays must be zero on entry (fresh StateBuffer, array.h:77-81). */
typedef struct {
  int32_t env_id, players_env_id, elapsed_step, step_type;
  uint8_t done, trunc;
  float reward, discount;
  uint8_t* cards;     /* [2*max_cards*39] */
  uint8_t* global;    /* [8] */
  uint8_t* actions;   /* [max_options*nf] */
  uint8_t* h_actions; /* [n_history*nf] */
  int32_t num_options, to_play, is_selfplay, win_reason;
  uint16_t* h_card_ids; /* [max_options*max_multi_select], 0-terminated rows */
} yo_out;

/* WriteState (:1828-1893) preceded by Env::Allocate (env.h:196-222). */
int yo_write_state(const ep_ygo_config* cfg, const ygo_card_data* table, int n_table, const ygo_raw_state* in,
                   yo_out* o) {
  Ctx x;
  x.cfg = cfg;
  x.nf = 9 + cfg->max_multi_select * 2;
  x.to_play = in->to_play;
  /* Env::Allocate */
  int done = in->done != 0;
  o->done = (uint8_t)done;
  o->discount = (float)(!done);
  o->step_type = in->elapsed_step == 0 ? 0 : done ? 2 : 1;
  o->trunc = (uint8_t)(done && (in->elapsed_step >= cfg->max_episode_steps));
  o->env_id = in->env_id;
  o->elapsed_step = in->elapsed_step;
  o->players_env_id = in->env_id;
  /* WriteState */
  int n_options = in->n_options;
  o->reward = in->reward;
  o->to_play = in->to_play;
  o->is_selfplay = in->is_selfplay;
  o->win_reason = in->win_reason;
  if (n_options == 0) {
    o->num_options = 1;
    o->global[7] = 1;
    return YO_OK;
  }
  SpecIndex* s2i = calloc(1, sizeof(SpecIndex));
  int rc = set_obs_cards(&x, table, n_table, in, o->cards, s2i);
  if (rc == YO_OK) rc = set_obs_global(in, o->global, in->to_play);
  if (n_options > cfg->max_options) n_options = cfg->max_options; /* options_.resize (:1849-1851) */
  for (int i = 0; i < n_options && rc == YO_OK; ++i) {
    /* spec2index is empty only when no card at all was listed; the reference would then take the
     * card_ids branch with an empty vector — treated as a throw here */
    if (s2i->n == 0) {
      rc = set_obs_action(&x, o->actions, i, in->msg, in->msg, in->options[i], NULL, NULL, 0);
    } else {
      rc = set_obs_action(&x, o->actions, i, in->msg, in->msg, in->options[i], s2i, NULL, 0);
    }
  }
  o->num_options = n_options;
  /* h_card_ids back-derivation, literally (:1863-1879) */
  for (int i = 0; i < n_options; ++i) {
    for (int j = 0; j < cfg->max_multi_select; ++j) {
      uint8_t spec_index = o->actions[(size_t)i * x.nf + 2 * j + 1];
      if (spec_index == 0) break;
      uint16_t c1 = o->cards[(size_t)(spec_index - 1) * 39 + 0];
      uint16_t c2 = o->cards[(size_t)(spec_index - 1) * 39 + 1];
      o->h_card_ids[(size_t)i * cfg->max_multi_select + j] = (uint16_t)((c1 << 8) + c2);
    }
  }
  /* history ring copy-out (:1881-1892) */
  const int ha_p = in->ha_p[in->to_play];
  const uint8_t* hist = in->history[in->to_play];
  const int n1 = cfg->n_history_actions - ha_p;
  memcpy(o->h_actions, hist + (size_t)ha_p * x.nf, (size_t)x.nf * n1);
  memcpy(o->h_actions + (size_t)n1 * x.nf, hist, (size_t)x.nf * ha_p);
  free(s2i);
  return rc;
}

/* update_history_actions (:1439-1450): the ring row is NOT cleared first (quirk Q10). */
int yo_history_push(const ep_ygo_config* cfg, uint8_t* ring, int32_t* ha_p, int msg, const char* option,
                    const uint16_t* card_ids, int n_card_ids) {
  Ctx x;
  x.cfg = cfg;
  x.nf = 9 + cfg->max_multi_select * 2;
  x.to_play = 0;
  int p = *ha_p - 1;
  if (p < 0) p = cfg->n_history_actions - 1;
  *ha_p = p;
  return set_obs_action(&x, ring, p, msg, msg, option, NULL, card_ids, n_card_ids);
}

/* Batch helper for tests / the CPU baseline: n raw states -> SoA outputs laid out like the
 * GPU's batch (arrays zeroed by the caller). */
int yo_write_batch(const ep_ygo_config* cfg, const ygo_card_data* table, int n_table, const ygo_raw_state* in,
                   int n, int32_t* env_id, int32_t* players_env_id, int32_t* elapsed, uint8_t* done, float* reward,
                   float* discount, int32_t* step_type, uint8_t* trunc, uint8_t* cards, uint8_t* global,
                   uint8_t* actions, uint8_t* h_actions, int32_t* num_options, int32_t* to_play,
                   int32_t* is_selfplay, int32_t* win_reason, uint16_t* h_card_ids, int32_t* status) {
  const int nf = 9 + cfg->max_multi_select * 2;
  int worst = YO_OK;
  for (int e = 0; e < n; ++e) {
    yo_out o;
    memset(&o, 0, sizeof(o));
    o.cards = cards + (size_t)e * cfg->max_cards * 2 * 39;
    o.global = global + (size_t)e * 8;
    o.actions = actions + (size_t)e * cfg->max_options * nf;
    o.h_actions = h_actions + (size_t)e * cfg->n_history_actions * nf;
    o.h_card_ids = h_card_ids + (size_t)e * cfg->max_options * cfg->max_multi_select;
    int rc = yo_write_state(cfg, table, n_table, &in[e], &o);
    env_id[e] = o.env_id; players_env_id[e] = o.players_env_id; elapsed[e] = o.elapsed_step; done[e] = o.done;
    reward[e] = o.reward; discount[e] = o.discount; step_type[e] = o.step_type; trunc[e] = o.trunc;
    num_options[e] = o.num_options; to_play[e] = o.to_play; is_selfplay[e] = o.is_selfplay;
    win_reason[e] = o.win_reason;
    if (status) status[e] = rc;
    if (rc) worst = rc;
  }
  return worst;
}
