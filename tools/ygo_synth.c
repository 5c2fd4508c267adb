/*
 * ygo_synth.c — synthetic YGOPro duel states in the REFERENCE's own data forms.
 *
 * The duel engine (ygopro-core + Lua scripts + cards.cdb) is not available offline, so the
 * observation-encode tail is exercised on synthetic engine output: raw query_field_card
 * buffers with the byte layout that get_cards_in_location parses
 * (envpool2/ygopro/ygopro.h:2060-2137), query_field_count values, option strings that follow
 * the grammar of _set_obs_action (:1674-1765) and reference existing specs, revealed_ strings
 * and per-player history rings. Distribution: SURVEY.md §8d "Cfg 3".
 * Used by tests/ (together with the oracle) and by bench.py (to feed ep_ygo_emit_record).
 * It contains no encode logic of its own.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/envpool_amd_ygo.h"

typedef struct { uint64_t s; } rng_t;
static uint64_t rnd(rng_t* r) {  /* splitmix64 */
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static int rint_(rng_t* r, int lo, int hi) { return lo + (int)(rnd(r) % (uint64_t)(hi - lo + 1)); }

static const uint32_t kTypeBits[25] = {0x1, 0x2, 0x4, 0x10, 0x20, 0x40, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000,
                                       0x2000, 0x4000, 0x10000, 0x20000, 0x40000, 0x80000, 0x100000, 0x200000,
                                       0x400000, 0x800000, 0x1000000, 0x2000000, 0x4000000};
static const uint8_t kLoc[7] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40};

void ys_make_table(int32_t n, uint64_t seed, ygo_card_data* out) {
  rng_t r = {seed * 77 + 1};
  for (int i = 0; i < n; ++i) {
    ygo_card_data* c = &out[i];
    memset(c, 0, sizeof(*c));
    c->code = 10000u + (uint32_t)i * 37u + (uint32_t)(rnd(&r) % 37);
    c->card_id = (uint16_t)(i + 1);
    c->type = kTypeBits[rnd(&r) % 25];
    int extra = rint_(&r, 0, 3);
    for (int k = 0; k < extra; ++k) c->type |= kTypeBits[rnd(&r) % 25];
    int a = rint_(&r, 0, 7);
    c->attribute = a == 0 ? 0 : (1u << (a - 1));
    int rc = rint_(&r, 0, 26);
    c->race = rc == 0 ? 0 : (1u << (rc - 1));
    c->level = (uint32_t)rint_(&r, 1, 12);
    c->attack = rint_(&r, 0, 20) == 0 ? -2 : 50 * rint_(&r, 0, 100);
    c->defense = rint_(&r, 0, 20) == 0 ? -2 : 50 * rint_(&r, 0, 100);
    if (c->type & 0x4000000) c->defense = 0; /* LINK: db_query_card zeroes it (:855-858) */
  }
}

#define YS_MAX_OPT 72
typedef struct {
  ygo_raw_state raw;
  uint8_t* buf[2][7];
  char opt_store[YS_MAX_OPT][64];
  const char* opt_ptr[YS_MAX_OPT];
  char rev_store[4][16];
  const char* rev_ptr[4];
  uint8_t* hist[2];
} ys_state;

typedef struct { uint8_t* p; int len, cap; } wbuf;
static void w32(wbuf* w, uint32_t v) {
  if (w->len + 4 > w->cap) { w->cap = w->cap * 2 + 64; w->p = realloc(w->p, w->cap); }
  memcpy(w->p + w->len, &v, 4);
  w->len += 4;
}
static void w8(wbuf* w, uint8_t v) {
  if (w->len + 1 > w->cap) { w->cap = w->cap * 2 + 64; w->p = realloc(w->p, w->cap); }
  w->p[w->len++] = v;
}

typedef struct { char s[16]; } specstr;

static void mk_spec(uint8_t loc, uint8_t seq, int overlay_idx, int opponent, char* out) {
  char* p = out;
  if (opponent) *p++ = 'o';
  switch (loc) {
    case 0x02: *p++ = 'h'; break;
    case 0x04: *p++ = 'm'; break;
    case 0x08: *p++ = 's'; break;
    case 0x10: *p++ = 'g'; break;
    case 0x20: *p++ = 'r'; break;
    case 0x40: *p++ = 'x'; break;
    default: break;
  }
  p += sprintf(p, "%d", seq + 1);
  if (overlay_idx >= 0) *p++ = (char)('a' + overlay_idx);
  *p = 0;
}

static uint32_t pick_code(rng_t* r, const ygo_card_data* t, int nt) { return t[rnd(r) % (uint64_t)nt].code; }

/* one card record in query_field_card layout */
static void put_card(wbuf* w, rng_t* r, const ygo_card_data* t, int nt, uint8_t ctrl, uint8_t loc, uint8_t seq,
                     uint8_t pos, int n_xyz) {
  const ygo_card_data* c = &t[rnd(r) % (uint64_t)nt];
  const int equip = rint_(r, 0, 7) == 0;
  const int n_counters = rint_(r, 0, 9) == 0 ? rint_(r, 1, 2) : 0;
  uint32_t flags = 0x1 | 0x2 | 0x10 | 0x20 | 0x100 | 0x200 | 0x10000 | 0x20000 | 0x200000 | 0x400000 | 0x800000;
  if (equip) flags |= 0x4000;
  const int words = 3 + 1 + 4 + equip + 1 + n_xyz + 1 + n_counters + 4;
  w32(w, (uint32_t)words * 4u);
  w32(w, flags);
  w32(w, c->code);
  w8(w, ctrl); w8(w, loc); w8(w, seq); w8(w, pos);
  uint32_t level = 0, rank = 0, link = 0, marker = 0;
  const int kind = rint_(r, 0, 9);
  if (kind < 6) level = c->level | (rint_(r, 0, 3) == 0 ? 0x00030000u : 0u);
  else if (kind < 8) rank = (uint32_t)rint_(r, 1, 12);
  else if (kind == 8) { link = (uint32_t)rint_(r, 1, 6); marker = (uint32_t)rint_(r, 1, 0x1ff); }
  w32(w, level);
  w32(w, rank);
  int atk = c->attack, def = c->defense;
  if (rint_(r, 0, 3) == 0) atk += 100 * rint_(r, -5, 30);
  if (rint_(r, 0, 3) == 0) def += 100 * rint_(r, -5, 30);
  if (rint_(r, 0, 40) == 0) atk = 70000 + rint_(r, 0, 999);
  w32(w, (uint32_t)atk);
  w32(w, (uint32_t)def);
  if (equip) w32(w, 0x0004u);
  w32(w, (uint32_t)n_xyz);
  for (int i = 0; i < n_xyz; ++i) w32(w, pick_code(r, t, nt));
  w32(w, (uint32_t)n_counters);
  for (int i = 0; i < n_counters; ++i) w32(w, 0x00011234u);
  w32(w, 0); /* lscale */
  w32(w, 0); /* rscale */
  w32(w, link);
  w32(w, marker);
}

static uint8_t pick_pos(rng_t* r, const uint8_t* set, int n) { return set[rnd(r) % (uint64_t)n]; }

ys_state* ys_gen(uint64_t seed, int64_t index, const ep_ygo_config* cfg, const ygo_card_data* table, int32_t n_table) {
  rng_t r = {seed * 1000003ull + (uint64_t)index * 7919ull + 12345};
  ys_state* s = calloc(1, sizeof(ys_state));
  ygo_raw_state* raw = &s->raw;
  const int nf = 9 + 2 * cfg->max_multi_select;
  static const int msgs[14] = {11, 16, 15, 20, 19, 12, 13, 10, 26, 14, 18, 23, 24, 141};
  raw->to_play = rint_(&r, 0, 1);
  raw->turn_player = rint_(&r, 0, 1);
  raw->msg = msgs[rnd(&r) % 14];
  raw->current_phase = 1 << rint_(&r, 0, 9);
  raw->is_selfplay = rint_(&r, 0, 1);
  { static const int wr[4] = {0, 1, 2, 255}; raw->win_reason = wr[rnd(&r) % 4]; }
  raw->lp[0] = rint_(&r, 0, 40) == 0 ? 66000 + rint_(&r, 0, 5000) : 100 * rint_(&r, 0, 120);
  raw->lp[1] = 100 * rint_(&r, 0, 120);
  { static const float rw[5] = {0.f, 1.f, -1.f, 2.f, -2.f}; raw->reward = rw[rnd(&r) % 5]; }
  raw->env_id = (int32_t)index;
  raw->elapsed_step = rint_(&r, 0, 30) == 0 ? 0 : rint_(&r, 1, 400);
  raw->done = rint_(&r, 0, 19) == 0;

  /* revealed_ is decided first: it changes which opponent specs are visible */
  raw->n_revealed = 0;
  if (rint_(&r, 0, 9) == 0) {
    raw->n_revealed = rint_(&r, 1, 3);
    for (int i = 0; i < raw->n_revealed; ++i) {
      if (rint_(&r, 0, 3) == 0) snprintf(s->rev_store[i], 16, "g%d", rint_(&r, 1, 5));
      else snprintf(s->rev_store[i], 16, "oh%d", rint_(&r, 1, 7));
      s->rev_ptr[i] = s->rev_store[i];
    }
  }
  raw->revealed = s->rev_ptr;

  /* visible specs, collected in the order _set_obs_cards walks the field (:1510-1552) */
  specstr* vis = malloc(sizeof(specstr) * 400);
  int nvis = 0;
  static const uint8_t pos_m[6] = {0x1, 0x2, 0x4, 0x8, 0x1, 0x8}, pos_s[4] = {0x1, 0xa, 0x5, 0x8};
  static const uint8_t pos_h[2] = {0xa, 0x5}, pos_g[2] = {0x5, 0x1}, pos_r[3] = {0x5, 0xa, 0x1};
  wbuf bufs[2][7];
  memset(bufs, 0, sizeof(bufs));
  for (int p = 0; p < 2; ++p) {
    int budget = cfg->max_cards;  /* rows this player may occupy */
    int n_hand = rint_(&r, 0, 7), n_m = rint_(&r, 0, 5), n_s = rint_(&r, 0, 5), n_g = rint_(&r, 0, 20);
    int n_r = rint_(&r, 0, 5), n_x = rint_(&r, 0, 15);
    int xyz[7] = {0};
    int n_mat = 0;
    for (int i = 0; i < n_m; ++i) { xyz[i] = rint_(&r, 0, 3) == 0 ? rint_(&r, 1, 2) : 0; n_mat += xyz[i]; }
    int used = n_hand + n_m + n_mat + n_s + n_g + n_r + n_x;
    int n_deck = 40 - (n_hand + n_m + n_s + n_g + n_r);
    if (n_deck < 0) n_deck = 0;
    if (used + n_deck > budget) n_deck = budget - used > 0 ? budget - used : 0;
    while (used + n_deck > budget && n_g > 0) { n_g--; used--; }
    const int counts[7] = {n_deck, n_hand, n_m, n_s, n_g, n_r, n_x};
    /* distinct zone sequences */
    int mseq[7] = {0, 1, 2, 3, 4, 5, 6}, sseq[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    for (int i = 6; i > 0; --i) { int j = (int)(rnd(&r) % (uint64_t)(i + 1)); int t = mseq[i]; mseq[i] = mseq[j]; mseq[j] = t; }
    for (int i = 7; i > 0; --i) { int j = (int)(rnd(&r) % (uint64_t)(i + 1)); int t = sseq[i]; sseq[i] = sseq[j]; sseq[j] = t; }
    for (int li = 0; li < 7; ++li) {
      wbuf* w = &bufs[p][li];
      raw->field_count[p][li] = counts[li];
      for (int i = 0; i < counts[li]; ++i) {
        if ((li == 2 || li == 3) && rint_(&r, 0, 5) == 0) w32(w, 4); /* an empty zone: LEN_EMPTY */
        uint8_t seq = (uint8_t)i, pos = 0xa;
        int nx = 0;
        switch (li) {
          case 0: pos = 0xa; break;
          case 1: pos = pick_pos(&r, pos_h, 2); break;
          case 2: seq = (uint8_t)mseq[i]; pos = pick_pos(&r, pos_m, 6); nx = xyz[i]; break;
          case 3: seq = (uint8_t)sseq[i]; pos = pick_pos(&r, pos_s, 4); break;
          case 4: pos = pick_pos(&r, pos_g, 2); break;
          case 5: pos = pick_pos(&r, pos_r, 3); break;
          default: pos = rint_(&r, 0, 4) == 0 ? 0x5 : 0xa; break;
        }
        put_card(w, &r, table, n_table, (uint8_t)p, kLoc[li], seq, pos, nx);
      }
    }
  }
  /* walk the field like _set_obs_cards to learn which specs spec2index will hold */
  for (int pi = 0; pi < 2; ++pi) {
    const int player = (raw->to_play + pi) % 2;
    for (int li = 0; li < 7; ++li) {
      int hidden = (li == 0 || li == 1 || li == 6);
      if (li == 1 && raw->n_revealed) hidden = 0;
      if (pi == 1 && hidden) continue;
      /* re-parse our own buffer just enough to list (loc, seq, n_xyz) */
      const uint8_t* b = bufs[player][li].p;
      int len = bufs[player][li].len, q = 0;
      while (q < len) {
        uint32_t f; memcpy(&f, b + q, 4); q += 4;
        if (f == 4) continue;
        uint32_t flags; memcpy(&flags, b + q, 4); q += 8; /* flags, code */
        uint8_t loc = b[q + 1], seq = b[q + 2]; q += 4;
        q += 16; /* level rank atk def */
        if (flags & 0x4000) q += 4;
        uint32_t nx; memcpy(&nx, b + q, 4); q += 4 + 4 * (int)nx;
        uint32_t nc; memcpy(&nc, b + q, 4); q += 4 + 4 * (int)nc;
        q += 16;
        for (uint32_t k = 0; k < nx; ++k) mk_spec(loc, seq, (int)k, pi == 1, vis[nvis++].s);
        mk_spec(loc, seq, -1, pi == 1, vis[nvis++].s);
      }
    }
  }
  for (int p = 0; p < 2; ++p)
    for (int li = 0; li < 7; ++li) {
      s->buf[p][li] = bufs[p][li].p;
      raw->field_buf[p][li] = bufs[p][li].p;
      raw->field_len[p][li] = bufs[p][li].len;
    }

  /* options */
  int n_opt = rint_(&r, 2, 16);
  const int roll = rint_(&r, 0, 99);
  if (roll == 0) n_opt = 0;
  else if (roll < 3 && cfg->max_options + 1 <= YS_MAX_OPT) n_opt = rint_(&r, cfg->max_options + 1, cfg->max_options + 8 < YS_MAX_OPT ? cfg->max_options + 8 : YS_MAX_OPT);
  else if (roll < 6) n_opt = 1;
  if (n_opt > YS_MAX_OPT) n_opt = YS_MAX_OPT;
  raw->n_options = n_opt;
  static const char* places[30] = {"m1", "m2", "m3", "m4", "m5", "m6", "m7", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8",
                                   "om1", "om2", "om3", "om4", "om5", "om6", "om7", "os1", "os2", "os3", "os4", "os5", "os6", "os7", "os8"};
  for (int i = 0; i < n_opt; ++i) {
    char* o = s->opt_store[i];
    const char* sp = nvis ? vis[rnd(&r) % (uint64_t)nvis].s : NULL;
    const int k = rint_(&r, 0, 9);
    switch (raw->msg) {
      case 11: /* IDLECMD */
        if (!sp || k == 0) strcpy(o, "b");
        else if (k == 1) strcpy(o, "e");
        else if (k < 6) sprintf(o, "%c %s", "trcsma"[rnd(&r) % 6], sp);
        else if (k < 8) sprintf(o, "v %s", sp);
        else sprintf(o, "v %s%c", sp, 'a' + rint_(&r, 0, 2));
        break;
      case 16: /* CHAIN */
        if (!sp || k == 0) strcpy(o, "c");
        else if (k < 6) sprintf(o, "%s", sp);
        else sprintf(o, "%s%c", sp, 'a' + rint_(&r, 0, 2));
        break;
      case 15: case 20: case 23: { /* CARD / TRIBUTE / SUM */
        if (!sp) { raw->n_options = i; i = n_opt; break; }
        int ns = rint_(&r, 1, cfg->max_multi_select);
        char* p = o;
        for (int j = 0; j < ns; ++j) p += sprintf(p, "%s%s", j ? " " : "", vis[rnd(&r) % (uint64_t)nvis].s);
        break;
      }
      case 26: /* UNSELECT_CARD */
        if (!sp || k == 0) strcpy(o, "f"); else strcpy(o, sp);
        break;
      case 19: sprintf(o, "%c", '1' + rint_(&r, 0, 3)); break;
      case 12:
        if (!sp) { raw->n_options = i; i = n_opt; break; }
        sprintf(o, "%c %s", k & 1 ? 'y' : 'n', sp);
        break;
      case 13: sprintf(o, "%c", k & 1 ? 'y' : 'n'); break;
      case 10: /* BATTLECMD */
        if (!sp || k == 0) strcpy(o, "m");
        else if (k == 1) strcpy(o, "e");
        else sprintf(o, "%c %s", k & 1 ? 'a' : 'v', sp);
        break;
      case 14: sprintf(o, "%c", '0' + rint_(&r, 0, 9)); break;
      case 18: case 24: strcpy(o, places[rnd(&r) % 30]); break;
      default: sprintf(o, "%c", '1' + rint_(&r, 0, 6)); break; /* ANNOUNCE_ATTRIB */
    }
  }
  for (int i = 0; i < YS_MAX_OPT; ++i) s->opt_ptr[i] = s->opt_store[i];
  raw->options = s->opt_ptr;

  /* history rings: arbitrary (stale) bytes, any pointer */
  for (int p = 0; p < 2; ++p) {
    const int nb = cfg->n_history_actions * nf;
    s->hist[p] = malloc(nb);
    const int filled = rint_(&r, 0, cfg->n_history_actions);
    memset(s->hist[p], 0, nb);
    for (int b = 0; b < filled * nf; ++b) s->hist[p][b] = (uint8_t)(rnd(&r) % 31);
    raw->history[p] = s->hist[p];
    raw->ha_p[p] = rint_(&r, 0, cfg->n_history_actions - 1);
  }
  free(vis);
  return s;
}

void ys_free(ys_state* s) {
  if (!s) return;
  for (int p = 0; p < 2; ++p) {
    for (int li = 0; li < 7; ++li) free(s->buf[p][li]);
    free(s->hist[p]);
  }
  free(s);
}

/* Batch: generates n states and copies their ygo_raw_state views into `out` (pointers stay
 * owned by the returned handle array; free with ys_free_batch). */
ys_state** ys_gen_batch(uint64_t seed, int64_t first, int32_t n, const ep_ygo_config* cfg, const ygo_card_data* table,
                        int32_t n_table, ygo_raw_state* out) {
  ys_state** h = malloc(sizeof(ys_state*) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    h[i] = ys_gen(seed, first + i, cfg, table, n_table);
    out[i] = h[i]->raw;
  }
  return h;
}

void ys_free_batch(ys_state** h, int32_t n) {
  for (int i = 0; i < n; ++i) ys_free(h[i]);
  free(h);
}

/* ---- stand-in duel engine for BASELINE config 5 ------------------------------------------------
 * `n_threads` producer threads; thread t owns the envs t, t + n_threads, ... of `states`. Per
 * env-step a thread spends `engine_us` microseconds "in the engine" (a busy wait: the Lua duel
 * engine is not part of this library), then takes the next row of the encoder's arrival-ordered
 * ring, emits the env's record IN PLACE and reports it done. The record carries the step index in
 * elapsed_step, so a test can key every received row by (env_id, elapsed_step). sink_only != 0:
 * records go to a thread-private scratch record instead of the ring (the cost of emission alone).
 * The ring's entry points come in as function pointers: this file does not link the product. */
#include <pthread.h>
#include <time.h>

typedef struct {
  int (*alloc_record)(void* enc, void** record, uint64_t* ticket);
  int (*emit_record)(const ep_ygo_config* cfg, const ygo_raw_state* in, void* record_out);
  int (*record_done)(void* enc, uint64_t ticket);
  int64_t (*record_stride)(const ep_ygo_config* cfg);
  int (*emit_arrival)(void* enc, const ygo_raw_state* in);  /* packed form: alloc + emit + done in one call */
  int (*put_arrival)(void* enc, const void* record);        /* packed form, record emitted beforehand */
} ys_ring_api;

/* sink_only argument of ys_run_producers: how a finished env-step reaches the encoder */
enum { YS_RING_IN_PLACE = 0, YS_SINK = 1, YS_PACKED_EMIT = 2, YS_PACKED_PUT = 3 };

typedef struct {
  void* enc;
  const ys_ring_api* api;
  const ep_ygo_config* cfg;
  const ygo_raw_state* states;
  int32_t n_envs, n_threads, tid, steps_per_env, sink_only;
  double engine_us;
  int rc;
  double emit_seconds; /* time inside alloc + emit + done */
  const uint8_t* pre_records; /* YS_PACKED_PUT: n_envs records emitted beforehand, at the fixed stride */
} ys_prod_arg;

static double ys_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* ys_producer(void* v) {
  ys_prod_arg* a = (ys_prod_arg*)v;
  void* scratch = NULL;
  const int64_t stride = a->api->record_stride(a->cfg);
  if (a->sink_only == YS_SINK || a->sink_only == YS_PACKED_PUT) scratch = malloc((size_t)stride);
  for (int s = 0; s < a->steps_per_env && a->rc == 0; ++s) {
    for (int e = a->tid; e < a->n_envs && a->rc == 0; e += a->n_threads) {
      if (a->engine_us > 0) {
        const double until = ys_now() + 1e-6 * a->engine_us;
        while (ys_now() < until) {}
      }
      ygo_raw_state st = a->states[e]; /* shallow copy: the buffers it points into are read-only */
      st.elapsed_step = s;
      const double t0 = ys_now();
      if (a->sink_only == YS_PACKED_EMIT) {
        int prc = a->api->emit_arrival(a->enc, &st);
        if (prc != 0) {
          ygo_raw_state empty = st;
          empty.n_options = 0;
          prc = a->api->emit_arrival(a->enc, &empty);
        }
        a->rc = prc;
        a->emit_seconds += ys_now() - t0;
        continue;
      }
      if (a->sink_only == YS_PACKED_PUT) {
        /* the record exists already (the engine + emitter cost is not what is measured): stamp the
         * step into its header copy and hand it over */
        memcpy(scratch, a->pre_records + (size_t)e * (size_t)stride, (size_t)stride);
        ((ygo_header*)scratch)->elapsed_step = s;
        a->rc = a->api->put_arrival(a->enc, scratch);
        a->emit_seconds += ys_now() - t0;
        continue;
      }
      void* rec = scratch;
      uint64_t ticket = 0;
      if (!a->sink_only && (a->rc = a->api->alloc_record(a->enc, &rec, &ticket)) != 0) break;
      int erc = a->api->emit_record(a->cfg, &st, rec);
      if (erc != 0) {
        /* a state the emitter refuses still has to fill its row: an empty record (no options) */
        ygo_raw_state empty = st;
        empty.n_options = 0;
        a->api->emit_record(a->cfg, &empty, rec);
      }
      if (!a->sink_only) a->rc = a->api->record_done(a->enc, ticket);
      a->emit_seconds += ys_now() - t0;
    }
  }
  free(scratch);
  return NULL;
}

int ys_run_producers(void* enc, const ys_ring_api* api, const ep_ygo_config* cfg, const ygo_raw_state* states,
                     int32_t n_envs, int32_t n_threads, int32_t steps_per_env, double engine_us, int32_t sink_only,
                     double* emit_seconds_out, const uint8_t* pre_records) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  pthread_t th[64];
  ys_prod_arg arg[64];
  int started = 0, rc = 0;
  for (int t = 0; t < n_threads; ++t) {
    arg[t] = (ys_prod_arg){enc, api, cfg, states, n_envs, n_threads, t, steps_per_env, sink_only, engine_us, 0, 0.0, pre_records};
    if (pthread_create(&th[t], NULL, ys_producer, &arg[t]) != 0) { rc = -1; break; }
    ++started;
  }
  double emit = 0;
  for (int t = 0; t < started; ++t) {
    pthread_join(th[t], NULL);
    if (arg[t].rc && !rc) rc = arg[t].rc;
    emit += arg[t].emit_seconds;
  }
  if (emit_seconds_out) *emit_seconds_out = emit;
  return rc;
}

/* ---- stand-in duel engine behind the ep_ygo_engine callbacks (ep_ygo_pool_create) ---------------
 * Deterministic per env: the k-th callback for env e produces the synthetic state number
 * e * 1000003 + k of the generator above; an episode of env e lasts 3 + e % 5 steps; the action of
 * a step is echoed in win_reason and reward so that a test can see which action reached which env.
 * A callback may spend `engine_us` microseconds "in the engine" first. States stay valid until the
 * env's next callback, as the callback contract requires. */
typedef struct {
  ep_ygo_config cfg;
  const ygo_card_data* table;
  int32_t n_table, num_envs;
  uint64_t seed;
  double engine_us;
  int64_t* calls;
  int32_t* t_ep;
  ys_state** cur;
  int32_t fail_env; /* >= 0: callbacks of this env report failure (tests) */
  int32_t env_off;  /* global id of env 0 (ys_engine_set_env_id_offset) */
  int32_t replay;   /* != 0: every callback of env e hands out the SAME pre-generated state (bench: the
                       cost of generating a synthetic state is not the engine's) */
} ys_engine;

void* ys_engine_create(const ep_ygo_config* cfg, const ygo_card_data* table, int32_t n_table, int32_t num_envs,
                       uint64_t seed, double engine_us) {
  ys_engine* g = calloc(1, sizeof(*g));
  g->cfg = *cfg;
  g->table = table;
  g->n_table = n_table;
  g->num_envs = num_envs;
  g->seed = seed;
  g->engine_us = engine_us;
  g->calls = calloc((size_t)num_envs, sizeof(int64_t));
  g->t_ep = calloc((size_t)num_envs, sizeof(int32_t));
  g->cur = calloc((size_t)num_envs, sizeof(ys_state*));
  g->fail_env = -1;
  return g;
}

void ys_engine_set_fail_env(void* ctx, int32_t env) { ((ys_engine*)ctx)->fail_env = env; }
void ys_engine_set_env_id_offset(void* ctx, int32_t off) { ((ys_engine*)ctx)->env_off = off; }

void ys_engine_set_replay(void* ctx, int32_t on) {
  ys_engine* g = (ys_engine*)ctx;
  g->replay = on;
  if (on)
    for (int i = 0; i < g->num_envs; ++i)
      if (!g->cur[i]) g->cur[i] = ys_gen(g->seed, (int64_t)i * 1000003, &g->cfg, g->table, g->n_table);
}

void ys_engine_destroy(void* ctx) {
  ys_engine* g = (ys_engine*)ctx;
  if (!g) return;
  for (int i = 0; i < g->num_envs; ++i) ys_free(g->cur[i]);
  free(g->calls); free(g->t_ep); free(g->cur); free(g);
}

static int ys_engine_advance(ys_engine* g, int32_t env_id, int reset, int32_t action, ygo_raw_state* out) {
  env_id -= g->env_off; /* the pool hands out global ids */
  if (env_id < 0 || env_id >= g->num_envs) return 1;
  if (g->engine_us > 0) {
    const double until = ys_now() + 1e-6 * g->engine_us;
    while (ys_now() < until) {}
  }
  if (env_id == g->fail_env) return 2;
  ys_state* s = g->cur[env_id];
  if (!g->replay) {
    ys_free(s);
    s = ys_gen(g->seed, (int64_t)env_id * 1000003 + g->calls[env_id], &g->cfg, g->table, g->n_table);
    g->cur[env_id] = s;
  }
  g->calls[env_id] += 1;
  g->t_ep[env_id] = reset ? 0 : g->t_ep[env_id] + 1;
  s->raw.done = g->t_ep[env_id] >= 3 + env_id % 5;
  s->raw.win_reason = reset ? 0 : action;
  s->raw.reward = reset ? 0.f : (float)(action % 3);
  *out = s->raw;
  return 0;
}

int ys_engine_reset(void* ctx, int32_t env_id, ygo_raw_state* out) { return ys_engine_advance((ys_engine*)ctx, env_id, 1, 0, out); }
int ys_engine_step(void* ctx, int32_t env_id, int32_t action, ygo_raw_state* out) {
  return ys_engine_advance((ys_engine*)ctx, env_id, 0, action, out);
}
