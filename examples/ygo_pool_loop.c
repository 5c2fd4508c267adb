/* ygo_pool_loop.c — the YGOPro boundary from plain C: ep_ygo_pool (AsyncEnvPool<YGOProEnv>) over a
 * duel engine that comes in as two callbacks. The "engine" here is a counter per env that offers one
 * option ("e": end the phase, MSG_SELECT_IDLECMD) on an empty field, so every key of the received
 * batches can be predicted by hand: the pool's own bookkeeping (info:env_id in sent order,
 * elapsed_step, auto-reset one step after done - env.h:181-222, async_envpool.h:125) and the encode
 * kernel's obs:actions_ row for that option (msg id 1, phase id 3 - ygopro.h:660-669, 1674-1765).
 *   gcc -O2 -std=c11 -Iinclude examples/ygo_pool_loop.c -Lenvpool_amd -lenvpool_amd \
 *       -Wl,-rpath,'$ORIGIN/../envpool_amd' -o examples/ygo_pool_loop
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "envpool_amd_ygo.h"

#define CHECK(call)                                                   \
  do {                                                                \
    int rc_ = (call);                                                 \
    if (rc_ != EP_OK) {                                               \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ep_last_error()); \
      return 1;                                                       \
    }                                                                 \
  } while (0)

enum { N = 64, STEPS = 40 };
static int32_t g_t[N];                      /* engine state: steps since the duel began */
static const char* const kOptions[1] = {"e"};

static int ep_len(int env) { return 2 + env % 3; }

static void describe(int env, int32_t action, ygo_raw_state* s) { /* what WriteState would read */
  memset(s, 0, sizeof(*s));
  s->to_play = env & 1;
  s->turn_player = env & 1;
  s->msg = 11;            /* MSG_SELECT_IDLECMD */
  s->current_phase = 0x04; /* PHASE_MAIN1 */
  s->lp[0] = s->lp[1] = 8000;
  s->win_reason = action;  /* (echoed so that the loop below can see its actions arrive) */
  s->done = g_t[env] >= ep_len(env);
  s->n_options = 1;
  s->options = kOptions;
}
static int eng_reset(void* ctx, int32_t env, ygo_raw_state* s) { (void)ctx; g_t[env] = 0; describe(env, -1, s); return 0; }
static int eng_step(void* ctx, int32_t env, int32_t action, ygo_raw_state* s) { (void)ctx; g_t[env] += 1; describe(env, action, s); return 0; }

int main(void) {
  ep_ygo_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.max_cards = 75; cfg.max_options = 16; cfg.n_history_actions = 16; cfg.max_multi_select = 5;
  cfg.max_episode_steps = 2147483647;
  ygo_card_data card;
  memset(&card, 0, sizeof(card));
  card.code = 1; card.type = 0x1; card.card_id = 1;
  ep_ygo_engine engine = {NULL, eng_reset, eng_step};
  ep_ygo_pool* pool = NULL;
  CHECK(ep_ygo_pool_create(&cfg, &engine, N, N, 4, -1, &card, 1, &pool));
  const int nf = ep_ygo_action_feats(&cfg), feat0 = 2 * cfg.max_multi_select;

  int32_t ids[N], act[N], elapsed[N], done[N];
  for (int i = 0; i < N; ++i) { ids[i] = N - 1 - i; elapsed[i] = 0; done[i] = 0; }   /* rows come back in THIS order */
  CHECK(ep_ygo_pool_reset(pool, N, ids));
  for (int t = 0; t <= STEPS; ++t) {
    ep_batch b;
    CHECK(ep_ygo_pool_recv(pool, EP_MEM_HOST, &b));
    const int32_t* env_id = (const int32_t*)b.arrays[0].data;
    const int32_t* el = (const int32_t*)b.arrays[2].data;
    const uint8_t* dn = (const uint8_t*)b.arrays[3].data;
    const uint8_t* actions = (const uint8_t*)b.arrays[10].data; /* obs:actions_ [N, 16, nf] */
    const int32_t* n_opt = (const int32_t*)b.arrays[12].data;
    const int32_t* to_play = (const int32_t*)b.arrays[13].data;
    const int32_t* echoed = (const int32_t*)b.arrays[15].data;
    const uint8_t* mask = (const uint8_t*)b.arrays[17].data;    /* aux:action_mask [N, 16] */
    if (b.num_rows != N) { fprintf(stderr, "batch of %d rows\n", b.num_rows); return 1; }
    for (int i = 0; i < N; ++i) {
      const int e = ids[i];
      const uint8_t* row = actions + ((size_t)i * cfg.max_options) * nf;
      const int want_done = elapsed[e] >= ep_len(e);
      if (env_id[i] != e || el[i] != elapsed[e] || dn[i] != want_done || n_opt[i] != 1 || to_play[i] != (e & 1) ||
          echoed[i] != (elapsed[e] == 0 ? -1 : 100 + e) || row[feat0 + 0] != 1 || row[feat0 + 3] != 3 ||
          mask[i * cfg.max_options] != 1 || mask[i * cfg.max_options + 1] != 0) {
        fprintf(stderr, "t=%d row %d env %d: elapsed %d (want %d) done %d (want %d) options %d echoed %d msg %d phase %d\n", t, i, e,
                el[i], elapsed[e], dn[i], want_done, n_opt[i], echoed[i], row[feat0], row[feat0 + 3]);
        return 1;
      }
      done[e] = want_done;
    }
    CHECK(ep_batch_release(&b));
    if (t == STEPS) break;
    for (int i = 0; i < N; ++i) {
      const int e = ids[i];
      act[i] = 100 + e;
      elapsed[e] = done[e] ? 0 : elapsed[e] + 1;   /* an env that reported done is reset by the pool on its next step */
    }
    CHECK(ep_ygo_pool_send(pool, N, ids, act));
  }
  CHECK(ep_ygo_pool_destroy(pool));
  printf("ygo_pool_loop: %d envs x %d steps: pool bookkeeping and obs:actions_ as predicted\n", N, STEPS);
  return 0;
}
