/* dummy_step_loop.c — the C ABI from plain C (gcc, no HIP headers, no C++): a sync Dummy-v0 pool of
 * 1024 envs stepped 50 times with host arrays in and host arrays out, checked against the closed
 * form of the env (with L = max(1, seed + env_id) and p = t mod (L + 1),
 * elapsed_step = obs:raw[:,0] = p and done = (p == L)).
 *   gcc -O2 -std=c11 -Iinclude examples/dummy_step_loop.c -Lenvpool_amd -lenvpool_amd \
 *       -Wl,-rpath,'$ORIGIN/../envpool_amd' -o examples/dummy_step_loop
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "envpool_amd.h"

#define CHECK(call)                                                         \
  do {                                                                      \
    int rc_ = (call);                                                       \
    if (rc_ != EP_OK) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ep_last_error());       \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main(void) {
  enum { N = 1024, STEPS = 50, SEED = 7 };
  ep_dummy_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.num_envs = N;
  cfg.batch_size = N;
  cfg.max_num_players = 1;
  cfg.thread_affinity_offset = -1;
  cfg.seed = SEED;
  cfg.max_episode_steps = 2147483647;
  cfg.state_num = 10;
  cfg.action_num = 6;
  ep_pool* pool = NULL;
  CHECK(ep_dummy_pool_create(&cfg, &pool));

  static int32_t ids[N], zeros[N];
  static double list_action[N][6];
  for (int i = 0; i < N; ++i) {
    ids[i] = i;
    for (int k = 0; k < 6; ++k) list_action[i][k] = 1.5;
  }
  const void* arrays[5] = {ids, ids, list_action, zeros, zeros}; /* _action_keys order */
  const int64_t dims[5] = {N, N, N, N, N};

  CHECK(ep_pool_reset(pool, ids, N, EP_MEM_HOST));
  for (int t = 0; t <= STEPS; ++t) {
    ep_batch b;
    CHECK(ep_pool_recv(pool, EP_MEM_HOST, &b));
    if (b.num_rows != N) { fprintf(stderr, "batch of %d rows\n", b.num_rows); return 1; }
    const int32_t* env_id = (const int32_t*)b.arrays[0].data;  /* info:env_id   */
    const int32_t* elapsed = (const int32_t*)b.arrays[2].data; /* elapsed_step  */
    const uint8_t* done = (const uint8_t*)b.arrays[3].data;    /* done          */
    const int32_t* raw = (const int32_t*)b.arrays[8].data;     /* obs:raw [N,10] */
    for (int i = 0; i < N; ++i) {
      const int L = SEED + i < 1 ? 1 : SEED + i, p = t % (L + 1);
      if (env_id[i] != i || elapsed[i] != p || raw[10 * i] != p || done[i] != (p == L)) {
        fprintf(stderr, "t=%d env %d: elapsed %d raw0 %d done %d, expected p=%d L=%d\n", t, i, elapsed[i], raw[10 * i], done[i], p, L);
        return 1;
      }
    }
    CHECK(ep_batch_release(&b));
    if (t < STEPS) CHECK(ep_pool_send(pool, 5, arrays, dims, EP_MEM_HOST));
  }
  CHECK(ep_pool_destroy(pool));
  printf("dummy_step_loop: %d envs x %d steps match the closed form\n", N, STEPS);
  return 0;
}
