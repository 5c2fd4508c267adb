/* tsan_driver.c — TEST INFRASTRUCTURE: runs the threaded restatement of the reference's
 * architecture (dummy_pool_mt.c: action queue -> worker threads -> state-buffer queue) under
 * ThreadSanitizer (`make -C oracle tsan`), sync and async, several workers. SURVEY.md §5 asks for
 * a -fsanitize=thread target on the CPU restatement; the reference itself has none. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct rp_pool rp_pool;
typedef struct {
  int32_t num_rows, num_player_rows, state_num, pad;
  int32_t* info_env_id; int32_t* info_players_env_id; int32_t* elapsed_step; uint8_t* done; float* reward; float* discount;
  int32_t* step_type; uint8_t* trunc; int32_t* obs_raw; int32_t* dyn_rows; int32_t* dyn_fill; uint8_t* players_done;
  int32_t* players_id;
} rp_batch;
rp_pool* rp_create(int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t);
void rp_destroy(rp_pool*);
void rp_send(rp_pool*, int32_t, const int32_t*, int32_t, const int32_t*, const double*);
void rp_reset(rp_pool*, const int32_t*, int32_t);
void rp_recv(rp_pool*, rp_batch*);
void rp_batch_free(rp_batch*);
uint64_t rp_bench_sync(int32_t, int32_t, int32_t, int32_t);

int main(void) {
  /* sync: 2000 envs, 4 workers, 30 lockstep steps */
  uint64_t sum = rp_bench_sync(2000, 3, 30, 4);
  /* async: 256 envs in batches of 64, 4 workers; every received batch is sent back */
  const int N = 256, B = 64;
  rp_pool* p = rp_create(N, B, 1, 2, 2147483647, 10, 4);
  int32_t* ids = malloc(sizeof(int32_t) * N);
  for (int i = 0; i < N; ++i) ids[i] = i;
  rp_reset(p, ids, N);
  /* action arrays must outlive the step they feed: one set per batch in flight */
  enum { SETS = 16 };
  int32_t* sent_ids[SETS];
  double* sent_la[SETS];
  for (int k = 0; k < SETS; ++k) { sent_ids[k] = malloc(sizeof(int32_t) * B); sent_la[k] = malloc(sizeof(double) * 6 * B); }
  for (int it = 0; it < 200; ++it) {
    rp_batch b;
    rp_recv(p, &b);
    if (b.num_rows != B) { fprintf(stderr, "batch of %d rows\n", b.num_rows); return 2; }
    const int k = it % SETS;
    for (int j = 0; j < B; ++j) {
      sent_ids[k][j] = b.info_env_id[j];
      for (int q = 0; q < 6; ++q) sent_la[k][6 * j + q] = 1.5;
      sum += (uint64_t)b.elapsed_step[j];
    }
    rp_batch_free(&b);
    rp_send(p, B, sent_ids[k], B, sent_ids[k], sent_la[k]);
  }
  for (int it = 0; it < N / B; ++it) { rp_batch b; rp_recv(p, &b); rp_batch_free(&b); }
  rp_destroy(p);
  for (int k = 0; k < SETS; ++k) { free(sent_ids[k]); free(sent_la[k]); }
  free(ids);
  printf("tsan_driver: done, checksum %llu; no ThreadSanitizer report\n", (unsigned long long)sum);
  return 0;
}
