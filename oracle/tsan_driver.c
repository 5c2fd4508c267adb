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
void rp_send_owned(rp_pool*, int32_t, const int32_t*, int32_t, const int32_t*, const double*,
                   void (*)(void*), void*);
void rp_reset(rp_pool*, const int32_t*, int32_t);
void rp_recv(rp_pool*, rp_batch*);
void rp_batch_free(rp_batch*);
uint64_t rp_bench_sync(int32_t, int32_t, int32_t, int32_t);

typedef struct { int32_t ids[64]; double la[6 * 64]; } sent_set;

int main(void) {
  /* sync: 2000 envs, 4 workers, 30 lockstep steps */
  uint64_t sum = rp_bench_sync(2000, 3, 30, 4);
  /* async: 256 envs in batches of 64, 4 workers; every received batch is sent back */
  const int N = 256, B = 64;
  rp_pool* p = rp_create(N, B, 1, 2, 2147483647, 10, 4);
  int32_t* ids = malloc(sizeof(int32_t) * N);
  for (int i = 0; i < N; ++i) ids[i] = i;
  rp_reset(p, ids, N);
  /* Action arrays live as long as the batch they were sent as: a worker that has dequeued an env
   * but not yet read its action row is only protected by the batch's reference count
   * (dummy_pool_mt.c:send_impl), not by anything the receiver can observe — a fixed rotation of
   * array sets ("one per batch in flight") was lapped by the other 255 envs while one worker was
   * descheduled, which ThreadSanitizer reported in round 2. So: one heap block per send, released
   * by the pool from the last unref, as the reference releases its numpy arrays
   * (py_envpool.h:88-100). */
  for (int it = 0; it < 200; ++it) {
    rp_batch b;
    rp_recv(p, &b);
    if (b.num_rows != B) { fprintf(stderr, "batch of %d rows\n", b.num_rows); return 2; }
    sent_set* s = malloc(sizeof(sent_set));
    for (int j = 0; j < B; ++j) {
      s->ids[j] = b.info_env_id[j];
      for (int q = 0; q < 6; ++q) s->la[6 * j + q] = 1.5;
      sum += (uint64_t)b.elapsed_step[j];
    }
    rp_batch_free(&b);
    rp_send_owned(p, B, s->ids, B, s->ids, s->la, free, s);
  }
  for (int it = 0; it < N / B; ++it) { rp_batch b; rp_recv(p, &b); rp_batch_free(&b); }
  rp_destroy(p);
  free(ids);
  printf("tsan_driver: done, checksum %llu; no ThreadSanitizer report\n", (unsigned long long)sum);
  return 0;
}
