/*
 * dummy_pool_mt.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The reference's ARCHITECTURE for the Dummy path restated in C with pthreads, for
 * bench.py's cpu_baseline leg (a second CPU figure beside dummy_oracle.c's sequential
 * shards) and for a thread-level cross-check of dummy_oracle.c in tests/. Only tests/ and
 * bench.py's cpu_baseline leg load it.
 *
 *   batch of actions -> ActionBufferQueue -> worker threads -> StateBufferQueue
 *   (async_envpool.h:35)
 *
 * What is restated, piece by piece:
 *   - LightweightSemaphore (third_party moodycamel, absent from the image): its published
 *     design, an atomic count that is spun on briefly before the thread sleeps in the
 *     kernel (here a futex). SURVEY.md §8(d) asks for exactly this so that the CPU side is
 *     not handicapped by a mutex/condvar stand-in.
 *   - ActionBufferQueue (action_buffer_queue.h:35-85): ring of 2N slices, one enqueuer at a
 *     time, `sem_` counts queued slices, `sem_dequeue_` serialises the pop.
 *   - StateBuffer (state_buffer.h:40-158): packed (player, shared) offset fetch_add,
 *     alloc / done counters, completion semaphore, sync `order`.
 *   - StateBufferQueue (state_buffer_queue.h:32-151): (N/B+2)*2 buffers, fresh ZEROED
 *     buffers made by max(1, cores/64) background threads into a CircularBuffer
 *     (circular_buffer.h:34-67) and swapped in at every Wait.
 *   - AsyncEnvPool (async_envpool.h:57-189): SendImpl / Reset / Recv / the worker loop.
 *   - Env::EnvStep + DummyEnv (env.h:112-222, dummy_envpool.h:125-215): as dummy_oracle.c.
 *
 * What is NOT restated: the per-step heap traffic of the reference's Array views
 * (shape vectors, shared_ptr control blocks, std::function; SURVEY.md §8a D3/D4 "where
 * time goes") and the obs:dyn payload ((env_id+1)*state_num ints per env-step, which
 * cannot run at N = 65536 at all): both omissions favour the CPU figure.
 */
#define _GNU_SOURCE
#include <linux/futex.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

/* ---- LightweightSemaphore: spin, then futex ------------------------------------------- */
typedef struct {
  _Atomic int32_t count; /* > 0: signals available; < 0: that many threads are (about to be) asleep */
  _Atomic int32_t wake;  /* wake-ups handed to sleepers, the futex word */
} lsem;

static void lsem_init(lsem* s, int initial) {
  atomic_store(&s->count, initial);
  atomic_store(&s->wake, 0);
}

static inline int lsem_try(lsem* s) {
  int32_t c = atomic_load_explicit(&s->count, memory_order_relaxed);
  while (c > 0)
    if (atomic_compare_exchange_weak_explicit(&s->count, &c, c - 1, memory_order_acquire,
                                              memory_order_relaxed))
      return 1;
  return 0;
}

static void lsem_wait(lsem* s) {
  for (int spin = 0; spin < 10000; ++spin) { /* moodycamel spins ~10000 times as well */
    if (lsem_try(s)) return;
    __builtin_ia32_pause();
  }
  if (atomic_fetch_sub_explicit(&s->count, 1, memory_order_acquire) > 0) return;
  for (;;) { /* registered as a sleeper: wait for a wake-up addressed to sleepers */
    int32_t w = atomic_load_explicit(&s->wake, memory_order_acquire);
    while (w > 0)
      if (atomic_compare_exchange_weak_explicit(&s->wake, &w, w - 1, memory_order_acquire, memory_order_relaxed)) return;
    syscall(SYS_futex, &s->wake, FUTEX_WAIT_PRIVATE, 0, NULL, NULL, 0);
  }
}

/* The signaller touches the semaphore's memory for the last time in the fetch_add unless a thread
 * is registered as asleep (count was negative): the consumer frees a StateBuffer, semaphore
 * included, as soon as its wait returns (state_buffer_queue.h:148), so a signaller that looked at
 * the object again after making the count positive would read freed memory (ThreadSanitizer found
 * exactly that in this file's first version). */
static void lsem_signal(lsem* s, int n) {
  const int32_t old = atomic_fetch_add_explicit(&s->count, n, memory_order_release);
  if (old >= 0) return;
  const int32_t to_release = -old < n ? -old : n;
  atomic_fetch_add_explicit(&s->wake, to_release, memory_order_release);
  syscall(SYS_futex, &s->wake, FUTEX_WAKE_PRIVATE, to_release, NULL, NULL, 0);
}

/* ---- ActionBufferQueue (action_buffer_queue.h:35-85) ------------------------------------ */
typedef struct {
  int32_t env_id, order;
  uint8_t force_reset;
} rp_slice;

typedef struct {
  _Atomic uint64_t alloc_ptr, done_ptr;
  size_t queue_size;
  rp_slice* queue;
  lsem sem, sem_enqueue, sem_dequeue;
} rp_abq;

static void abq_init(rp_abq* q, size_t num_envs) {
  atomic_store(&q->alloc_ptr, 0);
  atomic_store(&q->done_ptr, 0);
  q->queue_size = num_envs * 2; /* :53 */
  q->queue = calloc(q->queue_size, sizeof(rp_slice));
  lsem_init(&q->sem, 0);
  lsem_init(&q->sem_enqueue, 1);
  lsem_init(&q->sem_dequeue, 1);
}

static void abq_enqueue_bulk(rp_abq* q, const rp_slice* a, size_t n) { /* :59-69 */
  lsem_wait(&q->sem_enqueue);
  uint64_t pos = atomic_fetch_add(&q->alloc_ptr, n);
  for (size_t i = 0; i < n; ++i) q->queue[(pos + i) % q->queue_size] = a[i];
  lsem_signal(&q->sem, (int)n);
  lsem_signal(&q->sem_enqueue, 1);
}

static rp_slice abq_dequeue(rp_abq* q) { /* :71-80 */
  lsem_wait(&q->sem);
  lsem_wait(&q->sem_dequeue);
  uint64_t ptr = atomic_fetch_add(&q->done_ptr, 1);
  rp_slice r = q->queue[ptr % q->queue_size];
  lsem_signal(&q->sem_dequeue, 1);
  return r;
}

/* ---- StateBuffer (state_buffer.h:40-158) ------------------------------------------------- */
typedef struct {
  int32_t* info_env_id;
  int32_t* info_players_env_id;
  int32_t* elapsed_step;
  uint8_t* done;
  float* reward;
  float* discount;
  int32_t* step_type;
  uint8_t* trunc;
  int32_t* obs_raw;
  int32_t* dyn_rows; /* obs:dyn kept lazy: leading dim and fill value only */
  int32_t* dyn_fill;
  uint8_t* players_done;
  int32_t* players_id;
  _Atomic uint64_t offsets, alloc_count, done_count;
  lsem sem;
} rp_buffer;

/* what rp_recv hands over: the buffer's own arrays, truncated by the two counts (no copy,
 * state_buffer.h:146-156); rp_batch_free releases them as numpy's capsule would. */
typedef struct {
  int32_t num_rows, num_player_rows, state_num, pad;
  int32_t* info_env_id;
  int32_t* info_players_env_id;
  int32_t* elapsed_step;
  uint8_t* done;
  float* reward;
  float* discount;
  int32_t* step_type;
  uint8_t* trunc;
  int32_t* obs_raw;
  int32_t* dyn_rows;
  int32_t* dyn_fill;
  uint8_t* players_done;
  int32_t* players_id;
} rp_batch;

static rp_buffer* buffer_new(size_t B, size_t P, size_t S) {
  /* MakeArray(specs): one zero-initialised allocation per key (array.h:77-81) */
  size_t BP = B * P;
  rp_buffer* b = calloc(1, sizeof(rp_buffer));
  b->info_env_id = calloc(B, 4);
  b->info_players_env_id = calloc(BP, 4);
  b->elapsed_step = calloc(B, 4);
  b->done = calloc(B, 1);
  b->reward = calloc(BP, 4);
  b->discount = calloc(BP, 4);
  b->step_type = calloc(B, 4);
  b->trunc = calloc(B, 1);
  b->obs_raw = calloc(BP * S, 4);
  b->dyn_rows = calloc(BP, 4);
  b->dyn_fill = calloc(BP, 4);
  b->players_done = calloc(BP, 1);
  b->players_id = calloc(BP, 4);
  /* calloc of a large block hands back untouched zero pages; the reference's value-initialising
   * new[] writes them. Touch them the same way, on this (the background) thread. */
  memset(b->obs_raw, 0, BP * S * 4);
  memset(b->info_env_id, 0, B * 4);
  memset(b->info_players_env_id, 0, BP * 4);
  memset(b->elapsed_step, 0, B * 4);
  memset(b->done, 0, B);
  memset(b->reward, 0, BP * 4);
  memset(b->discount, 0, BP * 4);
  memset(b->step_type, 0, B * 4);
  memset(b->trunc, 0, B);
  memset(b->dyn_rows, 0, BP * 4);
  memset(b->dyn_fill, 0, BP * 4);
  memset(b->players_done, 0, BP);
  memset(b->players_id, 0, BP * 4);
  lsem_init(&b->sem, 0);
  return b;
}

static void buffer_free_arrays(rp_buffer* b) {
  free(b->info_env_id); free(b->info_players_env_id); free(b->elapsed_step); free(b->done);
  free(b->reward); free(b->discount); free(b->step_type); free(b->trunc); free(b->obs_raw);
  free(b->dyn_rows); free(b->dyn_fill); free(b->players_done); free(b->players_id);
}

/* ---- CircularBuffer (circular_buffer.h:34-67) -------------------------------------------- */
typedef struct {
  size_t size;
  lsem sem_get, sem_put;
  rp_buffer** buf;
  _Atomic uint64_t head, tail;
} rp_circ;

static void circ_init(rp_circ* c, size_t size) {
  c->size = size;
  lsem_init(&c->sem_get, 0);
  lsem_init(&c->sem_put, (int)size);
  c->buf = calloc(size, sizeof(rp_buffer*));
  atomic_store(&c->head, 0);
  atomic_store(&c->tail, 0);
}
static void circ_put(rp_circ* c, rp_buffer* v) {
  lsem_wait(&c->sem_put);
  uint64_t t = atomic_fetch_add(&c->tail, 1);
  c->buf[t % c->size] = v;
  lsem_signal(&c->sem_get, 1);
}
static rp_buffer* circ_get(rp_circ* c) {
  lsem_wait(&c->sem_get);
  uint64_t h = atomic_fetch_add(&c->head, 1);
  rp_buffer* v = c->buf[h % c->size];
  c->buf[h % c->size] = NULL;
  lsem_signal(&c->sem_put, 1);
  return v;
}

/* ---- the pool ------------------------------------------------------------------------------ */
typedef struct {
  const int32_t* players_env_id;
  const double* list_action;
  int32_t np_rows;
  _Atomic int32_t refs; /* the shared_ptr<vector<Array>> of async_envpool.h:62-63 */
  /* py_envpool.h:88-100: the numpy arrays behind the batch are released when the last env drops
   * it. NULL = the caller keeps the arrays alive itself (rp_send). */
  void (*release)(void*);
  void* release_ctx;
} rp_action_batch;

typedef struct {
  int32_t state, current_step, seed; /* dummy_envpool.h:118, env.h:71,94 */
  int32_t env_index;                  /* env.h:104-110 SetAction */
  rp_action_batch* batch;
} rp_env;

typedef struct rp_pool {
  int32_t N, B, P, S, seed, max_episode_steps, num_threads, is_sync;
  _Atomic int stop;
  _Atomic int64_t stepping_env_num;
  _Atomic int action_check_failed; /* the reference's CHECK_EQ would have aborted */
  rp_env* envs;
  rp_abq abq;
  /* StateBufferQueue */
  size_t Q;
  rp_buffer** queue;
  _Atomic uint64_t alloc_count, done_ptr;
  rp_circ stock;
  _Atomic int quit;
  int n_create;
  pthread_t create_threads[16];
  pthread_t* workers;
} rp_pool;

static void* create_buffer_thread(void* v) { /* state_buffer_queue.h:86-94 */
  rp_pool* p = (rp_pool*)v;
  for (;;) {
    circ_put(&p->stock, buffer_new(p->B, p->P, p->S));
    if (atomic_load(&p->quit)) break;
  }
  return NULL;
}

static void buffer_done(rp_pool* p, rp_buffer* b, uint64_t num) { /* state_buffer.h:124-129 */
  uint64_t dc = atomic_fetch_add(&b->done_count, num);
  if (dc + num == (uint64_t)p->B) lsem_signal(&b->sem, 1);
}

static inline int is_done(const rp_env* e) { return e->state >= e->seed; }

static void batch_unref(rp_action_batch* a) {
  if (a && atomic_fetch_sub(&a->refs, 1) == 1) {
    if (a->release) a->release(a->release_ctx);
    free(a);
  }
}

/* Env::EnvStep (env.h:162-172) for one slice, run by a worker */
static void env_step(rp_pool* p, int32_t env_id, int order, int reset) {
  rp_env* e = &p->envs[env_id];
  if (reset) e->current_step = 0; else e->current_step += 1; /* env.h:181-189 */
  int action_num = 0;
  if (reset) {
    e->state = 0;
  } else {
    const rp_action_batch* a = e->batch;
    if (p->P == 1) { /* env.h:150-158 */
      action_num = 1;
      if (a->players_env_id[e->env_index] != env_id) action_num = 0;
    } else { /* env.h:124-149 */
      int cnt = 0;
      for (int j = 0; j < a->np_rows; ++j)
        if (a->players_env_id[j] == env_id) ++cnt;
      action_num = cnt;
    }
    ++e->state;
    const double* la = a->list_action + (size_t)e->env_index * 6;
    for (int k = 0; k < 6; ++k)
      if (!(la[0] == la[k])) atomic_store(&p->action_check_failed, 1);
  }
  int num_players = p->P <= 1 ? 1 : e->state % (p->P - 1) + 1;

  uint64_t pos = atomic_fetch_add(&p->alloc_count, 1); /* state_buffer_queue.h:114-128 */
  rp_buffer* b = p->queue[(pos / (uint64_t)p->B) % p->Q];
  uint64_t ac = atomic_fetch_add(&b->alloc_count, 1); /* state_buffer.h:79-112 */
  if (ac >= (uint64_t)p->B) return;                    /* :111 out_of_range */
  uint64_t off = atomic_fetch_add(&b->offsets, ((uint64_t)num_players << 32) | 1);
  uint32_t player_off = (uint32_t)(off >> 32), shared_off = (uint32_t)off;
  if (order != -1 && p->P == 1) player_off = shared_off = (uint32_t)order;

  int done = is_done(e); /* env.h:196-222 */
  b->done[shared_off] = (uint8_t)done;
  b->discount[player_off] = (float)(!done);
  b->step_type[shared_off] = e->current_step == 0 ? 0 : done ? 2 : 1;
  b->trunc[shared_off] = (uint8_t)(done && (e->current_step >= p->max_episode_steps));
  b->info_env_id[shared_off] = env_id;
  b->elapsed_step[shared_off] = e->current_step;
  for (int i = 0; i < num_players; ++i) b->info_players_env_id[player_off + i] = env_id;
  for (int i = 0; i < num_players; ++i) { /* dummy_envpool.h:147-163, 199-209 */
    size_t r = (size_t)player_off + i;
    b->players_id[r] = i;
    b->players_done[r] = (uint8_t)done;
    b->obs_raw[r * p->S + 0] = e->state;
    b->obs_raw[r * p->S + 1] = reset ? 0 : action_num;
    b->reward[r] = (float)(-i);
    b->dyn_rows[r] = env_id + 1;
    b->dyn_fill[r] = env_id;
  }
  buffer_done(p, b, 1); /* env.h:191-194 */
}

static void* worker(void* v) { /* async_envpool.h:117-128 */
  rp_pool* p = (rp_pool*)v;
  for (;;) {
    rp_slice s = abq_dequeue(&p->abq);
    if (atomic_load(&p->stop) == 1) break;
    rp_env* e = &p->envs[s.env_id];
    int reset = s.force_reset || is_done(e);
    env_step(p, s.env_id, s.order, reset);
  }
  return NULL;
}

rp_pool* rp_create(int32_t num_envs, int32_t batch_size, int32_t max_num_players, int32_t seed,
                   int32_t max_episode_steps, int32_t state_num, int32_t num_threads) {
  if (num_envs < 1 || batch_size > num_envs || max_num_players < 1 || state_num < 2) return NULL;
  rp_pool* p = calloc(1, sizeof(rp_pool));
  p->N = num_envs;
  p->B = batch_size <= 0 ? num_envs : batch_size;
  p->P = max_num_players;
  p->S = state_num;
  p->seed = seed;
  p->max_episode_steps = max_episode_steps;
  p->is_sync = (p->B == p->N && p->P == 1);
  p->envs = calloc(num_envs, sizeof(rp_env));
  for (int i = 0; i < num_envs; ++i) {
    int32_t s = (int32_t)((uint32_t)seed + (uint32_t)i);
    p->envs[i].seed = s < 1 ? 1 : s;
    p->envs[i].current_step = -1;
  }
  abq_init(&p->abq, (size_t)num_envs);
  p->Q = ((size_t)num_envs / p->B + 2) * 2; /* state_buffer_queue.h:68 */
  p->queue = calloc(p->Q, sizeof(rp_buffer*));
  for (size_t i = 0; i < p->Q; ++i) p->queue[i] = buffer_new(p->B, p->P, p->S);
  circ_init(&p->stock, p->Q); /* :72 */
  long cores = sysconf(_SC_NPROCESSORS_ONLN);
  p->n_create = (int)(cores / 64 > 1 ? cores / 64 : 1); /* :82-84 */
  if (p->n_create > 16) p->n_create = 16;
  for (int i = 0; i < p->n_create; ++i) pthread_create(&p->create_threads[i], NULL, create_buffer_thread, p);
  if (num_threads <= 0) num_threads = (int32_t)(p->B < cores ? p->B : cores); /* async_envpool.h:113-115 */
  p->num_threads = num_threads;
  p->workers = calloc(num_threads, sizeof(pthread_t));
  for (int i = 0; i < num_threads; ++i) pthread_create(&p->workers[i], NULL, worker, p);
  return p;
}

int32_t rp_num_threads(rp_pool* p) { return p->num_threads; }
int32_t rp_action_check_failed(rp_pool* p) { return atomic_load(&p->action_check_failed); }

void rp_destroy(rp_pool* p) {
  if (!p) return;
  atomic_store(&p->stop, 1); /* async_envpool.h:144-154 */
  rp_slice* empty = calloc(p->num_threads, sizeof(rp_slice));
  abq_enqueue_bulk(&p->abq, empty, (size_t)p->num_threads);
  for (int i = 0; i < p->num_threads; ++i) pthread_join(p->workers[i], NULL);
  free(empty);
  atomic_store(&p->quit, 1); /* state_buffer_queue.h:98-107 */
  for (int i = 0; i < p->n_create; ++i) {
    rp_buffer* b = circ_get(&p->stock);
    if (b) { buffer_free_arrays(b); free(b); }
  }
  for (int i = 0; i < p->n_create; ++i) pthread_join(p->create_threads[i], NULL);
  for (size_t i = 0; i < p->stock.size; ++i)
    if (p->stock.buf[i]) { buffer_free_arrays(p->stock.buf[i]); free(p->stock.buf[i]); }
  for (size_t i = 0; i < p->Q; ++i) { buffer_free_arrays(p->queue[i]); free(p->queue[i]); }
  for (int i = 0; i < p->N; ++i) batch_unref(p->envs[i].batch);
  free(p->stock.buf);
  free(p->queue);
  free(p->abq.queue);
  free(p->workers);
  free(p->envs);
  free(p);
}

/* AsyncEnvPool::SendImpl (async_envpool.h:57-80). The action arrays are used in place, as the
 * reference uses the numpy memory, and like the reference's (py_envpool.h:88-100: incref at Send,
 * decref from the deleter of the shared batch) their lifetime is the batch's: `release(ctx)` runs
 * when the last env that was sent with them has been sent again (env.h:106-110) or the pool is
 * destroyed. A worker that has dequeued a slice but not yet read its action row holds no reference
 * of its own — the env's `batch` pointer does — so nothing but that count says when the arrays may
 * go: a caller that recycles them on any other schedule (a fixed rotation, "after the rows were
 * received") races with such a worker. An env must not be sent again before its previous step
 * was received (the same contract the reference's SetAction relies on). */
static void send_impl(rp_pool* p, int32_t n, const int32_t* env_id, int32_t np_rows,
                      const int32_t* players_env_id, const double* list_action,
                      void (*release)(void*), void* ctx) {
  rp_action_batch* a = calloc(1, sizeof(*a));
  a->players_env_id = players_env_id;
  a->list_action = list_action;
  a->np_rows = np_rows;
  a->release = release;
  a->release_ctx = ctx;
  atomic_store(&a->refs, n + 1);
  rp_slice* actions = malloc((size_t)(n > 0 ? n : 1) * sizeof(rp_slice));
  for (int i = 0; i < n; ++i) {
    rp_env* e = &p->envs[env_id[i]];
    batch_unref(e->batch); /* env.h:106-110: the previous batch is dropped here */
    e->batch = a;
    e->env_index = i;
    actions[i] = (rp_slice){env_id[i], p->is_sync ? i : -1, 0};
  }
  if (p->is_sync) atomic_fetch_add(&p->stepping_env_num, n);
  abq_enqueue_bulk(&p->abq, actions, (size_t)n);
  free(actions);
  batch_unref(a);
}

/* The caller keeps `players_env_id` / `list_action` alive for as long as the pool lives (the bench
 * leg: one set of arrays for the whole run). */
void rp_send(rp_pool* p, int32_t n, const int32_t* env_id, int32_t np_rows,
             const int32_t* players_env_id, const double* list_action) {
  send_impl(p, n, env_id, np_rows, players_env_id, list_action, NULL, NULL);
}

/* The pool releases the arrays: `release(ctx)` is called exactly once, from whichever thread drops
 * the batch's last reference. */
void rp_send_owned(rp_pool* p, int32_t n, const int32_t* env_id, int32_t np_rows,
                   const int32_t* players_env_id, const double* list_action,
                   void (*release)(void*), void* ctx) {
  send_impl(p, n, env_id, np_rows, players_env_id, list_action, release, ctx);
}

/* The pool works on its own copy of the arrays (one heap block, freed with the batch): what a
 * binding without reference counting — ctypes — uses. */
void rp_send_copy(rp_pool* p, int32_t n, const int32_t* env_id, int32_t np_rows,
                  const int32_t* players_env_id, const double* list_action) {
  size_t la_bytes = (size_t)(n > 0 ? n : 0) * 6 * sizeof(double);
  size_t pe_bytes = (size_t)(np_rows > 0 ? np_rows : 0) * sizeof(int32_t);
  char* blk = malloc(la_bytes + pe_bytes + 8);
  memcpy(blk, list_action, la_bytes);
  memcpy(blk + la_bytes, players_env_id, pe_bytes);
  send_impl(p, n, env_id, np_rows, (const int32_t*)(blk + la_bytes), (const double*)blk, free, blk);
}

void rp_reset(rp_pool* p, const int32_t* env_ids, int32_t n) { /* async_envpool.h:176-189 */
  rp_slice* actions = malloc((size_t)(n > 0 ? n : 1) * sizeof(rp_slice));
  for (int i = 0; i < n; ++i) actions[i] = (rp_slice){env_ids[i], p->is_sync ? i : -1, 1};
  if (p->is_sync) atomic_fetch_add(&p->stepping_env_num, n);
  abq_enqueue_bulk(&p->abq, actions, (size_t)n);
  free(actions);
}

/* AsyncEnvPool::Recv (async_envpool.h:162-174): blocks like the reference */
void rp_recv(rp_pool* p, rp_batch* out) {
  int64_t additional = 0;
  int64_t stepping = atomic_load(&p->stepping_env_num);
  if (p->is_sync && stepping < p->B) additional = p->B - stepping;
  rp_buffer* newbuf = circ_get(&p->stock); /* state_buffer_queue.h:139-150 */
  uint64_t pos = atomic_fetch_add(&p->done_ptr, 1);
  size_t offset = pos % p->Q;
  rp_buffer* b = p->queue[offset];
  if (additional > 0) buffer_done(p, b, (uint64_t)additional); /* state_buffer.h:135-157 */
  lsem_wait(&b->sem);
  uint64_t off = atomic_load(&b->offsets);
  if (additional > 0) atomic_fetch_add(&p->alloc_count, (uint64_t)additional);
  p->queue[offset] = newbuf;
  memset(out, 0, sizeof(*out));
  out->num_rows = (int32_t)(uint32_t)off;
  out->num_player_rows = (int32_t)(off >> 32);
  out->state_num = p->S;
  out->info_env_id = b->info_env_id;
  out->info_players_env_id = b->info_players_env_id;
  out->elapsed_step = b->elapsed_step;
  out->done = b->done;
  out->reward = b->reward;
  out->discount = b->discount;
  out->step_type = b->step_type;
  out->trunc = b->trunc;
  out->obs_raw = b->obs_raw;
  out->dyn_rows = b->dyn_rows;
  out->dyn_fill = b->dyn_fill;
  out->players_done = b->players_done;
  out->players_id = b->players_id;
  free(b);
  if (p->is_sync) atomic_fetch_sub(&p->stepping_env_num, (int64_t)out->num_rows);
}

void rp_batch_free(rp_batch* o) {
  free(o->info_env_id); free(o->info_players_env_id); free(o->elapsed_step); free(o->done);
  free(o->reward); free(o->discount); free(o->step_type); free(o->trunc); free(o->obs_raw);
  free(o->dyn_rows); free(o->dyn_fill); free(o->players_done); free(o->players_id);
  memset(o, 0, sizeof(*o));
}

/* bench.py cpu_baseline leg: reset + `steps` sync lockstep steps of all N envs through the
 * queues and the worker pool; returns a checksum so the work cannot be elided. */
uint64_t rp_bench_sync(int32_t num_envs, int32_t seed, int32_t steps, int32_t threads) {
  rp_pool* p = rp_create(num_envs, num_envs, 1, seed, 2147483647, 10, threads);
  if (!p) return 0;
  int N = num_envs;
  int32_t* ids = malloc((size_t)N * 4);
  double* la = malloc((size_t)N * 6 * 8);
  for (int i = 0; i < N; ++i) {
    ids[i] = i;
    for (int k = 0; k < 6; ++k) la[(size_t)i * 6 + k] = 1.5;
  }
  uint64_t sum = 0;
  rp_batch o;
  rp_reset(p, ids, N);
  rp_recv(p, &o);
  rp_batch_free(&o);
  for (int s = 0; s < steps; ++s) {
    rp_send(p, N, ids, N, ids, la);
    rp_recv(p, &o);
    sum += (uint64_t)o.elapsed_step[(size_t)s % (size_t)N] + o.done[0] + (uint64_t)o.obs_raw[1];
    rp_batch_free(&o);
  }
  rp_destroy(p);
  free(ids);
  free(la);
  return sum;
}
