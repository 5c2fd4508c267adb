/*
 * dummy_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded restatement of the reference's Dummy env +
 * ActionBufferQueue + StateBuffer(Queue) + AsyncEnvPool path. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (envpool_amd/) never does.
 *
 * Execution model restated: the reference runs worker threads that pop
 * ActionSlices off a FIFO (action_buffer_queue.h:71-80) and call EnvStep
 * (async_envpool.h:116-129). With num_threads == 1 that is strictly FIFO, which
 * is a valid execution of the reference and the one restated here: every slice
 * is processed at enqueue time, in order.
 *
 * PARITY UNPINNED. The reference's Dummy path needs glog and moodycamel's
 * lightweightsemaphore.h, neither of which is in this image, so it cannot be
 * compiled here, and the reference holds no test or fixture for
 * this path. The oracle is checked against the traces SURVEY.md §8a recorded
 * ("Closed form" / "Golden trace" / "Multi-player probe", tests/golden/), but
 * the survey's build of the reference used stand-in headers for those two
 * libraries, which by this project's rule pins nothing. What it rests on: the
 * line-by-line citations below, and agreement with the independently written
 * threaded restatement in dummy_pool_mt.c.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OD_OK 0
#define OD_ERR_WOULD_BLOCK 4
#define OD_ERR_RING_FULL 5
#define OD_ERR_ACTION_CHECK 6
#define OD_ERR_INVALID 1

/* One StateBuffer (state_buffer.h:40-158): arrays are zero-initialised
 * (array.h:77-81 value-initialising new[]). */
typedef struct {
  int32_t* info_env_id;         /* [B]      */
  int32_t* info_players_env_id; /* [B*P]    */
  int32_t* elapsed_step;        /* [B]      */
  uint8_t* done;                /* [B]      */
  float* reward;                /* [B*P]    */
  float* discount;              /* [B*P]    */
  int32_t* step_type;           /* [B]      */
  uint8_t* trunc;               /* [B]      */
  int32_t* obs_raw;             /* [B*P, S] */
  int32_t* dyn_rows;            /* [B*P] obs:dyn Container: 0 = None, else leading dim (env_id+1) */
  int32_t* dyn_fill;            /* [B*P] fill value (env_id) */
  uint8_t* players_done;        /* [B*P]    */
  int32_t* players_id;          /* [B*P]    */
  uint64_t alloc_count, done_count;
  uint32_t player_off, shared_off;
  int signalled;
} od_buffer;

typedef struct {
  int32_t state;        /* DummyEnv::state_ (dummy_envpool.h:118) */
  int32_t current_step; /* Env::current_step_{-1} (env.h:71) */
  int32_t seed;         /* Env::seed_ after the clamp (env.h:94, dummy_envpool.h:126-128) */
} od_env;

typedef struct {
  int32_t N, B, P, S, seed, max_episode_steps, env_id_offset;
  int is_sync;
  int64_t stepping_env_num; /* async_envpool.h:49 */
  od_env* envs;
  int Q;                        /* queue_size_ (state_buffer_queue.h:68) */
  od_buffer* queue;             /* [Q] */
  uint64_t alloc_count, done_ptr; /* state_buffer_queue.h:40 */
  int error;                    /* sticky first error */
} od_pool;

/* what od_recv hands back: the truncated arrays of one batch (caller frees with od_batch_free) */
typedef struct {
  int32_t num_rows, num_player_rows, state_num;
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
} od_batch;

static void buffer_init(od_buffer* b, int B, int P, int S) {
  size_t BP = (size_t)B * P;
  memset(b, 0, sizeof(*b));
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
}

static void buffer_free_arrays(od_buffer* b) {
  free(b->info_env_id); free(b->info_players_env_id); free(b->elapsed_step); free(b->done);
  free(b->reward); free(b->discount); free(b->step_type); free(b->trunc); free(b->obs_raw);
  free(b->dyn_rows); free(b->dyn_fill); free(b->players_done); free(b->players_id);
}

od_pool* od_create(int32_t num_envs, int32_t batch_size, int32_t max_num_players, int32_t seed,
                   int32_t max_episode_steps, int32_t state_num, int32_t env_id_offset) {
  /* env_spec.h:74-82 */
  if (num_envs < 1 || batch_size > num_envs || max_num_players < 1 || state_num < 2) return NULL;
  od_pool* p = calloc(1, sizeof(od_pool));
  p->N = num_envs;
  p->B = batch_size <= 0 ? num_envs : batch_size; /* async_envpool.h:91-92 */
  p->P = max_num_players;
  p->S = state_num;
  p->seed = seed;
  p->max_episode_steps = max_episode_steps;
  p->env_id_offset = env_id_offset;
  p->is_sync = (p->B == p->N && p->P == 1); /* async_envpool.h:95 */
  p->envs = calloc(num_envs, sizeof(od_env));
  for (int i = 0; i < num_envs; ++i) {
    /* env.h:94 seed_ = seed + env_id (int arithmetic; wraps like the two's-complement hardware
     * add the reference compiles to), dummy_envpool.h:126-128 clamp */
    int32_t s = (int32_t)((uint32_t)seed + (uint32_t)(env_id_offset + i));
    if (s < 1) s = 1;
    p->envs[i].seed = s;
    p->envs[i].state = 0;
    p->envs[i].current_step = -1;
  }
  p->Q = (num_envs / p->B + 2) * 2; /* state_buffer_queue.h:68 */
  p->queue = calloc(p->Q, sizeof(od_buffer));
  for (int i = 0; i < p->Q; ++i) buffer_init(&p->queue[i], p->B, p->P, p->S);
  return p;
}

void od_destroy(od_pool* p) {
  if (!p) return;
  for (int i = 0; i < p->Q; ++i) buffer_free_arrays(&p->queue[i]);
  free(p->queue);
  free(p->envs);
  free(p);
}

int od_queue_size(od_pool* p) { return p->Q; }
int od_is_sync(od_pool* p) { return p->is_sync; }

/* StateBuffer::Done (state_buffer.h:124-129) */
static void buffer_done(od_pool* p, od_buffer* b, uint64_t num) {
  uint64_t dc = b->done_count;
  b->done_count += num;
  if (dc + num == (uint64_t)p->B) b->signalled += 1;
}

static int is_done(const od_env* e) { return e->state >= e->seed; } /* dummy_envpool.h:215 */

/* Env::EnvStep for one slice (env.h:162-172) with DummyEnv::Reset/Step inlined.
 * `players_*`/`np_rows` describe the whole action batch (needed by the multi-player
 * ParseAction branch, env.h:124-159); `env_index` is this env's row in it. */
static int env_step(od_pool* p, int32_t env_id, int order, int force_reset, int env_index,
                    int np_rows, const int32_t* a_players_env_id, const double* a_list_action) {
  int local = env_id - p->env_id_offset;
  if (local < 0 || local >= p->N) return OD_ERR_INVALID;
  od_env* e = &p->envs[local];
  int reset = force_reset || is_done(e); /* async_envpool.h:125 auto-reset */
  /* PreProcess (env.h:181-189) */
  if (reset) e->current_step = 0; else e->current_step += 1;

  int action_num = 0;
  int status = OD_OK;
  if (reset) {
    e->state = 0; /* dummy_envpool.h:138 */
  } else {
    /* ParseAction (env.h:112-160) */
    if (p->P == 1) {
      /* single player: players.env_id row = Slice(env_index, env_index+1) */
      action_num = 1; /* Shape(0) of that slice (dummy_envpool.h:179) */
      if (a_players_env_id[env_index] != env_id) action_num = 0; /* :180-184 */
    } else {
      int cnt = 0;
      for (int j = 0; j < np_rows; ++j)
        if (a_players_env_id[j] == env_id) ++cnt; /* env.h:128-132 */
      action_num = cnt; /* every selected row equals env_id, so the :180-184 loop never zeroes it */
    }
    ++e->state; /* dummy_envpool.h:174 */
    /* :187-192 CHECK_EQ of the six list_action lanes (aborts the reference) */
    const double* la = a_list_action + (size_t)env_index * 6;
    for (int k = 0; k < 6; ++k)
      if (!(la[0] == la[k])) status = OD_ERR_ACTION_CHECK;
  }
  int num_players = p->P <= 1 ? 1 : e->state % (p->P - 1) + 1; /* :139-140, :175-176 */

  /* StateBufferQueue::Allocate (state_buffer_queue.h:114-128) */
  uint64_t pos = p->alloc_count++;
  od_buffer* b = &p->queue[(pos / (uint64_t)p->B) % (uint64_t)p->Q];
  /* StateBuffer::Allocate (state_buffer.h:79-112) */
  uint64_t ac = b->alloc_count++;
  if (ac >= (uint64_t)p->B) return OD_ERR_RING_FULL; /* :111 out_of_range */
  uint32_t player_off = b->player_off, shared_off = b->shared_off;
  b->player_off += (uint32_t)num_players; /* :85-88 packed fetch_add */
  b->shared_off += 1;
  if (order != -1 && p->P == 1) player_off = shared_off = (uint32_t)order; /* :92-95 */

  /* Env::Allocate common fields (env.h:196-222) */
  int done = is_done(e);
  b->done[shared_off] = (uint8_t)done;
  b->discount[player_off] = (float)(!done); /* scalar assign hits element 0 only (array.h:253-257) */
  b->step_type[shared_off] = e->current_step == 0 ? 0 : done ? 2 : 1;
  b->trunc[shared_off] = (uint8_t)(done && (e->current_step >= p->max_episode_steps));
  b->info_env_id[shared_off] = env_id;
  b->elapsed_step[shared_off] = e->current_step;
  for (int i = 0; i < num_players; ++i) b->info_players_env_id[player_off + i] = env_id;
  /* DummyEnv::Reset / Step writes (dummy_envpool.h:147-163, 199-209) */
  for (int i = 0; i < num_players; ++i) {
    size_t r = (size_t)player_off + i;
    b->players_id[r] = i;
    b->players_done[r] = (uint8_t)done;
    b->obs_raw[r * p->S + 0] = e->state;
    b->obs_raw[r * p->S + 1] = reset ? 0 : action_num;
    b->reward[r] = (float)(-i);
    b->dyn_rows[r] = env_id + 1;
    b->dyn_fill[r] = env_id;
  }
  /* PostProcess -> done_write -> Done() (env.h:191-194) */
  buffer_done(p, b, 1);
  return status;
}

/* AsyncEnvPool::Reset (async_envpool.h:176-189) */
int od_reset(od_pool* p, const int32_t* env_ids, int32_t n) {
  if (p->is_sync) p->stepping_env_num += n;
  int rc = OD_OK;
  for (int i = 0; i < n; ++i) {
    int r = env_step(p, env_ids[i], p->is_sync ? i : -1, 1, i, 0, NULL, NULL);
    if (r != OD_OK && rc == OD_OK) rc = r;
  }
  if (rc != OD_OK && p->error == OD_OK) p->error = rc;
  return rc;
}

/* AsyncEnvPool::SendImpl (async_envpool.h:57-80). Arrays in _action_keys order. */
int od_send(od_pool* p, int32_t n, const int32_t* env_id, int32_t np_rows,
            const int32_t* players_env_id, const double* list_action) {
  if (p->is_sync) p->stepping_env_num += n;
  int rc = OD_OK;
  for (int i = 0; i < n; ++i) {
    int r = env_step(p, env_id[i], p->is_sync ? i : -1, 0, i, np_rows, players_env_id, list_action);
    if (r != OD_OK && rc == OD_OK) rc = r;
  }
  if (rc != OD_OK && p->error == OD_OK) p->error = rc;
  return rc;
}

#define COPY(dst, src, n, T)                          \
  do {                                                \
    (dst) = malloc(((n) > 0 ? (n) : 1) * sizeof(T));  \
    memcpy((dst), (src), (size_t)(n) * sizeof(T));    \
  } while (0)

/* AsyncEnvPool::Recv (async_envpool.h:162-174) -> StateBufferQueue::Wait
 * (state_buffer_queue.h:139-150) -> StateBuffer::Wait (state_buffer.h:135-157) */
int od_recv(od_pool* p, od_batch* out) {
  int64_t additional = 0;
  if (p->is_sync && p->stepping_env_num < p->B) additional = p->B - p->stepping_env_num;
  od_buffer* b = &p->queue[p->done_ptr % (uint64_t)p->Q];
  /* would the semaphore ever be signalled? (the reference simply blocks) */
  if (b->done_count + (uint64_t)additional != (uint64_t)p->B && b->signalled == 0)
    return OD_ERR_WOULD_BLOCK;
  p->done_ptr++;
  if (additional > 0) buffer_done(p, b, (uint64_t)additional);
  b->signalled -= 1;
  uint32_t po = b->player_off, so = b->shared_off;
  memset(out, 0, sizeof(*out));
  out->num_rows = (int32_t)so;
  out->num_player_rows = (int32_t)po;
  out->state_num = p->S;
  COPY(out->info_env_id, b->info_env_id, so, int32_t);
  COPY(out->info_players_env_id, b->info_players_env_id, po, int32_t);
  COPY(out->elapsed_step, b->elapsed_step, so, int32_t);
  COPY(out->done, b->done, so, uint8_t);
  COPY(out->reward, b->reward, po, float);
  COPY(out->discount, b->discount, po, float);
  COPY(out->step_type, b->step_type, so, int32_t);
  COPY(out->trunc, b->trunc, so, uint8_t);
  COPY(out->obs_raw, b->obs_raw, (size_t)po * p->S, int32_t);
  COPY(out->dyn_rows, b->dyn_rows, po, int32_t);
  COPY(out->dyn_fill, b->dyn_fill, po, int32_t);
  COPY(out->players_done, b->players_done, po, uint8_t);
  COPY(out->players_id, b->players_id, po, int32_t);
  if (additional > 0) p->alloc_count += (uint64_t)additional; /* state_buffer_queue.h:144-147 */
  /* swap in a fresh zeroed buffer (state_buffer_queue.h:140,148) */
  buffer_free_arrays(b);
  buffer_init(b, p->B, p->P, p->S);
  if (p->is_sync) p->stepping_env_num -= so; /* async_envpool.h:170-172 */
  return OD_OK;
}

void od_batch_free(od_batch* o) {
  free(o->info_env_id); free(o->info_players_env_id); free(o->elapsed_step); free(o->done);
  free(o->reward); free(o->discount); free(o->step_type); free(o->trunc); free(o->obs_raw);
  free(o->dyn_rows); free(o->dyn_fill); free(o->players_done); free(o->players_id);
  memset(o, 0, sizeof(*o));
}

/* --- CPU-baseline helper ----------------------------------------------------
 * Runs `steps` sync-mode lockstep steps of all N envs (send + recv) with a
 * constant action batch and returns a checksum so the work cannot be elided.
 * Used only by bench.py's cpu_baseline leg. obs:dyn is kept lazy (shape + fill
 * only); the reference would allocate (env_id+1)*state_num ints per env-step
 * (dummy_envpool.h:205-208), which is O(N^2) per step and cannot run at
 * N = 65536 at all. */
uint64_t od_bench_sync(od_pool* p, int32_t steps) {
  int N = p->N;
  int32_t* ids = malloc((size_t)N * 4);
  double* la = malloc((size_t)N * 6 * 8);
  for (int i = 0; i < N; ++i) {
    ids[i] = p->env_id_offset + i;
    for (int k = 0; k < 6; ++k) la[(size_t)i * 6 + k] = 1.5;
  }
  uint64_t sum = 0;
  od_batch o;
  od_reset(p, ids, N);
  od_recv(p, &o);
  od_batch_free(&o);
  for (int s = 0; s < steps; ++s) {
    od_send(p, N, ids, N, ids, la);
    od_recv(p, &o);
    sum += (uint64_t)o.elapsed_step[(size_t)s % (size_t)N] + o.done[0] + (uint64_t)o.obs_raw[1];
    od_batch_free(&o);
  }
  free(ids);
  free(la);
  return sum;
}

/* ---- multi-threaded CPU baseline (bench.py cpu_baseline only) -----------------------------
 * The reference runs its envs on a thread pool (async_envpool.h:105-135). The port's
 * counterpart: `threads` independent shards of the same pool, one per core, each the sequential
 * program above over a contiguous env_id block (seed_ uses the global id, so the union is the
 * one big pool). No lockstep barrier between shards: this favours the CPU. */
#include <pthread.h>

typedef struct {
  int32_t n, off, seed, steps;
  uint64_t sum;
} od_mt_arg;

static void* od_mt_run(void* v) {
  od_mt_arg* a = (od_mt_arg*)v;
  od_pool* p = od_create(a->n, a->n, 1, a->seed, 2147483647, 10, a->off);
  a->sum = p ? od_bench_sync(p, a->steps) : 0;
  od_destroy(p);
  return NULL;
}

uint64_t od_bench_sync_mt(int32_t num_envs, int32_t seed, int32_t steps, int32_t threads) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t th[256];
  od_mt_arg arg[256];
  int32_t per = (num_envs + threads - 1) / threads, started = 0;
  for (int t = 0; t < threads; ++t) {
    int32_t off = t * per, n = num_envs - off < per ? num_envs - off : per;
    if (n <= 0) break;
    arg[t] = (od_mt_arg){n, off, seed, steps, 0};
    if (pthread_create(&th[t], NULL, od_mt_run, &arg[t]) != 0) break;
    ++started;
  }
  uint64_t sum = 0;
  for (int t = 0; t < started; ++t) {
    pthread_join(th[t], NULL);
    sum += arg[t].sum;
  }
  return sum;
}
