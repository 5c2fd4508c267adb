/*
 * envpool_amd.h — C ABI of the MI355X-native envpool2 hot path.
 *
 * This is the drop-in boundary. Everything above it (pybind11 classes
 * `_DummyEnvSpec` / `_DummyEnvPool`, the Python `make` / `send` / `recv` /
 * `step` / `reset` layer) is host glue; everything below it is HIP for gfx950.
 * No torch / pybind / C++ types cross this boundary: plain pointers, sizes and
 * status codes only.
 *
 * The reference has no FFI layer of its own for this path: its boundary is the
 * templated C++ class `PyEnvPool<AsyncEnvPool<DummyEnv>>`
 * (envpool2/core/py_envpool.h:204-252) whose three entry points are
 * `PySend` (:219-225), `PyRecv` (:230-241) and `PyReset` (:246-251). Each entry
 * point below cites the reference interface it replaces.
 *
 * Threading contract (same as the reference, state_buffer_queue.h:131-138 and
 * action_buffer_queue.h:60): `ep_pool_send` / `ep_pool_reset` may be called
 * from several threads (they are serialised inside); `ep_pool_recv` has a
 * single consumer.
 */
#ifndef ENVPOOL_AMD_H_
#define ENVPOOL_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
enum {
  EP_OK = 0,
  EP_ERR_INVALID = 1,      /* bad argument; std::invalid_argument in the reference */
  EP_ERR_HIP = 2,          /* a HIP runtime call failed */
  EP_ERR_NO_DEVICE = 3,    /* no gfx950 device visible: the product path has no CPU fallback */
  EP_ERR_WOULD_BLOCK = 4,  /* recv would wait forever (reference: deadlock in StateBuffer::Wait) */
  EP_ERR_RING_FULL = 5,    /* more un-received batches than ring slots (reference: std::out_of_range,
                              state_buffer.h:111) */
  EP_ERR_ACTION_CHECK = 6, /* Dummy list_action lanes differ (reference: glog CHECK_EQ abort,
                              dummy_envpool.h:189-192) or env_id out of range */
  EP_ERR_TOO_LARGE = 7,
  EP_ERR_DUP_ENV = 8       /* a device-memory send or rollout named an env twice in one launch: the two
                              items raced, the rows of that env are not the reference's (the batch is
                              still returned). Host-memory sends never report it: they are split on
                              the host so that such an env is stepped twice IN ORDER, which is what
                              the reference's worker does (async_envpool.h:116-129) */
};

/* ---- memory kinds --------------------------------------------------------- */
enum {
  EP_MEM_HOST = 0,  /* pointers are host memory (numpy); send copies, recv hands out a fresh host block */
  EP_MEM_DEVICE = 1 /* pointers are device memory on the pool's GPU; zero-copy both ways */
};

/* ---- element types -------------------------------------------------------- */
enum {
  EP_I32 = 0,
  EP_F32 = 1,
  EP_F64 = 2,
  EP_BOOL = 3, /* 1 byte, 0/1 */
  EP_U8 = 4,
  EP_CONTAINER_I32 = 5, /* ragged Container<int> (spec.h:96-109): no device array, see obs:dyn */
  EP_U16 = 6
};

#define EP_MAX_KEYS 20
#define EP_MAX_NDIM 4

typedef struct ep_pool ep_pool; /* opaque */

/* One returned array. Row-major, leading dim = rows actually returned
 * (state_buffer.h:146-156 Truncate). `data` is NULL for EP_CONTAINER_I32. */
typedef struct {
  void* data;
  int64_t shape[EP_MAX_NDIM];
  int32_t ndim;
  int32_t dtype;
} ep_array;

/* One received batch == one StateBuffer of the reference (state_buffer.h:40). */
typedef struct {
  int32_t num_keys;        /* arrays[0..num_keys) in `_state_keys` order */
  int32_t mem;             /* EP_MEM_HOST or EP_MEM_DEVICE */
  int32_t num_rows;        /* shared_offset: envs in this batch */
  int32_t num_player_rows; /* player_offset: players in this batch */
  uint64_t seq;            /* index of this batch since pool creation */
  void* owner;             /* host block to free with ep_batch_release (host mem), else NULL */
  ep_array arrays[EP_MAX_KEYS];
} ep_batch;

/* ---- Dummy-v0 ------------------------------------------------------------- */

/* Config of the Dummy pool. First eleven fields are the reference's config keys
 * (env_spec.h:25-30 ++ dummy_envpool.h:47-49) minus the two that are host-only
 * strings/flags (`base_path`, `gym_reset_return_info`, kept by the pybind
 * layer). Build-only keys come after them. */
typedef struct {
  int32_t num_envs;               /* envs owned by THIS pool (local shard) */
  int32_t batch_size;             /* 0 => num_envs (env_spec.h:80-82) */
  int32_t num_threads;            /* accepted, unused: there is no worker thread pool */
  int32_t max_num_players;
  int32_t thread_affinity_offset; /* accepted, unused */
  int32_t seed;
  int32_t max_episode_steps;
  int32_t state_num;
  int32_t action_num;
  /* build-only */
  int32_t device_id;     /* HIP device ordinal */
  int32_t env_id_offset; /* global env_id of local env 0 (multi-GPU shard: env.h:94 seed_ = seed + GLOBAL id) */
  int32_t ring_depth;    /* StateBuffer ring slots; 0 => reference rule (num_envs/batch+2)*2
                            (state_buffer_queue.h:68) */
} ep_dummy_config;

/* Replaces: AsyncEnvPool ctor (async_envpool.h:88-142) + StateBufferQueue ctor
 * (state_buffer_queue.h:48-96) + DummyEnv ctor (dummy_envpool.h:125-129).
 * Allocates the device-resident env state, the StateBuffer ring and the action
 * staging ring on `device_id`. Fails with EP_ERR_NO_DEVICE when no GPU is
 * visible. batch_size > num_envs => EP_ERR_INVALID (env_spec.h:74-79). */
int ep_dummy_pool_create(const ep_dummy_config* cfg, ep_pool** out);

/* Replaces: ~AsyncEnvPool (async_envpool.h:144-154). */
int ep_pool_destroy(ep_pool* pool);

/* Replaces: PyEnvPool::PyReset -> AsyncEnvPool::Reset (py_envpool.h:246-251,
 * async_envpool.h:176-189). Enqueues a forced reset of `n` envs; results appear
 * in the StateBuffer ring exactly like a step's. `env_ids` are GLOBAL ids. */
int ep_pool_reset(ep_pool* pool, const int32_t* env_ids, int32_t n, int32_t mem);

/* Replaces: PyEnvPool::PySend -> AsyncEnvPool::SendImpl (py_envpool.h:219-225,
 * async_envpool.h:57-80). `arrays[k]` is the k-th action array in
 * `_action_keys` order (env_spec.h:31-35: env_id, players.env_id, then the
 * env's own keys), already of the spec dtype and C-contiguous;
 * `leading_dims[k]` its leading dimension. Non-blocking: the step kernel is
 * enqueued on the pool's stream and the call returns. Host arrays are fully
 * consumed (copied to pinned staging) before return; device arrays must stay
 * untouched until the matching recv. An env named twice in one send is stepped
 * twice, in order, like the reference's single worker does (host arrays: the
 * send is split into launches at every repeat). Device arrays cannot be looked
 * at by the host: there the items of one launch run concurrently, a repeat is
 * detected by the kernels (per-env launch stamps) and the batch comes back with
 * EP_ERR_DUP_ENV from recv. */
int ep_pool_send(ep_pool* pool, int32_t n_arrays, const void* const* arrays,
                 const int64_t* leading_dims, int32_t mem);

/* Replaces: PyEnvPool::PyRecv -> AsyncEnvPool::Recv -> StateBufferQueue::Wait
 * -> StateBuffer::Wait (py_envpool.h:230-241, async_envpool.h:162-174,
 * state_buffer_queue.h:139-150, state_buffer.h:135-157). Blocks until the head
 * batch is complete - also for env-steps that other threads have yet to send,
 * like StateBuffer::Wait; the pool's lock is not held while waiting, so sends
 * from other threads go through - then describes it in `out`. EP_MEM_HOST: a
 * fresh host block holding every array (never reused; free with
 * ep_batch_release). EP_MEM_DEVICE: views into the ring slot, valid until
 * ring_depth-1 further batches have been received. How long it waits for rows
 * nobody has sent yet is set by ep_pool_set_recv_timeout (EP_ERR_WOULD_BLOCK
 * when the time is up). Returns EP_ERR_ACTION_CHECK when a step of this batch
 * failed the Dummy env's action self-check and EP_ERR_DUP_ENV for a repeated
 * env in a device-memory send (the batch is still returned in both cases). */
int ep_pool_recv(ep_pool* pool, int32_t mem, ep_batch* out);

/* One blocking lockstep step with device-resident actions: ep_pool_send(EP_MEM_DEVICE) followed
 * by ep_pool_recv(EP_MEM_DEVICE) (envpool.py:135-142 step() = send + recv), as ONE call. For a sync
 * pool stepped as a whole (every leading dim == num_envs, num_envs a multiple of 256, list_action
 * 16-byte aligned, nothing else outstanding) no kernel is launched at all: a resident kernel of the
 * pool (one workgroup per CU) waits on a doorbell - a word of device memory that the host writes
 * through the PCIe BAR - steps the envs and answers in words of host memory the host spins on; one
 * launch costs more than the step itself on this platform. The resident kernel is bound to the three
 * action arrays it was started for (a step with other pointers restarts it) and is stopped by any
 * other entry point of the pool; results are those of send + recv, views valid as for
 * ep_pool_recv(EP_MEM_DEVICE). Every other case takes the send + recv path. No counterpart in the
 * reference beyond step() itself.
 *
 * HAZARD - device-wide synchronisation. While the resident kernel sits on the GPU, anything that
 * waits for the whole device (hipDeviceSynchronize, torch.cuda.synchronize(), hipFree / hipMalloc
 * of some sizes, creating or destroying another pool) waits for that kernel to leave, and a kernel
 * enqueued on a stream that shares a hardware queue with the pool's server stream (a process has 4
 * by default, GPU_MAX_HW_QUEUES) can be held behind it the same way. The pool bounds that wait: an
 * idle watchdog (a host thread per pool that has used the server) stops the kernel once no step
 * has come for about ten times the gap between the last two steps, at least 200 us and at most
 * 50 ms, and a caller that keeps running into it (idle stops in a row) is served by the
 * launch-per-step path for the next 8, 16, ... 1024 steps before the server is tried again - so a
 * loop that synchronises the device after every step pays one stall of about 200 us now and then
 * and otherwise the 12 us of send + recv. Use stream-scoped waits (hipStreamSynchronize,
 * hipEventSynchronize, torch.cuda.current_stream().synchronize()) between steps to keep the
 * 6.5 us path; ep_pool_sync stops the server explicitly. It also occupies one workgroup slot
 * (256 threads, 12 KB of LDS) on every CU next to the caller's own kernels. */
int ep_pool_step(ep_pool* pool, int32_t n_arrays, const void* const* arrays, const int64_t* leading_dims,
                 ep_batch* out);

/* ep_pool_step in two halves, for a caller that steps several pools at once (one per GPU behind
 * one front-end: ring every pool's doorbell, then wait for each): begin = the send half (rings the
 * resident kernel, or enqueues the step kernel where the server does not apply), end = the blocking
 * recv half. Between the two, every other entry point of THIS pool is refused (EP_ERR_INVALID) when
 * the server was rung. ep_pool_step(...) == begin + end. */
int ep_pool_step_begin(ep_pool* pool, int32_t n_arrays, const void* const* arrays, const int64_t* leading_dims);
int ep_pool_step_end(ep_pool* pool, ep_batch* out);

/* Counters of the resident step server since pool creation: kernels launched, times the idle
 * watchdog stopped it, steps it served (the others took the launch path), and its current idle
 * limit in microseconds. Any pointer may be NULL. */
int ep_pool_server_stats(ep_pool* pool, uint64_t* launches, uint64_t* idle_stops, uint64_t* steps_served,
                         int64_t* idle_limit_us);
/* Diagnostic (pools created with EP_SERVER_TRACE=1 in the environment): stops the server and copies,
 * per workgroup of its last launch, four timestamps of the last step it served (100 MHz device wall
 * clock: doorbell seen, first tile staged, stores drained, counted in) into out[4 * workgroup + i].
 * Returns minus the number of workgroups written (0: none), or a positive status code. */
int ep_pool_server_trace(ep_pool* pool, uint64_t* out, int32_t max_workgroups);

/* How long ep_pool_recv / ep_pool_recv_many wait for env-steps that have not been sent yet:
 * < 0 (the default) for ever, as the reference does (a single-threaded caller that receives more
 * than it sent deadlocks there and here); 0 never - EP_ERR_WOULD_BLOCK at once; > 0 that many
 * milliseconds. Waiting for the GPU to finish rows that HAVE been sent is not affected. No
 * counterpart in the reference. */
int ep_pool_set_recv_timeout(ep_pool* pool, int64_t milliseconds);

/* Stream-ordered ep_pool_recv(EP_MEM_DEVICE): describes the head batch as device views at once
 * and makes `consumer_stream` (a hipStream_t; NULL or ep_pool_stream() = the pool's own step
 * stream, which needs no wait at all) wait for the kernels that write it - the host never
 * blocks, so a policy that runs on the GPU sees one kernel launch per step instead of a launch
 * plus a host round trip. Never waits for unsent rows (EP_ERR_WOULD_BLOCK at once). Differences: single-player
 * pools only (EP_ERR_INVALID otherwise); an action self-check failure of such a batch is reported
 * by the next ep_pool_sync, not here; the views stay valid until ring_depth-1 further batches
 * have been received AND the consumer's reads are ordered before the step stream's later work
 * (automatic when the consumer is the step stream; the library lets the host run at most about
 * two rings ahead). No counterpart in the reference (StateBufferQueue::Wait always blocks a host
 * thread, state_buffer_queue.h:139-150). */
int ep_pool_recv_on_stream(ep_pool* pool, void* consumer_stream, ep_batch* out);

/* `n` consecutive ep_pool_recv_on_stream calls in one: the T batches of a rollout handed to a
 * consumer on the GPU without the host waiting for any of them. */
int ep_pool_recv_many_on_stream(ep_pool* pool, void* consumer_stream, int32_t n, ep_batch* out);

/* One lockstep step in stream order, for a policy that lives on the GPU: the pool's step stream is
 * made to wait for everything submitted to `stream` (a hipStream_t; NULL = the pool's step stream
 * itself, as for ep_pool_recv_on_stream - a caller on the legacy default stream has the pool adopt
 * it first, ep_pool_set_stream(pool, NULL, 0)) so far - the kernels that wrote the actions, and the consumer's reads of the ring slot this
 * step refills - then ep_pool_send(EP_MEM_DEVICE), then ep_pool_recv_on_stream(stream): `stream`
 * waits for the rows, the host for nothing. What ep_pool_step does with a host round trip per step
 * (6.7 us), this does with none, at the price of ep_pool_recv_on_stream's differences (single-player
 * pools; errors of a step are reported later: ep_pool_poll_errors / ep_pool_sync). It is the form a
 * front-end over several pools on ONE device uses, where the resident kernel of ep_pool_step does
 * not apply. No counterpart in the reference. */
int ep_pool_step_on_stream(ep_pool* pool, void* stream, int32_t n_arrays, const void* const* arrays,
                           const int64_t* leading_dims, ep_batch* out);
/* The send half of it alone: ep_pool_send(EP_MEM_DEVICE) ordered behind everything submitted to
 * `stream` so far. A caller that steps several pools from one stream sends to ALL of them first and
 * only then receives (ep_pool_recv_on_stream): each pool then waits for the actions alone, not for the
 * rows of the pools before it, and their kernels run side by side. No counterpart in the reference
 * (AsyncEnvPool::Send, async_envpool.h:162-164, takes host arrays and returns at once: this is that
 * call for arrays a stream is still writing). */
int ep_pool_send_on_stream(ep_pool* pool, void* stream, int32_t n_arrays, const void* const* arrays,
                           const int64_t* leading_dims);

/* Makes `stream` (a hipStream_t of the pool's device; NULL = the legacy default stream) the pool's
 * step stream - the pool ADOPTS it: every later launch of the pool goes there, and ep_pool_stream()
 * returns it - or, with own != 0, returns to the stream the pool created. Waits for the pool's
 * outstanding work first. With the consumer's stream adopted, ep_pool_step_on_stream /
 * ep_pool_send_on_stream / ep_pool_recv_on_stream on that same stream need no event at all (one
 * stream is in order by itself): 3.4 us of host time per step instead of 71, 4.8 us per step
 * sustained instead of the 113 us that a step ordered against another stream by events in both
 * directions measured on this platform (profiles/r03/host_latency.txt) - what a front-end that
 * steps several pools from one stream saves K times per step. The caller keeps the stream alive until the pool is destroyed or
 * has returned to its own. No counterpart in the reference. */
int ep_pool_set_stream(ep_pool* pool, void* stream, int32_t own);

/* gymnasium's `terminated = done & ~truncated` (gymnasium_envpool.py:73-88) of a batch handed out as
 * device views, WITHOUT a launch: the one-launch-per-send step kernels and the resident kernel of
 * ep_pool_step write it beside the ring slot (one byte per row, valid as long as the batch's views).
 * *term = uint8[num_rows] on EP_OK; EP_ERR_INVALID when this batch has none - rows written by
 * ep_pool_rollout, from host memory into a pinned block, by a multi-player pool, or a slot that has
 * been handed out again since: compute it with ep_terminated_many then. */
int ep_pool_terminated_of(ep_pool* pool, const ep_batch* batch, void** term);

/* Reports (and clears) what the kernels of batches handed out in stream order have flagged SO FAR:
 * EP_OK, EP_ERR_ACTION_CHECK or EP_ERR_DUP_ENV. Never waits and touches no GPU API - a flag a
 * running kernel has yet to raise is seen by a later poll, or by ep_pool_sync, which waits first. */
int ep_pool_poll_errors(ep_pool* pool);

/* `n` consecutive ep_pool_recv calls in one (the natural companion of
 * ep_pool_rollout). `out` points at n ep_batch structs. */
int ep_pool_recv_many(ep_pool* pool, int32_t mem, int32_t n, ep_batch* out);

/* Frees the host block of a EP_MEM_HOST batch (the reference frees it when the
 * numpy capsule dies, py_envpool.h:42-46). No-op for device views. */
int ep_batch_release(ep_batch* batch);

/* Result memory that the CALLER lays out, for a front-end that puts several pools' batches into one
 * array per key (python/sharded.py: K shard pools, one numpy array per key over all shards, no copy
 * to stitch them). No counterpart in the reference: its one pool writes one StateBuffer
 * (state_buffer.h:92-95); this is what keeps K pools looking like that to the host.
 *
 * ep_host_blocks is a recycling set of pinned blocks of `bytes` each, mapped into every device of the
 * process. Blocks are reference-counted: ep_host_blocks_take hands one out held once,
 * ep_host_blocks_give drops that hold, and a pool given the block as a target holds it until its
 * batch is done with it - the memory goes back to the set (or is freed, after
 * ep_host_blocks_destroy) when the last hold is dropped, never while a kernel may write it. */
typedef struct ep_host_blocks ep_host_blocks;
int ep_host_blocks_create(size_t bytes, ep_host_blocks** out);
int ep_host_blocks_take(ep_host_blocks* blocks, void** block);
int ep_host_blocks_give(ep_host_blocks* blocks, void* block);
int ep_host_blocks_destroy(ep_host_blocks* blocks);

/* The next batch that ep_pool_send / ep_pool_reset START on this pool is delivered into `block`:
 * state key k (slot order: info:env_id, info:players.env_id, elapsed_step, done, reward, discount,
 * step_type, trunc, obs:raw, info:players.done, info:players.id) is an array at block + key_off[k]
 * (16-byte aligned), and row r of the batch goes to row row_offset + r of it - K pools given the
 * same key_off and row_offset = 0, n, 2n, ... fill one array per key between them. Host-memory sends have their kernels write the block directly (as they
 * write the pool's own block otherwise); batches written into the ring are copied there by
 * ep_pool_recv(EP_MEM_HOST). That recv returns arrays pointing into the block with `owner` NULL.
 * A batch received as device views after all is complete in its ring slot; what the block holds
 * then is unspecified (its rows, if the kernels had written them there already). One target is pending at a time
 * (a new one replaces it; hb = block = NULL cancels it). Refused with EP_ERR_INVALID when a key
 * would not fit (rows row_offset .. row_offset + batch_size of the array at key_off[k], inside `bytes`), for a block that is not out of
 * `blocks`, and for multi-player pools (their player rows have no fixed place). */
int ep_pool_host_target(ep_pool* pool, ep_host_blocks* blocks, void* block, const size_t* key_off, int64_t row_offset);

/* Fused lockstep loop: T consecutive sync-mode send+step rounds of ALL envs in
 * one kernel launch (the "persistent-kernel step loop" of the north star; the
 * reference's counterpart is T iterations of envpool.py:135-142 step()).
 * `arrays[k]` is the k-th action array in `_action_keys` order with an extra
 * leading step dimension, laid out [T][leading_dims[k]][...] on the device;
 * `step_stride_rows[k]` is the row stride between consecutive steps (0 = reuse
 * the same action every step). The T result batches are queued in the ring and
 * are handed out by the next T ep_pool_recv calls. Requires sync mode
 * (batch_size == num_envs, max_num_players == 1), the queue on a batch boundary
 * and T free ring slots (so with ring_depth >= 2T the next rollout can be
 * enqueued while the previous one's batches are still being received). */
int ep_pool_rollout(ep_pool* pool, int32_t T, int32_t n_arrays, const void* const* arrays,
                    const int64_t* leading_dims, const int64_t* step_stride_rows);

/* One stateless send or reset that allocates nothing. No counterpart in the reference; it is for
 * callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc. It is ep_pool_reset / ep_pool_send (async_envpool.h:57-80, 176-189) without the
 * pool object: the caller owns the env state and the StateBuffer arrays, everything runs on
 * `stream` and the call returns without waiting. The kernels are the pool's multi-player kernels
 * (which also handle max_num_players == 1); their scratch (histogram counters, look-back granules,
 * tile ticket) is a static device array inside the library, about 2.1 MB per GPU, which limits
 * num_envs to EP_NOALLOC_MAX_ENVS. Calls no hipMalloc / hipFree or allocating routine; the first
 * call on a device looks the array's address up (do that once outside a graph capture).
 *   cfg         num_envs, max_num_players, seed, max_episode_steps, state_num, env_id_offset,
 *               device_id are read; the rest is ignored
 *   env_state   device, num_envs x {int32 state, int32 cur_step}; in/out; {0, -1} for a fresh env
 *   n, env_id, np_rows, players_env_id, list_action   device action arrays as in ep_pool_send
 *               (players.action / players.id are never read by the Dummy env); force_reset != 0:
 *               only env_id is read (ep_pool_reset)
 *   out         12 ep_array in _state_keys order with DEVICE data pointers and shape[0] = the rows
 *               the array can take: at least n for the per-env keys, n * max_num_players for the
 *               per-player keys (obs:dyn, index 9, is ignored); item i writes row i, player rows in
 *               item order (one batch, like a pool with batch_size = n receives it)
 *   result      device (or host-mapped) words, cleared by the call and written by the kernels
 * All calls on one GPU share the scratch array: calls that overlap in time on one GPU race
 * (serialise them on one stream). An env named twice in one call is not detected here. */
#define EP_NOALLOC_MAX_ENVS 262144
typedef struct {
  uint32_t num_player_rows; /* player_offset of the batch (state_buffer.h:85-88) */
  uint32_t action_check;    /* != 0: list_action lanes differed (EP_ERR_ACTION_CHECK of ep_pool_recv) */
  uint32_t env_id_range;    /* != 0: an env_id outside [env_id_offset, env_id_offset + num_envs) */
  uint32_t reserved;
} ep_noalloc_result;
int ep_dummy_send_noalloc(const ep_dummy_config* cfg, void* env_state, int32_t n, const int32_t* env_id,
                          int32_t np_rows, const int32_t* players_env_id, const double* list_action,
                          int32_t force_reset, const ep_array* out, int32_t n_out, ep_noalloc_result* result,
                          void* stream);

/* Wait for everything enqueued on the pool's streams. */
int ep_pool_sync(ep_pool* pool);

/* Per-episode statistics of the most recently finished episode of every local
 * env (new: the reference keeps none). fin_return[e] = sum of the
 * env's rewards over that episode, fin_length[e] = its elapsed_step at done,
 * ep_count[e] = episodes finished so far. Any pointer may be NULL. */
int ep_pool_episode_stats(ep_pool* pool, float* fin_return, int32_t* fin_length, int32_t* ep_count,
                          int32_t mem);

/* Stream-ordered variant for device consumers (the RCCL all-gather of the
 * multi-GPU bench): a snapshot kernel is enqueued behind the pool's pending
 * launches, writes into the given DEVICE buffers, and `consumer_stream`
 * (a hipStream_t) is made to wait for it. No host synchronisation. */
int ep_pool_episode_stats_async(ep_pool* pool, float* fin_return, int32_t* fin_length,
                                int32_t* ep_count, void* consumer_stream);

/* Snapshot / restore the whole device-resident env state (new: SURVEY.md §8f-4).
 * `buf` holds ep_pool_checkpoint_bytes() bytes of host memory. */
int64_t ep_pool_checkpoint_bytes(ep_pool* pool);
int ep_pool_checkpoint_save(ep_pool* pool, void* buf);
int ep_pool_checkpoint_restore(ep_pool* pool, const void* buf);

/* Introspection. */
int32_t ep_pool_num_envs(ep_pool* pool);
int32_t ep_pool_batch_size(ep_pool* pool);
int32_t ep_pool_ring_depth(ep_pool* pool);
int32_t ep_pool_is_sync(ep_pool* pool);
void* ep_pool_stream(ep_pool* pool); /* hipStream_t the step kernels run on */
/* Name of the kernel the last ep_pool_rollout launched ("" before the first one): which of the
 * rollout kernels runs depends on state_num, num_envs % 64 and the alignment of list_action. */
const char* ep_pool_last_rollout_kernel(ep_pool* pool);
/* Average device time in microseconds of the step / rollout kernels launched
 * since the last call with reset != 0 (HIP events on the pool's stream;
 * recording is enabled by ep_pool_set_timing). */
int ep_pool_set_timing(ep_pool* pool, int32_t enabled);
int ep_pool_kernel_time_us(ep_pool* pool, double* total_us, int64_t* launches, int32_t reset);

/* gymnasium's `terminated = done & ~truncated` (gymnasium_envpool.py:73-88, gym_envpool.py:74-95 new
 * step API) for k batches in ONE launch on `stream`: out[i][j] = done[i][j] & ~trunc[i][j] over n[i]
 * bytes of device memory each (the `done` / `trunc` keys of k batches, e.g. one per shard pool of a
 * front-end). `device`: the arrays' device (made current), or -1 for the current one. */
int ep_terminated_many(int32_t k, const void* const* done, const void* const* trunc, void* const* out,
                       const int64_t* n, int32_t device, void* stream);
/* hipStreamSynchronize(stream) on `device` (-1: the current one), for front-ends that hold a raw
 * stream handle and must see the producer of the action arrays finished before a blocking
 * ep_pool_step (which reads them from a resident kernel, outside any stream's order). No
 * counterpart in the reference. */
int ep_stream_synchronize(int32_t device, void* stream);
int32_t ep_device_count(void);
const char* ep_last_error(void); /* thread-local text of the last failure */
const char* ep_version(void);

/* ---- YGOPro observation-encode tail --------------------------------------- */
/* declared in envpool_amd_ygo.h */

#ifdef __cplusplus
}
#endif
#endif /* ENVPOOL_AMD_H_ */
