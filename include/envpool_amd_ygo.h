/*
 * envpool_amd_ygo.h — C ABI of the YGOPro observation-encode tail on MI355X.
 *
 * What moves to the GPU (BASELINE.json north_star): everything YGOProEnv::WriteState
 * (envpool2/ygopro/ygopro.h:1828-1893) does after the duel engine has stopped at a
 * decision — obs:cards_ (:1508-1603), obs:global_ (:1605-1620), obs:actions_
 * (:1674-1765, :1821-1826), the obs:h_actions_ ring copy-out (:1881-1892), the scalar
 * info keys and the common StateBuffer fields (envpool2/core/env.h:196-222).
 * The Lua/SQLite duel engine (ygopro-core) stays on host threads and is NOT part of
 * this library: its output reaches the GPU as one fixed-stride POD record per env
 * (layout below), written by the host into pinned memory.
 *
 * The reference has no struct at this seam (WriteState reads engine state through
 * query_field_card / query_field_count and the options_ strings), so the record is a new
 * contract; `ep_ygo_emit_record` is the host-side adapter that builds it from exactly the
 * reference-form data (raw query buffers, option strings, revealed_ strings), and the
 * parity tests prove oracle(reference form) == GPU(emit(reference form)).
 */
#ifndef ENVPOOL_AMD_YGO_H_
#define ENVPOOL_AMD_YGO_H_

#include <stdint.h>

#include "envpool_amd.h"
#include "envpool_amd_ygo_consts.h" /* the ygopro-core values this path assumes + ep_ygo_check_constants */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- config (ygopro.h:1179-1186 defaults) --------------------------------- */
typedef struct {
  int32_t max_cards;         /* per player; obs:cards_ has 2*max_cards rows (default 75) */
  int32_t max_options;       /* default 16 */
  int32_t n_history_actions; /* default 16 */
  int32_t max_multi_select;  /* default 5; n_action_feats = 9 + 2*max_multi_select (:1189) */
  int32_t max_episode_steps; /* common config (env_spec.h:30), for `trunc` */
  int32_t device_id;
  int32_t ring_depth;        /* in-flight batches of the submit/collect pipeline (0 => 4) */
  int32_t max_batch;         /* largest n of one featurize/submit call */
} ep_ygo_config;

#define EP_YGO_CARD_FEATS 39 /* ygopro.h:1191 */
#define EP_YGO_MAX_SPECS 8   /* >= max_multi_select */

/* ---- the per-env record ---------------------------------------------------
 * record := ygo_header (64 B)
 *           | history ring of the to_play player, n_history_actions*n_action_feats bytes,
 *             raw ring order (ygopro.h:1304-1314), padded to 16 B
 *           | ygo_card[n_cards], n_cards = sum(loc_count of the listed locations) <= 2*max_cards
 *           | min(num_options, max_options) x { ygo_option_head, uint32_t spec_ref[max_multi_select] }
 * so the used part of a record is one contiguous prefix (about 1.5 KB in a mid-game state): the
 * device fetches that prefix only - from HBM or, in the zero-copy mode, over PCIe. Records sit at
 * a fixed stride, the maximum size padded to a multiple of 256 B: ep_ygo_record_stride()
 * (3328 B at the default config). */
typedef struct {
  uint8_t to_play;      /* to_play_ (:1281) */
  uint8_t turn_player;  /* tp_ (:1275) */
  uint8_t msg;          /* msg_ (:1279), raw MSG_SELECT_* / MSG_ANNOUNCE_ATTRIB value */
  uint8_t num_options;  /* options_.size() BEFORE the max_options truncation (:1849-1851), capped at 255 */
  uint8_t is_selfplay;  /* play_mode_ == kSelfPlay (:1834) */
  uint8_t revealed_any; /* revealed_.size() != 0 (:1522-1525) */
  uint8_t ha_p;         /* ring pointer of to_play_ (:1883) */
  uint8_t done;         /* Env::IsDone() (env.h:199) */
  int32_t env_id;       /* env.h:208 */
  int32_t elapsed_step; /* env.h:209 */
  int32_t lp[2];        /* lp_ by ABSOLUTE player (:1272) */
  int32_t current_phase;/* current_phase_ (:1276), raw PHASE_* bit */
  int32_t win_reason;   /* value passed to WriteState (:1493 / :1422) */
  float reward;
  uint64_t revealed_hand_mask; /* bit k set <=> "oh<k+1>" is in revealed_ (:1541-1545) */
  /* per pi (0 = to_play, 1 = opponent) and location in the order DECK, HAND, MZONE, SZONE,
   * GRAVE, REMOVED, EXTRA (:1514-1519): number of rows that location contributes — the
   * query_field_count for a hidden opponent location (:1527), else the number of ygo_card
   * entries (overlay materials included, :2106-2115). */
  uint8_t loc_count[2][7];
  uint8_t pad[2];
} ygo_header; /* 64 bytes */

/* One card of a listed location, in get_cards_in_location order (:2060-2137): overlay
 * materials come before their host monster and carry location|0x80, position = index. */
typedef struct {
  uint32_t code;      /* CardCode */
  uint8_t controller; /* ABSOLUTE player id */
  uint8_t location;   /* LOCATION_* | LOCATION_OVERLAY(0x80) */
  uint8_t sequence;
  uint8_t position;   /* POS_* ; overlay index for materials */
  uint8_t level;      /* low byte of the queried level (0 if not > 0)  (:2090-2093) */
  uint8_t rank;       /* low byte of the queried rank                   (:2094-2097) */
  uint8_t link;       /* low byte of the queried link rating            (:2126-2130) */
  uint8_t is_material;/* 1 for overlay materials: every dynamic field below is absent (:2109-2114) */
  /* low 16 bits of the queried attack / defense: float_transform (:704-710) takes x % 65536 and
   * keeps two bytes, so nothing above bit 15 can reach an observation */
  uint16_t attack;    /* queried attack (:2098) */
  uint16_t defense;   /* queried defense, or link_marker when that is > 0 (:2099, :2131-2133) */
} ygo_card; /* 16 bytes */

/* One option after the host has applied the per-message grammar of _set_obs_action
 * (:1674-1765): the nine spec-independent feature bytes (columns 2*mms .. 2*mms+8) and the
 * referenced specs as keys to be resolved against the card rows on the device. */
typedef struct {
  uint8_t feats[9];  /* msg, act(+offset), yes/no, phase, cancel/finish, position, option, place, attrib */
  uint8_t n_specs;   /* 0..max_multi_select */
  uint8_t pad[2];
} ygo_option_head; /* 12 bytes, followed by uint32_t spec_ref[max_multi_select], one per referenced
   * spec (spec_to_ls, :352-400: an 'o' prefix = opponent relative to to_play, a location letter,
   * a 1-based sequence, an optional overlay letter). With loc = LOCATION_* & 0x7f, seq 0-based
   * and suf = letter - 'a' + 1 (0 without one):
   *   plain spec, seq < max_cards:              (opp*7 + location_id-1) * max_cards + seq
   *   overlay spec on MZONE, seq < 8, suf <= 8: 14*max_cards + (opp*8 + seq)*8 + suf-1
   *   any other spec:                           0x80000000 | opp | loc<<1 | seq<<8 | suf<<16
   *   not a spec at all:                        0xfffffffe (matches no row)
   * The first two are the row's slot in the device's per-env spec table (the reference's
   * spec2index, :1549); ep_ygo_emit_record computes them. */

/* ---- static card table (replaces init_module's cards_/card_ids_, ygopro.h:1030-1064) -- */
typedef struct {
  uint32_t code;
  uint32_t type;      /* TYPE_* bits */
  uint32_t level;     /* already masked & 0xff by db_query_card (:849) */
  uint32_t attribute; /* ATTRIBUTE_* bit */
  uint32_t race;      /* RACE_* bit */
  int32_t attack;     /* static atk: what an overlay material shows (c_get_card copy, :2109) */
  int32_t defense;    /* static def; 0 for LINK monsters (:855-858) */
  uint16_t card_id;   /* code-list index (:1034-1041) */
  uint16_t pad;
} ygo_card_data; /* 32 bytes */

typedef struct ep_ygo_encoder ep_ygo_encoder;

int ep_ygo_encoder_create(const ep_ygo_config* cfg, ep_ygo_encoder** out);
int ep_ygo_encoder_destroy(ep_ygo_encoder* enc);

/* Uploads the static table as an open-addressing hash keyed by code (device resident). */
int ep_ygo_set_card_table(ep_ygo_encoder* enc, const ygo_card_data* cards, int32_t n);

int64_t ep_ygo_record_stride(const ep_ygo_config* cfg);
int32_t ep_ygo_action_feats(const ep_ygo_config* cfg);

/* Replaces: WriteState (ygopro.h:1828-1893) + Env::Allocate (env.h:196-222) for a batch of
 * `n` records. `records` is host (pageable or pinned) or device memory per `mem_in`; the
 * result is described in `out` as in ep_pool_recv, keys in the reference's `_state_keys`
 * order (8 common keys, then obs:cards_, obs:global_, obs:actions_, obs:h_actions_,
 * info:num_options, info:to_play, info:is_selfplay, info:win_reason) followed by two
 * additive keys: aux:h_card_ids uint16 [n, max_options, max_multi_select] (the
 * back-derivation of :1863-1879, so the host ring update needs no GPU read-back of obs) and
 * aux:action_mask bool [n, max_options], the legal-action mask the reference leaves implicit
 * (row i of obs:actions_ is an option iff i < info:num_options).
 * Blocking. Submit and collect as ONE operation: refused with EP_ERR_INVALID while batches of
 * ep_ygo_submit / ep_ygo_commit_slot are in flight (it returns the batch it was given, not the
 * oldest one). Returns EP_ERR_ACTION_CHECK when a record tripped a condition on which the
 * reference throws (unknown card code / attribute / race / position / phase / message,
 * spec not on the field, more rows than max_cards); the batch is still returned. */
int ep_ygo_featurize(ep_ygo_encoder* enc, const void* records, int32_t n, int32_t mem_in,
                     int32_t mem_out, ep_batch* out);

/* Pipelined form: submit copies the host records into the next pinned staging slot,
 * enqueues the H2D copy on a side stream and the kernel on the compute stream, and returns;
 * collect hands out the oldest submitted batch. Up to ring_depth batches in flight, so the
 * copy of batch k+1 overlaps the kernel of batch k (north_star: "pinned hipMemcpyAsync
 * overlap of host duel results on a side stream"). */
int ep_ygo_submit(ep_ygo_encoder* enc, const void* host_records, int32_t n);
int ep_ygo_collect(ep_ygo_encoder* enc, int32_t mem_out, ep_batch* out);
/* ep_ygo_submit for records that already sit in device memory (n x ep_ygo_record_stride bytes):
 * the kernel reads the caller's buffer, which must stay untouched until the batch is collected.
 * Non-blocking like ep_ygo_submit; results come back through ep_ygo_collect in FIFO order with the
 * host-record batches. (ep_ygo_featurize with EP_MEM_DEVICE input is this + ep_ygo_collect.) */
int ep_ygo_submit_device(ep_ygo_encoder* enc, const void* device_records, int32_t n);
/* Direct access to the next pinned staging slot, so host duel threads can emit records in
 * place (no memcpy): acquire -> fill n records -> commit. */
int ep_ygo_acquire_slot(ep_ygo_encoder* enc, void** pinned_records);
int ep_ygo_commit_slot(ep_ygo_encoder* enc, int32_t n);
/* Arrival-ordered record ring: BASELINE config 5 (async mode, batch_size < num_envs). Duel threads
 * finish env-steps in no particular order; each takes the next row of the batch being filled,
 * emits its record IN PLACE into the pinned slot and reports it done; the batch is encoded as soon
 * as its batch_size-th record has landed, and batches are received in order. The counterparts:
 *   ep_ygo_async_begin   puts the encoder into this mode (no batch may be in flight); batch_size
 *                        <= max_batch; up to ring_depth - 1 batches can be filled ahead of the
 *                        consumer
 *   ep_ygo_alloc_record  StateBufferQueue::Allocate + StateBuffer::Allocate
 *                        (state_buffer_queue.h:114-128, state_buffer.h:79-112): one atomic add
 *                        hands out (batch, row) in ARRIVAL order; *record points at that row of the
 *                        pinned slot (ep_ygo_record_stride bytes, to be filled by
 *                        ep_ygo_emit_record), *ticket names it. Any number of threads. Waits while
 *                        the ring is full (the reference throws out_of_range there, state_buffer.h:111)
 *   ep_ygo_record_done   StateBuffer::Done (state_buffer.h:124-129): the call that completes a
 *                        batch enqueues its encode kernel (zero-copy or staged, ep_ygo_set_zero_copy)
 *   ep_ygo_recv          StateBufferQueue::Wait (state_buffer_queue.h:139-150): blocks until the
 *                        next batch in order is complete and encoded (ep_ygo_set_recv_timeout as for
 *                        pools), then describes it like ep_ygo_collect; single consumer. Device views
 *                        stay valid until the next ep_ygo_recv. Rows are in arrival order: compare
 *                        with the reference as sets keyed by info:env_id. */
int ep_ygo_async_begin(ep_ygo_encoder* enc, int32_t batch_size);
/* Shutdown: producers asleep on a full ring and a consumer asleep on an incomplete batch return
 * EP_ERR_WOULD_BLOCK (the reference has no counterpart: its destructor joins workers that cannot
 * be blocked there, because a full queue throws, state_buffer.h:111). */
int ep_ygo_async_end(ep_ygo_encoder* enc);
int ep_ygo_alloc_record(ep_ygo_encoder* enc, void** record, uint64_t* ticket);
int ep_ygo_record_done(ep_ygo_encoder* enc, uint64_t ticket);
int ep_ygo_recv(ep_ygo_encoder* enc, int32_t mem_out, ep_batch* out);
int ep_ygo_set_recv_timeout(ep_ygo_encoder* enc, int64_t milliseconds);
/* Packed form of the arrival-ordered ring (after ep_ygo_async_begin): records lie back to back in
 * the pinned slot in the order their producers finished, with a per-row offset table in front, and
 * the batch reaches the device as ONE copy-engine transfer of exactly the bytes in use (about 1.6 KB
 * per mid-game record instead of the 3328-byte stride; large PCIe reads instead of the kernel's own
 * 64-byte ones).
 *   ep_ygo_set_packed       switches the form (no record may be outstanding)
 *   ep_ygo_emit_arrival     one call per finished env-step from any thread: ep_ygo_emit_record into
 *                           a thread-local buffer, then row + bytes are taken in arrival order
 *                           (two atomic adds), the record is copied in and reported done
 *   ep_ygo_put_arrival      the same for a record that has been emitted already
 *   ep_ygo_record_used_bytes  the length of the used prefix of a record (16-byte multiple)
 * Batches are received with ep_ygo_recv. Results are identical in every form. */
int ep_ygo_set_packed(ep_ygo_encoder* enc, int32_t enabled);
int ep_ygo_put_arrival(ep_ygo_encoder* enc, const void* record);
/* Arrival-ordered mode: info:env_id of the batch ep_ygo_recv handed out last, row by row, in HOST
 * memory (the producers noted them as their rows arrived), valid until the next ep_ygo_recv. For a
 * consumer that keeps the batch on the device (EP_MEM_DEVICE) and only needs to know whom to send the
 * next actions to. No counterpart in the reference (its batches are host memory). */
int ep_ygo_last_env_ids(ep_ygo_encoder* enc, const int32_t** ids, int32_t* n);
int64_t ep_ygo_record_used_bytes(const ep_ygo_config* cfg, const void* record);
/* The packed form for ONE producer (submit / collect mode): the caller fills the slot it got from
 * ep_ygo_acquire_slot with n records back to back (record i at byte offsets[i], 16-byte aligned,
 * used_bytes in all) and commits; or lets ep_ygo_emit_records_packed (below, with the raw-state
 * type) emit n states that way. Collected with ep_ygo_collect like any other batch. With nothing
 * else in flight the kernel reads the slot in place (one launch); a batch committed behind another
 * one is moved by the copy engine under that one's kernel. */
int ep_ygo_commit_slot_packed(ep_ygo_encoder* enc, int32_t n, const uint32_t* offsets, int64_t used_bytes);
/* ep_ygo_commit_slot in parts, for producers that finish a batch's rows in stages: rows
 * [row0, row0 + n_rows) of the slot from ep_ygo_acquire_slot are complete (fixed stride, row i at
 * i * stride) and may be encoded now, while the other rows of the batch of n_total are still being
 * written. Parts may come from any thread, in any order, each row once; the call that brings the
 * parts to n_total rows sets *completed = 1 and the batch is then collected like any other. The
 * StateBuffer of the reference fills row by row in the same way (state_buffer.h:79-129); there the
 * rows are final observations, here each part's encode runs under the emission of the rest. */
int ep_ygo_commit_slot_rows(ep_ygo_encoder* enc, int32_t row0, int32_t n_rows, int32_t n_total, int32_t* completed);
/* Gives up a batch some of whose parts went out through ep_ygo_commit_slot_rows (one part's launch
 * failed, so the batch can never complete): waits for the parts that were launched and forgets them;
 * the next ep_ygo_acquire_slot / commit starts a clean batch in the same slot. Call it once no other
 * thread is still committing parts of that batch. */
int ep_ygo_abort_slot(ep_ygo_encoder* enc);
/* Counters of the arrival-ordered ring since ep_ygo_async_begin: the time producer threads spent
 * waiting for a free slot (the only place where the GPU side can hold the duel threads up), the
 * records handed out and the batches received. Any pointer may be NULL. */
int ep_ygo_ring_stats(ep_ygo_encoder* enc, double* producer_wait_seconds, uint64_t* records, uint64_t* batches_received);
/* How a committed slot reaches the kernel. enabled != 0 (the default; EP_YGO_ZERO_COPY=0 in the
 * environment changes the default): the kernel reads the USED part of every record straight from
 * the pinned slot over PCIe - no staging copy, about half the PCIe bytes of a fixed-stride copy;
 * consecutive batches alternate between two streams. enabled == 0: the slot is first copied to
 * device memory on a side stream (whole records), under the kernel of the batch before. Results
 * are identical. Call with no batch in flight. No counterpart in the reference (its records never
 * leave the host). */
int ep_ygo_set_zero_copy(ep_ygo_encoder* enc, int32_t enabled);
int ep_ygo_sync(ep_ygo_encoder* enc);
void* ep_ygo_stream(ep_ygo_encoder* enc);
int ep_ygo_set_timing(ep_ygo_encoder* enc, int32_t enabled);
int ep_ygo_kernel_time_us(ep_ygo_encoder* enc, double* total_us, int64_t* launches, int32_t reset);

/* ---- host-side adapter from the reference's own data forms ----------------
 * Everything WriteState reads from YGOProEnv, as plain pointers. */
typedef struct {
  int32_t to_play, turn_player, msg, current_phase, is_selfplay, win_reason;
  int32_t lp[2];
  float reward;
  int32_t env_id, elapsed_step, done;
  int32_t n_options;
  const char* const* options;  /* options_ (:1280) */
  int32_t n_revealed;
  const char* const* revealed; /* revealed_ (:1316) */
  /* by ABSOLUTE player and location order DECK..EXTRA: query_field_count (:1527) and the raw
   * query_field_card buffer with the flags of :2061-2064 (may be NULL when count-only) */
  int32_t field_count[2][7];
  const uint8_t* field_buf[2][7];
  int32_t field_len[2][7];
  const uint8_t* history[2]; /* history_actions_0_/1_ (:1307,:1312), n_history*n_feats bytes each */
  int32_t ha_p[2];           /* ha_p_0_/ha_p_1_ */
} ygo_raw_state;

/* Parses the raw query buffers (get_cards_in_location, :2060-2137) and the option strings
 * (grammar of _set_obs_action, :1674-1765) into one record. Host only, no GPU. Returns
 * EP_ERR_INVALID on input the reference would throw on while parsing; record_out then holds the
 * state's EMPTY observation (what emitting it with n_options = 0 gives: the early-out of WriteState),
 * so a batch that contains it can still be encoded and keeps its row. */
int ep_ygo_emit_record(const ep_ygo_config* cfg, const ygo_raw_state* in, void* record_out);

/* Packed arrival-ordered ring, one call per finished env-step (see ep_ygo_set_packed above). */
int ep_ygo_emit_arrival(ep_ygo_encoder* enc, const ygo_raw_state* in);

/* acquire + n packed emits + commit for a single producer (see ep_ygo_commit_slot_packed). */
int ep_ygo_emit_records_packed(ep_ygo_encoder* enc, const ygo_raw_state* in, int32_t n, int32_t* status);

/* `n` consecutive ep_ygo_emit_record calls; record i goes to records_out + i*stride.
 * status[i] (may be NULL) receives each call's code; returns the first non-OK code. */
int ep_ygo_emit_records(const ep_ygo_config* cfg, const ygo_raw_state* in, int32_t n, void* records_out,
                        int32_t* status);

/* Host ring update of YGOProEnv::update_history_actions (:1439-1450) with the card ids of
 * the chosen option (aux:h_card_ids row): moves *ha_p and encodes `option` into the ring.
 * Untouched columns keep their old bytes (the reference never clears a reused row). */
int ep_ygo_history_push(const ep_ygo_config* cfg, uint8_t* ring, int32_t* ha_p, int32_t msg,
                        const char* option, const uint16_t* card_ids, int32_t n_card_ids);

/* ---- a YGOPro env pool over a caller-supplied duel engine ---------------------------------
 * AsyncEnvPool<YGOProEnv> (envpool2/core/async_envpool.h:40-190) with the duel engine behind two
 * callbacks. The engine (ygopro-core, Lua card scripts, the SQLite card database) runs on the host
 * and is not part of this library; the callbacks run it for one env up to the point where the
 * reference calls WriteState (YGOProEnv::Reset, ygopro.h:1374-1423; Step, :1452-1494) and describe
 * what WriteState would read. The pool does the rest: action queue and worker threads
 * (action_buffer_queue.h:35-85, async_envpool.h:116-129), auto-reset one step after `done`
 * (:125), current_step_ / info:env_id / elapsed_step (env.h:181-189, 196-222), record emission on
 * the worker thread, the pinned ring, the encode kernel, and batches handed out in order.
 *   callbacks   are called from worker threads (and from inside ep_ygo_pool_recv of a sync pool, whose
 *               caller helps instead of sleeping), never twice at once for the same env; return 0 or
 *               non-zero for failure (the env's row is then an empty observation, the env counts as
 *               done and the next recv reports EP_ERR_INVALID beside the batch). Pointers placed in
 *               *state must stay valid until the same env's next callback returns. env_id,
 *               elapsed_step and the history of `done` are the pool's: what the engine writes into
 *               state->env_id / elapsed_step is ignored, state->done is IsDone() (:1428).
 *   send        SendImpl (:57-80): host arrays env_id[n], action[n] (the reference's `action` key,
 *               an option index). An env whose previous step is still running on a worker cannot
 *               be sent again (EP_ERR_INVALID), and an env may appear once per send
 *               (EP_ERR_DUP_ENV): its duel would otherwise run on two threads at once.
 *   reset       Reset (:176-189): forced reset of the listed envs.
 *   recv        Recv (:162-174): the next batch, arrays in _state_keys order + aux:h_card_ids as for
 *               ep_ygo_collect. batch_size < num_envs: rows in arrival order; batch_size ==
 *               num_envs (sync): row i is position i of the sent env_id array (:69), and a sync pool
 *               takes one send / reset per recv.
 * num_threads <= 0: min(batch_size, CPUs this process may use) (:113-115), one fewer for a sync pool,
 * whose receiving thread steps envs too; thread_affinity_offset < 0: none. */
typedef struct {
  void* ctx;
  int (*reset)(void* ctx, int32_t env_id, ygo_raw_state* state);
  int (*step)(void* ctx, int32_t env_id, int32_t action, ygo_raw_state* state);
} ep_ygo_engine;
typedef struct ep_ygo_pool ep_ygo_pool;
int ep_ygo_pool_create(const ep_ygo_config* cfg, const ep_ygo_engine* engine, int32_t num_envs, int32_t batch_size,
                       int32_t num_threads, int32_t thread_affinity_offset, const ygo_card_data* cards, int32_t n_cards,
                       ep_ygo_pool** out);
int ep_ygo_pool_destroy(ep_ygo_pool* pool);
int ep_ygo_pool_send(ep_ygo_pool* pool, int32_t n, const int32_t* env_id, const int32_t* action);
int ep_ygo_pool_reset(ep_ygo_pool* pool, int32_t n, const int32_t* env_id);
int ep_ygo_pool_recv(ep_ygo_pool* pool, int32_t mem_out, ep_batch* out);
int ep_ygo_pool_set_recv_timeout(ep_ygo_pool* pool, int64_t milliseconds); /* as ep_pool_set_recv_timeout */
/* One pool per GPU, env ids sharded in contiguous blocks (BASELINE config 5: 8192 envs over 8 GPUs =
 * 1024 per pool): this pool owns the global ids [offset, offset + num_envs). send / reset take global
 * ids, the callbacks get global ids, info:env_id is global. Before the first send / reset. */
int ep_ygo_pool_set_env_id_offset(ep_ygo_pool* pool, int32_t env_id_offset);
/* info:env_id of the batch ep_ygo_pool_recv handed out last (global ids), in host memory, valid until
 * the next ep_ygo_pool_recv: what the next ep_ygo_pool_send of an async actor loop takes, without
 * reading the batch's own info:env_id back from the device. No counterpart in the reference. */
int ep_ygo_pool_last_env_ids(ep_ygo_pool* pool, const int32_t** ids, int32_t* n);
ep_ygo_encoder* ep_ygo_pool_encoder(ep_ygo_pool* pool); /* for ep_ygo_set_timing / ep_ygo_ring_stats */
int32_t ep_ygo_pool_num_threads(ep_ygo_pool* pool);

#ifdef __cplusplus
}
#endif
#endif /* ENVPOOL_AMD_YGO_H_ */
