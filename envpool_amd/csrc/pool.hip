// pool.hip — device-resident replacement of envpool2/core's AsyncEnvPool /
// ActionBufferQueue / StateBufferQueue for Dummy-v0, behind the C ABI of
// include/envpool_amd.h.
//
// The reference's moving parts and what stands in for them here:
//   ActionBufferQueue (action_buffer_queue.h:35-85)   -> the HIP stream: a send IS a kernel
//       launch, FIFO by construction; a pinned staging ring + copy stream carries host actions.
//   worker threads (async_envpool.h:116-129)          -> one GPU thread per env-step.
//   StateBufferQueue ring (state_buffer_queue.h:32-151)-> `ring_depth` slots in ONE HBM
//       allocation; the host mirrors alloc_count_/done_ptr_ and per-buffer offsets, so row
//       assignment costs no device atomics (every slice yields exactly one shared row).
//   background zero-fill thread (:82-95)              -> gone; kernels write whole rows.
//   semaphores / StateBuffer::Wait (state_buffer.h:135-157) -> one hipEvent per slot.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#if defined(__linux__)
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#endif
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "dummy_kernels.hpp"
#include "dummy_mp_kernels.hpp"
#include "dummy_rollout_wave.hpp"
#include "dummy_server.hpp"

#include "common.hpp"

namespace {
using epc::align_up;
using epc::fail;
using epc::HostBlockOwner;
using epc::HostBlocks;

// ep_host_blocks: pinned, portable, device-mapped blocks that a caller hands to pools as result
// targets. Every block out of the set is reference-counted: the taker holds it once, a pool that
// was given it as a target holds it until its batch is done with it, and it goes back to the free
// list (or is freed, after ep_host_blocks_destroy) when the last hold is dropped - so a block is
// never unmapped while a kernel may still write it.
struct HostBlockSet {
  std::mutex mu;
  std::vector<void*> free_list;
  std::unordered_map<void*, int> holds;
  size_t bytes = 0;
  bool alive = true;
  bool held(void* blk) {
    std::lock_guard<std::mutex> lk(mu);
    return holds.count(blk) != 0;
  }
  void hold(void* blk) {
    std::lock_guard<std::mutex> lk(mu);
    ++holds[blk];
  }
  void drop(void* blk) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = holds.find(blk);
    if (it == holds.end() || --it->second > 0) return;
    holds.erase(it);
    if (alive && free_list.size() < 8) free_list.push_back(blk);
    else (void)hipHostFree(blk);
  }
};

struct SlotHost {
  uint64_t alloc_count = 0;  // StateBuffer::alloc_count_
  uint32_t shared_off = 0;   // low half of StateBuffer::offsets_
  std::vector<std::pair<int64_t, int64_t>> written;  // shared rows written, [a,b)
  bool has_work = false;
  uint64_t launch_id = 0;  // id of the last launch that wrote this slot
  int mp_word = -1;        // multi-player pools: index of the player-total word of this batch
  void* host_blk = nullptr;  // direct host mode: the kernels wrote this batch straight into a pinned block, not the ring
  bool term_ok = false;      // every row so far came from a step kernel that also wrote d_term (ep_pool_terminated_of)
  // ep_pool_host_target: this batch is delivered into the caller's arrays (direct host mode: the
  // kernels write them, host_blk = ext_host[0] as the "direct" mark; else recv copies into them)
  bool ext = false;
  int64_t ext_row0 = 0;  // row of the caller's arrays that row 0 of this batch goes to
  char* ext_host[11]{};
  char* ext_dev[11]{};
  std::shared_ptr<HostBlockSet> ext_home;  // the pool's hold on the block, dropped when the batch is done with it
  void* ext_block = nullptr;
};

}  // namespace

struct ep_host_blocks {
  std::shared_ptr<HostBlockSet> set;
};

struct ep_pool {
  ep_dummy_config cfg{};
  int N = 0, B = 0, P = 0, S = 0, Q = 0;
  bool is_sync = false;
  bool direct_host = true;  // EP_DIRECT_HOST=0: always stage host batches through the copy engine
  bool host_const = true;   // EP_HOST_CONST=0: the kernels write the all-zero arrays of a host-memory batch too
  int direct_min_rows = 1;  // EP_DIRECT_MIN_ROWS (4096 until round 2: the staged copies cost a small pool 24 us per step)
  int device = 0;
  hipStream_t stream = nullptr, copy_stream = nullptr;
  hipStream_t own_stream = nullptr;  // the stream the pool created; `stream` is another one while adopted (ep_pool_set_stream)
  // device state
  epk::EnvRec* d_env = nullptr;
  uint8_t* d_term = nullptr;            // [Q][term_stride]: terminated = done & ~trunc of the rows of each ring slot (step kernels)
  size_t term_stride = 0;
  std::vector<uint64_t> slot_term_seq;  // [Q]: seq + 1 of the received batch whose d_term rows are valid, else 0
  epk::StatPtrs stats{};
  char* d_ring = nullptr;
  size_t slot_stride = 0;
  size_t key_off[11]{};
  size_t key_rowbytes[11]{};
  bool key_is_player[11]{};
  // multi-player scratch (dummy_mp_kernels.hpp)
  epk::MpScratch mp_sc{};              // histogram counters [N], look-back granules, ticket
  uint32_t mp_ticket_total = 0;        // tiles launched so far (the ticket counter never goes back)
  uint32_t mp_barrier_total = 0;       // workgroups of folded-histogram launches so far (that counter never goes back either)
  bool mp_fold = false;                // EP_MP_FOLD=1: the histogram as a phase of the fused kernel (one launch per send)
  uint64_t mp_count_epoch = 0;         // sends so far: low 32 bits tag the row histogram of a send
  uint32_t* d_slot_players = nullptr;  // [Q] player_offset of each slot (device running total)
  uint32_t* h_slot_players = nullptr;  // [EQ] host-mapped: final player_offset per batch generation
  uint32_t* d_host_players = nullptr;  // device alias of h_slot_players
  // error words (host-mapped), one per batch GENERATION: batch bi reports into word bi % EQ with
  // EQ = 2Q, so a batch handed out by ep_pool_recv_on_stream (whose word is only looked at by the
  // next ep_pool_sync) keeps its word while its ring slot is already being refilled
  int EQ = 0;
  epk::ErrWords* h_err = nullptr;  // [EQ]
  epk::ErrWords* d_err = nullptr;  // device alias of h_err
  std::vector<uint64_t> deferred_lid;  // [EQ] last launch that may write a deferred word
  hipEvent_t producer_ev = nullptr;    // ep_pool_step_on_stream: the caller's stream as it was when the step was asked for
  uint32_t sticky_err = 0;  // deferred words folded in when their index came round again: bit 0 action check, 1 id range, 2 duplicate env
  // "written exactly once" (SURVEY.md §5): launch stamps per env
  uint32_t* d_stamp = nullptr;    // [N] device-memory sends: checked by the kernels
  std::vector<uint32_t> h_stamp;  // [N] host-memory sends: checked (and the send split) on the host
  uint32_t h_stamp_seq = 0;
  std::vector<int> seg_cuts;      // scratch of the host check
  // action staging for host sends
  int A = 0;
  size_t stage_bytes = 0;
  char* h_stage = nullptr;
  char* d_stage = nullptr;
  std::vector<hipEvent_t> stage_free;  // recorded on `stream` after the consuming kernel
  std::vector<hipEvent_t> stage_copied;
  uint64_t stage_idx = 0;
  // host mirror of the queue counters
  uint64_t alloc_count = 0, done_ptr = 0;  // StateBufferQueue::alloc_count_/done_ptr_ (in rows / batches)
  int64_t stepping_env_num = 0;            // async_envpool.h:49
  uint64_t recv_seq = 0;
  struct InFlight {
    uint64_t first;     // launch id
    hipEvent_t second;  // completion event
    bool owned;         // false: the event belongs to the `timed` list (it doubles as the timing stop event)
  };
  std::deque<InFlight> inflight;
  std::vector<hipEvent_t> done_ev_pool;
  uint64_t launch_seq = 0;       // launches enqueued on `stream`
  uint64_t completed_launch = 0; // highest launch id known to have finished (stream is in-order)
  hipEvent_t stats_ev = nullptr, consumer_ev = nullptr, recv_ev = nullptr;
  std::vector<uint8_t> deferred_err;  // [EQ] batches handed out by ep_pool_recv_on_stream, error word not looked at yet
  std::vector<SlotHost> slots;
  std::mutex mu;
  std::condition_variable cv;    // signalled when a send/reset has added rows (StateBuffer::Done -> sem_.signal)
  int64_t recv_timeout_ms = -1;  // ep_pool_set_recv_timeout: < 0 wait like the reference, 0 never wait, > 0 bounded
  std::shared_ptr<HostBlocks> blocks;
  // ep_pool_host_target: where the next batch started by a send / reset is to be delivered
  bool target_pending = false;
  int64_t target_row0 = 0;
  char* target_host[11]{};
  char* target_dev[11]{};
  std::shared_ptr<HostBlockSet> target_home;
  void* target_block = nullptr;
  const char* last_rollout_kernel = "";  // which rollout kernel the last ep_pool_rollout launched
  // work on `stream` that no launch id covers (memsets, copies, the stats snapshot) / anything on
  // copy_stream: ep_pool_sync only has to call hipStreamSynchronize (8-10 us even on an idle stream,
  // ROCm 7.2) when one of these is set or a launch is not yet known to have finished
  bool stream_extra = true, copy_dirty = true;
  // resident step server (ep_pool_step, dummy_server.hpp)
  hipStream_t server_stream = nullptr;
  epk::ServerCtl* h_ctl = nullptr;  // host-mapped doorbell / answer lines
  epk::ServerCtl* d_ctl = nullptr;
  unsigned long long* d_server_words = nullptr;  // device flag line + fan-in counter lines
  unsigned long long* bar_bell = nullptr;  // fine-grained DEVICE word the host writes through the PCIe BAR: the doorbell every
                                           // workgroup polls (null: no large BAR, or EP_SERVER_BAR=0: round 2's leader wake-up)
  bool server_running = false;
  uint32_t server_seq = 0;
  const void* server_arrays[3]{};
  bool spin_pause = false;  // EP_SERVER_PAUSE=1: a PAUSE in the host's wait loop (round 2 had one)
  bool server_two_level = false;  // EP_SERVER_JOIN=2: round 2's two-level join (one host word)
  int server_shards = 0;       // shard words the host waits for (min(workgroups, kServerShards))
  int step_mode = 0;           // ep_pool_step_begin .. ep_pool_step_end: 0 none, 1 the server has been rung, 2 send + recv
  bool step_warm = false;      // the step in flight found the server running (it was not started for it)
  uint64_t step_bi = 0;
  // idle watchdog: stops the server when no step has come for idle_limit_us (about 10 x the last
  // gap between two steps, 200 us .. 50 ms), so that a device-wide synchronisation of the caller,
  // which has to wait for the resident kernel to end, is held up for that long at most
  std::thread wd;
  std::condition_variable wd_cv;
  bool wd_started = false, wd_quit = false;
  std::chrono::steady_clock::time_point last_ring{};
  int64_t idle_limit_us = 200;
  bool counted_on_device = false;
  bool server_shared_ok = false;  // EP_SERVER_SHARED_DEVICE=1
  bool server_fresh = false;    // started, and no warm step served yet
  bool gap_tainted = true;      // the time since the last step holds an idle stop (or nothing yet): not a cadence to learn from
  uint32_t idle_streak = 0;     // idle stops in a row with no warm step between them
  uint32_t server_backoff = 0;  // steps that take the launch path before the server is tried again
  uint64_t server_launches = 0, server_idle_stops = 0, server_steps = 0;
  unsigned long long* d_server_trace = nullptr;  // EP_SERVER_TRACE=1: per workgroup 4 timestamps of its last step
  int server_wgs = 0;
  int cu_count = 0;
  bool use_server = true;  // EP_STEP_SERVER=0: ep_pool_step always takes the launch-per-step path
  bool poisoned = false;  // a launch failed half-way through a send: the host mirror of the queue is no longer trustworthy
  int rollout_variant = 2;  // EP_ROLLOUT_VARIANT: 2 = LDS-DMA D=4 (default), 0 = LDS-DMA D=6, 1 = register queue
  bool force_generic = false;  // EP_FORCE_GENERIC=1: use the generic kernels everywhere (tests)
  bool mp_skip_count = false;  // EP_MP_SKIP_COUNT=1: DIAGNOSTIC, wrong results: no histogram launch (what that launch and its boundary cost)
  bool mp_force_ticket = false;  // EP_MP_FORCE_TICKET=1: multi-player tile ids from the ticket counter even when the launch fits (tests)
  // kernel timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;
  std::vector<hipEvent_t> ev_pool;
  double timed_total_us = 0;
  int64_t timed_launches = 0;
};

namespace epk_mp {
int enqueue_mp(ep_pool* p, const int32_t* d_env_id, const int32_t* d_penv, const double* d_la, int n,
               int np_rows, bool force_reset, const std::vector<int>* cuts, bool host_checked,
               bool host_mode, const uint32_t* la_ok);
}

namespace {

// state-key index (in _state_keys order, 12 keys) -> slot key index (obs:dyn has no array)
const int kStateToSlot[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, -1, 9, 10};
const int kSlotDtype[11] = {EP_I32, EP_I32, EP_I32, EP_BOOL, EP_F32, EP_F32,
                            EP_I32, EP_BOOL, EP_I32, EP_BOOL, EP_I32};
const int kSlotElem[11] = {4, 4, 4, 1, 4, 4, 4, 1, 4, 1, 4};

epk::SlotPtrs slot_ptrs_at(ep_pool* p, char* b) {  // the key arrays of one batch laid out at `b`
  epk::SlotPtrs s;
  s.env_id = (int32_t*)(b + p->key_off[0]);
  s.penv = (int32_t*)(b + p->key_off[1]);
  s.elapsed = (int32_t*)(b + p->key_off[2]);
  s.done = (uint8_t*)(b + p->key_off[3]);
  s.reward = (float*)(b + p->key_off[4]);
  s.discount = (float*)(b + p->key_off[5]);
  s.step_type = (int32_t*)(b + p->key_off[6]);
  s.trunc = (uint8_t*)(b + p->key_off[7]);
  s.raw = (int32_t*)(b + p->key_off[8]);
  s.pdone = (uint8_t*)(b + p->key_off[9]);
  s.pid = (int32_t*)(b + p->key_off[10]);
  return s;
}
epk::SlotPtrs slot_ptrs_of(char* const* k) {  // the key arrays of one batch, each where the caller put it
  epk::SlotPtrs s;
  s.env_id = (int32_t*)k[0];
  s.penv = (int32_t*)k[1];
  s.elapsed = (int32_t*)k[2];
  s.done = (uint8_t*)k[3];
  s.reward = (float*)k[4];
  s.discount = (float*)k[5];
  s.step_type = (int32_t*)k[6];
  s.trunc = (uint8_t*)k[7];
  s.raw = (int32_t*)k[8];
  s.pdone = (uint8_t*)k[9];
  s.pid = (int32_t*)k[10];
  return s;
}
epk::SlotPtrs host_slot_ptrs(ep_pool* p, int slot) { return slot_ptrs_at(p, p->d_ring + (size_t)slot * p->slot_stride); }

epk::PoolConst pool_const(ep_pool* p) {
  epk::PoolConst pc;
  pc.N = p->N;
  pc.env_off = p->cfg.env_id_offset;
  pc.seed = p->cfg.seed;
  pc.max_episode_steps = p->cfg.max_episode_steps;
  pc.S = p->S;
  return pc;
}

hipEvent_t get_event(ep_pool* p) {
  if (!p->ev_pool.empty()) {
    hipEvent_t e = p->ev_pool.back();
    p->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

struct TimedScope {
  ep_pool* p;
  hipEvent_t a = nullptr, b = nullptr;
  explicit TimedScope(ep_pool* pool) : p(pool) {
    if (p->timing) {
      a = get_event(p);
      b = get_event(p);
      if (a && b) (void)hipEventRecord(a, p->stream);
    }
  }
  ~TimedScope() {
    if (a && b) {
      (void)hipEventRecord(b, p->stream);
      p->timed.emplace_back(a, b);
    }
  }
};

int drain_timed(ep_pool* p) {
  for (auto& ab : p->timed) {
    HIP_TRY(hipEventSynchronize(ab.second));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ab.first, ab.second));
    p->timed_total_us += (double)ms * 1e3;
    p->timed_launches += 1;
    p->ev_pool.push_back(ab.first);
    p->ev_pool.push_back(ab.second);
  }
  p->timed.clear();
  // completion entries that borrowed a timing event: those launches are finished, and their events
  // are back in the pool now
  for (auto it = p->inflight.begin(); it != p->inflight.end();) {
    if (!it->owned) {
      p->completed_launch = std::max(p->completed_launch, it->first);
      it = p->inflight.erase(it);
    } else {
      ++it;
    }
  }
  return EP_OK;
}

// Is ring slot `slot` free for batch index `bi`? (the batch that used it before, bi - Q, must
// have been received; reference: StateBuffer::Allocate throws once alloc_count_ >= batch_)
bool slot_free_for(ep_pool* p, uint64_t bi) { return bi < p->done_ptr + (uint64_t)p->Q; }

hipEvent_t take_done_event(ep_pool* p) {
  hipEvent_t e = nullptr;
  if (!p->done_ev_pool.empty()) {
    e = p->done_ev_pool.back();
    p->done_ev_pool.pop_back();
  } else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
    return nullptr;
  }
  return e;
}

// Records a completion event for launch `lid` on the step stream.
int mark_launch(ep_pool* p, uint64_t lid) {
  hipEvent_t e = take_done_event(p);
  if (!e) return fail(EP_ERR_HIP, "hipEventCreate failed");
  HIP_TRY(hipEventRecord(e, p->stream));
  p->inflight.push_back({lid, e, true});
  return EP_OK;
}

void retire_front(ep_pool* p) {  // the oldest launch is known to be finished
  if (p->inflight.front().owned) p->done_ev_pool.push_back(p->inflight.front().second);
  p->inflight.pop_front();
}

// Events of one kernel launch. The completion event rides on the kernel's own dispatch packet
// (hipExtLaunchKernelGGL stop event): no event packet of its own behind the kernel, about 1 us
// less on the host and on the queue per launch. With timing on it doubles as the timing stop
// event, and the start event is an ordinary record in front of the kernel (handing it to
// hipExtLaunchKernelGGL as the start event costs 20 us of host time per launch on ROCm 7.2).
struct LaunchEvents {
  hipEvent_t start = nullptr, stop = nullptr;
  bool timed = false;
};
int begin_launch(ep_pool* p, LaunchEvents* le, hipStream_t st = nullptr) {
  if (p->timing) {
    le->start = get_event(p);
    le->stop = get_event(p);
    le->timed = true;
    if (!le->start || !le->stop) return fail(EP_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(le->start, st ? st : p->stream));
  } else {
    le->stop = take_done_event(p);
    if (!le->stop) return fail(EP_ERR_HIP, "hipEventCreate failed");
  }
  return EP_OK;
}
void abandon_launch(ep_pool* p, const LaunchEvents& le) {
  if (le.timed) { p->ev_pool.push_back(le.start); p->ev_pool.push_back(le.stop); }
  else p->done_ev_pool.push_back(le.stop);
}
void end_launch(ep_pool* p, const LaunchEvents& le, uint64_t lid) {
  p->inflight.push_back({lid, le.stop, !le.timed});
  if (le.timed) p->timed.emplace_back(le.start, le.stop);
}

// Blocks until launch `lid` has finished (the stream is in-order, so everything older has too).
// `lk` (the caller's hold on p->mu), when given, is released around the wait: a consumer sitting
// in recv must not keep other threads from sending (state_buffer.h:135-140 waits on a semaphore
// with no lock held).
int wait_launch(ep_pool* p, uint64_t lid, std::unique_lock<std::mutex>* lk = nullptr) {
  if (lid <= p->completed_launch) return EP_OK;
  while (!p->inflight.empty() && p->inflight.front().first < lid) retire_front(p);  // older launch: done once `lid` is
  hipEvent_t ev = nullptr;
  bool owned = true;
  if (!p->inflight.empty() && p->inflight.front().first == lid) {
    ev = p->inflight.front().second;
    owned = p->inflight.front().owned;
    p->inflight.pop_front();
  }
  // (the event, not the stream: hipStreamSynchronize costs 4-5 us more per wait on ROCm 7.2. What it
  // buys - the runtime then knows the stream is idle, so a later hipDeviceSynchronize of the caller
  // is free instead of 9 us - is ep_pool_sync's job: callers that are about to synchronise the
  // device call it BEFORE receiving)
  if (lk) lk->unlock();
  const hipError_t e = ev ? hipEventSynchronize(ev) : hipStreamSynchronize(p->stream);
  if (lk) lk->lock();
  if (ev && owned) p->done_ev_pool.push_back(ev);
  if (e != hipSuccess) return fail(EP_ERR_HIP, "waiting for launch %llu failed: %s", (unsigned long long)lid, hipGetErrorString(e));
  p->completed_launch = std::max(p->completed_launch, lid);
  return EP_OK;
}

// hipStreamSynchronize(p->stream) with p->mu released; afterwards every launch enqueued before the
// call is known to be finished.
int sync_stream_unlocked(ep_pool* p, std::unique_lock<std::mutex>* lk) {
  const uint64_t upto = p->launch_seq;
  p->stream_extra = false;  // (whatever other threads enqueue during the wait sets it again)
  if (lk) lk->unlock();
  const hipError_t e = hipStreamSynchronize(p->stream);
  if (lk) lk->lock();
  if (e != hipSuccess) return fail(EP_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
  while (!p->inflight.empty() && p->inflight.front().first <= upto) retire_front(p);
  p->completed_launch = std::max(p->completed_launch, upto);
  return EP_OK;
}

// ---- error words ----------------------------------------------------------------------------
int err_index(ep_pool* p, uint64_t bi) { return (int)(bi % (uint64_t)p->EQ); }

uint32_t take_err_bits(ep_pool* p, int w) {  // reads and clears word w
  epk::ErrWords& e = p->h_err[w];
  const uint32_t bits = (e.action_check ? 1u : 0u) | (e.env_id_range ? 2u : 0u) | (e.dup_env ? 4u : 0u);
  e.action_check = 0;
  e.env_id_range = 0;
  e.dup_env = 0;
  p->deferred_err[w] = 0;
  return bits;
}

int status_of_err_bits(uint32_t bits, const char* what) {
  if (bits & 1u)
    return fail(EP_ERR_ACTION_CHECK,
                "%s: Dummy action self-check failed: list_action lanes differ (the reference aborts here, "
                "dummy_envpool.h:189-192)", what);
  if (bits & 2u) return fail(EP_ERR_ACTION_CHECK, "%s: an env_id in the sent batch is outside this pool's shard", what);
  if (bits & 4u)
    return fail(EP_ERR_DUP_ENV,
                "%s: a device-memory send (or rollout) named an env twice; its two items were stepped "
                "concurrently and the rows of that env are not the reference's", what);
  return EP_OK;
}

// Batch `bi` is about to get its first rows: if its error word still belongs to the batch 2Q
// earlier (handed out by ep_pool_recv_on_stream and never looked at), fold that one into the
// pool's sticky flags, which the next ep_pool_sync reports.
int claim_err_word(ep_pool* p, uint64_t bi) {
  const int w = err_index(p, bi);
  if (!p->deferred_err[w]) return EP_OK;
  if (int rc = wait_launch(p, p->deferred_lid[w])) return rc;
  p->sticky_err |= take_err_bits(p, w);
  return EP_OK;
}

uint32_t next_stamp(ep_pool* p) {  // stamp of the launch about to be enqueued: its id, never 0
  const uint32_t v = (uint32_t)(p->launch_seq + 1);
  return v ? v : 1u;
}

// ---- resident step server (dummy_server.hpp) ---------------------------------------------------
// Asks a running server to leave and waits until its kernel has ended. Every entry point that
// enqueues work on the pool's streams or reads the env state calls this first: while the server
// runs, it alone owns the env records and the ring.
// Resident servers running per device in this process: each holds a (low-priority) hardware queue.
std::atomic<int> g_servers_running[64];
std::atomic<int> g_pools_on_device[64];  // live pools per device in this process
int max_servers_per_device() {  // EP_SERVER_MAX_PER_DEVICE (read when a pool asks to start its server)
  const char* e = getenv("EP_SERVER_MAX_PER_DEVICE");
  return e ? std::max(0, atoi(e)) : 1;
}
// Several pools on ONE GPU (a sharded front-end rehearsed on a one-GPU box): a resident kernel
// beside other pools' launches made every step slower, not faster (4 shards of 16 384 envs:
// 89 us per front-end step with launches only, 142-177 us with one to four servers resident,
// profiles/r03/sharded_step_time_*.txt), so the server is for a pool that has its GPU to itself -
// the deployment it is meant for. EP_SERVER_SHARED_DEVICE=1 at pool creation lifts that (measurements, tests).
void ring_bell(ep_pool* p, unsigned long long v) {
  if (p->bar_bell) {
    // device memory through the BAR: a write-combining mapping, so the store is pushed out at once
    *(volatile unsigned long long*)p->bar_bell = v;
#if defined(__HIP_DEVICE_COMPILE__)
    // (host-only function: nothing to emit in the device pass)
#elif defined(__x86_64__)
    _mm_sfence();
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
  } else {
    __atomic_store_n(&p->h_ctl->bell, v, __ATOMIC_RELEASE);
  }
}

int server_stop(ep_pool* p) {
  if (p->step_mode == 1)
    return fail(EP_ERR_INVALID, "a step is in flight (ep_pool_step_begin): call ep_pool_step_end first");
  if (!p->server_running) return EP_OK;
  ring_bell(p, (unsigned long long)p->server_seq | epk::kServerQuit);
  HIP_TRY(hipStreamSynchronize(p->server_stream));  // (the kernel leaves within a poll of seeing the flag, or has left already)
  ring_bell(p, (unsigned long long)p->server_seq);
  p->server_running = false;
  g_servers_running[p->device & 63].fetch_sub(1);
  return EP_OK;
}

// The idle watchdog (one thread per pool, started with the first server): see ep_pool::wd.
void server_watchdog(ep_pool* p) {
  std::unique_lock<std::mutex> lk(p->mu);
  while (!p->wd_quit) {
    if (!p->server_running) { p->wd_cv.wait(lk); continue; }
    // (a caller that keeps running into idle stops gets half the patience each time: the stall of
    // one that synchronises the device after every step is this limit)
    // A server that has just been started gets 2 ms before its first step-to-step gap is judged:
    // starting one costs some 50 us, and a front-end that starts several (one per shard) in a row
    // would otherwise find the first ones stopped again before its next step.
    const auto limit = std::chrono::microseconds(
        std::max<int64_t>(p->server_fresh ? 2000 : 200, p->idle_limit_us >> std::min(p->idle_streak, 8u)));
    if (p->step_mode == 1) { p->wd_cv.wait_for(lk, limit); continue; }
    const auto deadline = p->last_ring + limit;
    if (std::chrono::steady_clock::now() < deadline) { p->wd_cv.wait_until(lk, deadline); continue; }
    if (hipSetDevice(p->device) != hipSuccess || server_stop(p) != EP_OK) { p->wd_cv.wait_for(lk, std::chrono::milliseconds(50)); continue; }
    ++p->server_idle_stops;
    p->gap_tainted = true;
    // a caller that keeps letting the server run idle (one that synchronises the device after every
    // step, or steps rarely) is served by the launch path for a while: 8, 16, ... 1024 steps
    p->idle_streak = std::min(p->idle_streak + 1, 8u);
    p->server_backoff = 4u << p->idle_streak;
  }
}

// Starts the server for the action arrays `arrs` (env_id, players.env_id, list_action on the
// device); the first step it serves is number server_seq + 1, batch index bi0.
int server_start(ep_pool* p, const void* const* arrs) {
  // everything the launch-per-step path has in flight must be finished: the server reads the env
  // records those kernels write, and writes ring slots they may still be filling
  HIP_TRY(hipStreamSynchronize(p->copy_stream));
  if (int rc = sync_stream_unlocked(p, nullptr)) return rc;
  HIP_TRY(hipMemsetAsync(p->d_server_words, 0, 128 * (2 + epk::kServerShards), p->server_stream));
  p->h_ctl->alive = 1;
  for (int s = 0; s < epk::kServerShards; ++s) __atomic_store_n(&p->h_ctl->done[s], p->server_seq, __ATOMIC_RELEASE);
  ring_bell(p, (unsigned long long)p->server_seq);
  epk::ServerArgs a;
  a.ctl = p->d_ctl;
  a.bell_dev = p->bar_bell;
  a.dev_flag = p->d_server_words;
  a.counters = reinterpret_cast<uint32_t*>(p->d_server_words + 16);  // the line after the flag's
  a.env_id = (const int32_t*)arrs[0];
  a.players_env_id = (const int32_t*)arrs[1];
  a.list_action = (const double*)arrs[2];
  a.term = p->d_term;
  a.term_stride = (uint32_t)p->term_stride;
  a.env = p->d_env;
  a.stats = p->stats;
  a.ring.ring = p->d_ring;
  a.ring.stride = p->slot_stride;
  for (int k = 0; k < 11; ++k) a.ring.off[k] = (uint32_t)p->key_off[k];
  a.ring.depth = p->Q;
  a.err = p->d_err;
  a.err_depth = p->EQ;
  a.stamp = p->d_stamp;
  a.pc = pool_const(p);
  a.first_step = p->server_seq + 1;
  a.bi0 = p->alloc_count / (uint64_t)p->B;
  // the kernel's own limit is a last resort (the host stops the server: every other entry point,
  // the idle watchdog, ep_pool_destroy): about a second of polls
  a.idle_polls = p->bar_bell ? 1500000u : 800000u;
  a.num_tiles = p->N / epk::kBlock;
  const int n_wg = std::min(a.num_tiles, std::max(1, p->cu_count));
  a.tiles_per_wg = (a.num_tiles + n_wg - 1) / n_wg;
  a.trace = p->d_server_trace;
  a.two_level_join = p->server_two_level ? 1 : 0;
  p->server_wgs = n_wg;
  if (p->S == 10) hipLaunchKernelGGL(epk::dummy_step_server<10>, dim3(n_wg), dim3(epk::kBlock), 0, p->server_stream, a);
  else hipLaunchKernelGGL(epk::dummy_step_server<0>, dim3(n_wg), dim3(epk::kBlock), 0, p->server_stream, a);
  HIP_TRY(hipGetLastError());
  p->server_running = true;
  g_servers_running[p->device & 63].fetch_add(1);
  p->server_fresh = true;
  p->server_shards = p->server_two_level ? 1 : std::min(n_wg, epk::kServerShards);
  ++p->server_launches;
  p->last_ring = std::chrono::steady_clock::now();
  for (int k = 0; k < 3; ++k) p->server_arrays[k] = arrs[k];
  if (!p->wd_started) {
    p->wd_started = true;
    p->wd = std::thread(server_watchdog, p);
  } else {
    p->wd_cv.notify_one();
  }
  return EP_OK;
}

// Launch the single-player kernel for items [i0, i0+n) of one send/reset that land in `slot`
// at rows row0...
// `done_ev`, when given, rides on the kernel's own dispatch packet as its completion event (no
// separate event packet behind the kernel: about 1 us less on the host and on the queue).
int launch_sp(ep_pool* p, const int32_t* d_env_id, const int32_t* d_penv, const double* d_la, int n,
              int slot, int err_w, int row0, bool force_reset, bool check_dup, const LaunchEvents& le,
              char* out_base = nullptr, hipStream_t st = nullptr, const uint32_t* la_ok = nullptr, int la_bit0 = 0,
              char* const* ext_dev = nullptr, int host_keys = 0) {
  if (!st) st = p->stream;
  epk::StepArgs a;
  a.env_id = d_env_id;
  a.players_env_id = d_penv;
  a.list_action = d_la;
  a.la_ok = la_ok;
  a.la_bit0 = la_bit0;
  a.env = p->d_env;
  a.stats = p->stats;
  a.slot = ext_dev ? slot_ptrs_of(ext_dev) : out_base ? slot_ptrs_at(p, out_base) : host_slot_ptrs(p, slot);
  a.err = p->d_err + err_w;
  a.stamp = check_dup ? p->d_stamp : nullptr;
  a.stamp_val = next_stamp(p);
  a.pc = pool_const(p);
  a.n = n;
  a.row0 = row0;
  a.force_reset = force_reset ? 1 : 0;
  a.term = (out_base || ext_dev) ? nullptr : p->d_term + (size_t)slot * p->term_stride;
  a.host_keys = host_keys;
  dim3 grid((n + epk::kBlock - 1) / epk::kBlock), block(epk::kBlock);
  const bool dev = !force_reset && d_la && a.stamp && !p->force_generic;  // (see dummy_step_sp)
  if (p->S == 10 && dev)
    hipExtLaunchKernelGGL((epk::dummy_step_sp<10, true>), grid, block, 0, st, nullptr, le.stop, 0, a);
  else if (p->S == 10)
    hipExtLaunchKernelGGL((epk::dummy_step_sp<10, false>), grid, block, 0, st, nullptr, le.stop, 0, a);
  else if (dev)
    hipExtLaunchKernelGGL((epk::dummy_step_sp<0, true>), grid, block, 0, st, nullptr, le.stop, 0, a);
  else
    hipExtLaunchKernelGGL((epk::dummy_step_sp<0, false>), grid, block, 0, st, nullptr, le.stop, 0, a);
  HIP_TRY(hipGetLastError());
  return EP_OK;
}

// Common tail of send/reset once the action arrays are on the device.
// `n` items; multi-player: `np_rows` rows in players.env_id.
// `cuts`: positions (ascending, in (0, n)) at which the items must be split into separate launches
// because an env shows up again (host-validated sends); null for device-memory sends, whose
// kernels check the per-env launch stamps instead.
// `row_base`: host mode sends one batch as several calls (row_base = items already sent).
// `host_mode`: the send comes from host memory and will most likely be received as host memory: a
// batch whose first rows arrive this way gets a pinned host block, and every kernel that writes
// rows of that batch (this send's or a later one's, host or device memory) writes them into the
// block over PCIe instead of into the ring slot; recv hands the block out without a copy.
// `la_ok` / `la_bit_base`: host-compared list_action lanes, bit la_bit_base + i for item i.
int enqueue_slices(ep_pool* p, const int32_t* d_env_id, const int32_t* d_penv, const double* d_la,
                   int n, int np_rows, bool force_reset, const std::vector<int>* cuts, bool host_checked,
                   int row_base = 0, bool host_mode = false, hipStream_t st = nullptr,
                   const uint32_t* la_ok = nullptr, int la_bit_base = 0, const int32_t* h_ids = nullptr) {
  if (n == 0) return EP_OK;
  if (p->poisoned) return fail(EP_ERR_HIP, "the pool is in a failed state after an earlier launch error");
  const uint64_t B = (uint64_t)p->B;
  // ring capacity check for every batch this call touches
  uint64_t first_bi = p->alloc_count / B, last_bi = (p->alloc_count + (uint64_t)n - 1) / B;
  for (uint64_t bi = first_bi; bi <= last_bi; ++bi)
    if (!slot_free_for(p, bi))
      return fail(EP_ERR_RING_FULL,
                  "StateBuffer ring full: %d un-received batches (ring_depth=%d); call recv", p->Q,
                  p->Q);
  for (uint64_t bi = first_bi; bi <= last_bi; ++bi)
    if (p->slots[bi % (uint64_t)p->Q].alloc_count == 0)
      if (int rc = claim_err_word(p, bi)) return rc;
  if (p->is_sync) p->stepping_env_num += n;  // async_envpool.h:73-75 / :185-187

  int rc = EP_OK;
  if (p->P > 1) {
    rc = epk_mp::enqueue_mp(p, d_env_id, d_penv, d_la, n, np_rows, force_reset, cuts, host_checked, host_mode, la_ok);
    p->cv.notify_all();
    return rc;
  }
  int i0 = 0;
  size_t next_cut = 0;
  while (i0 < n) {
    uint64_t pos = p->alloc_count;
    uint64_t bi = pos / B;
    int slot = (int)(bi % (uint64_t)p->Q);
    int room = (int)(B - pos % B);
    int cnt = std::min(room, n - i0);
    if (cuts) {
      while (next_cut < cuts->size() && (*cuts)[next_cut] <= i0) ++next_cut;
      if (next_cut < cuts->size()) cnt = std::min(cnt, (*cuts)[next_cut] - i0);
    }
    // state_buffer.h:92-95: sync single-player => row = order (index in the sent batch);
    // otherwise rows follow the allocation order.
    int row0 = p->is_sync ? row_base + i0 : (int)(pos % B);
    SlotHost& sh = p->slots[slot];
    if (p->target_pending && sh.alloc_count == 0 && !sh.has_work && !sh.host_blk) {
      // the caller's arrays for this batch (ep_pool_host_target): the hold on the block moves to the slot
      sh.ext = true;
      sh.ext_row0 = p->target_row0;
      memcpy(sh.ext_host, p->target_host, sizeof(sh.ext_host));
      memcpy(sh.ext_dev, p->target_dev, sizeof(sh.ext_dev));
      sh.ext_home = std::move(p->target_home);
      sh.ext_block = p->target_block;
      p->target_pending = false;
      p->target_block = nullptr;
      if (host_mode) sh.host_blk = sh.ext_host[0];  // direct: the kernels write the caller's arrays (rows from ext_row0)
    }
    if (host_mode && !sh.host_blk && !sh.ext && sh.alloc_count == 0 && !sh.has_work) {
      void* blk = nullptr;
      if ((rc = epc::take_block(p->blocks, &blk))) break;
      sh.host_blk = blk;
    }
    char* out_base = nullptr;
    if (sh.host_blk && !sh.ext) {
      void* dv = nullptr;
      if (hipHostGetDevicePointer(&dv, sh.host_blk, 0) != hipSuccess) { rc = fail(EP_ERR_HIP, "pinned result block is not device-mappable"); break; }
      out_base = (char*)dv;
    }
    LaunchEvents le;
    if ((rc = begin_launch(p, &le, st))) break;
    const bool ext_direct = sh.ext && sh.host_blk;
    // rows that the kernel writes over PCIe: the two arrays that are zero for every env (reward,
    // info:players.id: dummy_envpool.h:148,152 / :200,204) are zeroed HERE instead, below, while the
    // kernel runs - 8 of the 72 bytes per env do not cross the link, which is what bounds this path
    // (and the two that hold the ids that were sent - validated before they got here - are copied from
    // the caller's array: 16 of 72 bytes in all)
    const int host_keys = (p->host_const && sh.host_blk != nullptr) ? (h_ids ? 3 : 1) : 0;
    rc = launch_sp(p, d_env_id + i0, d_penv ? d_penv + i0 : nullptr,
                   d_la ? d_la + (size_t)i0 * 6 : nullptr, cnt, slot, err_index(p, bi), row0 + (ext_direct ? (int)sh.ext_row0 : 0), force_reset,
                   !host_checked && !force_reset, le, out_base, st, la_ok, la_bit_base + i0,
                   ext_direct ? sh.ext_dev : nullptr, host_keys);
    if (rc != EP_OK) { abandon_launch(p, le); break; }
    sh.term_ok = (sh.has_work ? sh.term_ok : true) && out_base == nullptr && !ext_direct;
    sh.alloc_count += cnt;
    sh.shared_off += cnt;
    sh.written.emplace_back(row0, row0 + cnt);
    sh.has_work = true;
    sh.launch_id = ++p->launch_seq;
    end_launch(p, le, sh.launch_id);
    if (host_keys) {
      const size_t r = (size_t)row0 + (sh.ext ? (size_t)sh.ext_row0 : 0);
      auto key = [&](int k) { return (sh.ext ? sh.ext_host[k] : (char*)sh.host_blk + p->key_off[k]) + 4 * r; };
      memset(key(4), 0, 4 * (size_t)cnt);
      memset(key(10), 0, 4 * (size_t)cnt);
      if (host_keys & 2) {
        memcpy(key(0), h_ids + i0, 4 * (size_t)cnt);
        memcpy(key(1), h_ids + i0, 4 * (size_t)cnt);
      }
    }
    p->alloc_count += cnt;
    i0 += cnt;
  }
  if (rc != EP_OK && i0 > 0) p->poisoned = true;  // part of the send is on the stream, the queue counters are not the reference's any more
  p->cv.notify_all();
  return rc;
}

// Copies `bytes` of host action data into the next pinned staging slot and enqueues the H2D
// copy on the copy stream; the compute stream then waits for it.
int stage_begin(ep_pool* p, int* out_idx) {
  int a = (int)(p->stage_idx++ % (uint64_t)p->A);
  HIP_TRY(hipEventSynchronize(p->stage_free[a]));  // the kernel that read this slot is done
  *out_idx = a;
  return EP_OK;
}

int stage_commit(ep_pool* p, int a, size_t bytes) {
  char* h = p->h_stage + (size_t)a * p->stage_bytes;
  char* d = p->d_stage + (size_t)a * p->stage_bytes;
  p->copy_dirty = true;
  HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, p->copy_stream));
  HIP_TRY(hipEventRecord(p->stage_copied[a], p->copy_stream));
  HIP_TRY(hipStreamWaitEvent(p->stream, p->stage_copied[a], 0));
  return EP_OK;
}

int stage_release(ep_pool* p, int a) {
  HIP_TRY(hipEventRecord(p->stage_free[a], p->stream));
  return EP_OK;
}

// bit j of bits[j / 32]: the six list_action lanes of item j are equal (IEEE ==, as the device
// compares them: NaN differs from itself, -0 equals +0). Host-only code.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target("avx2"))) void la_lanes_equal_avx2(const double* la, int cnt, uint32_t* bits) {
  for (int w0 = 0; w0 < cnt; w0 += 32) {
    uint32_t m = 0;
    const int lim = std::min(32, cnt - w0);
    for (int j = 0; j < lim; ++j) {
      const double* v = la + 6 * (size_t)(w0 + j);
      const __m256d first = _mm256_broadcast_sd(v);
      const __m256d eq = _mm256_and_pd(_mm256_cmp_pd(_mm256_loadu_pd(v), first, _CMP_EQ_OQ),
                                       _mm256_cmp_pd(_mm256_loadu_pd(v + 2), first, _CMP_EQ_OQ));
      m |= (_mm256_movemask_pd(eq) == 0xf ? 1u : 0u) << j;
    }
    bits[w0 >> 5] = m;
  }
}
#endif
void la_lanes_equal(const double* la, int cnt, uint32_t* bits) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) return la_lanes_equal_avx2(la, cnt, bits);
#endif
  for (int w0 = 0; w0 < cnt; w0 += 32) {
    uint32_t m = 0;
    const int lim = std::min(32, cnt - w0);
    for (int j = 0; j < lim; ++j) {
      const double* v = la + 6 * (size_t)(w0 + j);
      const bool ok = (v[0] == v[1]) & (v[0] == v[2]) & (v[0] == v[3]) & (v[0] == v[4]) & (v[0] == v[5]);
      m |= (ok ? 1u : 0u) << j;
    }
    bits[w0 >> 5] = m;
  }
}

// Host-memory ids: the shard range check, plus the rule that one launch steps an env at most
// once. The reference's single worker steps an env named twice in one send twice, in order;
// here `cuts` receives the positions at which the send is split into separate launches (in order
// on one stream) so that the same happens. Ascending ids - the usual case - need one vectorised
// pass; anything else goes through the per-env stamp array.
int validate_ids_host(ep_pool* p, const int32_t* ids, int n, std::vector<int>* cuts) {
  if (cuts) cuts->clear();
  if (n <= 0) return EP_OK;
  const uint32_t N = (uint32_t)p->N, off = (uint32_t)p->cfg.env_id_offset;
  uint32_t bad = 0, not_inc = 0;
  for (int i = 0; i < n; ++i) bad |= ((uint32_t)ids[i] - off >= N) ? 1u : 0u;
  if (bad) {
    for (int i = 0; i < n; ++i)
      if ((uint32_t)ids[i] - off >= N)
        return fail(EP_ERR_INVALID, "env_id %d at position %d is outside this pool's shard [%d, %d)",
                    ids[i], i, p->cfg.env_id_offset, p->cfg.env_id_offset + p->N);
  }
  if (!cuts) return EP_OK;
  for (int i = 1; i < n; ++i) not_inc |= (ids[i] <= ids[i - 1]) ? 1u : 0u;
  if (!not_inc) return EP_OK;
  if (p->h_stamp.empty()) p->h_stamp.assign((size_t)p->N, 0u);
  auto fresh = [&]() -> uint32_t {
    if (++p->h_stamp_seq == 0) {  // wrapped: forget every old stamp
      std::fill(p->h_stamp.begin(), p->h_stamp.end(), 0u);
      p->h_stamp_seq = 1;
    }
    return p->h_stamp_seq;
  };
  uint32_t cur = fresh();
  uint32_t* st = p->h_stamp.data();
  for (int i = 0; i < n; ++i) {
    const uint32_t local = (uint32_t)ids[i] - off;
    if (st[local] == cur) {
      cuts->push_back(i);
      cur = fresh();
    }
    st[local] = cur;
  }
  return EP_OK;
}

}  // namespace

#include "dummy_mp_host.inl"

extern "C" {

const char* ep_last_error(void) { return epc::g_last_error.c_str(); }
const char* ep_version(void) { return "envpool_amd 0.1 (gfx950)"; }

namespace {
constexpr int kTermParts = 16;
struct TermArgs {
  const uint8_t* done[kTermParts];
  const uint8_t* trunc[kTermParts];
  uint8_t* out[kTermParts];
  int64_t n[kTermParts];
};
// out = done & ~trunc, bytewise over up to 16 arrays in one launch: blockIdx.y picks the array, a
// thread takes 16 bytes (one 16-byte access per array when all three are aligned, bytes otherwise)
__global__ void terminated_many_kernel(TermArgs a) {
  const int part = blockIdx.y;
  const int64_t n = a.n[part];
  const uint8_t* __restrict__ d = a.done[part];
  const uint8_t* __restrict__ t = a.trunc[part];
  uint8_t* __restrict__ o = a.out[part];
  const bool aligned = ((((uintptr_t)d) | ((uintptr_t)t) | ((uintptr_t)o)) & 15) == 0;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (int64_t)gridDim.x * blockDim.x * 16) {
    if (aligned && i + 16 <= n) {
      const uint4 dv = *reinterpret_cast<const uint4*>(d + i), tv = *reinterpret_cast<const uint4*>(t + i);
      *reinterpret_cast<uint4*>(o + i) = make_uint4(dv.x & ~tv.x, dv.y & ~tv.y, dv.z & ~tv.z, dv.w & ~tv.w);
    } else {
      for (int64_t j = i; j < n && j < i + 16; ++j) o[j] = d[j] & (uint8_t)~t[j];
    }
  }
}
}  // namespace

int ep_terminated_many(int32_t k, const void* const* done, const void* const* trunc, void* const* out, const int64_t* n,
                       int32_t device, void* stream) {
  ::epc::DeviceScope _dev_scope;
  if (device >= 0) HIP_TRY(_dev_scope.enter(device));
  if (k < 0 || (k > 0 && (!done || !trunc || !out || !n))) return fail(EP_ERR_INVALID, "bad argument");
  for (int i0 = 0; i0 < k; i0 += kTermParts) {
    TermArgs a{};
    const int cnt = std::min(kTermParts, k - i0);
    int64_t longest = 0;
    for (int i = 0; i < cnt; ++i) {
      if (n[i0 + i] < 0 || (n[i0 + i] > 0 && (!done[i0 + i] || !trunc[i0 + i] || !out[i0 + i])))
        return fail(EP_ERR_INVALID, "array %d: null pointer or negative length", i0 + i);
      a.done[i] = (const uint8_t*)done[i0 + i];
      a.trunc[i] = (const uint8_t*)trunc[i0 + i];
      a.out[i] = (uint8_t*)out[i0 + i];
      a.n[i] = n[i0 + i];
      longest = std::max(longest, n[i0 + i]);
    }
    if (longest == 0) continue;
    const int64_t groups = std::min<int64_t>((longest + 4095) / 4096, 1024);
    hipLaunchKernelGGL(terminated_many_kernel, dim3((unsigned)groups, (unsigned)cnt), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
  }
  return EP_OK;
}

int ep_stream_synchronize(int32_t device, void* stream) {
  ::epc::DeviceScope _dev_scope;
  if (device >= 0) HIP_TRY(_dev_scope.enter(device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return EP_OK;
}

int32_t ep_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ep_dummy_pool_create(const ep_dummy_config* cfg, ep_pool** out) {
  if (!cfg || !out) return fail(EP_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->num_envs < 1) return fail(EP_ERR_INVALID, "num_envs must be >= 1, got %d", cfg->num_envs);
  if (cfg->batch_size > cfg->num_envs)  // env_spec.h:74-79
    return fail(EP_ERR_INVALID,
                "It is required that batch_size <= num_envs, got num_envs = %d, batch_size = %d",
                cfg->num_envs, cfg->batch_size);
  if (cfg->batch_size < 0) return fail(EP_ERR_INVALID, "batch_size must be >= 0");
  if (cfg->max_num_players < 1) return fail(EP_ERR_INVALID, "max_num_players must be >= 1");
  if (cfg->state_num < 2) return fail(EP_ERR_INVALID, "state_num must be >= 2 (obs:raw has two live columns)");
  int ndev = ep_device_count();
  if (ndev <= 0)
    return fail(EP_ERR_NO_DEVICE,
                "no HIP device visible: envpool_amd has no CPU fallback (the CPU oracle lives in "
                "oracle/ and is test infrastructure only)");
  if (cfg->device_id < 0 || cfg->device_id >= ndev)
    return fail(EP_ERR_INVALID, "device_id %d out of range [0,%d)", cfg->device_id, ndev);

  std::unique_ptr<ep_pool> p(new ep_pool());
  p->cfg = *cfg;
  p->N = cfg->num_envs;
  p->B = cfg->batch_size == 0 ? cfg->num_envs : cfg->batch_size;  // env_spec.h:80-82
  p->cfg.batch_size = p->B;
  p->P = cfg->max_num_players;
  p->S = cfg->state_num;
  p->is_sync = (p->B == p->N && p->P == 1);  // async_envpool.h:95
  int ref_depth = (p->N / p->B + 2) * 2;     // state_buffer_queue.h:68
  p->Q = cfg->ring_depth > 0 ? cfg->ring_depth : ref_depth;
  if (p->Q < 2) return fail(EP_ERR_INVALID, "ring_depth must be >= 2");
  p->device = cfg->device_id;
  if (const char* fg = getenv("EP_FORCE_GENERIC")) p->force_generic = fg[0] == '1';
  if (const char* ft = getenv("EP_MP_FORCE_TICKET")) p->mp_force_ticket = ft[0] != '0';
  if (const char* sk = getenv("EP_MP_SKIP_COUNT")) p->mp_skip_count = sk[0] == '1';
  if (const char* fo = getenv("EP_MP_FOLD")) p->mp_fold = fo[0] == '1';
  if (const char* dh = getenv("EP_DIRECT_HOST")) p->direct_host = dh[0] != '0';
  if (const char* dm = getenv("EP_DIRECT_MIN_ROWS")) p->direct_min_rows = atoi(dm);
  if (const char* hc = getenv("EP_HOST_CONST")) p->host_const = hc[0] != '0';
  if (const char* rv = getenv("EP_ROLLOUT_VARIANT")) p->rollout_variant = atoi(rv);
  EP_ENTER_DEVICE(p->device);
  HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  p->own_stream = p->stream;
  HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));

  // slot layout
  const size_t B = p->B, BP = (size_t)p->B * p->P;
  const size_t rows[11] = {B, BP, B, B, BP, BP, B, B, BP, BP, BP};
  size_t off = 0;
  for (int k = 0; k < 11; ++k) {
    p->key_off[k] = off;
    p->key_rowbytes[k] = (size_t)kSlotElem[k] * (k == 8 ? (size_t)p->S : 1);
    p->key_is_player[k] = rows[k] == BP && (k == 1 || k == 4 || k == 5 || k >= 8);
    off += align_up(rows[k] * p->key_rowbytes[k]);
  }
  p->slot_stride = off;
  if (p->slot_stride * (size_t)p->Q > ((size_t)200 << 30))
    return fail(EP_ERR_TOO_LARGE, "StateBuffer ring of %zu bytes does not fit", p->slot_stride * (size_t)p->Q);
  HIP_TRY(hipMalloc((void**)&p->d_ring, p->slot_stride * (size_t)p->Q));
  HIP_TRY(hipMemsetAsync(p->d_ring, 0, p->slot_stride * (size_t)p->Q, p->stream));

  // env state: state_ = 0, current_step_ = -1 (dummy_envpool.h:118, env.h:71)
  HIP_TRY(hipMalloc((void**)&p->d_env, sizeof(epk::EnvRec) * (size_t)p->N));
  {
    std::vector<epk::EnvRec> init((size_t)p->N, epk::EnvRec{0, -1});
    HIP_TRY(hipMemcpy(p->d_env, init.data(), sizeof(epk::EnvRec) * (size_t)p->N, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMalloc((void**)&p->stats.ep_return, 4 * (size_t)p->N));
  HIP_TRY(hipMalloc((void**)&p->stats.fin_return, 4 * (size_t)p->N));
  HIP_TRY(hipMalloc((void**)&p->stats.fin_length, 4 * (size_t)p->N));
  HIP_TRY(hipMalloc((void**)&p->stats.ep_count, 4 * (size_t)p->N));
  HIP_TRY(hipMemsetAsync(p->stats.ep_return, 0, 4 * (size_t)p->N, p->stream));
  HIP_TRY(hipMemsetAsync(p->stats.fin_return, 0, 4 * (size_t)p->N, p->stream));
  HIP_TRY(hipMemsetAsync(p->stats.fin_length, 0, 4 * (size_t)p->N, p->stream));
  HIP_TRY(hipMemsetAsync(p->stats.ep_count, 0, 4 * (size_t)p->N, p->stream));

  // error words: host-mapped so that a (rare) failure is visible after the slot's event
  p->EQ = 2 * p->Q;
  HIP_TRY(hipHostMalloc((void**)&p->h_err, sizeof(epk::ErrWords) * (size_t)p->EQ, hipHostMallocMapped));
  memset(p->h_err, 0, sizeof(epk::ErrWords) * (size_t)p->EQ);
  HIP_TRY(hipHostGetDevicePointer((void**)&p->d_err, p->h_err, 0));
  HIP_TRY(hipMalloc((void**)&p->d_stamp, 4 * (size_t)p->N));
  HIP_TRY(hipMemsetAsync(p->d_stamp, 0, 4 * (size_t)p->N, p->stream));

  // multi-player scratch
  if (p->P > 1) {
    if ((uint64_t)p->B * (uint64_t)p->P >= (1ull << 30))
      return fail(EP_ERR_TOO_LARGE, "batch_size * max_num_players must stay below 2^30 player rows");
    const size_t tiles = ((size_t)p->N + epk::kMpTile - 1) / epk::kMpTile;
    HIP_TRY(hipMalloc((void**)&p->mp_sc.counts, 8 * (size_t)p->N));
    HIP_TRY(hipMemsetAsync(p->mp_sc.counts, 0, 8 * (size_t)p->N, p->stream));
    HIP_TRY(hipMalloc((void**)&p->mp_sc.tiles, 8 * tiles));
    HIP_TRY(hipMemsetAsync(p->mp_sc.tiles, 0, 8 * tiles, p->stream));
    HIP_TRY(hipMalloc((void**)&p->mp_sc.ticket, 256));  // [0] the ticket, [32] (its own 128 bytes) the folded histogram's arrival counter
    HIP_TRY(hipMemsetAsync(p->mp_sc.ticket, 0, 256, p->stream));
    p->mp_sc.barrier = p->mp_sc.ticket + 32;
    HIP_TRY(hipMalloc((void**)&p->d_slot_players, 4 * (size_t)p->Q));
    HIP_TRY(hipMemsetAsync(p->d_slot_players, 0, 4 * (size_t)p->Q, p->stream));
    HIP_TRY(hipHostMalloc((void**)&p->h_slot_players, 4 * (size_t)p->EQ, hipHostMallocMapped));
    memset(p->h_slot_players, 0, 4 * (size_t)p->EQ);
    HIP_TRY(hipHostGetDevicePointer((void**)&p->d_host_players, p->h_slot_players, 0));
  }

  // action staging ring for host sends: env_id + players.env_id + list_action + 2 unused keys
  p->A = 4;
  p->stage_bytes = align_up(4 * (size_t)p->N) + align_up(4 * (size_t)p->N * p->P) +
                   align_up(48 * (size_t)p->N) + 2 * align_up(4 * (size_t)p->N * p->P);
  HIP_TRY(hipHostMalloc((void**)&p->h_stage, p->stage_bytes * (size_t)p->A, hipHostMallocMapped));
  HIP_TRY(hipMalloc((void**)&p->d_stage, p->stage_bytes * (size_t)p->A));
  p->stage_free.resize(p->A);
  p->stage_copied.resize(p->A);
  for (int a = 0; a < p->A; ++a) {
    HIP_TRY(hipEventCreateWithFlags(&p->stage_free[a], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->stage_copied[a], hipEventDisableTiming));
  }
  {
    // The resident kernel occupies the hardware queue its stream is mapped to for as long as it
    // runs, and a process has few of them (GPU_MAX_HW_QUEUES, 4 by default) shared by all its
    // streams: anything enqueued on a stream that shares that queue would sit behind the server
    // until it leaves. The runtime keeps a separate set of hardware queues per stream priority, so
    // the server stream is created with the LOWEST priority: it then shares queues only with other
    // pools' server streams (EP_SERVER_MAX_PER_DEVICE of which run at once), never with the
    // pool's own streams, torch's or the caller's.
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    if (least != greatest) HIP_TRY(hipStreamCreateWithPriority(&p->server_stream, hipStreamNonBlocking, least));
    else HIP_TRY(hipStreamCreateWithFlags(&p->server_stream, hipStreamNonBlocking));
  }
  HIP_TRY(hipHostMalloc((void**)&p->h_ctl, sizeof(epk::ServerCtl), hipHostMallocMapped));
  memset(p->h_ctl, 0, sizeof(epk::ServerCtl));
  HIP_TRY(hipHostGetDevicePointer((void**)&p->d_ctl, p->h_ctl, 0));
  HIP_TRY(hipMalloc((void**)&p->d_server_words, 128 * (2 + epk::kServerShards)));
  HIP_TRY(hipDeviceGetAttribute(&p->cu_count, hipDeviceAttributeMultiprocessorCount, p->device));
  if (const char* sv = getenv("EP_STEP_SERVER")) p->use_server = sv[0] != '0';
  {
    // the server's doorbell: a word of fine-grained device memory the host can write (large BAR)
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, p->device) != hipSuccess) large_bar = 0;
    if (const char* tr = getenv("EP_SERVER_TRACE"))
      if (tr[0] == '1') {
        HIP_TRY(hipMalloc((void**)&p->d_server_trace, 8 * 4 * 1024));
        HIP_TRY(hipMemset(p->d_server_trace, 0, 8 * 4 * 1024));
#if defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
        // where the host side of the doorbell round trip lives: the CPU this thread runs on and the
        // NUMA node of the host-mapped answer line (a GPU write into the far socket's memory is slower)
        void* page = (void*)((uintptr_t)p->h_ctl & ~(uintptr_t)4095);
        int node = -1;
        (void)syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &node, 0);
        fprintf(stderr, "[ep trace] pool %p: cpu %d, answer line on NUMA node %d\n", (void*)p.get(), sched_getcpu(), node);
#endif
      }
    if (const char* sj = getenv("EP_SERVER_JOIN")) p->server_two_level = sj[0] == '2';
    if (const char* sd = getenv("EP_SERVER_SHARED_DEVICE")) p->server_shared_ok = sd[0] == '1';
    if (const char* sp = getenv("EP_SERVER_PAUSE")) p->spin_pause = sp[0] == '1';
    const char* sb = getenv("EP_SERVER_BAR");
    if (large_bar && !(sb && sb[0] == '0')) {
      if (hipExtMallocWithFlags((void**)&p->bar_bell, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
        HIP_TRY(hipMemset(p->bar_bell, 0, 4096));
        HIP_TRY(hipDeviceSynchronize());
        // the attribute promises that a host store reaches this word; see it arrive once before a
        // step's completion depends on it (otherwise: the leader wake-up through host memory)
        const unsigned long long probe = 0x0ddba11c0ffeeull;
        unsigned long long back = 0;
        *(volatile unsigned long long*)p->bar_bell = probe;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
        _mm_sfence();
#endif
        for (int tries = 0; tries < 3 && back != probe; ++tries) {
          if (tries) std::this_thread::sleep_for(std::chrono::milliseconds(1));
          HIP_TRY(hipMemcpy(&back, p->bar_bell, sizeof(back), hipMemcpyDeviceToHost));
        }
        if (back == probe) {
          HIP_TRY(hipMemset(p->bar_bell, 0, 4096));
          HIP_TRY(hipDeviceSynchronize());
        } else {
          (void)hipFree(p->bar_bell);
          p->bar_bell = nullptr;
        }
      } else {
        (void)hipGetLastError();
        p->bar_bell = nullptr;
      }
    }
  }
  p->slots.resize(p->Q);
  p->term_stride = ((size_t)p->B + 255) & ~(size_t)255;
  HIP_TRY(hipMalloc((void**)&p->d_term, p->term_stride * (size_t)p->Q));
  HIP_TRY(hipMemsetAsync(p->d_term, 0, p->term_stride * (size_t)p->Q, p->stream));
  p->slot_term_seq.assign((size_t)p->Q, 0);
  HIP_TRY(hipEventCreateWithFlags(&p->stats_ev, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&p->consumer_ev, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&p->recv_ev, hipEventDisableTiming));
  p->deferred_err.assign((size_t)p->EQ, 0);
  p->deferred_lid.assign((size_t)p->EQ, 0);
  p->blocks = std::make_shared<HostBlocks>();
  p->blocks->bytes = p->slot_stride;
  HIP_TRY(hipStreamSynchronize(p->stream));
  g_pools_on_device[p->device & 63].fetch_add(1);
  p->counted_on_device = true;
  *out = p.release();
  return EP_OK;
}

int ep_pool_destroy(ep_pool* p) {
  if (!p) return EP_OK;
  ::epc::DeviceScope _dev_scope;
  (void)_dev_scope.enter(p->device);
  if (p->wd_started) {
    { std::lock_guard<std::mutex> lk(p->mu); p->wd_quit = true; }
    p->wd_cv.notify_all();
    p->wd.join();
  }
  p->step_mode = 0;
  (void)server_stop(p);
  if (p->counted_on_device) g_pools_on_device[p->device & 63].fetch_sub(1);
  (void)hipFree(p->bar_bell);
  (void)hipFree(p->d_server_trace);
  (void)hipStreamSynchronize(p->stream);
  (void)hipStreamSynchronize(p->copy_stream);
  if (p->server_stream) (void)hipStreamDestroy(p->server_stream);
  if (p->h_ctl) (void)hipHostFree(p->h_ctl);
  (void)hipFree(p->d_server_words);
  for (auto& ab : p->timed) { (void)hipEventDestroy(ab.first); (void)hipEventDestroy(ab.second); }
  for (auto e : p->ev_pool) (void)hipEventDestroy(e);
  for (auto& fe : p->inflight) (void)hipEventDestroy(fe.second);
  for (auto e : p->done_ev_pool) (void)hipEventDestroy(e);
  if (p->stats_ev) (void)hipEventDestroy(p->stats_ev);
  if (p->consumer_ev) (void)hipEventDestroy(p->consumer_ev);
  if (p->recv_ev) (void)hipEventDestroy(p->recv_ev);
  if (p->producer_ev) (void)hipEventDestroy(p->producer_ev);
  for (auto e : p->stage_free) (void)hipEventDestroy(e);
  for (auto e : p->stage_copied) (void)hipEventDestroy(e);
  (void)hipFree(p->d_ring);
  (void)hipFree(p->d_env);
  (void)hipFree(p->d_term);
  (void)hipFree(p->d_stamp);
  (void)hipFree(p->stats.ep_return);
  (void)hipFree(p->stats.fin_return);
  (void)hipFree(p->stats.fin_length);
  (void)hipFree(p->stats.ep_count);
  (void)hipFree(p->mp_sc.counts);
  (void)hipFree(p->mp_sc.tiles);
  (void)hipFree(p->mp_sc.ticket);
  (void)hipFree(p->d_slot_players);
  if (p->h_slot_players) (void)hipHostFree(p->h_slot_players);
  if (p->h_err) (void)hipHostFree(p->h_err);
  if (p->h_stage) (void)hipHostFree(p->h_stage);
  (void)hipFree(p->d_stage);
  for (auto& sh : p->slots) {
    if (sh.host_blk && !sh.ext) (void)hipHostFree(sh.host_blk);  // a direct-mode batch nobody received
    if (sh.ext_home) sh.ext_home->drop(sh.ext_block);            // (the stream was drained above: nothing writes it any more)
  }
  if (p->target_home) p->target_home->drop(p->target_block);
  epc::retire_blocks(p->blocks);
  (void)hipStreamDestroy(p->own_stream);
  (void)hipStreamDestroy(p->copy_stream);
  delete p;
  return EP_OK;
}

int ep_pool_reset(ep_pool* p, const int32_t* env_ids, int32_t n, int32_t mem) {
  if (!p || (!env_ids && n > 0) || n < 0) return fail(EP_ERR_INVALID, "bad reset arguments");
  if (epc::bad_mem(mem)) return fail(EP_ERR_INVALID, "mem = %d is neither EP_MEM_HOST nor EP_MEM_DEVICE", mem);
  std::lock_guard<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  if ((rc = server_stop(p))) return rc;
  if (n == 0) return EP_OK;
  if (n > p->N) return fail(EP_ERR_INVALID, "reset of %d envs on a pool of %d", n, p->N);
  // (an env named twice in one reset is harmless: both items write the same record and the same rows)
  if (mem == EP_MEM_DEVICE) return enqueue_slices(p, env_ids, nullptr, nullptr, n, 0, true, nullptr, false);
  if ((rc = validate_ids_host(p, env_ids, n, nullptr))) return rc;
  int a = 0;
  if ((rc = stage_begin(p, &a))) return rc;
  char* h = p->h_stage + (size_t)a * p->stage_bytes;
  memcpy(h, env_ids, 4 * (size_t)n);
  void* stage_dev = nullptr;
  if (p->direct_host && n >= p->direct_min_rows && hipHostGetDevicePointer(&stage_dev, h, 0) == hipSuccess) {
    // host mode (see ep_pool_send): the kernel reads the ids from the pinned staging area itself
    rc = enqueue_slices(p, (const int32_t*)stage_dev, nullptr, nullptr, n, 0, true, nullptr, true, 0, /*host_mode=*/true, nullptr,
                        nullptr, 0, env_ids);
  } else {
    if ((rc = stage_commit(p, a, 4 * (size_t)n))) return rc;
    rc = enqueue_slices(p, (const int32_t*)(p->d_stage + (size_t)a * p->stage_bytes), nullptr, nullptr, n, 0, true, nullptr, true);
  }
  int rc2 = stage_release(p, a);
  return rc ? rc : rc2;
}

int ep_pool_send(ep_pool* p, int32_t n_arrays, const void* const* arrays, const int64_t* leading_dims,
                 int32_t mem) {
  if (!p || !arrays || !leading_dims) return fail(EP_ERR_INVALID, "null argument");
  if (epc::bad_mem(mem)) return fail(EP_ERR_INVALID, "mem = %d is neither EP_MEM_HOST nor EP_MEM_DEVICE", mem);
  if (n_arrays != 5)
    return fail(EP_ERR_INVALID, "Dummy expects 5 action arrays in _action_keys order, got %d", n_arrays);
  const int64_t n = leading_dims[0], np_rows = leading_dims[1];
  if (n < 0 || n > p->N) return fail(EP_ERR_INVALID, "env_id leading dim %lld not in [0,%d]", (long long)n, p->N);
  if (leading_dims[2] != n)
    return fail(EP_ERR_INVALID, "list_action leading dim %lld != env_id leading dim %lld",
                (long long)leading_dims[2], (long long)n);
  if (p->P == 1 && np_rows < n)
    return fail(EP_ERR_INVALID, "players.env_id has %lld rows, need %lld", (long long)np_rows, (long long)n);
  if (np_rows > (int64_t)p->N * p->P)
    return fail(EP_ERR_INVALID, "players.env_id has %lld rows, more than num_envs*max_num_players", (long long)np_rows);
  std::lock_guard<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  if ((rc = server_stop(p))) return rc;
  if (n == 0) return EP_OK;
  if (!arrays[0] || !arrays[1] || !arrays[2])
    return fail(EP_ERR_INVALID, "env_id, players.env_id and list_action must not be NULL");
  if (mem == EP_MEM_DEVICE) {
    if (((uintptr_t)arrays[0] & 3) || ((uintptr_t)arrays[1] & 3) || ((uintptr_t)arrays[2] & 7))
      return fail(EP_ERR_INVALID, "device action arrays must be aligned to their element size");
    return enqueue_slices(p, (const int32_t*)arrays[0], (const int32_t*)arrays[1], (const double*)arrays[2],
                          (int)n, (int)np_rows, false, nullptr, false);
  }
  if ((rc = validate_ids_host(p, (const int32_t*)arrays[0], (int)n, &p->seg_cuts))) return rc;
  const std::vector<int>* cuts = p->seg_cuts.empty() ? nullptr : &p->seg_cuts;
  int a = 0;
  if ((rc = stage_begin(p, &a))) return rc;
  char* h = p->h_stage + (size_t)a * p->stage_bytes;
  char* d = p->d_stage + (size_t)a * p->stage_bytes;
  size_t o0 = 0, o1 = align_up(4 * (size_t)n), o2 = o1 + align_up(4 * (size_t)np_rows);
  size_t total = o2 + 48 * (size_t)n;
  // Host mode: a send from host buffers, most likely to be received as host buffers (the numpy
  // step() / send() of the reference's API). No copy engine in either direction: the send goes in
  // pieces, each piece's ids copied into the pinned staging area and stepped by a kernel that
  // READS the staging area and WRITES its rows into the batch's pinned result block over PCIe
  // itself - the two directions and the host's work on the next piece overlap. (A later recv that
  // asks for device views copies the block into the ring.)
  {
    void* stage_dev = nullptr;
    if (p->direct_host && n >= p->direct_min_rows && hipHostGetDevicePointer(&stage_dev, h, 0) == hipSuccess) {
      const char* sd = (const char*)stage_dev;
      if (p->P > 1) {
        // multi-player: ParseAction scans the WHOLE players.env_id array for every env
        // (env.h:128-132), so the send stays in one piece
        memcpy(h + o0, arrays[0], 4 * (size_t)n);
        memcpy(h + o1, arrays[1], 4 * (size_t)np_rows);
        la_lanes_equal((const double*)arrays[2], (int)n, (uint32_t*)(h + o2));
        rc = enqueue_slices(p, (const int32_t*)(sd + o0), (const int32_t*)(sd + o1), nullptr, (int)n, (int)np_rows, false, cuts,
                            true, 0, /*host_mode=*/true, nullptr, (const uint32_t*)(sd + o2), 0);
        int rc2 = stage_release(p, a);
        return rc ? rc : rc2;
      }
      const int pieces = cuts ? 1 : n >= 32768 ? 4 : n >= 4096 ? 2 : 1;  // (a piece is worth its launch only when the host's share of it is)
      const int step = (((int)n + pieces - 1) / pieces + 255) & ~255;
      for (int c0 = 0; c0 < (int)n && rc == EP_OK; c0 += step) {
        const int cnt = std::min(step, (int)n - c0);
        memcpy(h + o0 + 4 * (size_t)c0, (const int32_t*)arrays[0] + c0, 4 * (size_t)cnt);
        memcpy(h + o1 + 4 * (size_t)c0, (const int32_t*)arrays[1] + c0, 4 * (size_t)cnt);
        // list_action only matters through "are its six lanes equal" (dummy_envpool.h:187-192):
        // the comparison is done here, while the 48 bytes per env pass through the host's cache
        // anyway, and one bit per env crosses PCIe instead (whether a failure counts - only on
        // the Step branch - is still decided by the kernel, which knows the env's state)
        la_lanes_equal((const double*)arrays[2] + 6 * (size_t)c0, cnt, (uint32_t*)(h + o2) + (c0 >> 5));
        rc = enqueue_slices(p, (const int32_t*)(sd + o0) + c0, (const int32_t*)(sd + o1) + c0, nullptr, cnt, cnt, false,
                            cuts, true, c0, /*host_mode=*/true, nullptr, (const uint32_t*)(sd + o2), c0, (const int32_t*)arrays[0] + c0);
      }
      int rc2 = stage_release(p, a);
      return rc ? rc : rc2;
    }
  }
  // players.action / players.id (arrays[3], arrays[4]) are never read by DummyEnv::Step
  // (dummy_envpool.h:173-210), so they are not shipped to the device.
  memcpy(h + o0, arrays[0], 4 * (size_t)n);
  memcpy(h + o1, arrays[1], 4 * (size_t)np_rows);
  constexpr size_t kChunk = 768 << 10;
  if (total <= kChunk) {
    memcpy(h + o2, arrays[2], 48 * (size_t)n);
    if ((rc = stage_commit(p, a, total))) return rc;
  } else {
    // large batches: the pageable -> pinned memcpy of one piece runs while the copy engine ships
    // the piece before it (the H2D copy used to start only after the whole memcpy)
    p->copy_dirty = true;
    HIP_TRY(hipMemcpyAsync(d, h, o2, hipMemcpyHostToDevice, p->copy_stream));
    const size_t la_bytes = 48 * (size_t)n;
    for (size_t off = 0; off < la_bytes; off += kChunk) {
      const size_t len = std::min(kChunk, la_bytes - off);
      memcpy(h + o2 + off, (const char*)arrays[2] + off, len);
      HIP_TRY(hipMemcpyAsync(d + o2 + off, h + o2 + off, len, hipMemcpyHostToDevice, p->copy_stream));
    }
    HIP_TRY(hipEventRecord(p->stage_copied[a], p->copy_stream));
    HIP_TRY(hipStreamWaitEvent(p->stream, p->stage_copied[a], 0));
  }
  rc = enqueue_slices(p, (const int32_t*)(d + o0), (const int32_t*)(d + o1), (const double*)(d + o2), (int)n,
                      (int)np_rows, false, cuts, true);
  int rc2 = stage_release(p, a);
  return rc ? rc : rc2;
}

int ep_pool_rollout(ep_pool* p, int32_t T, int32_t n_arrays, const void* const* arrays,
                    const int64_t* leading_dims, const int64_t* step_stride_rows) {
  if (!p || !arrays || !leading_dims || !step_stride_rows) return fail(EP_ERR_INVALID, "null argument");
  if (n_arrays != 5) return fail(EP_ERR_INVALID, "Dummy expects 5 action arrays, got %d", n_arrays);
  std::lock_guard<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  if ((rc = server_stop(p))) return rc;
  if (!p->is_sync)
    return fail(EP_ERR_INVALID, "rollout needs sync mode (batch_size == num_envs, max_num_players == 1)");
  if (T < 1 || T > p->Q) return fail(EP_ERR_INVALID, "rollout of %d steps needs 1 <= T <= ring_depth (%d)", T, p->Q);
  if (leading_dims[0] != p->N || leading_dims[1] != p->N || leading_dims[2] != p->N)
    return fail(EP_ERR_INVALID, "rollout steps ALL %d envs every step; got leading dims %lld/%lld/%lld", p->N,
                (long long)leading_dims[0], (long long)leading_dims[1], (long long)leading_dims[2]);
  if (step_stride_rows[0] != 0)
    return fail(EP_ERR_INVALID, "rollout keeps the env_id array fixed across steps (step stride 0)");
  if (!arrays[0] || !arrays[1] || !arrays[2])
    return fail(EP_ERR_INVALID, "env_id, players.env_id and list_action must not be NULL");
  if (((uintptr_t)arrays[0] & 3) || ((uintptr_t)arrays[1] & 3) || ((uintptr_t)arrays[2] & 7))
    return fail(EP_ERR_INVALID, "device action arrays must be aligned to their element size");
  if (p->poisoned) return fail(EP_ERR_HIP, "the pool is in a failed state after an earlier launch error");
  if (p->alloc_count % (uint64_t)p->B != 0)
    return fail(EP_ERR_INVALID, "rollout needs the queue on a batch boundary: receive the partial batch first");
  {
    uint64_t bi = p->alloc_count / (uint64_t)p->B;
    if (!slot_free_for(p, bi + (uint64_t)T - 1))
      return fail(EP_ERR_RING_FULL, "rollout of %d steps does not fit: %llu un-received batches in a ring of %d",
                  T, (unsigned long long)(bi - p->done_ptr), p->Q);
  }
  epk::RolloutArgs a;
  a.env_id = (const int32_t*)arrays[0];
  a.players_env_id = (const int32_t*)arrays[1];
  a.list_action = (const double*)arrays[2];
  a.penv_stride = step_stride_rows[1];
  a.la_stride = step_stride_rows[2];
  a.env = p->d_env;
  a.stats = p->stats;
  a.ring.ring = p->d_ring;
  a.ring.stride = p->slot_stride;
  for (int k = 0; k < 11; ++k) a.ring.off[k] = (uint32_t)p->key_off[k];
  a.ring.depth = p->Q;
  uint64_t bi0 = p->alloc_count / (uint64_t)p->B;
  for (int t = 0; t < T; ++t)
    if ((rc = claim_err_word(p, bi0 + (uint64_t)t))) return rc;
  a.slot0 = (int)(bi0 % (uint64_t)p->Q);
  a.err = p->d_err;  // a failed self-check is reported with the batch of the step that failed
  a.err0 = err_index(p, bi0);
  a.err_depth = p->EQ;
  a.stamp = p->d_stamp;
  a.stamp_val = next_stamp(p);
  a.pc = pool_const(p);
  a.T = T;
  // the wave kernels load list_action as 16-byte pieces (LDS-DMA dwordx4 / double2): a view that is
  // only 8-byte aligned, or a step stride that breaks the alignment, takes the generic kernel
  const bool wide_ok = ((uintptr_t)arrays[2] & 15) == 0 && ((step_stride_rows[2] * 48) & 15) == 0;
  p->last_rollout_kernel = "dummy_rollout_sp";
  LaunchEvents le;
  if ((rc = begin_launch(p, &le))) return rc;
  if (p->S == 10 && p->N % epk::kWave == 0 && !p->force_generic && wide_ok) {
    p->last_rollout_kernel = p->rollout_variant == 1 ? "dummy_rollout_wave" : p->rollout_variant == 2 ? "dummy_rollout_dma<4>" : "dummy_rollout_dma<6>";
    // wave-autonomous software-pipelined loop (dummy_rollout_wave.hpp)
    const dim3 g(p->N / epk::kWave), b(epk::kWave);
    if (p->rollout_variant == 1) {
      if (T % 4 == 0) hipExtLaunchKernelGGL(epk::dummy_rollout_wave<4>, g, b, 0, p->stream, nullptr, le.stop, 0, a);
      else if (T % 2 == 0) hipExtLaunchKernelGGL(epk::dummy_rollout_wave<2>, g, b, 0, p->stream, nullptr, le.stop, 0, a);
      else hipExtLaunchKernelGGL(epk::dummy_rollout_wave<1>, g, b, 0, p->stream, nullptr, le.stop, 0, a);
    } else if (p->rollout_variant == 2) {
      hipExtLaunchKernelGGL(epk::dummy_rollout_dma<4>, g, b, 0, p->stream, nullptr, le.stop, 0, a);
    } else {
      hipExtLaunchKernelGGL(epk::dummy_rollout_dma<6>, g, b, 0, p->stream, nullptr, le.stop, 0, a);
    }
  } else {
    dim3 grid((p->N + epk::kBlock - 1) / epk::kBlock), block(epk::kBlock);
    if (p->S == 10)
      hipExtLaunchKernelGGL(epk::dummy_rollout_sp<10>, grid, block, 0, p->stream, nullptr, le.stop, 0, a);
    else
      hipExtLaunchKernelGGL(epk::dummy_rollout_sp<0>, grid, block, 0, p->stream, nullptr, le.stop, 0, a);
  }
  if (hipError_t e = hipGetLastError(); e != hipSuccess) {
    abandon_launch(p, le);
    return fail(EP_ERR_HIP, "rollout launch failed: %s", hipGetErrorString(e));
  }
  const uint64_t lid = ++p->launch_seq;
  for (int t = 0; t < T; ++t) {
    int slot = (int)((bi0 + (uint64_t)t) % (uint64_t)p->Q);
    SlotHost& sh = p->slots[slot];
    sh.alloc_count += p->B;
    sh.shared_off += p->B;
    sh.written.emplace_back(0, p->B);
    sh.has_work = true;
    sh.term_ok = false;  // (the rollout kernels do not write d_term)
    sh.launch_id = lid;
  }
  end_launch(p, le, lid);
  p->alloc_count += (uint64_t)T * (uint64_t)p->B;
  p->stepping_env_num += (int64_t)T * p->B;
  return EP_OK;
}

}  // extern "C"

namespace {
// StateBuffer::Wait (state_buffer.h:135-140): is the head batch complete? `additional` is the
// credit AsyncEnvPool::Recv gives a sync pool whose outstanding env-steps do not fill a batch
// (async_envpool.h:163-166).
bool head_ready(ep_pool* p, int64_t* additional) {
  const SlotHost& sh = p->slots[(size_t)(p->done_ptr % (uint64_t)p->Q)];
  *additional = 0;
  if (p->is_sync && p->stepping_env_num < p->B) *additional = p->B - p->stepping_env_num;
  return (int64_t)sh.alloc_count + *additional == p->B;
}

// `consumer` != nullptr: stream-ordered form (ep_pool_recv_on_stream) - device views at once, the
// consumer stream is made to wait instead of the host. `lk` holds p->mu; it is released around
// every wait (for rows to be sent, for the GPU), so other threads can keep sending meanwhile -
// the reference's consumer blocks on a semaphore with no lock held (state_buffer.h:135-140).
int recv_locked(ep_pool* p, int32_t mem, ep_batch* out, std::unique_lock<std::mutex>& lk,
                hipStream_t* consumer = nullptr) {
  int rc = EP_OK;
  if ((rc = server_stop(p))) return rc;
  int64_t additional = 0;
  if (!head_ready(p, &additional)) {
    auto would_block = [&]() {
      return fail(EP_ERR_WOULD_BLOCK,
                  "recv would block: head batch has %llu of %d env-steps outstanding (the reference waits in "
                  "StateBuffer::Wait here; ep_pool_set_recv_timeout chooses how long this library does)",
                  (unsigned long long)p->slots[(size_t)(p->done_ptr % (uint64_t)p->Q)].alloc_count, p->B);
    };
    if (consumer || p->recv_timeout_ms == 0) return would_block();
    auto pred = [&]() { return head_ready(p, &additional) || p->poisoned; };
    if (p->recv_timeout_ms < 0) p->cv.wait(lk, pred);
    else if (!p->cv.wait_for(lk, std::chrono::milliseconds(p->recv_timeout_ms), pred)) return would_block();
    if (!head_ready(p, &additional)) return fail(EP_ERR_HIP, "the pool is in a failed state after an earlier launch error");
    EP_ENTER_DEVICE(p->device);
  }
  const uint64_t bi = p->done_ptr;
  const int slot = (int)(bi % (uint64_t)p->Q);
  const int ew = err_index(p, bi);
  SlotHost& sh = p->slots[slot];
  const uint32_t nrows = sh.shared_off;
  uint32_t nplayers = nrows;
  if (consumer && p->P > 1)
    return fail(EP_ERR_INVALID, "ep_pool_recv_on_stream needs max_num_players == 1 (the player row count of a "
                                "multi-player batch is only known on the device)");
  // host output of a single-player pool: the D2H copy below is ordered behind the kernels on the
  // step stream and one synchronisation at its end serves both (no host round trip in between)
  auto drop_target = [&]() {  // the pool is done with the caller's arrays of this batch
    if (sh.ext_home) sh.ext_home->drop(sh.ext_block);
    sh.ext_home.reset();
    sh.ext_block = nullptr;
    sh.ext = false;
  };
  if (sh.host_blk && (mem != EP_MEM_HOST || consumer)) {
    // the batch was written straight into a pinned block (direct host mode) but is wanted as
    // device views after all: bring it into its ring slot
    if (sh.ext) {
      for (int k = 0; k < 11; ++k)
        if (nrows) HIP_TRY(hipMemcpyAsync(p->d_ring + (size_t)slot * p->slot_stride + p->key_off[k],
                                          sh.ext_host[k] + (size_t)sh.ext_row0 * p->key_rowbytes[k],
                                          (size_t)nrows * p->key_rowbytes[k], hipMemcpyHostToDevice, p->stream));
    } else {
      HIP_TRY(hipMemcpyAsync(p->d_ring + (size_t)slot * p->slot_stride, sh.host_blk, p->slot_stride, hipMemcpyHostToDevice, p->stream));
    }
    if ((rc = sync_stream_unlocked(p, &lk))) return rc;
    if (!sh.ext) epc::give_block(p->blocks, sh.host_blk);
    sh.host_blk = nullptr;
  }
  if (sh.ext && (mem != EP_MEM_HOST || consumer)) drop_target();  // received as device views: the target stays as it was
  const bool filled_blk = sh.host_blk != nullptr;  // (still set: wanted as host memory, by the host)
  const bool one_sync = mem == EP_MEM_HOST && p->P == 1 && !consumer && !filled_blk;
  if (sh.has_work && !consumer && !one_sync)
    if ((rc = wait_launch(p, sh.launch_id, &lk))) return rc;
  if (p->P > 1) nplayers = sh.mp_word >= 0 ? p->h_slot_players[sh.mp_word] : 0;
  // rows inside [0, nrows) that no kernel wrote must read as zero (fresh zeroed StateBuffer,
  // array.h:77-81); only the reference's "several sync sends before one recv" corner gets here.
  if (p->P == 1 && nrows > 0) {
    std::sort(sh.written.begin(), sh.written.end());
    int64_t cur = 0;
    bool any = false;
    auto zero_rows = [&](int64_t a, int64_t b) -> int {
      if (filled_blk) {  // the kernels of this batch are done (waited for above): plain host stores
        for (int k = 0; k < 11; ++k)
          memset((sh.ext ? sh.ext_host[k] + (size_t)sh.ext_row0 * p->key_rowbytes[k] : (char*)sh.host_blk + p->key_off[k]) +
                     (size_t)a * p->key_rowbytes[k],
                 0, (size_t)(b - a) * p->key_rowbytes[k]);
        return EP_OK;
      }
      p->stream_extra = true;
      for (int k = 0; k < 11; ++k) {
        char* base = p->d_ring + (size_t)slot * p->slot_stride + p->key_off[k];
        HIP_TRY(hipMemsetAsync(base + (size_t)a * p->key_rowbytes[k], 0, (size_t)(b - a) * p->key_rowbytes[k], p->stream));
      }
      HIP_TRY(hipMemsetAsync(p->d_term + (size_t)slot * p->term_stride + (size_t)a, 0, (size_t)(b - a), p->stream));
      any = true;
      return EP_OK;
    };
    for (auto& w : sh.written) {
      if (w.first > cur && cur < (int64_t)nrows)
        if ((rc = zero_rows(cur, std::min<int64_t>(w.first, nrows)))) return rc;
      cur = std::max(cur, w.second);
    }
    if (cur < (int64_t)nrows)
      if ((rc = zero_rows(cur, nrows))) return rc;
    if (any && !consumer && !one_sync)
      if ((rc = sync_stream_unlocked(p, &lk))) return rc;
  }
  int status = EP_OK;
  if (consumer) {
    // everything this batch needs is already on the step stream: another stream waits for it
    // there, the step stream itself is in order anyway. The batch's error word is looked at later
    // (ep_pool_sync), when the kernels have run.
    if (*consumer != p->stream) {
      HIP_TRY(hipEventRecord(p->recv_ev, p->stream));
      HIP_TRY(hipStreamWaitEvent(*consumer, p->recv_ev, 0));
    }
    p->deferred_err[ew] = 1;
    p->deferred_lid[ew] = sh.launch_id;
    // nobody waits for launches on the host in this mode: retire their completion events here,
    // which also keeps the host from running more than about two rings ahead of the device
    while (p->inflight.size() > (size_t)(2 * p->Q + 8)) {
      const uint64_t lid = p->inflight.front().first;
      if ((rc = wait_launch(p, lid, &lk))) return rc;
    }
  }
  if (!consumer && !one_sync) status = status_of_err_bits(take_err_bits(p, ew), "recv");

  memset(out, 0, sizeof(*out));
  out->num_keys = 12;
  out->mem = mem;
  out->num_rows = (int32_t)nrows;
  out->num_player_rows = (int32_t)nplayers;
  out->seq = p->recv_seq++;
  char* base = nullptr;
  if (mem == EP_MEM_HOST) {
    void* blk = sh.host_blk;  // direct host mode: the kernels are writing (or wrote) it already
    sh.host_blk = nullptr;
    const bool filled = blk != nullptr;
    if (!filled && !sh.ext && (rc = epc::take_block(p->blocks, &blk))) return rc;
    char* src = p->d_ring + (size_t)slot * p->slot_stride;
    if (filled) {
      // nothing to copy, and the launches that wrote it were waited for above
    } else if (sh.ext) {  // the caller's arrays, filled from the ring slot
      for (int k = 0; k < 11; ++k)
        if (nrows) HIP_TRY(hipMemcpyAsync(sh.ext_host[k] + (size_t)sh.ext_row0 * p->key_rowbytes[k], src + p->key_off[k],
                                          (size_t)nrows * p->key_rowbytes[k], hipMemcpyDeviceToHost, p->stream));
    } else if ((size_t)nrows * 2 >= (size_t)p->B) {
      HIP_TRY(hipMemcpyAsync(blk, src, p->slot_stride, hipMemcpyDeviceToHost, p->stream));
    } else {
      for (int k = 0; k < 11; ++k) {
        size_t r = p->key_is_player[k] ? nplayers : nrows;
        if (r) HIP_TRY(hipMemcpyAsync((char*)blk + p->key_off[k], src + p->key_off[k], r * p->key_rowbytes[k], hipMemcpyDeviceToHost, p->stream));
      }
    }
    if (!filled && (rc = sync_stream_unlocked(p, &lk))) { if (!sh.ext) epc::give_block(p->blocks, blk); return rc; }
    if (one_sync) status = status_of_err_bits(take_err_bits(p, ew), "recv");
    if (!sh.ext) out->owner = new HostBlockOwner{p->blocks, blk};  // (the caller's arrays: no owner, ep_batch_release has nothing to do)
    base = (char*)blk;
  } else {
    base = p->d_ring + (size_t)slot * p->slot_stride;
  }
  for (int k = 0; k < 12; ++k) {
    ep_array& a = out->arrays[k];
    int sk = kStateToSlot[k];
    if (sk < 0) {  // obs:dyn — implicit: row r is full((info:players.env_id[r]+1, S), that id)
      a.data = nullptr;
      a.dtype = EP_CONTAINER_I32;
      a.ndim = 1;
      a.shape[0] = nplayers;
      continue;
    }
    a.data = (sh.ext && mem == EP_MEM_HOST) ? sh.ext_host[sk] + (size_t)sh.ext_row0 * p->key_rowbytes[sk] : base + p->key_off[sk];
    a.dtype = kSlotDtype[sk];
    a.ndim = 1;
    a.shape[0] = p->key_is_player[sk] ? nplayers : nrows;
    if (sk == 8) { a.ndim = 2; a.shape[1] = p->S; }
  }
  // StateBufferQueue::Wait tail (state_buffer_queue.h:141-149) + AsyncEnvPool::Recv (:170-172)
  p->done_ptr++;
  if (additional > 0) p->alloc_count += (uint64_t)additional;
  sh.alloc_count = 0;
  sh.shared_off = 0;
  sh.written.clear();
  sh.has_work = false;
  sh.mp_word = -1;
  if (sh.ext) drop_target();
  p->slot_term_seq[slot] = (sh.term_ok && mem == EP_MEM_DEVICE && p->P == 1) ? out->seq + 1 : 0;
  sh.term_ok = false;
  if (p->is_sync) p->stepping_env_num -= nrows;
  return status;
}
}  // namespace

extern "C" {

int ep_pool_recv(ep_pool* p, int32_t mem, ep_batch* out) {
  if (!p || !out) return fail(EP_ERR_INVALID, "null argument");
  if (epc::bad_mem(mem)) return fail(EP_ERR_INVALID, "mem = %d is neither EP_MEM_HOST nor EP_MEM_DEVICE", mem);
  std::unique_lock<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  return recv_locked(p, mem, out, lk);
}

int ep_pool_recv_on_stream(ep_pool* p, void* consumer_stream, ep_batch* out) {
  if (!p || !out) return fail(EP_ERR_INVALID, "null argument");
  std::unique_lock<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  hipStream_t cs = consumer_stream ? (hipStream_t)consumer_stream : p->stream;
  return recv_locked(p, EP_MEM_DEVICE, out, lk, &cs);
}

int ep_pool_recv_many_on_stream(ep_pool* p, void* consumer_stream, int32_t n, ep_batch* out) {
  if (!p || !out || n < 0) return fail(EP_ERR_INVALID, "bad argument");
  std::unique_lock<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  hipStream_t cs = consumer_stream ? (hipStream_t)consumer_stream : p->stream;
  for (int i = 0; i < n; ++i)
    if ((rc = recv_locked(p, EP_MEM_DEVICE, out + i, lk, &cs))) return rc;
  return EP_OK;
}

int ep_pool_set_stream(ep_pool* p, void* stream, int32_t own) {
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  std::unique_lock<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  if (p->step_mode != 0) return fail(EP_ERR_INVALID, "ep_pool_set_stream between ep_pool_step_begin and ep_pool_step_end");
  hipStream_t next = own ? p->own_stream : (hipStream_t)stream;
  if (next == p->stream) return EP_OK;
  if ((rc = server_stop(p))) return rc;
  // everything enqueued so far finishes on the stream it was enqueued on: launches, completion
  // events and timing events all refer to that stream
  // (the lock is released around the wait: whatever another thread enqueued meanwhile is waited for too)
  for (uint64_t upto = ~0ull; upto != p->launch_seq;) {
    upto = p->launch_seq;
    if ((rc = sync_stream_unlocked(p, &lk))) return rc;
  }
  if ((rc = drain_timed(p))) return rc;
  if (p->step_mode != 0) return fail(EP_ERR_INVALID, "ep_pool_set_stream raced with ep_pool_step_begin");
  p->stream = next;
  return EP_OK;
}

int ep_pool_send_on_stream(ep_pool* p, void* stream, int32_t n_arrays, const void* const* arrays,
                           const int64_t* leading_dims) {
  if (!p) return fail(EP_ERR_INVALID, "null argument");
  {
    std::unique_lock<std::mutex> lk(p->mu);
    EP_ENTER_DEVICE(p->device);
    [[maybe_unused]] int rc = EP_OK;
    hipStream_t cs = stream ? (hipStream_t)stream : p->stream;  // (NULL: the pool's step stream, as for ep_pool_recv_on_stream)
    if (cs != p->stream) {
      // the actions (and the consumer's reads of the slot this step will refill) come first
      if (!p->producer_ev) HIP_TRY(hipEventCreateWithFlags(&p->producer_ev, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(p->producer_ev, cs));
      HIP_TRY(hipStreamWaitEvent(p->stream, p->producer_ev, 0));
    }
  }
  return ep_pool_send(p, n_arrays, arrays, leading_dims, EP_MEM_DEVICE);
}

int ep_pool_step_on_stream(ep_pool* p, void* stream, int32_t n_arrays, const void* const* arrays,
                           const int64_t* leading_dims, ep_batch* out) {
  if (!p || !out) return fail(EP_ERR_INVALID, "null argument");
  if (p->P > 1) return fail(EP_ERR_INVALID, "ep_pool_step_on_stream needs max_num_players == 1 (see ep_pool_recv_on_stream)");
  int rc = ep_pool_send_on_stream(p, stream, n_arrays, arrays, leading_dims);
  if (rc) return rc;
  return ep_pool_recv_on_stream(p, stream, out);
}

int ep_pool_terminated_of(ep_pool* p, const ep_batch* b, void** term) {
  if (!p || !b || !term) return fail(EP_ERR_INVALID, "null argument");
  *term = nullptr;
  std::lock_guard<std::mutex> lk(p->mu);
  const char* first = (const char*)b->arrays[0].data;
  if (b->mem != EP_MEM_DEVICE || !first || first < p->d_ring || first >= p->d_ring + p->slot_stride * (size_t)p->Q)
    return fail(EP_ERR_INVALID, "not a batch of device views of this pool");
  const size_t slot = (size_t)(first - p->d_ring) / p->slot_stride;
  if (p->slot_term_seq[slot] != b->seq + 1)
    return fail(EP_ERR_INVALID, "no terminated array for this batch (its rows came from a rollout, from host memory or a "
                                "multi-player pool, or the slot has been handed out again): use ep_terminated_many");
  *term = p->d_term + slot * p->term_stride;
  return EP_OK;
}

int ep_pool_poll_errors(ep_pool* p) {
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  std::lock_guard<std::mutex> lk(p->mu);
  // the words are host memory the kernels write: whatever has arrived is reported, what has not
  // is seen by the next poll (or by ep_pool_sync, which waits first)
  uint32_t bits = p->sticky_err;
  p->sticky_err = 0;
  for (int w = 0; w < p->EQ; ++w) {
    if (!p->deferred_err[w]) continue;
    epk::ErrWords& e = p->h_err[w];
    if (__atomic_load_n(&e.action_check, __ATOMIC_RELAXED)) { bits |= 1u; e.action_check = 0; }
    if (__atomic_load_n(&e.env_id_range, __ATOMIC_RELAXED)) { bits |= 2u; e.env_id_range = 0; }
    if (__atomic_load_n(&e.dup_env, __ATOMIC_RELAXED)) { bits |= 4u; e.dup_env = 0; }
    if (p->deferred_lid[w] <= p->completed_launch) p->deferred_err[w] = 0;  // its kernels are known to be through
  }
  return status_of_err_bits(bits, "poll");
}

int ep_pool_recv_many(ep_pool* p, int32_t mem, int32_t n, ep_batch* out) {
  if (!p || !out || n < 0) return fail(EP_ERR_INVALID, "bad argument");
  if (epc::bad_mem(mem)) return fail(EP_ERR_INVALID, "mem = %d is neither EP_MEM_HOST nor EP_MEM_DEVICE", mem);
  std::unique_lock<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  int status = EP_OK;
  std::string first_msg;
  for (int i = 0; i < n; ++i) {
    rc = recv_locked(p, mem, out + i, lk);
    if (rc == EP_ERR_ACTION_CHECK || rc == EP_ERR_DUP_ENV) {  // the batch is there; keep going, report the first
      if (status == EP_OK) { status = rc; first_msg = epc::g_last_error + " (batch " + std::to_string(i) + " of this call)"; }
      continue;
    }
    if (rc != EP_OK) return rc;
  }
  if (status != EP_OK) epc::g_last_error = first_msg;
  return status;
}

int ep_pool_step_begin(ep_pool* p, int32_t n_arrays, const void* const* arrays, const int64_t* leading_dims) {
  if (!p || !arrays || !leading_dims) return fail(EP_ERR_INVALID, "null argument");
  if (n_arrays != 5) return fail(EP_ERR_INVALID, "Dummy expects 5 action arrays in _action_keys order, got %d", n_arrays);
  {
    std::unique_lock<std::mutex> lk(p->mu);
    EP_ENTER_DEVICE(p->device);
    [[maybe_unused]] int rc = EP_OK;
    if (p->step_mode != 0) return fail(EP_ERR_INVALID, "the previous ep_pool_step_begin has not been ended");
    const bool whole = leading_dims[0] == p->N && leading_dims[1] == p->N && leading_dims[2] == p->N;
    const bool aligned = arrays[0] && arrays[1] && arrays[2] && !((uintptr_t)arrays[0] & 3) && !((uintptr_t)arrays[1] & 3) &&
                         !((uintptr_t)arrays[2] & 15);
    const bool idle = p->alloc_count == p->done_ptr * (uint64_t)p->B && p->stepping_env_num == 0;
    bool serve = p->use_server && !p->force_generic && p->is_sync && whole && aligned && idle && p->N % epk::kBlock == 0 && !p->poisoned;
    const auto now = std::chrono::steady_clock::now();
    if (serve) {
      // the server's patience follows the caller's cadence: about ten gaps between consecutive
      // steps (whichever path served them), 200 us .. 50 ms. A gap that holds an idle stop is not a
      // cadence: the wait of a caller that synchronises the device is the idle limit itself, and
      // learning from it would lengthen the next one.
      if (!p->gap_tainted) {
        const int64_t gap = std::chrono::duration_cast<std::chrono::microseconds>(now - p->last_ring).count();
        p->idle_limit_us = std::min<int64_t>(50000, std::max<int64_t>(200, 10 * gap));
      }
      p->gap_tainted = false;
      p->last_ring = now;
    }
    if (serve && !p->server_running &&
        (g_servers_running[p->device & 63].load() >= max_servers_per_device() ||
         (g_pools_on_device[p->device & 63].load() > 1 && !p->server_shared_ok)))
      serve = false;  // (several pools on one GPU: the launch path)
    if (serve && !p->server_running && p->server_backoff > 0) {
      --p->server_backoff;  // the server ran idle lately (server_watchdog): the launch path for now
      serve = false;
    }
    if (serve) {
      // ---- the resident server: no launch ---------------------------------------------
      if (p->server_running && __atomic_load_n(&p->h_ctl->alive, __ATOMIC_ACQUIRE) == 0) {
        // its own poll limit ran out (the watchdog should have stopped it long before): wait for the kernel to end
        HIP_TRY(hipStreamSynchronize(p->server_stream));
        p->server_running = false;
        g_servers_running[p->device & 63].fetch_sub(1);
      }
      if (p->server_running && (p->server_arrays[0] != arrays[0] || p->server_arrays[1] != arrays[1] || p->server_arrays[2] != arrays[2]))
        if ((rc = server_stop(p))) return rc;  // other action arrays: the server is started again for them
      const uint64_t bi = p->alloc_count / (uint64_t)p->B;
      if ((rc = claim_err_word(p, bi))) return rc;
      const bool warm = p->server_running;
      if (!warm && (rc = server_start(p, arrays))) return rc;
      p->last_ring = std::chrono::steady_clock::now();
      p->step_warm = warm;
      p->step_bi = bi;
      ring_bell(p, (unsigned long long)++p->server_seq);
      p->step_mode = 1;
      return EP_OK;
    }
  }
  // ---- anything else: one launch per step (send now, blocking recv in ep_pool_step_end) -----------
  int rc = ep_pool_send(p, n_arrays, arrays, leading_dims, EP_MEM_DEVICE);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(p->mu);
  p->step_mode = 2;
  return EP_OK;
}

int ep_pool_step_end(ep_pool* p, ep_batch* out) {
  if (!p || !out) return fail(EP_ERR_INVALID, "null argument");
  {
    std::unique_lock<std::mutex> lk(p->mu);
    EP_ENTER_DEVICE(p->device);
    [[maybe_unused]] int rc = EP_OK;
    if (p->step_mode == 0) return fail(EP_ERR_INVALID, "ep_pool_step_end without ep_pool_step_begin");
    if (p->step_mode == 1) {
      const uint32_t seq = p->server_seq;
      const int ns = p->server_shards;
      auto all_done = [&]() {
        for (int s = 0; s < ns; ++s)
          if (__atomic_load_n(&p->h_ctl->done[s], __ATOMIC_ACQUIRE) != seq) return false;
        return true;
      };
      const auto t0 = std::chrono::steady_clock::now();
      for (uint32_t spins = 0;; ++spins) {
        if (all_done()) break;
        if ((spins & 1023u) == 1023u) {
          if (__atomic_load_n(&p->h_ctl->alive, __ATOMIC_ACQUIRE) == 0) {
            // the kernel's own poll limit ran out as the doorbell rang: wait for the kernel to end
            HIP_TRY(hipStreamSynchronize(p->server_stream));
            p->server_running = false;
            g_servers_running[p->device & 63].fetch_sub(1);
            if (all_done()) break;
            if (p->bar_bell) {
              // (each workgroup decides for itself there: some may have served the step, some not)
              p->step_mode = 0;
              p->poisoned = true;
              return fail(EP_ERR_HIP, "the step server gave up waiting while a step was rung: the pool's state is undefined");
            }
            // leader wake-up: the leader decides for all, so the step was not served at all: a new server for this very step
            --p->server_seq;
            p->step_mode = 0;
            if ((rc = server_start(p, p->server_arrays))) return rc;
            p->step_mode = 1;
            ring_bell(p, (unsigned long long)++p->server_seq);
          }
          if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
            p->step_mode = 0;
            p->poisoned = true;
            return fail(EP_ERR_HIP, "the step server did not answer within 5 s");
          }
        }
#if defined(__x86_64__)
        if (p->spin_pause) __builtin_ia32_pause();
#endif
      }
      p->step_mode = 0;
      ++p->server_steps;
      if (p->step_warm) { p->idle_streak = 0; p->server_fresh = false; }
      const uint64_t bi = p->step_bi;
      const int slot = (int)(bi % (uint64_t)p->Q);
      const int status = status_of_err_bits(take_err_bits(p, err_index(p, bi)), "step");
      p->alloc_count += (uint64_t)p->B;  // one whole batch allocated ...
      p->done_ptr += 1;                  // ... and received
      memset(out, 0, sizeof(*out));
      out->num_keys = 12;
      out->mem = EP_MEM_DEVICE;
      out->num_rows = p->N;
      out->num_player_rows = p->N;
      out->seq = p->recv_seq++;
      p->slot_term_seq[slot] = out->seq + 1;  // (the resident kernel writes d_term as well)
      char* base = p->d_ring + (size_t)slot * p->slot_stride;
      for (int k = 0; k < 12; ++k) {
        ep_array& a = out->arrays[k];
        const int sk = kStateToSlot[k];
        a.ndim = 1;
        a.shape[0] = p->N;
        if (sk < 0) { a.data = nullptr; a.dtype = EP_CONTAINER_I32; continue; }
        a.data = base + p->key_off[sk];
        a.dtype = kSlotDtype[sk];
        if (sk == 8) { a.ndim = 2; a.shape[1] = p->S; }
      }
      return status;
    }
    p->step_mode = 0;
  }
  return ep_pool_recv(p, EP_MEM_DEVICE, out);
}

int ep_pool_step(ep_pool* p, int32_t n_arrays, const void* const* arrays, const int64_t* leading_dims, ep_batch* out) {
  if (!out) return fail(EP_ERR_INVALID, "null argument");
  int rc = ep_pool_step_begin(p, n_arrays, arrays, leading_dims);
  if (rc) return rc;
  return ep_pool_step_end(p, out);
}

int ep_pool_server_trace(ep_pool* p, uint64_t* out, int32_t max_workgroups) {
  if (!p || !out) return fail(EP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  if (!p->d_server_trace) return fail(EP_ERR_INVALID, "no trace: create the pool with EP_SERVER_TRACE=1 in the environment");
  if ((rc = server_stop(p))) return rc;
  const int n = std::min(std::min(max_workgroups, p->server_wgs), 1024);
  if (n > 0) HIP_TRY(hipMemcpy(out, p->d_server_trace, 32 * (size_t)n, hipMemcpyDeviceToHost));
  return n < 0 ? 0 : -n;  // (negative count: the number of workgroups written; 0 = none)
}

int ep_pool_server_stats(ep_pool* p, uint64_t* launches, uint64_t* idle_stops, uint64_t* steps_served, int64_t* idle_limit_us) {
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  std::lock_guard<std::mutex> lk(p->mu);
  if (launches) *launches = p->server_launches;
  if (idle_stops) *idle_stops = p->server_idle_stops;
  if (steps_served) *steps_served = p->server_steps;
  if (idle_limit_us) *idle_limit_us = p->idle_limit_us;
  return EP_OK;
}

int ep_pool_set_recv_timeout(ep_pool* p, int64_t milliseconds) {
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  std::lock_guard<std::mutex> lk(p->mu);
  p->recv_timeout_ms = milliseconds;
  return EP_OK;
}

int ep_pool_episode_stats_async(ep_pool* p, float* fin_return, int32_t* fin_length, int32_t* ep_count,
                                void* consumer_stream) {
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  std::lock_guard<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  if ((rc = server_stop(p))) return rc;
  hipStream_t cs = (hipStream_t)consumer_stream;
  // destination buffers may still be read by the consumer's previous work (e.g. an all-gather)
  HIP_TRY(hipEventRecord(p->consumer_ev, cs));
  HIP_TRY(hipStreamWaitEvent(p->stream, p->consumer_ev, 0));
  p->stream_extra = true;
  hipLaunchKernelGGL(epk::snapshot_stats, dim3((p->N + 255) / 256), dim3(256), 0, p->stream, p->stats, p->N,
                     fin_return, fin_length, ep_count);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(p->stats_ev, p->stream));
  HIP_TRY(hipStreamWaitEvent(cs, p->stats_ev, 0));
  return EP_OK;
}

int ep_host_blocks_create(size_t bytes, ep_host_blocks** out) {
  if (!out || bytes == 0) return fail(EP_ERR_INVALID, "ep_host_blocks_create: null result pointer or zero block size");
  auto* hb = new ep_host_blocks;
  hb->set = std::make_shared<HostBlockSet>();
  hb->set->bytes = bytes;
  *out = hb;
  return EP_OK;
}

int ep_host_blocks_take(ep_host_blocks* hb, void** block) {
  if (!hb || !block) return fail(EP_ERR_INVALID, "null argument");
  void* blk = nullptr;
  {
    std::lock_guard<std::mutex> lk(hb->set->mu);
    if (!hb->set->free_list.empty()) {
      blk = hb->set->free_list.back();
      hb->set->free_list.pop_back();
    }
  }
  // portable: every device of the process can write it, whichever was current when it was made
  if (!blk) HIP_TRY(hipHostMalloc(&blk, hb->set->bytes, hipHostMallocMapped | hipHostMallocPortable));
  hb->set->hold(blk);
  *block = blk;
  return EP_OK;
}

int ep_host_blocks_give(ep_host_blocks* hb, void* block) {
  if (!hb || !block) return fail(EP_ERR_INVALID, "null argument");
  if (!hb->set->held(block)) return fail(EP_ERR_INVALID, "ep_host_blocks_give: not a block that is out of this set");
  hb->set->drop(block);
  return EP_OK;
}

int ep_host_blocks_destroy(ep_host_blocks* hb) {
  if (!hb) return EP_OK;
  {
    std::lock_guard<std::mutex> lk(hb->set->mu);
    hb->set->alive = false;
    for (void* b : hb->set->free_list) (void)hipHostFree(b);
    hb->set->free_list.clear();
  }
  delete hb;  // (blocks still held - by the caller's arrays or by a pool - are freed when the last hold goes)
  return EP_OK;
}

int ep_pool_host_target(ep_pool* p, ep_host_blocks* hb, void* block, const size_t* key_off, int64_t row_offset) {
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  std::lock_guard<std::mutex> lk(p->mu);
  if (!hb && !block) {  // cancel
    if (p->target_home) p->target_home->drop(p->target_block);
    p->target_home.reset();
    p->target_block = nullptr;
    p->target_pending = false;
    return EP_OK;
  }
  if (!hb || !block || !key_off) return fail(EP_ERR_INVALID, "ep_pool_host_target: null argument");
  if (p->P > 1)
    return fail(EP_ERR_INVALID, "ep_pool_host_target needs max_num_players == 1 (the player rows of a multi-player batch have no "
                                "fixed place)");
  if (!hb->set->held(block)) return fail(EP_ERR_INVALID, "ep_pool_host_target: not a block that is out of this set");
  if (row_offset < 0 || row_offset > (int64_t)1 << 30) return fail(EP_ERR_INVALID, "ep_pool_host_target: row_offset = %lld", (long long)row_offset);
  EP_ENTER_DEVICE(p->device);
  char* host[11];
  char* dev[11];
  for (int k = 0; k < 11; ++k) {
    const size_t need = ((size_t)row_offset + (size_t)p->B) * p->key_rowbytes[k];
    if (key_off[k] > hb->set->bytes || need > hb->set->bytes - key_off[k])
      return fail(EP_ERR_INVALID, "ep_pool_host_target: key %d needs %zu bytes at offset %zu of a %zu-byte block", k, need, key_off[k],
                  hb->set->bytes);
    // (the kernels choose their 16-byte stores by the ROW they start at: the arrays themselves must be 16-byte aligned)
    if (key_off[k] % 16 != 0) return fail(EP_ERR_INVALID, "ep_pool_host_target: key %d at offset %zu is not 16-byte aligned", k, key_off[k]);
    host[k] = (char*)block + key_off[k];
  }
  void* dv = nullptr;
  if (hipHostGetDevicePointer(&dv, block, 0) != hipSuccess) {
    (void)hipGetLastError();
    return fail(EP_ERR_HIP, "ep_pool_host_target: the block is not mapped into device %d", p->device);
  }
  for (int k = 0; k < 11; ++k) dev[k] = (char*)dv + key_off[k];
  if (p->target_home) p->target_home->drop(p->target_block);  // a target nobody used is replaced
  hb->set->hold(block);
  p->target_home = hb->set;
  p->target_block = block;
  memcpy(p->target_host, host, sizeof(host));
  memcpy(p->target_dev, dev, sizeof(dev));
  p->target_row0 = row_offset;
  p->target_pending = true;
  return EP_OK;
}

int ep_batch_release(ep_batch* b) {
  if (!b || !b->owner) return EP_OK;
  HostBlockOwner* o = (HostBlockOwner*)b->owner;
  {
    std::lock_guard<std::mutex> lk(o->home->mu);
    if (o->home->alive && o->home->free_list.size() < 8) o->home->free_list.push_back(o->ptr);
    else (void)hipHostFree(o->ptr);
  }
  delete o;
  b->owner = nullptr;
  return EP_OK;
}

int ep_pool_sync(ep_pool* p) {
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  std::unique_lock<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  if ((rc = server_stop(p))) return rc;
  if (p->copy_dirty) {
    p->copy_dirty = false;
    lk.unlock();
    hipError_t e = hipStreamSynchronize(p->copy_stream);
    lk.lock();
    if (e != hipSuccess) return fail(EP_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
  }
  // (a stream whose every launch is already known to have finished is not synchronised again:
  // hipStreamSynchronize costs 8-10 us even then)
  if (p->stream_extra || p->completed_launch < p->launch_seq)
    if ((rc = sync_stream_unlocked(p, &lk))) return rc;
  // batches handed out by ep_pool_recv_on_stream: their error words are final now
  uint32_t bits = p->sticky_err;
  p->sticky_err = 0;
  for (int w = 0; w < p->EQ; ++w)
    if (p->deferred_err[w]) bits |= take_err_bits(p, w);
  return status_of_err_bits(bits, "a batch received with ep_pool_recv_on_stream");
}

int ep_pool_episode_stats(ep_pool* p, float* fin_return, int32_t* fin_length, int32_t* ep_count, int32_t mem) {
  if (epc::bad_mem(mem)) return fail(EP_ERR_INVALID, "mem = %d is neither EP_MEM_HOST nor EP_MEM_DEVICE", mem);
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  std::lock_guard<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  if ((rc = server_stop(p))) return rc;
  hipMemcpyKind kind = mem == EP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  size_t bytes = 4 * (size_t)p->N;
  if (fin_return) HIP_TRY(hipMemcpyAsync(fin_return, p->stats.fin_return, bytes, kind, p->stream));
  if (fin_length) HIP_TRY(hipMemcpyAsync(fin_length, p->stats.fin_length, bytes, kind, p->stream));
  if (ep_count) HIP_TRY(hipMemcpyAsync(ep_count, p->stats.ep_count, bytes, kind, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return EP_OK;
}

int64_t ep_pool_checkpoint_bytes(ep_pool* p) {
  if (!p) return 0;
  return (int64_t)((sizeof(epk::EnvRec) + 16) * (size_t)p->N);
}

int ep_pool_checkpoint_save(ep_pool* p, void* buf) {
  if (!p || !buf) return fail(EP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  if ((rc = server_stop(p))) return rc;
  char* o = (char*)buf;
  size_t n = (size_t)p->N;
  HIP_TRY(hipStreamSynchronize(p->stream));
  HIP_TRY(hipMemcpy(o, p->d_env, 8 * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(o + 8 * n, p->stats.ep_return, 4 * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(o + 12 * n, p->stats.fin_return, 4 * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(o + 16 * n, p->stats.fin_length, 4 * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(o + 20 * n, p->stats.ep_count, 4 * n, hipMemcpyDeviceToHost));
  return EP_OK;
}

int ep_pool_checkpoint_restore(ep_pool* p, const void* buf) {
  if (!p || !buf) return fail(EP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  if ((rc = server_stop(p))) return rc;
  const char* o = (const char*)buf;
  size_t n = (size_t)p->N;
  HIP_TRY(hipStreamSynchronize(p->stream));
  HIP_TRY(hipMemcpy(p->d_env, o, 8 * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->stats.ep_return, o + 8 * n, 4 * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->stats.fin_return, o + 12 * n, 4 * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->stats.fin_length, o + 16 * n, 4 * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->stats.ep_count, o + 20 * n, 4 * n, hipMemcpyHostToDevice));
  return EP_OK;
}

// ---- ep_dummy_send_noalloc ------------------------------------------------------------------
// The one entry point that allocates nothing (include/envpool_amd.h). Its kernels are the pool's
// (mp_count_rows, mp_step_fused); their scratch - histogram counters, look-back granules, the tile
// ticket - lives in this file-scope device array instead of a pool's allocation. `scratch` is a
// host-side stand-in for the symbol, never a device address: the per-device address comes from
// hipGetSymbolAddress, once per device, and only that pointer is handed to kernels and memsets.
}  // extern "C"

namespace {
constexpr int kNoallocMaxEnvs = EP_NOALLOC_MAX_ENVS;
constexpr int kNoallocMaxTiles = (kNoallocMaxEnvs + epk::kMpTile - 1) / epk::kMpTile;
constexpr size_t kNoallocHead = 256;  // ticket word, padded
constexpr size_t kNoallocTilesOff = kNoallocHead;
constexpr size_t kNoallocCountsOff = kNoallocTilesOff + 8 * (size_t)kNoallocMaxTiles;
constexpr size_t kNoallocBytes = kNoallocCountsOff + 8 * (size_t)kNoallocMaxEnvs;
}  // namespace
static __device__ __align__(256) unsigned char scratch[kNoallocBytes];
namespace {
constexpr int kNoallocMaxDevices = 64;
std::mutex g_noalloc_mu;
unsigned char* g_noalloc_ptr[kNoallocMaxDevices];  // per device: where ITS copy of `scratch` lives
int g_noalloc_cus[kNoallocMaxDevices];             // per device: compute units
}  // namespace

extern "C" {

int ep_dummy_send_noalloc(const ep_dummy_config* cfg, void* env_state, int32_t n, const int32_t* env_id,
                          int32_t np_rows, const int32_t* players_env_id, const double* list_action,
                          int32_t force_reset, const ep_array* out, int32_t n_out, ep_noalloc_result* result,
                          void* stream) {
  if (!cfg || !env_state || !out || !result || (n > 0 && !env_id))
    return fail(EP_ERR_INVALID, "null argument");
  const int N = cfg->num_envs, P = cfg->max_num_players, S = cfg->state_num;
  if (N < 1 || P < 1 || S < 2) return fail(EP_ERR_INVALID, "bad config: num_envs %d, max_num_players %d, state_num %d", N, P, S);
  if (N > kNoallocMaxEnvs)
    return fail(EP_ERR_TOO_LARGE, "ep_dummy_send_noalloc handles at most %d envs (its scratch is a fixed static array); got num_envs = %d",
                kNoallocMaxEnvs, N);
  if (n < 0 || n > N) return fail(EP_ERR_TOO_LARGE, "n = %d outside [0, num_envs = %d]", n, N);
  if (np_rows < 0 || (int64_t)np_rows > (int64_t)N * P)
    return fail(EP_ERR_TOO_LARGE, "players.env_id has %d rows, more than num_envs * max_num_players", np_rows);
  if ((int64_t)n * P >= (1ll << 30)) return fail(EP_ERR_TOO_LARGE, "n * max_num_players must stay below 2^30 player rows");
  if (n_out != 12) return fail(EP_ERR_INVALID, "need the 12 output arrays in _state_keys order, got %d", n_out);
  if (!force_reset && n > 0) {
    if (!list_action) return fail(EP_ERR_INVALID, "list_action must not be NULL for a step");
    if (P == 1 ? (!players_env_id || np_rows < n) : (np_rows > 0 && !players_env_id))
      return fail(EP_ERR_INVALID, "players.env_id is NULL or too short");
    if (((uintptr_t)list_action & 7) || ((uintptr_t)players_env_id & 3))
      return fail(EP_ERR_INVALID, "device action arrays must be aligned to their element size");
  }
  for (int k = 0; k < 12; ++k) {
    const int sk = kStateToSlot[k];
    if (sk < 0) continue;  // obs:dyn has no array
    const bool player = sk == 1 || sk == 4 || sk == 5 || sk >= 8;
    const int64_t need = player ? (int64_t)n * P : (int64_t)n;
    if (need > 0 && (!out[k].data || out[k].shape[0] < need))
      return fail(EP_ERR_INVALID, "output array %d is NULL or holds %lld rows, need %lld (player-row keys: n * max_num_players)",
                  k, (long long)out[k].shape[0], (long long)need);
    if ((uintptr_t)out[k].data & (sk == 8 ? 15 : (size_t)kSlotElem[sk] - 1))
      return fail(EP_ERR_INVALID, "output array %d is misaligned (obs:raw needs 16 bytes)", k);
    // the kernels write state_num int32 per obs:raw row (as 16- and 8-byte stores): an array laid
    // out for another state_num would be overrun without a word
    if (sk == 8 ? (out[k].ndim != 2 || out[k].shape[1] != S) : out[k].ndim != 1)
      return fail(EP_ERR_INVALID, sk == 8 ? "output array %d (obs:raw) must have shape [rows, state_num = %d]"
                                          : "output array %d must be one-dimensional", k, S);
  }
  if (cfg->device_id < 0 || cfg->device_id >= kNoallocMaxDevices) return fail(EP_ERR_INVALID, "device_id %d out of range", cfg->device_id);
  EP_ENTER_DEVICE(cfg->device_id);
  unsigned char* sc = nullptr;
  int cus = 0;
  {
    std::lock_guard<std::mutex> lk(g_noalloc_mu);
    if (!g_noalloc_ptr[cfg->device_id]) {  // set-up, once per device, with that device current
      unsigned char* ptr = nullptr;
      HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
      if (!ptr) return fail(EP_ERR_HIP, "hipGetSymbolAddress returned a null address for the scratch array");
      HIP_TRY(hipDeviceGetAttribute(&g_noalloc_cus[cfg->device_id], hipDeviceAttributeMultiprocessorCount, cfg->device_id));
      g_noalloc_ptr[cfg->device_id] = ptr;
    }
    sc = g_noalloc_ptr[cfg->device_id];
    cus = g_noalloc_cus[cfg->device_id];
  }
  hipStream_t st = (hipStream_t)stream;
  // what must start at zero: the result words, the ticket and (multi-player steps) the histogram
  HIP_TRY(hipMemsetAsync(result, 0, sizeof(ep_noalloc_result), st));
  if (n == 0) return EP_OK;
  const int num_tiles = (n + epk::kMpTile - 1) / epk::kMpTile;
  HIP_TRY(hipMemsetAsync(sc, 0, kNoallocTilesOff + 8 * (size_t)num_tiles, st));
  epk::MpScratch ms;
  ms.ticket = (uint32_t*)sc;
  ms.tiles = (unsigned long long*)(sc + kNoallocTilesOff);
  ms.counts = (unsigned long long*)(sc + kNoallocCountsOff);
  ms.barrier = nullptr;  // (no folded histogram here: two launches, as the pool did when this entry point was written)
  const uint32_t epoch = 1;  // the words were just cleared: no tag from an earlier call can match
  if (P > 1 && !force_reset) {
    // every call uses epoch 1, so the previous call's counters carry a matching tag: they are
    // cleared on every multi-player step, also one that brings no player rows at all (its envs
    // must see action_num = 0, not the last call's histogram)
    HIP_TRY(hipMemsetAsync(ms.counts, 0, 8 * (size_t)N, st));
    if (np_rows > 0) {
      hipLaunchKernelGGL(epk::mp_count_rows, dim3((np_rows + 255) / 256), dim3(256), 0, st, players_env_id, np_rows,
                         cfg->env_id_offset, N, ms.counts, epoch);
      HIP_TRY(hipGetLastError());
    }
  }
  epk::MpFusedArgs a;
  a.env_id = env_id;
  a.players_env_id = players_env_id;
  a.list_action = force_reset ? nullptr : list_action;
  a.la_ok = nullptr;
  a.la_bit0 = 0;
  a.sc = ms;
  a.count_epoch = epoch;
  a.tile_epoch = epoch;
  a.ticket_base = 0;
  a.env = (epk::EnvRec*)env_state;
  a.stats = epk::StatPtrs{nullptr, nullptr, nullptr, nullptr};
  a.slot.env_id = (int32_t*)out[0].data;
  a.slot.penv = (int32_t*)out[1].data;
  a.slot.elapsed = (int32_t*)out[2].data;
  a.slot.done = (uint8_t*)out[3].data;
  a.slot.reward = (float*)out[4].data;
  a.slot.discount = (float*)out[5].data;
  a.slot.step_type = (int32_t*)out[6].data;
  a.slot.trunc = (uint8_t*)out[7].data;
  a.slot.raw = (int32_t*)out[8].data;
  a.slot.pdone = (uint8_t*)out[10].data;
  a.slot.pid = (int32_t*)out[11].data;
  a.err = (epk::ErrWords*)&result->action_check;
  a.stamp = nullptr;
  a.stamp_val = 0;
  a.pc.N = N;
  a.pc.env_off = cfg->env_id_offset;
  a.pc.seed = cfg->seed;
  a.pc.max_episode_steps = cfg->max_episode_steps;
  a.pc.S = S;
  a.slot_players = &result->num_player_rows;
  a.host_players = nullptr;
  a.fold_penv = nullptr;
  a.fold_rows = 0;
  a.fold_base = 0;
  a.row0 = 0;
  a.force_reset = force_reset ? 1 : 0;
  a.P = P;
  a.first_segment = 1;
  return epk_mp::launch_fused(st, a, n, cus, false);
}

int32_t ep_pool_num_envs(ep_pool* p) { return p ? p->N : 0; }
int32_t ep_pool_batch_size(ep_pool* p) { return p ? p->B : 0; }
int32_t ep_pool_ring_depth(ep_pool* p) { return p ? p->Q : 0; }
int32_t ep_pool_is_sync(ep_pool* p) { return p ? (p->is_sync ? 1 : 0) : 0; }
void* ep_pool_stream(ep_pool* p) { return p ? (void*)p->stream : nullptr; }
const char* ep_pool_last_rollout_kernel(ep_pool* p) { return p ? p->last_rollout_kernel : ""; }

int ep_pool_set_timing(ep_pool* p, int32_t enabled) {
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  std::lock_guard<std::mutex> lk(p->mu);
  p->timing = enabled != 0;
  return EP_OK;
}

int ep_pool_kernel_time_us(ep_pool* p, double* total_us, int64_t* launches, int32_t reset) {
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  std::lock_guard<std::mutex> lk(p->mu);
  EP_ENTER_DEVICE(p->device);
  [[maybe_unused]] int rc = EP_OK;
  if ((rc = drain_timed(p))) return rc;
  if (total_us) *total_us = p->timed_total_us;
  if (launches) *launches = p->timed_launches;
  if (reset) { p->timed_total_us = 0; p->timed_launches = 0; }
  return EP_OK;
}

}  // extern "C"
