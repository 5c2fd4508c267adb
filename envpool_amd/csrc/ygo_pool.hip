// ygo_pool.hip — AsyncEnvPool<YGOProEnv> over a caller-supplied duel engine.
//
// The reference runs every YGOPro env on a host worker thread: pop an ActionSlice, run the duel
// engine until the next decision (YGOProEnv::Reset / Step), then WriteState encodes the
// observation into the env's StateBuffer row (envpool2/core/async_envpool.h:57-189,
// envpool2/ygopro/ygopro.h:1374-1423, 1452-1494, 1828-1893). The engine (ygopro-core + Lua +
// SQLite) stays on the host and is not part of this library: it comes in as two callbacks that
// run it up to the point where the reference calls WriteState and describe what WriteState would
// read (ygo_raw_state). Everything after that point is this library's: the record is emitted on
// the worker thread, travels to the GPU in the encoder's pinned ring and is encoded there.
//
//   send / reset  -> action queue (action_buffer_queue.h:35-85) -> worker threads
//   worker        -> engine callback -> ep_ygo_emit_record -> row of the batch being filled
//                    async pools (batch_size < num_envs): rows in ARRIVAL order
//                    (state_buffer_queue.h:114-128); sync pools: row = position in the sent
//                    env_id array (`order`, async_envpool.h:69, state_buffer.h:92-95)
//   recv          -> the next batch in order, encoded (state_buffer_queue.h:139-150)
//
// The framework half of Env (env.h:162-222) is here as well: auto-reset of an env one step after
// it reported done (async_envpool.h:125), current_step_ (env.h:181-189) and the common keys
// info:env_id / elapsed_step / done of Env::Allocate (env.h:196-222) are the pool's, not the
// engine's. Host-only code (no kernels in this file); C ABI: include/envpool_amd_ygo.h.
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/envpool_amd_ygo.h"
#include "common.hpp"

using epc::fail;

namespace {

struct Slice {  // ActionBufferQueue::ActionSlice (+ the action itself: YGOPro's is one int32)
  int32_t env_id, order, action;
  bool force_reset, quit;
  uint64_t batch;  // sync pools: which send this slice belongs to
};

struct EnvSlot {
  int32_t current_step = -1;  // Env::current_step_ (env.h:71)
  bool done = true;           // IsDone(): a fresh env resets on its first step (ygopro.h:1269, done_{true})
  std::atomic<int> in_flight{0};
};

}  // namespace

struct ep_ygo_pool {
  ep_ygo_config cfg{};
  ep_ygo_engine engine{};
  ep_ygo_encoder* enc = nullptr;
  int N = 0, B = 0;
  int32_t env_off = 0;  // global id of local env 0 (one pool per GPU, env ids sharded in blocks)
  bool is_sync = false;
  int64_t stride = 0;
  std::unique_ptr<EnvSlot[]> envs;
  std::vector<uint32_t> stamp;  // duplicate check of one send
  uint32_t stamp_val = 0;
  // action queue
  std::mutex qmu;
  std::condition_variable qcv;
  std::deque<Slice> queue;
  alignas(64) std::atomic<int> q_size{0};  // mirror of queue.size() for the workers' spin phase
  std::vector<std::thread> workers;
  // sync pools: the one batch being filled
  std::mutex smu;
  std::condition_variable scv;
  uint64_t sync_sent = 0, sync_committed = 0, sync_received = 0;
  char* sync_slot = nullptr;
  int sync_n = 0;
  alignas(64) std::atomic<int> sync_left{0};        // (own cache lines: every record bumps these two)
  alignas(64) std::atomic<uint32_t> sync_bytes{0};
  alignas(64) int pad_after_counters = 0;
  std::vector<uint32_t> sync_offs;
  int sync_commit_rc = EP_OK;
  // outcome of callbacks and emits since the last recv
  std::atomic<int> engine_failed{0}, emit_refused{0};
  int64_t recv_timeout_ms = -1;
};

namespace {

// One env-step on a worker thread: EnvStep (env.h:162-172) with the engine behind the callbacks.
void run_slice(ep_ygo_pool* p, const Slice& s, std::vector<uint8_t>& scratch) {
  EnvSlot& e = p->envs[s.env_id - p->env_off];
  const bool reset = s.force_reset || e.done;  // async_envpool.h:125
  e.current_step = reset ? 0 : e.current_step + 1;  // env.h:181-189
  ygo_raw_state st;
  memset(&st, 0, sizeof(st));
  int rc = reset ? p->engine.reset(p->engine.ctx, s.env_id, &st) : p->engine.step(p->engine.ctx, s.env_id, s.action, &st);
  if (rc != 0) {
    // the row still has to arrive or the batch never completes: an empty record (the no-options
    // early-out of WriteState), flagged at the next recv
    p->engine_failed.store(1, std::memory_order_relaxed);
    memset(&st, 0, sizeof(st));
    st.done = 1;
  }
  e.done = st.done != 0;
  st.env_id = s.env_id;              // env.h:208
  st.elapsed_step = e.current_step;  // env.h:209
  if (scratch.size() < (size_t)p->stride) scratch.resize((size_t)p->stride);
  int erc = ep_ygo_emit_record(&p->cfg, &st, scratch.data());
  if (erc != EP_OK) {
    p->emit_refused.store(1, std::memory_order_relaxed);
    ygo_raw_state empty = st;
    empty.n_options = 0;
    (void)ep_ygo_emit_record(&p->cfg, &empty, scratch.data());
  }
  e.in_flight.store(0, std::memory_order_release);
  if (!p->is_sync) {
    if (ep_ygo_put_arrival(p->enc, scratch.data()) != EP_OK) p->engine_failed.store(1, std::memory_order_relaxed);
    return;
  }
  // sync: bytes back to back in completion order, the offset table puts the record at row `order`
  const int64_t used = ep_ygo_record_used_bytes(&p->cfg, scratch.data());
  const uint32_t off = p->sync_bytes.fetch_add((uint32_t)used, std::memory_order_relaxed);
  memcpy(p->sync_slot + off, scratch.data(), (size_t)used);
  p->sync_offs[(size_t)s.order] = off;
  if (p->sync_left.fetch_sub(1, std::memory_order_acq_rel) == 1) {
    const int rc2 = ep_ygo_commit_slot_packed(p->enc, p->sync_n, p->sync_offs.data(), (int64_t)p->sync_bytes.load());
    std::lock_guard<std::mutex> lk(p->smu);
    p->sync_commit_rc = rc2;
    p->sync_committed = s.batch;
    p->scv.notify_all();
  }
}

// Takes a fair share of what is queued, at most 8 slices per visit (one lock round trip per slice
// is what limits a pool whose engine steps are short). `block`: sleep until there is something.
int take_slices(ep_ygo_pool* p, Slice* mine, bool block) {
  const size_t nw = p->workers.capacity();  // (reserved to the worker count before the first thread starts)
  int k = 0;
  std::unique_lock<std::mutex> lk(p->qmu);
  if (block) p->qcv.wait(lk, [&] { return !p->queue.empty(); });
  const size_t share = std::min<size_t>(8, std::max<size_t>(1, p->queue.size() / (2 * nw)));
  while ((size_t)k < share && !p->queue.empty()) {
    if (!block && p->queue.front().quit) break;  // quit slices are the workers'
    mine[k] = p->queue.front();
    p->queue.pop_front();
    if (mine[k++].quit) break;  // one quit slice per worker: leave the others theirs
  }
  p->q_size.store((int)p->queue.size(), std::memory_order_relaxed);
  return k;
}

void worker_loop(ep_ygo_pool* p) {  // async_envpool.h:117-128
  std::vector<uint8_t> scratch;
  Slice mine[8];
  for (;;) {
    // like the reference's LightweightSemaphore: look for work for a few tens of microseconds
    // before sleeping in the kernel (a futex wake-up costs more than a short engine step)
    for (int spin = 0; spin < 4000 && p->q_size.load(std::memory_order_relaxed) == 0; ++spin) __builtin_ia32_pause();
    const int k = take_slices(p, mine, true);
    for (int i = 0; i < k; ++i) {
      if (mine[i].quit) return;
      run_slice(p, mine[i], scratch);
    }
  }
}

// std::thread::hardware_concurrency() (what async_envpool.h:103,114 uses) counts the machine's CPUs;
// a job confined by its affinity mask or by a cgroup CPU quota (a container, one GPU's share of a
// node) would start several workers per CPU it may actually use, and they spin before they sleep.
unsigned usable_cpus() {
  unsigned n = std::thread::hardware_concurrency();
  if (n == 0) n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    const int c = CPU_COUNT(&set);
    if (c > 0) n = std::min<unsigned>(n, (unsigned)c);
  }
  auto quota = [](const char* path, bool v2) -> double {
    FILE* f = fopen(path, "r");
    if (!f) return 0.0;
    char a[64] = {0};
    long long q = -1, per = 100000;
    double cpus = 0.0;
    if (v2) {
      if (fscanf(f, "%63s %lld", a, &per) == 2 && strcmp(a, "max") != 0 && per > 0) cpus = (double)atoll(a) / (double)per;
    } else if (fscanf(f, "%lld", &q) == 1 && q > 0) {
      FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
      if (g) { if (fscanf(g, "%lld", &per) != 1) per = 100000; fclose(g); }
      if (per > 0) cpus = (double)q / (double)per;
    }
    fclose(f);
    return cpus;
  };
  double lim = quota("/sys/fs/cgroup/cpu.max", true);
  if (lim <= 0.0) lim = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);
  if (lim >= 1.0) n = std::min<unsigned>(n, (unsigned)(lim + 0.5));
  if (const char* e = getenv("EP_YGO_POOL_THREADS")) { const int v = atoi(e); if (v > 0) n = (unsigned)v; }
  return std::max(1u, n);
}

int enqueue(ep_ygo_pool* p, int32_t n, const int32_t* env_id, const int32_t* action, bool force_reset) {
  if (!p || (n > 0 && !env_id) || (n > 0 && !force_reset && !action)) return fail(EP_ERR_INVALID, "null argument");
  if (n < 0 || n > p->N) return fail(EP_ERR_TOO_LARGE, "n = %d outside [0, num_envs = %d]", n, p->N);
  if (n == 0) return EP_OK;
  // one sender at a time (action_buffer_queue.h:60-68)
  std::unique_lock<std::mutex> sl(p->smu);
  if (++p->stamp_val == 0) {
    std::fill(p->stamp.begin(), p->stamp.end(), 0u);
    p->stamp_val = 1;
  }
  for (int i = 0; i < n; ++i) {
    const int32_t e = env_id[i] - p->env_off;
    if (e < 0 || e >= p->N)
      return fail(EP_ERR_INVALID, "env_id[%d] = %d outside this pool's [%d, %d)", i, env_id[i], p->env_off, p->env_off + p->N);
    if (p->stamp[(size_t)e] == p->stamp_val)
      return fail(EP_ERR_DUP_ENV, "env %d is named twice in one send: its duel would run on two threads at once", env_id[i]);
    p->stamp[(size_t)e] = p->stamp_val;
    if (p->envs[e].in_flight.load(std::memory_order_acquire))
      return fail(EP_ERR_INVALID, "env %d is still stepping: receive its state before sending it again", env_id[i]);
  }
  uint64_t batch = 0;
  if (p->is_sync) {
    if (p->sync_sent != p->sync_received)
      return fail(EP_ERR_INVALID, "sync pool: receive the batch of the previous send / reset first");
    void* slot = nullptr;
    int rc = ep_ygo_acquire_slot(p->enc, &slot);
    if (rc) return rc;
    p->sync_slot = (char*)slot;
    p->sync_n = n;
    p->sync_left.store(n, std::memory_order_relaxed);
    p->sync_bytes.store(0, std::memory_order_relaxed);
    p->sync_offs.assign((size_t)n, 0u);
    batch = ++p->sync_sent;
  }
  for (int i = 0; i < n; ++i) p->envs[env_id[i] - p->env_off].in_flight.store(1, std::memory_order_relaxed);
  {
    std::lock_guard<std::mutex> lk(p->qmu);
    for (int i = 0; i < n; ++i)
      p->queue.push_back(Slice{env_id[i], p->is_sync ? i : -1, force_reset ? 0 : action[i], force_reset, false, batch});
    p->q_size.store((int)p->queue.size(), std::memory_order_relaxed);
  }
  p->qcv.notify_all();
  return EP_OK;
}

}  // namespace

extern "C" {

int ep_ygo_pool_create(const ep_ygo_config* cfg, const ep_ygo_engine* engine, int32_t num_envs, int32_t batch_size,
                       int32_t num_threads, int32_t thread_affinity_offset, const ygo_card_data* cards, int32_t n_cards,
                       ep_ygo_pool** out) {
  if (!cfg || !engine || !out || !engine->reset || !engine->step) return fail(EP_ERR_INVALID, "null argument");
  if (num_envs < 1) return fail(EP_ERR_INVALID, "num_envs = %d", num_envs);
  if (batch_size > num_envs)  // env_spec.h:74-79
    return fail(EP_ERR_INVALID, "It is required that batch_size <= num_envs, got num_envs = %d, batch_size = %d", num_envs, batch_size);
  std::unique_ptr<ep_ygo_pool> p(new ep_ygo_pool);
  p->cfg = *cfg;
  p->engine = *engine;
  p->N = num_envs;
  p->B = batch_size <= 0 ? num_envs : batch_size;  // async_envpool.h:91-92
  p->is_sync = p->B == p->N;                        // :95 (max_num_players is 1 for YGOPro)
  p->cfg.max_batch = p->B;
  if (p->cfg.ring_depth <= 0) p->cfg.ring_depth = std::max(4, (num_envs / p->B + 2) * 2);  // state_buffer_queue.h:68
  if (p->cfg.ring_depth < 2) p->cfg.ring_depth = 2;
  int rc = ep_ygo_encoder_create(&p->cfg, &p->enc);
  if (rc) return rc;
  if ((rc = ep_ygo_set_card_table(p->enc, cards, n_cards))) { ep_ygo_encoder_destroy(p->enc); return rc; }
  p->stride = ep_ygo_record_stride(&p->cfg);
  if (!p->is_sync) {
    if ((rc = ep_ygo_async_begin(p->enc, p->B)) || (rc = ep_ygo_set_packed(p->enc, 1))) { ep_ygo_encoder_destroy(p->enc); return rc; }
  }
  p->envs.reset(new EnvSlot[(size_t)num_envs]);
  p->stamp.assign((size_t)num_envs, 0u);
  const unsigned hw = usable_cpus();
  if (num_threads <= 0) num_threads = (int32_t)std::min<unsigned>((unsigned)p->B, hw);  // async_envpool.h:113-115
  p->workers.reserve((size_t)num_threads);
  for (int t = 0; t < num_threads; ++t) {
    p->workers.emplace_back(worker_loop, p.get());
    if (thread_affinity_offset >= 0) {  // async_envpool.h:130-141
      cpu_set_t set;
      CPU_ZERO(&set);
      CPU_SET(((unsigned)thread_affinity_offset + (unsigned)t) % std::max(1u, std::thread::hardware_concurrency()), &set);
      (void)pthread_setaffinity_np(p->workers.back().native_handle(), sizeof(set), &set);
    }
  }
  *out = p.release();
  return EP_OK;
}

int ep_ygo_pool_destroy(ep_ygo_pool* p) {
  if (!p) return EP_OK;
  {  // async_envpool.h:144-154: one empty slice per worker
    std::lock_guard<std::mutex> lk(p->qmu);
    for (size_t i = 0; i < p->workers.size(); ++i) p->queue.push_back(Slice{0, -1, 0, false, true, 0});
    p->q_size.store((int)p->queue.size(), std::memory_order_relaxed);
  }
  p->qcv.notify_all();
  if (!p->is_sync) (void)ep_ygo_async_end(p->enc);  // workers asleep on a full ring (nobody received) must come back
  for (auto& w : p->workers) w.join();
  int rc = ep_ygo_encoder_destroy(p->enc);
  delete p;
  return rc;
}

int ep_ygo_pool_send(ep_ygo_pool* p, int32_t n, const int32_t* env_id, const int32_t* action) {
  return enqueue(p, n, env_id, action, false);
}

int ep_ygo_pool_reset(ep_ygo_pool* p, int32_t n, const int32_t* env_id) { return enqueue(p, n, env_id, nullptr, true); }

int ep_ygo_pool_set_recv_timeout(ep_ygo_pool* p, int64_t milliseconds) {
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  {
    std::lock_guard<std::mutex> lk(p->smu);
    p->recv_timeout_ms = milliseconds;
  }
  return p->is_sync ? EP_OK : ep_ygo_set_recv_timeout(p->enc, milliseconds);
}

int ep_ygo_pool_recv(ep_ygo_pool* p, int32_t mem_out, ep_batch* out) {
  if (!p || !out) return fail(EP_ERR_INVALID, "null argument");
  int rc;
  if (p->is_sync) {
    {
      std::lock_guard<std::mutex> lk(p->smu);
      if (p->sync_sent == p->sync_received) return fail(EP_ERR_WOULD_BLOCK, "nothing was sent");
    }
    if (p->recv_timeout_ms != 0) {
      // the caller would only sleep until the batch is complete: let it step envs meanwhile
      static thread_local std::vector<uint8_t> scratch;
      Slice mine[8];
      for (int k; (k = take_slices(p, mine, false)) > 0;)
        for (int i = 0; i < k; ++i) run_slice(p, mine[i], scratch);
    }
    {
      std::unique_lock<std::mutex> lk(p->smu);
      const uint64_t want = p->sync_sent;
      auto ready = [&] { return p->sync_committed == want; };
      if (!ready()) {
        if (p->recv_timeout_ms == 0) return fail(EP_ERR_WOULD_BLOCK, "the batch is not complete yet");
        if (p->recv_timeout_ms < 0) p->scv.wait(lk, ready);
        else if (!p->scv.wait_for(lk, std::chrono::milliseconds(p->recv_timeout_ms), ready))
          return fail(EP_ERR_WOULD_BLOCK, "the batch is not complete yet");
      }
      if (p->sync_commit_rc != EP_OK) {
        p->sync_received = want;
        return fail(p->sync_commit_rc, "the encoder refused the batch");
      }
    }
    rc = ep_ygo_collect(p->enc, mem_out, out);
    std::lock_guard<std::mutex> lk(p->smu);
    p->sync_received = p->sync_sent;
  } else {
    rc = ep_ygo_recv(p->enc, mem_out, out);
  }
  if (rc != EP_OK && rc != EP_ERR_ACTION_CHECK) return rc;
  // the batch is there; say so if something on the way to it went wrong
  if (p->engine_failed.exchange(0))
    return fail(EP_ERR_INVALID, "an engine callback reported failure: its env's row is an empty observation (no options), the env is marked done");
  if (p->emit_refused.exchange(0))
    return fail(EP_ERR_ACTION_CHECK, "a duel state could not be parsed (the reference throws on it): its row is an empty observation");
  return rc;
}

int ep_ygo_pool_set_env_id_offset(ep_ygo_pool* p, int32_t env_id_offset) {
  if (!p) return fail(EP_ERR_INVALID, "null pool");
  std::lock_guard<std::mutex> lk(p->smu);
  std::lock_guard<std::mutex> lq(p->qmu);
  if (!p->queue.empty() || p->sync_sent != 0) return fail(EP_ERR_INVALID, "set the env id offset before the first send / reset");
  for (int i = 0; i < p->N; ++i)
    if (p->envs[i].current_step != -1) return fail(EP_ERR_INVALID, "set the env id offset before the first send / reset");
  p->env_off = env_id_offset;
  return EP_OK;
}

ep_ygo_encoder* ep_ygo_pool_encoder(ep_ygo_pool* p) { return p ? p->enc : nullptr; }

int32_t ep_ygo_pool_num_threads(ep_ygo_pool* p) { return p ? (int32_t)p->workers.size() : 0; }

}  // extern "C"
