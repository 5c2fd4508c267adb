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
//   send / reset  -> action queue (action_buffer_queue.h:35-85) -> worker threads. The queue is a ring
//                    of 2 * num_envs cells like the reference's; cells are claimed a few at a time
//                    with one compare-and-swap on the head index (no lock on the way to an env-step)
//   worker        -> engine callback -> ep_ygo_emit_record -> row of the batch being filled
//                    async pools (batch_size < num_envs): rows in ARRIVAL order
//                    (state_buffer_queue.h:114-128); sync pools: row = position in the sent
//                    env_id array (`order`, async_envpool.h:69, state_buffer.h:92-95), emitted in
//                    place, and handed to the encoder in parts of consecutive rows as they complete
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
  bool force_reset;
};

// One cell of the action ring: a Slice as two words that are read while other threads may be
// claiming it (a reader that loses the claim throws its copy away).
struct Cell {
  std::atomic<uint64_t> a{0}, b{0};
  void put(const Slice& s) {
    a.store((uint64_t)(uint32_t)s.env_id | (uint64_t)(uint32_t)s.action << 32, std::memory_order_relaxed);
    b.store((uint64_t)(uint32_t)s.order | (uint64_t)(s.force_reset ? 1 : 0) << 32, std::memory_order_relaxed);
  }
  Slice get() const {
    const uint64_t x = a.load(std::memory_order_relaxed), y = b.load(std::memory_order_relaxed);
    return Slice{(int32_t)(uint32_t)x, (int32_t)(uint32_t)y, (int32_t)(uint32_t)(x >> 32), (y >> 32 & 1) != 0};
  }
};

constexpr int kVisit = 8;  // most slices one visit to the queue takes

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
  // action queue: cells [q_head, q_tail) are waiting. At most one slice per env is ever waiting
  // (an env that is in flight is refused by send), so q_tail - q_head <= num_envs < ring_cap and a
  // cell is never written while it can still be claimed.
  std::unique_ptr<Cell[]> ring;
  uint64_t ring_cap = 0;
  alignas(64) std::atomic<uint64_t> q_head{0};
  alignas(64) std::atomic<uint64_t> q_tail{0};
  alignas(64) std::atomic<int> sleepers{0};
  std::atomic<bool> quitting{false};
  std::mutex qmu;  // only for going to sleep and waking up
  std::condition_variable qcv;
  std::vector<std::thread> workers;
  // sync pools: the one batch being filled
  std::mutex smu;
  std::condition_variable scv;
  uint64_t sync_sent = 0, sync_committed = 0, sync_received = 0;  // (one batch at a time: sync_sent names the one being filled)
  char* sync_slot = nullptr;
  int sync_n = 0;
  std::vector<int32_t> sync_ids[2];  // the ids of the last two sends ([sync_sent & 1]): ep_ygo_pool_last_env_ids
  const int32_t* last_ids = nullptr;
  int last_ids_n = 0;
  // The batch goes to the encoder in up to kMaxParts parts of consecutive rows, each as soon as its
  // rows are emitted: slices are claimed in sent order, so the early rows are done while the late
  // ones are still being stepped, and their encode (launch, PCIe reads, kernel) runs under that.
  struct alignas(64) PartLeft { std::atomic<int> v{0}; };  // (own cache lines: every visit counts one down)
  static constexpr int kMaxParts = 8;
  PartLeft sync_left[kMaxParts];
  int sync_parts = 1, sync_part_rows = 0;
  int max_parts = 4;  // EP_YGO_POOL_PARTS
  alignas(64) int pad_after_counters = 0;
  int sync_commit_rc = EP_OK;
  std::atomic<int> sync_fail_rc{EP_OK};   // first failing part's code of the batch in flight
  std::atomic<int> sync_parts_done{0};    // parts of that batch whose count-down has reached zero
  std::string sync_fail_msg;
  // outcome of callbacks and emits since the last recv
  std::atomic<int> engine_failed{0}, emit_refused{0};
  int64_t recv_timeout_ms = -1;
  int spin_pauses = 4000;
  // EP_YGO_POOL_TRACE=1: where a step's time goes, printed at destroy (nanoseconds, summed)
  bool trace = false;
  std::atomic<int64_t> t_engine{0}, t_emit{0}, t_commit{0}, t_slices{0};
  int64_t t_send = 0, t_drain = 0, t_wait = 0, t_collect = 0, n_recv = 0;  // EP_YGO_POOL_SPIN: how long an idle worker looks for work before it sleeps
};

namespace {

inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The env-steps of one visit to the queue, on a worker thread (or on the receiving thread while it
// waits): EnvStep (env.h:162-172) with the engine behind the callbacks. Sync pools emit each
// record straight into its row of the pinned slot (row = position in the sent env_id array; the
// kernel fetches only the used part of a row over PCIe) and count the visit down ONCE (the counter
// is shared by all threads: per record it was most of a short step's time). Async pools emit into
// this thread's scratch record and hand it to the arrival ring.
void run_slices(ep_ygo_pool* p, const Slice* sl, int k, std::vector<uint8_t>& scratch) {
  const size_t stride = (size_t)p->stride;
  if (!p->is_sync && scratch.size() < stride) scratch.resize(stride);
  int64_t v_engine = 0, v_emit = 0;  // trace: this visit's sums, added to the pool's once per visit
  for (int i = 0; i < k; ++i) {
    const Slice& s = sl[i];
    uint8_t* const rec = p->is_sync ? (uint8_t*)p->sync_slot + stride * (size_t)s.order : scratch.data();
    EnvSlot& e = p->envs[s.env_id - p->env_off];
    const bool reset = s.force_reset || e.done;  // async_envpool.h:125
    e.current_step = reset ? 0 : e.current_step + 1;  // env.h:181-189
    ygo_raw_state st;
    memset(&st, 0, sizeof(st));
    const int64_t c0 = p->trace ? now_ns() : 0;
    int rc = reset ? p->engine.reset(p->engine.ctx, s.env_id, &st) : p->engine.step(p->engine.ctx, s.env_id, s.action, &st);
    const int64_t c1 = p->trace ? now_ns() : 0;
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
    // (a state the emitter cannot parse comes back as its empty observation: the row still arrives)
    if (ep_ygo_emit_record(&p->cfg, &st, rec) != EP_OK) p->emit_refused.store(1, std::memory_order_relaxed);
    e.in_flight.store(0, std::memory_order_release);
    if (p->trace) { v_engine += c1 - c0; v_emit += now_ns() - c1; }
    if (!p->is_sync && ep_ygo_put_arrival(p->enc, rec) != EP_OK) p->engine_failed.store(1, std::memory_order_relaxed);
  }
  if (p->trace) { p->t_engine += v_engine; p->t_emit += v_emit; p->t_slices += k; }
  if (!p->is_sync) return;
  // The batch's set-up is read BEFORE this visit's first count-down: once the last one is done the
  // batch may complete, be received, and the next send rewrite it.
  const uint64_t batch = p->sync_sent;  // (written before the slices were published)
  const int parts = p->sync_parts, part_rows = p->sync_part_rows, n_total = p->sync_n;
  int per_part[ep_ygo_pool::kMaxParts] = {0};
  for (int i = 0; i < k; ++i) per_part[sl[i].order / part_rows] += 1;
  for (int part = 0; part < parts; ++part) {
    const int mine = per_part[part];
    if (mine == 0 || p->sync_left[part].v.fetch_sub(mine, std::memory_order_acq_rel) != mine) continue;
    // this visit finished the part: its rows go to the encoder now
    const int row0 = part * part_rows, rows = std::min(part_rows, n_total - row0);
    const int64_t c3 = p->trace ? now_ns() : 0;
    int32_t completed = 0;
    // (once a part has failed the batch cannot complete: the later parts are not launched at all)
    int rc2 = p->sync_fail_rc.load(std::memory_order_acquire);
    if (rc2 == EP_OK) {
      rc2 = ep_ygo_commit_slot_rows(p->enc, row0, rows, n_total, &completed);
      if (rc2 != EP_OK) {
        int none = EP_OK;
        p->sync_fail_rc.compare_exchange_strong(none, rc2);
        p->sync_fail_msg = epc::g_last_error;  // (only the thread that set the code writes it; read after the last part)
      }
    }
    if (p->trace) p->t_commit += now_ns() - c3;
    // The batch is reported when it is complete, or - after a failure - when the LAST part's
    // count-down has reached zero: until then other workers are still emitting rows into the slot,
    // and the next send would rewrite the batch's set-up (slot, sizes, count-downs) under them.
    const bool last_part = p->sync_parts_done.fetch_add(1, std::memory_order_acq_rel) + 1 == parts;
    if (rc2 == EP_OK && !completed) continue;
    if (rc2 != EP_OK && !last_part) continue;
    const int final_rc = p->sync_fail_rc.load(std::memory_order_acquire);
    if (final_rc != EP_OK) (void)ep_ygo_abort_slot(p->enc);  // the parts that did go out are forgotten: the slot starts clean
    std::lock_guard<std::mutex> lk(p->smu);  // the batch is on its way, or never will be
    if (p->sync_committed != batch) {
      p->sync_commit_rc = final_rc;
      p->sync_committed = batch;
      p->scv.notify_all();
    }
  }
}

// Claims a fair share of what is waiting, at most kVisit slices (one visit per slice is what limits
// a pool whose engine steps are short). The cells are copied BEFORE the claim: once the head has
// moved past them the sender may reuse them. 0: nothing waiting.
int take_slices(ep_ygo_pool* p, Slice* mine) {
  const uint64_t nw = p->workers.capacity();  // (reserved to the worker count before the first thread starts)
  uint64_t h = p->q_head.load(std::memory_order_acquire);
  for (;;) {
    const uint64_t t = p->q_tail.load(std::memory_order_acquire);
    if (h >= t) return 0;
    const uint64_t k = std::min<uint64_t>(std::min<uint64_t>((uint64_t)kVisit, t - h), std::max<uint64_t>(1, (t - h) / (2 * nw)));
    for (uint64_t i = 0; i < k; ++i) mine[i] = p->ring[(h + i) % p->ring_cap].get();
    if (p->q_head.compare_exchange_weak(h, h + k, std::memory_order_acq_rel, std::memory_order_acquire)) return (int)k;
  }
}

void worker_loop(ep_ygo_pool* p) {  // async_envpool.h:117-128
  std::vector<uint8_t> scratch;
  Slice mine[kVisit];
  auto waiting = [p] { return p->q_head.load() != p->q_tail.load(); };  // (sequentially consistent: pairs with the sender's publish-then-look)
  for (;;) {
    // like the reference's LightweightSemaphore: look for work for a few tens of microseconds
    // before sleeping in the kernel (a futex wake-up costs more than a short engine step)
    for (int spin = 0; spin < p->spin_pauses && !waiting() && !p->quitting.load(std::memory_order_relaxed); ++spin)
      __builtin_ia32_pause();
    const int k = take_slices(p, mine);
    if (k > 0) {
      run_slices(p, mine, k, scratch);
      continue;
    }
    // what was sent before destroy is still stepped (async_envpool.h:144-154 queues its empty
    // slices BEHIND the pending ones)
    if (p->quitting.load(std::memory_order_acquire)) return;
    std::unique_lock<std::mutex> lk(p->qmu);
    p->sleepers.fetch_add(1);  // (sequentially consistent, like the sender's look at it after publishing)
    p->qcv.wait(lk, [&] { return waiting() || p->quitting.load(std::memory_order_acquire); });
    p->sleepers.fetch_sub(1);
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
  const int64_t s0 = p->trace ? now_ns() : 0;
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
  if (p->is_sync) {
    if (p->sync_sent != p->sync_received)
      return fail(EP_ERR_INVALID, "sync pool: receive the batch of the previous send / reset first");
    void* slot = nullptr;
    int rc = ep_ygo_acquire_slot(p->enc, &slot);
    if (rc) return rc;
    p->sync_slot = (char*)slot;
    p->sync_n = n;
    // parts of at least 128 rows (below that a launch costs more than it hides)
    p->sync_parts = std::max(1, std::min(p->max_parts, n / 128));
    p->sync_part_rows = (n + p->sync_parts - 1) / p->sync_parts;
    p->sync_parts = (n + p->sync_part_rows - 1) / p->sync_part_rows;
    for (int part = 0; part < p->sync_parts; ++part)
      p->sync_left[part].v.store(std::min(p->sync_part_rows, n - part * p->sync_part_rows), std::memory_order_relaxed);
    p->sync_fail_rc.store(EP_OK, std::memory_order_relaxed);
    p->sync_parts_done.store(0, std::memory_order_relaxed);
    ++p->sync_sent;
    p->sync_ids[p->sync_sent & 1].assign(env_id, env_id + n);
  }
  const uint64_t t = p->q_tail.load(std::memory_order_relaxed);  // senders are serialised by smu
  for (int i = 0; i < n; ++i) {
    p->envs[env_id[i] - p->env_off].in_flight.store(1, std::memory_order_relaxed);
    p->ring[(t + (uint64_t)i) % p->ring_cap].put(Slice{env_id[i], p->is_sync ? i : -1, force_reset ? 0 : action[i], force_reset});
  }
  p->q_tail.store(t + (uint64_t)n);  // publishes the cells, the in-flight flags and the sync batch's set-up
  if (p->sleepers.load() > 0) {
    { std::lock_guard<std::mutex> lk(p->qmu); }  // a worker between its last look and its sleep is past it now
    p->qcv.notify_all();
  }
  if (p->trace) p->t_send += now_ns() - s0;
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
  p->ring_cap = 2 * (uint64_t)num_envs + 16;  // action_buffer_queue.h: 2 * num_envs
  p->ring.reset(new Cell[p->ring_cap]);
  if (const char* e = getenv("EP_YGO_POOL_TRACE")) p->trace = atoi(e) != 0;
  if (const char* e = getenv("EP_YGO_POOL_PARTS")) p->max_parts = std::max(1, std::min((int)ep_ygo_pool::kMaxParts, atoi(e)));
  if (const char* e = getenv("EP_YGO_POOL_SPIN")) p->spin_pauses = std::max(0, atoi(e));
  const unsigned hw = usable_cpus();
  // async_envpool.h:113-115: min(batch, hardware_concurrency). A sync pool's receiving thread steps envs
  // itself while it waits (ep_ygo_pool_recv), so it counts as one of them: with a worker per CPU
  // besides, the sender would share a CPU with a spinning worker.
  if (num_threads <= 0) num_threads = (int32_t)std::min<unsigned>((unsigned)p->B, p->is_sync ? std::max(1u, hw - 1) : hw);
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
  {  // async_envpool.h:144-154 (there: one empty slice per worker behind the pending ones)
    std::lock_guard<std::mutex> lk(p->qmu);
    p->quitting.store(true);
  }
  p->qcv.notify_all();
  if (!p->is_sync) (void)ep_ygo_async_end(p->enc);  // workers asleep on a full ring (nobody received) must come back
  for (auto& w : p->workers) w.join();
  if (p->trace && p->n_recv > 0) {
    const double k = 1e-3 / (double)p->n_recv, ns = (double)std::max<int64_t>(1, p->t_slices.load());
    fprintf(stderr, "[ep_ygo_pool trace] %lld sync recvs, %zu workers; per step: send %.1f us, recv drain (helping) %.1f, wait for commit %.1f, "
            "collect %.1f; per env-step: engine callback %.2f us, emit %.2f; commit call %.1f us per step\n",
            (long long)p->n_recv, p->workers.size(), p->t_send * k, p->t_drain * k, p->t_wait * k, p->t_collect * k,
            p->t_engine.load() / ns * 1e-3, p->t_emit.load() / ns * 1e-3, p->t_commit.load() * k);
  }
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
  if (epc::bad_mem(mem_out)) return fail(EP_ERR_INVALID, "mem_out = %d is neither EP_MEM_HOST nor EP_MEM_DEVICE", mem_out);
  int rc;
  if (p->is_sync) {
    {
      std::lock_guard<std::mutex> lk(p->smu);
      if (p->sync_sent == p->sync_received) return fail(EP_ERR_WOULD_BLOCK, "nothing was sent");
    }
    const int64_t r0 = p->trace ? now_ns() : 0;
    if (p->recv_timeout_ms != 0) {
      // the caller would only sleep until the batch is complete: let it step envs meanwhile
      static thread_local std::vector<uint8_t> scratch;
      Slice mine[kVisit];
      for (int k; (k = take_slices(p, mine)) > 0;) run_slices(p, mine, k, scratch);
    }
    const int64_t r1 = p->trace ? now_ns() : 0;
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
        return fail(p->sync_commit_rc, "the encoder refused the batch: %s", p->sync_fail_msg.c_str());
      }
    }
    const int64_t r2 = p->trace ? now_ns() : 0;
    rc = ep_ygo_collect(p->enc, mem_out, out);
    std::lock_guard<std::mutex> lk(p->smu);
    p->sync_received = p->sync_sent;
    p->last_ids = p->sync_ids[p->sync_sent & 1].data();
    p->last_ids_n = p->sync_n;
    if (p->trace) { p->t_drain += r1 - r0; p->t_wait += r2 - r1; p->t_collect += now_ns() - r2; ++p->n_recv; }
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
  if (p->q_tail.load() != 0 || p->sync_sent != 0) return fail(EP_ERR_INVALID, "set the env id offset before the first send / reset");
  for (int i = 0; i < p->N; ++i)
    if (p->envs[i].current_step != -1) return fail(EP_ERR_INVALID, "set the env id offset before the first send / reset");
  p->env_off = env_id_offset;
  return EP_OK;
}

int ep_ygo_pool_last_env_ids(ep_ygo_pool* p, const int32_t** ids, int32_t* n) {
  if (!p || !ids || !n) return fail(EP_ERR_INVALID, "null argument");
  if (!p->is_sync) return ep_ygo_last_env_ids(p->enc, ids, n);  // rows in arrival order: the ring noted them
  std::lock_guard<std::mutex> lk(p->smu);
  if (!p->last_ids) return fail(EP_ERR_WOULD_BLOCK, "no batch has been received yet");
  *ids = p->last_ids;  // rows in sent order: the ids of that send
  *n = p->last_ids_n;
  return EP_OK;
}

ep_ygo_encoder* ep_ygo_pool_encoder(ep_ygo_pool* p) { return p ? p->enc : nullptr; }

int32_t ep_ygo_pool_num_threads(ep_ygo_pool* p) { return p ? (int32_t)p->workers.size() : 0; }

}  // extern "C"
