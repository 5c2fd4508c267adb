// tsan_ygo_pool.cpp — ThreadSanitizer target for the HOST logic of ep_ygo_pool
// (envpool_amd/csrc/ygo_pool.hip: action queue, worker threads, in-flight flags, the sync batch's
// hand-off to the worker that finishes last, the helping receiver, shutdown). ThreadSanitizer and
// the ROCm runtime do not coexist, so ygo_pool.hip is compiled as plain host C++ and linked against
// the MOCK encoder below instead of ygo.hip: same entry points, same blocking behaviour (arrival
// ring of `depth` batches, producers wait when it is full, recv waits for a complete batch), no
// GPU. The duel engine is a pair of callbacks that keep PLAIN (non-atomic) per-env state: the pool
// promises never to run two callbacks of one env at once, and ThreadSanitizer checks that promise.
//   make tsan-pool      (see the Makefile)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <atomic>
#include <mutex>
#include <random>
#include <vector>

#include "../include/envpool_amd_ygo.h"

// ---- mock encoder ------------------------------------------------------------------------------
namespace {
constexpr int kStride = 256;
struct MockRec { int32_t env_id, elapsed, done, action; };
}  // namespace

struct ep_ygo_encoder {
  ep_ygo_config cfg{};
  std::mutex mu;
  std::condition_variable cv_space, cv_ready;
  int B = 0;
  size_t depth = 4;
  bool closing = false;
  std::vector<MockRec> filling;
  std::deque<std::vector<MockRec>> ready;
  int64_t timeout_ms = -1;
  // sync (acquire / commit / collect)
  std::vector<char> slot;
  std::vector<MockRec> committed;
  bool has_committed = false;
  int rows_in = 0;
  // what the last recv / collect handed out (the views stay valid until the next one)
  std::vector<int32_t> out_ids, out_elapsed, out_action;
  std::vector<uint8_t> out_done;
};

static void describe(ep_ygo_encoder* e, const std::vector<MockRec>& rows, ep_batch* out) {
  e->out_ids.clear(); e->out_elapsed.clear(); e->out_action.clear(); e->out_done.clear();
  for (const MockRec& r : rows) { e->out_ids.push_back(r.env_id); e->out_elapsed.push_back(r.elapsed); e->out_action.push_back(r.action); e->out_done.push_back((uint8_t)r.done); }
  memset(out, 0, sizeof(*out));
  out->num_keys = 4;
  out->num_rows = (int32_t)rows.size();
  out->arrays[0].data = e->out_ids.data();
  out->arrays[2].data = e->out_elapsed.data();
  out->arrays[3].data = e->out_done.data();
  out->arrays[15].data = e->out_action.data();  // info:win_reason carries the action in this mock
}

extern "C" {
const char* ep_last_error(void) { return ""; }
int ep_batch_release(ep_batch*) { return EP_OK; }
int ep_ygo_encoder_create(const ep_ygo_config* cfg, ep_ygo_encoder** out) {
  auto* e = new ep_ygo_encoder;
  e->cfg = *cfg;
  e->depth = (size_t)(cfg->ring_depth > 1 ? cfg->ring_depth : 2);
  e->slot.resize((size_t)cfg->max_batch * kStride + 1024);
  *out = e;
  return EP_OK;
}
int ep_ygo_encoder_destroy(ep_ygo_encoder* e) { delete e; return EP_OK; }
int ep_ygo_set_card_table(ep_ygo_encoder*, const ygo_card_data*, int32_t) { return EP_OK; }
int64_t ep_ygo_record_stride(const ep_ygo_config*) { return kStride; }
int64_t ep_ygo_record_used_bytes(const ep_ygo_config*, const void*) { return 64; }
int ep_ygo_emit_record(const ep_ygo_config*, const ygo_raw_state* in, void* rec) {
  MockRec r{in->env_id, in->elapsed_step, in->done, in->win_reason};
  memset(rec, 0, 64);
  memcpy(rec, &r, sizeof(r));
  return in->n_options < 0 ? EP_ERR_INVALID : EP_OK;
}
int ep_ygo_async_begin(ep_ygo_encoder* e, int32_t B) { std::lock_guard<std::mutex> lk(e->mu); e->B = B; e->closing = false; return EP_OK; }
int ep_ygo_set_packed(ep_ygo_encoder*, int32_t) { return EP_OK; }
int ep_ygo_async_end(ep_ygo_encoder* e) {
  std::lock_guard<std::mutex> lk(e->mu);
  e->closing = true;
  e->cv_space.notify_all();
  e->cv_ready.notify_all();
  return EP_OK;
}
int ep_ygo_set_recv_timeout(ep_ygo_encoder* e, int64_t ms) { std::lock_guard<std::mutex> lk(e->mu); e->timeout_ms = ms; return EP_OK; }
int ep_ygo_put_arrival(ep_ygo_encoder* e, const void* rec) {
  std::unique_lock<std::mutex> lk(e->mu);
  e->cv_space.wait(lk, [&] { return e->closing || e->ready.size() + 1 < e->depth; });
  if (e->closing) return EP_ERR_WOULD_BLOCK;
  MockRec r;
  memcpy(&r, rec, sizeof(r));
  e->filling.push_back(r);
  if ((int)e->filling.size() == e->B) {
    e->ready.push_back(std::move(e->filling));
    e->filling.clear();
    e->cv_ready.notify_all();
  }
  return EP_OK;
}
int ep_ygo_recv(ep_ygo_encoder* e, int32_t, ep_batch* out) {
  std::unique_lock<std::mutex> lk(e->mu);
  auto ready = [&] { return e->closing || !e->ready.empty(); };
  if (!ready()) {
    if (e->timeout_ms == 0) return EP_ERR_WOULD_BLOCK;
    if (e->timeout_ms < 0) e->cv_ready.wait(lk, ready);
    else if (!e->cv_ready.wait_for(lk, std::chrono::milliseconds(e->timeout_ms), ready)) return EP_ERR_WOULD_BLOCK;
  }
  if (e->ready.empty()) return EP_ERR_WOULD_BLOCK;
  std::vector<MockRec> rows = std::move(e->ready.front());
  e->ready.pop_front();
  e->cv_space.notify_all();
  describe(e, rows, out);
  return EP_OK;
}
int ep_ygo_acquire_slot(ep_ygo_encoder* e, void** p) { *p = e->slot.data(); return EP_OK; }
int ep_ygo_commit_slot_packed(ep_ygo_encoder* e, int32_t n, const uint32_t* offs, int64_t used) {
  std::lock_guard<std::mutex> lk(e->mu);
  e->committed.clear();
  for (int i = 0; i < n; ++i) {
    if ((int64_t)offs[i] + 64 > used) return EP_ERR_INVALID;
    MockRec r;
    memcpy(&r, e->slot.data() + offs[i], sizeof(r));
    e->committed.push_back(r);
  }
  e->has_committed = true;
  return EP_OK;
}
int ep_ygo_commit_slot(ep_ygo_encoder* e, int32_t n) {  // fixed stride: row i at i * kStride
  std::lock_guard<std::mutex> lk(e->mu);
  e->committed.clear();
  for (int i = 0; i < n; ++i) {
    MockRec r;
    memcpy(&r, e->slot.data() + (size_t)i * kStride, sizeof(r));
    e->committed.push_back(r);
  }
  e->has_committed = true;
  return EP_OK;
}
static std::atomic<int> g_fail_one_part{0};  // the next commit of a part that is not the batch's first fails
int ep_ygo_abort_slot(ep_ygo_encoder* e) {
  std::lock_guard<std::mutex> lk(e->mu);
  e->rows_in = 0;
  return EP_OK;
}
int ep_ygo_commit_slot_rows(ep_ygo_encoder* e, int32_t row0, int32_t n_rows, int32_t n_total, int32_t* completed) {
  std::lock_guard<std::mutex> lk(e->mu);
  *completed = 0;
  if (row0 > 0 && g_fail_one_part.exchange(0)) return EP_ERR_HIP;
  if (e->rows_in == 0) e->committed.assign((size_t)n_total, MockRec{});
  for (int i = row0; i < row0 + n_rows; ++i) memcpy(&e->committed[(size_t)i], e->slot.data() + (size_t)i * kStride, sizeof(MockRec));
  e->rows_in += n_rows;
  *completed = 0;
  if (e->rows_in == n_total) { e->rows_in = 0; e->has_committed = true; *completed = 1; }
  return EP_OK;
}
int ep_ygo_last_env_ids(ep_ygo_encoder* e, const int32_t** ids, int32_t* n) { std::lock_guard<std::mutex> lk(e->mu); *ids = e->out_ids.data(); *n = (int32_t)e->out_ids.size(); return EP_OK; }
int ep_ygo_collect(ep_ygo_encoder* e, int32_t, ep_batch* out) {
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->has_committed) return EP_ERR_WOULD_BLOCK;
  e->has_committed = false;
  describe(e, e->committed, out);
  return EP_OK;
}
}  // extern "C"

// ---- engine: plain per-env state ------------------------------------------------------------------
struct Engine {
  std::vector<int64_t> calls;   // callbacks per env so far (plain ints: never two at once per env)
  std::vector<int32_t> t_ep;
  int fail_env = -1;
};
static int eng_advance(void* ctx, int32_t env, bool reset, int32_t action, ygo_raw_state* s) {
  Engine* g = (Engine*)ctx;
  if (env == g->fail_env) return 1;
  g->calls[(size_t)env] += 1;
  g->t_ep[(size_t)env] = reset ? 0 : g->t_ep[(size_t)env] + 1;
  volatile int spin = 0;
  for (int i = 0; i < 50 + (env % 7) * 40; ++i) spin += i;  // uneven step lengths
  memset(s, 0, sizeof(*s));
  s->done = g->t_ep[(size_t)env] >= 3 + env % 5;
  s->win_reason = reset ? -1 : action;
  s->n_options = 1;
  return 0;
}
static int eng_reset(void* ctx, int32_t env, ygo_raw_state* s) { return eng_advance(ctx, env, true, 0, s); }
static int eng_step(void* ctx, int32_t env, int32_t action, ygo_raw_state* s) { return eng_advance(ctx, env, false, action, s); }

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #c); exit(2); } } while (0)

int main() {
  ep_ygo_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.max_cards = 75; cfg.max_options = 16; cfg.n_history_actions = 16; cfg.max_multi_select = 5;
  std::mt19937 rng(1);
  for (const int N : {64, 600}) {  // sync pool: rows in sent order, partial sends, forced resets, an engine failure
    // (600 envs: the batch goes to the encoder in up to four parts, ep_ygo_commit_slot_rows)
    Engine g{std::vector<int64_t>(N, 0), std::vector<int32_t>(N, 0)};
    ep_ygo_engine eng{&g, eng_reset, eng_step};
    ep_ygo_pool* p = nullptr;
    REQUIRE(ep_ygo_pool_create(&cfg, &eng, N, N, 4, -1, nullptr, 0, &p) == EP_OK);
    std::vector<int32_t> ids(N), act(N);
    for (int i = 0; i < N; ++i) ids[i] = i;
    std::vector<int32_t> cur(N, -1);
    std::vector<char> done(N, 1);
    for (int it = 0; it < (N > 64 ? 120 : 300); ++it) {
      std::shuffle(ids.begin(), ids.end(), rng);
      // (it == 50, 600 envs: one part's commit fails - the batch is reported only after every part's
      // count-down is through, the slot is given up, and the next batch starts clean)
      const bool inject = N > 64 && it == 50;
      const int n = (it % 3 || inject || (N > 64 && it == 51)) ? N : 1 + (int)(rng() % N);
      const bool reset = it % 17 == 0 || (N > 64 && it == 51);
      if (inject) g_fail_one_part.store(1);
      if (it == 100) g.fail_env = 5;
      if (it == 103) g.fail_env = -1;
      for (int i = 0; i < n; ++i) act[i] = (int32_t)(rng() % 16);
      REQUIRE((reset ? ep_ygo_pool_reset(p, n, ids.data()) : ep_ygo_pool_send(p, n, ids.data(), act.data())) == EP_OK);
      ep_batch b;
      const int rc = ep_ygo_pool_recv(p, EP_MEM_HOST, &b);
      if (inject) {
        REQUIRE(rc == EP_ERR_HIP);
        REQUIRE(g_fail_one_part.load() == 0);
        continue;  // (the envs were stepped, their rows are lost: the next iteration resets every env)
      }
      REQUIRE(rc == EP_OK || rc == EP_ERR_INVALID);
      REQUIRE(b.num_rows == n);
      const int32_t* got = (const int32_t*)b.arrays[0].data;
      const int32_t* el = (const int32_t*)b.arrays[2].data;
      const uint8_t* dn = (const uint8_t*)b.arrays[3].data;
      for (int i = 0; i < n; ++i) {
        const int e = ids[i];
        REQUIRE(got[i] == e);  // row i = position i of the sent array
        const int want = (reset || done[e]) ? 0 : cur[e] + 1;
        REQUIRE(el[i] == want);
        cur[e] = want;
        done[e] = (char)dn[i];
      }
    }
    REQUIRE(ep_ygo_pool_destroy(p) == EP_OK);
  }
  {  // async pool: one thread receives and sends back, with four workers
    const int N = 256, B = 64;
    Engine g{std::vector<int64_t>(N, 0), std::vector<int32_t>(N, 0)};
    ep_ygo_engine eng{&g, eng_reset, eng_step};
    ep_ygo_pool* p = nullptr;
    REQUIRE(ep_ygo_pool_create(&cfg, &eng, N, B, 4, -1, nullptr, 0, &p) == EP_OK);
    std::vector<int32_t> ids(N);
    for (int i = 0; i < N; ++i) ids[i] = i;
    REQUIRE(ep_ygo_pool_reset(p, N, ids.data()) == EP_OK);
    std::vector<int32_t> cur(N, -1), back(B), act(B, 3);
    std::vector<char> done(N, 1), first(N, 1);
    for (int it = 0; it < 400; ++it) {
      ep_batch b;
      REQUIRE(ep_ygo_pool_recv(p, EP_MEM_HOST, &b) == EP_OK);
      REQUIRE(b.num_rows == B);
      const int32_t* got = (const int32_t*)b.arrays[0].data;
      const int32_t* el = (const int32_t*)b.arrays[2].data;
      const uint8_t* dn = (const uint8_t*)b.arrays[3].data;
      for (int j = 0; j < B; ++j) {
        const int e = got[j];
        const int want = (first[e] || done[e]) ? 0 : cur[e] + 1;
        REQUIRE(el[j] == want);
        cur[e] = want; done[e] = (char)dn[j]; first[e] = 0;
        back[j] = e;
      }
      REQUIRE(ep_ygo_pool_send(p, B, back.data(), act.data()) == EP_OK);
    }
    REQUIRE(ep_ygo_pool_destroy(p) == EP_OK);  // with steps still in flight
  }
  {  // async pool, one receiver and TWO sender threads (senders are serialised by the pool,
     // action_buffer_queue.h:60-68): the receiver deals each batch's envs out to the senders by parity
    const int N = 256, B = 32;
    Engine g{std::vector<int64_t>(N, 0), std::vector<int32_t>(N, 0)};
    ep_ygo_engine eng{&g, eng_reset, eng_step};
    ep_ygo_pool* p = nullptr;
    REQUIRE(ep_ygo_pool_create(&cfg, &eng, N, B, 4, -1, nullptr, 0, &p) == EP_OK);
    std::vector<int32_t> ids(N);
    for (int i = 0; i < N; ++i) ids[i] = i;
    REQUIRE(ep_ygo_pool_reset(p, N, ids.data()) == EP_OK);
    struct Box { std::mutex mu; std::condition_variable cv; std::deque<std::vector<int32_t>> q; bool stop = false; } box[2];
    std::vector<std::thread> senders;
    for (int s = 0; s < 2; ++s)
      senders.emplace_back([&, s] {
        for (;;) {
          std::vector<int32_t> mine;
          {
            std::unique_lock<std::mutex> lk(box[s].mu);
            box[s].cv.wait(lk, [&] { return box[s].stop || !box[s].q.empty(); });
            if (box[s].q.empty()) return;
            mine = std::move(box[s].q.front());
            box[s].q.pop_front();
          }
          std::vector<int32_t> a(mine.size(), s);
          if (!mine.empty()) REQUIRE(ep_ygo_pool_send(p, (int32_t)mine.size(), mine.data(), a.data()) == EP_OK);
        }
      });
    std::vector<int32_t> seen(N, 0);
    for (int it = 0; it < 300; ++it) {
      ep_batch b;
      REQUIRE(ep_ygo_pool_recv(p, EP_MEM_HOST, &b) == EP_OK);
      REQUIRE(b.num_rows == B);
      const int32_t* got = (const int32_t*)b.arrays[0].data;
      const int32_t* last = nullptr;
      int32_t n_last = 0;
      REQUIRE(ep_ygo_pool_last_env_ids(p, &last, &n_last) == EP_OK && n_last == B);
      std::vector<int32_t> part[2];
      for (int j = 0; j < B; ++j) {
        REQUIRE(last[j] == got[j]);
        seen[got[j]] += 1;
        part[got[j] & 1].push_back(got[j]);
      }
      for (int s = 0; s < 2; ++s) {
        std::lock_guard<std::mutex> lk(box[s].mu);
        box[s].q.push_back(std::move(part[s]));
        box[s].cv.notify_one();
      }
    }
    for (int s = 0; s < 2; ++s) {
      std::lock_guard<std::mutex> lk(box[s].mu);
      box[s].stop = true;
      box[s].cv.notify_one();
    }
    for (auto& th : senders) th.join();
    REQUIRE(ep_ygo_pool_destroy(p) == EP_OK);
  }
  {  // shutdown with a full ring and nobody receiving
    const int N = 64, B = 8;
    Engine g{std::vector<int64_t>(N, 0), std::vector<int32_t>(N, 0)};
    ep_ygo_engine eng{&g, eng_reset, eng_step};
    ep_ygo_config c2 = cfg;
    c2.ring_depth = 2;
    ep_ygo_pool* p = nullptr;
    REQUIRE(ep_ygo_pool_create(&c2, &eng, N, B, 3, -1, nullptr, 0, &p) == EP_OK);
    std::vector<int32_t> ids(N);
    for (int i = 0; i < N; ++i) ids[i] = i;
    REQUIRE(ep_ygo_pool_reset(p, N, ids.data()) == EP_OK);
    REQUIRE(ep_ygo_pool_destroy(p) == EP_OK);
  }
  printf("tsan_ygo_pool: sync, async and shutdown cases done; no ThreadSanitizer report\n");
  return 0;
}
