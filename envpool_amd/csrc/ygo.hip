// ygo.hip — host side of the YGOPro observation-encode tail: the encoder object (pinned
// staging ring, side-stream H2D copies overlapped with the encode kernel), the static card
// table upload, and the host adapter from the reference's own data forms to the POD record.
// C ABI: include/envpool_amd_ygo.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "ygo_kernels.hpp"

using epc::align_up;
using epc::fail;

namespace {

struct Layout {  // record layout for one config
  int nf, off_hist, off_cards;  // the options follow the record's own cards: off_cards + 16 * n_cards
  int64_t stride;
};

Layout make_layout(const ep_ygo_config& c) {
  Layout l;
  l.nf = 9 + 2 * c.max_multi_select;
  l.off_hist = 64;
  l.off_cards = l.off_hist + ((c.n_history_actions * l.nf + 15) & ~15);
  l.stride = (int64_t)align_up((size_t)l.off_cards + (size_t)2 * c.max_cards * sizeof(ygo_card) +
                               (size_t)c.max_options * ygk::option_bytes(c.max_multi_select));
  return l;
}

int check_config(const ep_ygo_config* c) {
  if (!c) return fail(EP_ERR_INVALID, "null config");
  if (c->max_cards < 1 || c->max_cards > 127) return fail(EP_ERR_INVALID, "max_cards must be in [1,127] (row indices travel in one byte, ygopro.h:1869)");
  if (c->max_options < 1 || c->max_options > 64) return fail(EP_ERR_INVALID, "max_options must be in [1,64]");
  if (c->n_history_actions < 1 || c->n_history_actions > 255) return fail(EP_ERR_INVALID, "n_history_actions must be in [1,255]");
  if (c->max_multi_select < 1 || c->max_multi_select > EP_YGO_MAX_SPECS)
    return fail(EP_ERR_INVALID, "max_multi_select must be in [1,%d]", EP_YGO_MAX_SPECS);
  return EP_OK;
}

struct Slot {
  // one allocation each on the host (pinned) and on the device: [offset table, max_batch x u32,
  // padded to 256 B][records]. h_rec / d_rec point at the records.
  char* h_tab = nullptr;
  char* d_tab = nullptr;
  char* h_rec = nullptr;   // pinned
  char* d_rec = nullptr;
  char* d_out = nullptr;   // all output keys of one batch
  hipEvent_t copied = nullptr, done = nullptr;
  int n = 0;
  std::unique_ptr<int32_t[]> h_ids;  // arrival-ordered mode: info:env_id of row i, in host memory (ep_ygo_last_env_ids)
  void* out_blk = nullptr;  // host-output mode: the pinned block this batch's kernels write their rows into
  int rows_launched = 0;  // ep_ygo_commit_slot_rows: rows of the batch being filled that are on their way
  bool busy = false;
};

constexpr int kNumKeys = 18;  // the reference's 16 state keys + aux:h_card_ids + aux:action_mask

}  // namespace

struct ep_ygo_encoder {
  ep_ygo_config cfg{};
  Layout lay{};
  int device = 0;
  hipStream_t stream = nullptr, copy_stream = nullptr;
  hipStream_t copy_stream2 = nullptr;  // packed batches alternate between two copy streams: the fixed cost of one transfer hides under the data of the other
  ygk::HashEntry* d_table = nullptr;
  ygk::HashAux* d_table_aux = nullptr;
  uint32_t table_mask = 0, table_mul_b = 0;
  std::vector<Slot> slots;
  uint32_t* h_err = nullptr;  // [ring] host-mapped
  uint32_t* d_err = nullptr;
  unsigned long long* d_stamps = nullptr;  // diagnostic builds (-DYGO_STAMPS)
  size_t key_off[kNumKeys]{};
  size_t key_row[kNumKeys]{};
  int key_dtype[kNumKeys]{};
  size_t out_bytes = 0;
  uint64_t submit_idx = 0, collect_idx = 0;
  int lds_bytes = 0;
  std::mutex mu;
  std::recursive_mutex featurize_mu;  // ep_ygo_featurize = submit + collect, atomically with respect to the other submitters
  std::shared_ptr<epc::HostBlocks> blocks;
  bool timing = false;
  // Where a batch's kernels write: the slot's device arrays, or - when the consumer took the batch
  // before this one as host arrays - straight into the pinned host block that recv / collect will
  // hand out (over PCIe, under the kernel: no device-to-host copy afterwards). Follows the consumer.
  bool host_out = false;
  bool zero_copy = true;  // the kernel reads the pinned slot over PCIe itself (no staging copy); EP_YGO_ZERO_COPY=0: stage through d_rec
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;
  double timed_total_us = 0;
  int64_t timed_launches = 0;
  // arrival-ordered record ring (ep_ygo_async_begin / alloc_record / record_done / recv): the
  // counters of StateBufferQueue / StateBuffer (state_buffer_queue.h:114-150, state_buffer.h:79-129)
  int async_B = 0;                             // batch_size of this mode; 0 = submit/collect mode
  std::atomic<uint64_t> a_alloc{0};            // records handed out (StateBufferQueue::alloc_count_)
  std::atomic<uint64_t> a_recv{0};             // batches received   (StateBufferQueue::done_ptr_)
  std::unique_ptr<std::atomic<int>[]> a_done;  // [ring] records finished in the slot (StateBuffer::done_count_)
  std::vector<uint64_t> a_committed;           // [ring] 1 + index of the batch whose kernel is enqueued in the slot, 0 = none
  std::condition_variable cv_space, cv_ready;
  int64_t a_recv_timeout_ms = -1;
  std::atomic<bool> a_closing{false};  // ep_ygo_async_end: waiters leave with an error instead of sleeping on
  std::atomic<uint64_t> a_space_wait_ns{0};
  size_t tab_bytes = 0;
  int a_last_slot = -1;                           // slot of the batch ep_ygo_recv handed out last
  bool a_packed = false;                          // arrival-ordered mode: records packed back to back (ep_ygo_emit_arrival)
  std::unique_ptr<std::atomic<uint32_t>[]> a_bytes;  // [ring] bytes of packed records in the slot    // time producers spent waiting for a free slot (ring full)
};

namespace {

void out_ptrs(ep_ygo_encoder* e, char* base, ygk::Outs* o) {
  o->env_id = (int32_t*)(base + e->key_off[0]);
  o->penv = (int32_t*)(base + e->key_off[1]);
  o->elapsed = (int32_t*)(base + e->key_off[2]);
  o->done = (uint8_t*)(base + e->key_off[3]);
  o->reward = (float*)(base + e->key_off[4]);
  o->discount = (float*)(base + e->key_off[5]);
  o->step_type = (int32_t*)(base + e->key_off[6]);
  o->trunc = (uint8_t*)(base + e->key_off[7]);
  o->cards = (uint8_t*)(base + e->key_off[8]);
  o->global = (uint8_t*)(base + e->key_off[9]);
  o->actions = (uint8_t*)(base + e->key_off[10]);
  o->h_actions = (uint8_t*)(base + e->key_off[11]);
  o->num_options = (int32_t*)(base + e->key_off[12]);
  o->to_play = (int32_t*)(base + e->key_off[13]);
  o->is_selfplay = (int32_t*)(base + e->key_off[14]);
  o->win_reason = (int32_t*)(base + e->key_off[15]);
  o->h_card_ids = (uint16_t*)(base + e->key_off[16]);
  o->action_mask = (uint8_t*)(base + e->key_off[17]);
}

// Rows [row0, row0 + n) of the batch in `slot`; `d_records` is the batch's first record.
int launch(ep_ygo_encoder* e, const char* d_records, int n, int slot, bool lean_fetch = false, hipStream_t st = nullptr,
           const uint32_t* d_offsets = nullptr, int row0 = 0) {
  if (!st) st = e->stream;
  ygk::Args a;
  a.records = (const uint8_t*)d_records;
  a.offsets = d_offsets;
  a.stride = e->lay.stride;
  a.n = row0 + n;
  a.row0 = row0;
  a.table = e->d_table;
  a.table_aux = e->d_table_aux;
  a.table_mask = e->table_mask;
  a.table_mul_b = e->table_mul_b;
  a.max_cards = e->cfg.max_cards;
  a.max_options = e->cfg.max_options;
  a.n_hist = e->cfg.n_history_actions;
  a.mms = e->cfg.max_multi_select;
  a.nf = e->lay.nf;
  a.max_episode_steps = e->cfg.max_episode_steps;
  a.lean_fetch = lean_fetch ? 1 : 0;
  a.cv = ygk::carve(e->cfg.max_cards, e->cfg.max_options, e->cfg.n_history_actions, e->cfg.max_multi_select);
  {
    Slot& sl = e->slots[slot];
    if (e->host_out && !sl.out_blk)
      if (int rc = epc::take_block(e->blocks, &sl.out_blk)) return rc;
    char* base = sl.d_out;
    if (sl.out_blk) {  // (set by the batch's first launch: every part of a batch writes to one place)
      void* dev_view = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dev_view, sl.out_blk, 0));
      base = (char*)dev_view;
    }
    out_ptrs(e, base, &a.out);
  }
  a.err = e->d_err + slot;
  a.stamps = e->d_stamps;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  if (e->timing) {
    HIP_TRY(hipEventCreate(&t0));
    HIP_TRY(hipEventCreate(&t1));
    HIP_TRY(hipEventRecord(t0, st));
  }
  const int groups = (n + ygk::kGroup - 1) / ygk::kGroup;
  hipLaunchKernelGGL(ygk::ygo_encode, dim3(groups), dim3(ygk::kThreads), e->lds_bytes, st, a);
  HIP_TRY(hipGetLastError());
  if (e->timing) {
    HIP_TRY(hipEventRecord(t1, st));
    e->timed.emplace_back(t0, t1);
  }
  return EP_OK;
}

int describe(ep_ygo_encoder* e, int slot, int32_t mem_out, ep_batch* out) {
  Slot& s = e->slots[slot];
  const int n = s.n;
  memset(out, 0, sizeof(*out));
  out->num_keys = kNumKeys;
  out->mem = mem_out;
  out->num_rows = n;
  out->num_player_rows = n;
  out->seq = e->collect_idx;
  char* base = s.d_out;
  e->host_out = mem_out == EP_MEM_HOST;  // the batches launched from now on are written where this consumer reads
  if (s.out_blk && mem_out == EP_MEM_HOST) {
    // the kernels wrote the block themselves (and the caller has waited for them)
    out->owner = new epc::HostBlockOwner{e->blocks, s.out_blk};
    base = (char*)s.out_blk;
    s.out_blk = nullptr;
  } else if (s.out_blk) {
    // written for a host consumer, wanted as device views: the block goes to the slot's device arrays
    HIP_TRY(hipMemcpyAsync(s.d_out, s.out_blk, e->out_bytes, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    epc::give_block(e->blocks, s.out_blk);
    s.out_blk = nullptr;
  } else if (mem_out == EP_MEM_HOST) {
    void* blk = nullptr;
    int rc = epc::take_block(e->blocks, &blk);
    if (rc) return rc;
    if (n * 2 >= e->cfg.max_batch) {
      HIP_TRY(hipMemcpyAsync(blk, s.d_out, e->out_bytes, hipMemcpyDeviceToHost, e->stream));
    } else {
      for (int k = 0; k < kNumKeys; ++k)
        if (n) HIP_TRY(hipMemcpyAsync((char*)blk + e->key_off[k], s.d_out + e->key_off[k], (size_t)n * e->key_row[k], hipMemcpyDeviceToHost, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    out->owner = new epc::HostBlockOwner{e->blocks, blk};
    base = (char*)blk;
  }
  const ep_ygo_config& c = e->cfg;
  for (int k = 0; k < kNumKeys; ++k) {
    ep_array& a = out->arrays[k];
    a.data = base + e->key_off[k];
    a.dtype = e->key_dtype[k];
    a.ndim = 1;
    a.shape[0] = n;
  }
  auto set3 = [&](int k, int d1, int d2) { out->arrays[k].ndim = 3; out->arrays[k].shape[1] = d1; out->arrays[k].shape[2] = d2; };
  set3(8, 2 * c.max_cards, EP_YGO_CARD_FEATS);
  out->arrays[9].ndim = 2; out->arrays[9].shape[1] = 8;
  set3(10, c.max_options, e->lay.nf);
  set3(11, c.n_history_actions, e->lay.nf);
  set3(16, c.max_options, c.max_multi_select);
  out->arrays[17].ndim = 2; out->arrays[17].shape[1] = c.max_options;
  return EP_OK;
}

}  // namespace

extern "C" {

int64_t ep_ygo_record_stride(const ep_ygo_config* cfg) {
  if (check_config(cfg)) return -1;
  return make_layout(*cfg).stride;
}

int32_t ep_ygo_action_feats(const ep_ygo_config* cfg) { return cfg ? 9 + 2 * cfg->max_multi_select : 0; }

int ep_ygo_encoder_create(const ep_ygo_config* cfg, ep_ygo_encoder** out) {
  if (!out) return fail(EP_ERR_INVALID, "null argument");
  *out = nullptr;
  int rc = check_config(cfg);
  if (rc) return rc;
  if (cfg->max_batch < 1) return fail(EP_ERR_INVALID, "max_batch must be >= 1");
  int ndev = ep_device_count();
  if (ndev <= 0) return fail(EP_ERR_NO_DEVICE, "no HIP device visible: envpool_amd has no CPU fallback");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(EP_ERR_INVALID, "device_id %d out of range", cfg->device_id);
  std::unique_ptr<ep_ygo_encoder> e(new ep_ygo_encoder());
  e->cfg = *cfg;
  if (e->cfg.ring_depth <= 0) e->cfg.ring_depth = 4;
  e->lay = make_layout(*cfg);
  e->device = cfg->device_id;
  EP_ENTER_DEVICE(e->device);
  HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  {
    // a different priority class puts the second stream on a hardware queue of its own: the two
    // streams of the zero-copy mode (and the staging copy of the other mode) then really overlap,
    // however many other streams the process has (HIP spreads equal-priority streams over only a
    // few hardware queues)
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(hipStreamCreateWithPriority(&e->copy_stream, hipStreamNonBlocking, hi));
    HIP_TRY(hipStreamCreateWithPriority(&e->copy_stream2, hipStreamNonBlocking, hi));
  }
  const ygk::Carve cv = ygk::carve(cfg->max_cards, cfg->max_options, cfg->n_history_actions, cfg->max_multi_select);
  e->lds_bytes = cv.total;
  if (e->lds_bytes > 160 * 1024) return fail(EP_ERR_TOO_LARGE, "config needs %d bytes of LDS per workgroup (max 163840)", e->lds_bytes);
  if (cv.off_hist != e->lay.off_hist || cv.off_cards != e->lay.off_cards || cv.rec_bytes > e->lay.stride)
    return fail(EP_ERR_INVALID, "internal: kernel and host record layouts disagree");
  // output block layout
  const size_t B = (size_t)cfg->max_batch;
  const int nf = e->lay.nf;
  const size_t rowb[kNumKeys] = {4, 4, 4, 1, 4, 4, 4, 1, (size_t)2 * cfg->max_cards * EP_YGO_CARD_FEATS, 8,
                                 (size_t)cfg->max_options * nf, (size_t)cfg->n_history_actions * nf, 4, 4, 4, 4,
                                 (size_t)cfg->max_options * cfg->max_multi_select * 2, (size_t)cfg->max_options};
  const int dt[kNumKeys] = {EP_I32, EP_I32, EP_I32, EP_BOOL, EP_F32, EP_F32, EP_I32, EP_BOOL, EP_U8, EP_U8,
                            EP_U8, EP_U8, EP_I32, EP_I32, EP_I32, EP_I32, EP_U16, EP_BOOL};
  size_t off = 0;
  for (int k = 0; k < kNumKeys; ++k) {
    e->key_off[k] = off;
    e->key_row[k] = rowb[k];
    e->key_dtype[k] = dt[k];
    off += align_up(B * rowb[k]);
  }
  e->out_bytes = off;
  e->slots.resize(e->cfg.ring_depth);
  if (const char* z = getenv("EP_YGO_ZERO_COPY")) e->zero_copy = atoi(z) != 0;
  e->tab_bytes = align_up(4 * B);
  for (auto& s : e->slots) {
    // (+1 KiB: the kernel's first fetch of a record is a whole KiB, also of the last packed one)
    HIP_TRY(hipHostMalloc((void**)&s.h_tab, e->tab_bytes + (size_t)e->lay.stride * B + 1024, hipHostMallocMapped));
    HIP_TRY(hipMalloc((void**)&s.d_tab, e->tab_bytes + (size_t)e->lay.stride * B + 1024));
    s.h_rec = s.h_tab + e->tab_bytes;
    s.d_rec = s.d_tab + e->tab_bytes;
    HIP_TRY(hipMalloc((void**)&s.d_out, e->out_bytes));
    HIP_TRY(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  }
  HIP_TRY(hipHostMalloc((void**)&e->h_err, 4 * e->slots.size(), hipHostMallocMapped));
  memset(e->h_err, 0, 4 * e->slots.size());
  HIP_TRY(hipHostGetDevicePointer((void**)&e->d_err, e->h_err, 0));
  e->blocks = std::make_shared<epc::HostBlocks>();
  e->blocks->bytes = e->out_bytes;
#ifdef YGO_STAMPS
  HIP_TRY(hipMalloc((void**)&e->d_stamps, 16 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(e->d_stamps, 0, 16 * sizeof(unsigned long long)));
#endif
  *out = e.release();
  return EP_OK;
}

int ep_ygo_encoder_destroy(ep_ygo_encoder* e) {
  if (!e) return EP_OK;
  ::epc::DeviceScope _dev_scope;
  (void)_dev_scope.enter(e->device);
  (void)hipStreamSynchronize(e->copy_stream);
  (void)hipStreamSynchronize(e->stream);
#ifdef YGO_STAMPS
  {
    unsigned long long st[16];
    (void)hipMemcpy(st, e->d_stamps, sizeof(st), hipMemcpyDeviceToHost);
    unsigned long long tot = 0;
    for (int k = 0; k < 8; ++k) tot += st[k];
    for (int k = 0; k < 8; ++k) fprintf(stderr, "[ygo stamps] phase %d: %5.1f%%\n", k, 100.0 * (double)st[k] / (double)(tot ? tot : 1));
    (void)hipFree(e->d_stamps);
  }
#endif
  for (auto& s : e->slots) {
    if (s.out_blk) epc::give_block(e->blocks, s.out_blk);
    if (s.h_tab) (void)hipHostFree(s.h_tab);
    (void)hipFree(s.d_tab);
    (void)hipFree(s.d_out);
    if (s.copied) (void)hipEventDestroy(s.copied);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  for (auto& ab : e->timed) { (void)hipEventDestroy(ab.first); (void)hipEventDestroy(ab.second); }
  if (e->h_err) (void)hipHostFree(e->h_err);
  (void)hipFree(e->d_table);
  (void)hipFree(e->d_table_aux);
  epc::retire_blocks(e->blocks);
  (void)hipStreamDestroy(e->stream);
  (void)hipStreamSynchronize(e->copy_stream2);
  (void)hipStreamDestroy(e->copy_stream2);
  (void)hipStreamDestroy(e->copy_stream);
  delete e;
  return EP_OK;
}

int ep_ygo_set_card_table(ep_ygo_encoder* e, const ygo_card_data* cards, int32_t n) {
  if (!e || (!cards && n > 0) || n < 0) return fail(EP_ERR_INVALID, "bad card table");
  std::lock_guard<std::mutex> lk(e->mu);
  EP_ENTER_DEVICE(e->device);
  // entries first (a repeated code keeps the last one, like map[code] = ...)
  std::vector<ygk::HashEntry> ent;
  std::vector<ygk::HashAux> ent_aux;
  {
    std::unordered_map<uint32_t, size_t> last;
    for (int i = 0; i < n; ++i) last[cards[i].code] = (size_t)i;
    for (int i = 0; i < n; ++i) {
      const ygo_card_data& c = cards[i];
      if (last[c.code] != (size_t)i) continue;
      ygk::HashEntry h;
      memset(&h, 0, sizeof(h));
      h.code = c.code;
      h.type = ygk::kEntryValid;
      for (int j = 0; j < 25; ++j)  // type_to_ids order (ygopro.h:635-642)
        if (c.type & ygk::type_bit(j)) h.type |= 1u << j;
      h.card_id = c.card_id;
      h.level = (uint8_t)c.level;
      // attribute2id (ygopro.h:563-571): {0,1,2,4,8,16,32,64} -> 0..7
      h.attr_id = 0xff;
      if (c.attribute == 0) h.attr_id = 0;
      else if ((c.attribute & (c.attribute - 1)) == 0 && c.attribute <= EP_YGO_ATTRIBUTE_DEVINE) h.attr_id = (uint8_t)(__builtin_ctz(c.attribute) + 1);
      // race2id (:575-605): {0, bit0..bit25} -> 0..26
      h.race_id = 0xff;
      if (c.race == 0) h.race_id = 0;
      else if ((c.race & (c.race - 1)) == 0 && c.race <= (1u << 25)) h.race_id = (uint8_t)(__builtin_ctz(c.race) + 1);
      ent.push_back(h);
      ent_aux.push_back(ygk::HashAux{c.attack, c.defense});
    }
  }
  // cuckoo build: every code ends up in slot_a or slot_b. Load <= 0.4; on a cycle try the next
  // multiplier for the second function, then a larger table.
  uint32_t cap = 64, mul_b = 0;
  while (cap < (uint32_t)ent.size() * 5u / 2u) cap <<= 1;
  std::vector<ygk::HashEntry> tab;
  std::vector<ygk::HashAux> aux;
  static const uint32_t kMulB[8] = {0x85EBCA6Bu, 0xC2B2AE35u, 0x27D4EB2Fu, 0x165667B1u, 0x9E3779B1u, 0xD6E8FEB9u, 0xCC9E2D51u, 0x1B873593u};
  for (int attempt = 0;; ++attempt) {
    if (attempt == 64) return fail(EP_ERR_INVALID, "card table: no two-choice placement found");
    if (attempt && attempt % 8 == 0) cap <<= 1;
    mul_b = kMulB[attempt % 8];
    tab.assign(cap, ygk::HashEntry{});
    aux.assign(cap, ygk::HashAux{0, 0});
    bool ok = true;
    for (size_t i = 0; i < ent.size() && ok; ++i) {
      ygk::HashEntry h = ent[i];
      ygk::HashAux x = ent_aux[i];
      uint32_t slot = ygk::slot_a(h.code, cap - 1);
      ok = false;
      for (int kick = 0; kick < 512; ++kick) {
        if (!(tab[slot].type & ygk::kEntryValid)) {
          tab[slot] = h;
          aux[slot] = x;
          ok = true;
          break;
        }
        std::swap(h, tab[slot]);
        std::swap(x, aux[slot]);
        // the evicted entry moves to its OTHER slot
        const uint32_t sa = ygk::slot_a(h.code, cap - 1), sb = ygk::slot_b(h.code, mul_b, cap - 1);
        slot = slot == sa ? sb : sa;
      }
    }
    if (ok) break;
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  (void)hipFree(e->d_table);
  (void)hipFree(e->d_table_aux);
  e->d_table = nullptr;
  e->d_table_aux = nullptr;
  HIP_TRY(hipMalloc((void**)&e->d_table, sizeof(ygk::HashEntry) * cap));
  HIP_TRY(hipMalloc((void**)&e->d_table_aux, sizeof(ygk::HashAux) * cap));
  HIP_TRY(hipMemcpy(e->d_table, tab.data(), sizeof(ygk::HashEntry) * cap, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d_table_aux, aux.data(), sizeof(ygk::HashAux) * cap, hipMemcpyHostToDevice));
  e->table_mask = cap - 1;
  e->table_mul_b = mul_b;
  return EP_OK;
}

int ep_ygo_acquire_slot(ep_ygo_encoder* e, void** pinned_records) {
  if (!e || !pinned_records) return fail(EP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->async_B) return fail(EP_ERR_INVALID, "the encoder is in arrival-ordered mode (ep_ygo_async_begin)");
  if (e->submit_idx - e->collect_idx >= e->slots.size())
    return fail(EP_ERR_RING_FULL, "%zu batches in flight: collect one first", e->slots.size());
  *pinned_records = e->slots[e->submit_idx % e->slots.size()].h_rec;
  return EP_OK;
}

}  // extern "C"
static int commit_slot_impl(ep_ygo_encoder* e, int32_t n, const uint32_t* offsets, int64_t used_bytes);
extern "C" {

int ep_ygo_commit_slot(ep_ygo_encoder* e, int32_t n) { return commit_slot_impl(e, n, nullptr, 0); }

int ep_ygo_commit_slot_packed(ep_ygo_encoder* e, int32_t n, const uint32_t* offsets, int64_t used_bytes) {
  if (!offsets && n > 0) return fail(EP_ERR_INVALID, "null offsets");
  return commit_slot_impl(e, n, n > 0 ? offsets : nullptr, used_bytes);
}

}  // extern "C"

// `offsets` != null: the caller filled the slot in the packed form - record i at byte offsets[i]
// (16-byte aligned) of the slot, `used_bytes` in all.
static int commit_slot_impl(ep_ygo_encoder* e, int32_t n, const uint32_t* offsets, int64_t used_bytes) {
  if (!e) return fail(EP_ERR_INVALID, "null encoder");
  std::lock_guard<std::recursive_mutex> whole(e->featurize_mu);
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->async_B) return fail(EP_ERR_INVALID, "the encoder is in arrival-ordered mode (ep_ygo_async_begin)");
  if (n < 0 || n > e->cfg.max_batch) return fail(EP_ERR_INVALID, "n = %d outside [0, max_batch = %d]", n, e->cfg.max_batch);
  if (!e->d_table) return fail(EP_ERR_INVALID, "no card table: call ep_ygo_set_card_table first");
  if (e->submit_idx - e->collect_idx >= e->slots.size())
    return fail(EP_ERR_RING_FULL, "%zu batches in flight: collect one first", e->slots.size());
  EP_ENTER_DEVICE(e->device);
  const int si = (int)(e->submit_idx % e->slots.size());
  Slot& s = e->slots[si];
  if (s.rows_launched != 0)
    return fail(EP_ERR_INVALID, "%d rows of this batch went out through ep_ygo_commit_slot_rows: finish it there", s.rows_launched);
  s.n = n;
  hipStream_t st = e->stream;
  if (n > 0 && offsets) {
    if (used_bytes < 0 || used_bytes > (int64_t)n * e->lay.stride) return fail(EP_ERR_INVALID, "used_bytes %lld out of range", (long long)used_bytes);
    for (int i = 0; i < n; ++i)
      if ((offsets[i] & 15u) || (int64_t)offsets[i] + 64 > used_bytes)
        return fail(EP_ERR_INVALID, "offsets[%d] = %u is misaligned or outside the %lld used bytes", i, offsets[i], (long long)used_bytes);
    // offset table + exactly the bytes in use: one copy-engine transfer, under the kernel of the batch before
    memcpy(s.h_tab, offsets, 4 * (size_t)n);
    static const int packed_in_place = getenv("EP_YGO_PACKED_IN_PLACE") ? atoi(getenv("EP_YGO_PACKED_IN_PLACE")) : -1;
    if (e->zero_copy && (packed_in_place == 1 || (packed_in_place != 0 && e->submit_idx == e->collect_idx))) {
      // nothing else is in flight to hide a copy behind (a sync pool: commit, then wait): the kernel
      // reads the offset table and the used bytes from the pinned slot itself - one launch instead
      // of copy + event + wait + launch, and the transfer starts with the kernel
      void* dev_tab = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dev_tab, s.h_tab, 0));
      if (int rc = launch(e, (const char*)dev_tab + e->tab_bytes, n, si, /*lean_fetch=*/true, st, (const uint32_t*)dev_tab)) return rc;
    } else {
      hipStream_t cs = (e->submit_idx & 1) ? e->copy_stream2 : e->copy_stream;
      HIP_TRY(hipMemcpyAsync(s.d_tab, s.h_tab, e->tab_bytes + (size_t)used_bytes, hipMemcpyHostToDevice, cs));
      HIP_TRY(hipEventRecord(s.copied, cs));
      HIP_TRY(hipStreamWaitEvent(e->stream, s.copied, 0));
      if (int rc = launch(e, s.d_rec, n, si, false, st, (const uint32_t*)s.d_tab)) return rc;
    }
  } else if (n > 0) {
    int rc;
    if (e->zero_copy) {
      // the kernel fetches each record's USED part straight from the pinned slot (it is
      // device-mapped): less PCIe traffic than copying whole fixed-stride records, and no copy
      // step. Consecutive batches alternate between two streams, so the PCIe round trips at the
      // head and tail of one kernel run under the bulk of its neighbour.
      void* dev_view = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dev_view, s.h_rec, 0));
      st = (e->submit_idx & 1) ? e->copy_stream : e->stream;
      rc = launch(e, (const char*)dev_view, n, si, /*lean_fetch=*/true, st);
    } else {
      // side-stream copy; the compute stream only waits for THIS batch's copy, so the copy of
      // batch k+1 runs under the kernel of batch k
      HIP_TRY(hipMemcpyAsync(s.d_rec, s.h_rec, (size_t)n * e->lay.stride, hipMemcpyHostToDevice, e->copy_stream));
      HIP_TRY(hipEventRecord(s.copied, e->copy_stream));
      HIP_TRY(hipStreamWaitEvent(e->stream, s.copied, 0));
      rc = launch(e, s.d_rec, n, si);
    }
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(s.done, st));
  s.busy = true;
  e->submit_idx++;
  return EP_OK;
}

extern "C" {

int ep_ygo_commit_slot_rows(ep_ygo_encoder* e, int32_t row0, int32_t n_rows, int32_t n_total, int32_t* completed) {
  if (!e || !completed) return fail(EP_ERR_INVALID, "null argument");
  *completed = 0;
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->async_B) return fail(EP_ERR_INVALID, "the encoder is in arrival-ordered mode (ep_ygo_async_begin)");
  if (n_total < 1 || n_total > e->cfg.max_batch || row0 < 0 || n_rows < 1 || row0 > n_total - n_rows)
    return fail(EP_ERR_INVALID, "rows [%d, %d + %d) of a batch of %d (max_batch = %d)", row0, row0, n_rows, n_total, e->cfg.max_batch);
  if (!e->d_table) return fail(EP_ERR_INVALID, "no card table: call ep_ygo_set_card_table first");
  if (e->submit_idx - e->collect_idx >= e->slots.size())
    return fail(EP_ERR_RING_FULL, "%zu batches in flight: collect one first", e->slots.size());
  EP_ENTER_DEVICE(e->device);
  const int si = (int)(e->submit_idx % e->slots.size());
  Slot& s = e->slots[si];
  if (s.rows_launched + n_rows > n_total)
    return fail(EP_ERR_INVALID, "%d rows of this batch of %d are committed already", s.rows_launched, n_total);
  // every part of one batch goes to ONE stream (consecutive batches alternate, as in ep_ygo_commit_slot):
  // the event behind the last part covers them all
  hipStream_t st = (e->submit_idx & 1) ? e->copy_stream : e->stream;
  int rc;
  if (e->zero_copy) {
    void* dev_view = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dev_view, s.h_rec, 0));
    rc = launch(e, (const char*)dev_view, n_rows, si, /*lean_fetch=*/true, st, nullptr, row0);
  } else {
    st = e->stream;
    const size_t off = (size_t)row0 * (size_t)e->lay.stride;
    HIP_TRY(hipMemcpyAsync(s.d_rec + off, s.h_rec + off, (size_t)n_rows * e->lay.stride, hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(hipEventRecord(s.copied, e->copy_stream));
    HIP_TRY(hipStreamWaitEvent(e->stream, s.copied, 0));
    rc = launch(e, s.d_rec, n_rows, si, false, st, nullptr, row0);
  }
  if (rc) return rc;
  s.rows_launched += n_rows;
  if (s.rows_launched < n_total) return EP_OK;
  s.rows_launched = 0;
  s.n = n_total;
  HIP_TRY(hipEventRecord(s.done, st));
  s.busy = true;
  e->submit_idx++;
  *completed = 1;
  return EP_OK;
}

// Gives up the batch being committed in parts (a part's launch failed): waits for the parts that
// did go out and forgets them, so that the slot starts clean for the next batch.
int ep_ygo_abort_slot(ep_ygo_encoder* e) {
  if (!e) return fail(EP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  Slot& s = e->slots[(size_t)(e->submit_idx % e->slots.size())];
  if (s.rows_launched == 0) return EP_OK;
  EP_ENTER_DEVICE(e->device);
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipStreamSynchronize(e->copy_stream));
  s.rows_launched = 0;
  return EP_OK;
}

// Emits `n` states straight into the next pinned slot in the packed form and commits it: the
// single-producer counterpart of ep_ygo_emit_arrival (a sync pool whose host thread has all n
// duel results at hand, BASELINE config 3). status[i] (may be NULL) receives each emit's code; a
// state the emitter refuses becomes an empty record (no options), so the batch keeps its n rows.
int ep_ygo_emit_records_packed(ep_ygo_encoder* e, const ygo_raw_state* in, int32_t n, int32_t* status) {
  if (!e || (!in && n > 0)) return fail(EP_ERR_INVALID, "null argument");
  if (n < 0 || n > e->cfg.max_batch) return fail(EP_ERR_INVALID, "n = %d outside [0, max_batch = %d]", n, e->cfg.max_batch);
  std::lock_guard<std::recursive_mutex> whole(e->featurize_mu);
  void* slot = nullptr;
  int rc = ep_ygo_acquire_slot(e, &slot);
  if (rc) return rc;
  static thread_local std::vector<uint32_t> offs;
  static thread_local std::vector<uint8_t> scratch;
  offs.resize((size_t)std::max(n, 1));
  if (scratch.size() < (size_t)e->lay.stride) scratch.resize((size_t)e->lay.stride);
  int64_t used = 0;
  int first = EP_OK;
  for (int i = 0; i < n; ++i) {
    // (the emitter clears a whole stride: it writes into the scratch record, the used prefix moves on)
    int erc = ep_ygo_emit_record(&e->cfg, in + i, scratch.data());
    if (status) status[i] = erc;
    if (erc && !first) first = erc;  // (the record is then the state's empty observation)
    const int64_t len = ep_ygo_record_used_bytes(&e->cfg, scratch.data());
    memcpy((char*)slot + used, scratch.data(), (size_t)len);
    offs[(size_t)i] = (uint32_t)used;
    used += len;
  }
  (void)first;  // per-record codes travel in `status`; the batch itself is complete
  return commit_slot_impl(e, n, n > 0 ? offs.data() : nullptr, used);
}

int ep_ygo_submit(ep_ygo_encoder* e, const void* host_records, int32_t n) {
  if (!e || (!host_records && n > 0)) return fail(EP_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> whole(e->featurize_mu);  // acquire + copy + commit: one slot, one owner
  if (n < 0 || n > e->cfg.max_batch) return fail(EP_ERR_INVALID, "n = %d outside [0, max_batch = %d]", n, e->cfg.max_batch);
  void* dst = nullptr;
  int rc = ep_ygo_acquire_slot(e, &dst);
  if (rc) return rc;
  memcpy(dst, host_records, (size_t)n * e->lay.stride);
  return ep_ygo_commit_slot(e, n);
}

int ep_ygo_collect(ep_ygo_encoder* e, int32_t mem_out, ep_batch* out) {
  if (!e || !out) return fail(EP_ERR_INVALID, "null argument");
  if (epc::bad_mem(mem_out)) return fail(EP_ERR_INVALID, "mem_out = %d is neither EP_MEM_HOST nor EP_MEM_DEVICE", mem_out);
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->collect_idx == e->submit_idx) return fail(EP_ERR_WOULD_BLOCK, "nothing submitted");
  EP_ENTER_DEVICE(e->device);
  const int si = (int)(e->collect_idx % e->slots.size());
  Slot& s = e->slots[si];
  HIP_TRY(hipEventSynchronize(s.done));
  int rc = describe(e, si, mem_out, out);
  if (rc) return rc;
  s.busy = false;
  e->collect_idx++;
  if (e->h_err[si]) {
    e->h_err[si] = 0;
    return fail(EP_ERR_ACTION_CHECK,
                "a record hit a condition on which the reference throws (unknown card code / attribute / race / "
                "position / phase, a spec that is not on the field, or more rows than max_cards)");
  }
  return EP_OK;
}

int ep_ygo_featurize(ep_ygo_encoder* e, const void* records, int32_t n, int32_t mem_in, int32_t mem_out, ep_batch* out) {
  if (!e || !out || (!records && n > 0)) return fail(EP_ERR_INVALID, "null argument");
  if (epc::bad_mem(mem_in) || epc::bad_mem(mem_out))
    return fail(EP_ERR_INVALID, "mem_in = %d / mem_out = %d: each must be EP_MEM_HOST or EP_MEM_DEVICE", mem_in, mem_out);
  if (n < 0 || n > e->cfg.max_batch) return fail(EP_ERR_INVALID, "n = %d outside [0, max_batch = %d]", n, e->cfg.max_batch);
  // submit + collect as ONE operation: collect hands out the OLDEST batch in flight, so with
  // another batch still in flight (or another thread submitting in between) the caller would get
  // somebody else's results for its records
  std::lock_guard<std::recursive_mutex> whole(e->featurize_mu);
  {
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->async_B) return fail(EP_ERR_INVALID, "the encoder is in arrival-ordered mode (ep_ygo_async_begin)");
    if (e->submit_idx != e->collect_idx)
      return fail(EP_ERR_INVALID, "%llu batches are in flight from ep_ygo_submit / ep_ygo_commit_slot: collect them first "
                                  "(ep_ygo_featurize returns the batch it was given, ep_ygo_collect the oldest one)",
                  (unsigned long long)(e->submit_idx - e->collect_idx));
  }
  if (mem_in == EP_MEM_HOST) {
    int rc = ep_ygo_submit(e, records, n);
    if (rc) return rc;
    return ep_ygo_collect(e, mem_out, out);
  }
  int rc = ep_ygo_submit_device(e, records, n);
  if (rc) return rc;
  return ep_ygo_collect(e, mem_out, out);
}

int ep_ygo_submit_device(ep_ygo_encoder* e, const void* device_records, int32_t n) {
  if (!e || (!device_records && n > 0)) return fail(EP_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> whole(e->featurize_mu);
  if (n < 0 || n > e->cfg.max_batch) return fail(EP_ERR_INVALID, "n = %d outside [0, max_batch = %d]", n, e->cfg.max_batch);
  // device-resident records: no staging, the kernel reads the caller's buffer
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->d_table) return fail(EP_ERR_INVALID, "no card table: call ep_ygo_set_card_table first");
  if (e->submit_idx - e->collect_idx >= e->slots.size())
    return fail(EP_ERR_RING_FULL, "%zu batches in flight: collect one first", e->slots.size());
  EP_ENTER_DEVICE(e->device);
  const int si = (int)(e->submit_idx % e->slots.size());
  Slot& s = e->slots[si];
  s.n = n;
  if (n > 0) {
    int rc = launch(e, (const char*)device_records, n, si);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(s.done, e->stream));
  s.busy = true;
  e->submit_idx++;
  return EP_OK;
}

// ---- arrival-ordered record ring ----------------------------------------------------------
// BASELINE config 5: duel threads finish their env-steps in no particular order; each takes the
// next row of the batch being filled (one atomic add: StateBufferQueue::Allocate,
// state_buffer_queue.h:114-128), writes its record IN PLACE into the pinned slot, and reports it
// done (StateBuffer::Done, state_buffer.h:124-129); the thread whose record completes the batch
// enqueues the encode kernel. Batches are handed out in order by ep_ygo_recv
// (StateBufferQueue::Wait, :139-150). Nothing is copied on the host.
int ep_ygo_async_begin(ep_ygo_encoder* e, int32_t batch_size) {
  if (!e) return fail(EP_ERR_INVALID, "null encoder");
  std::lock_guard<std::mutex> lk(e->mu);
  if (batch_size < 1 || batch_size > e->cfg.max_batch)
    return fail(EP_ERR_INVALID, "batch_size = %d outside [1, max_batch = %d]", batch_size, e->cfg.max_batch);
  if (e->submit_idx != e->collect_idx) return fail(EP_ERR_INVALID, "batches in flight: collect them first");
  if (e->async_B && e->a_alloc.load() != e->a_recv.load() * (uint64_t)e->async_B)
    return fail(EP_ERR_INVALID, "records of the previous arrival-ordered run are still outstanding");
  if (!e->d_table) return fail(EP_ERR_INVALID, "no card table: call ep_ygo_set_card_table first");
  if (e->slots.size() < 2) return fail(EP_ERR_INVALID, "arrival-ordered mode needs ring_depth >= 2");
  e->async_B = batch_size;
  e->a_alloc.store(0);
  e->a_recv.store(0);
  e->a_space_wait_ns.store(0);
  e->a_done.reset(new std::atomic<int>[e->slots.size()]);
  e->a_bytes.reset(new std::atomic<uint32_t>[e->slots.size()]);
  for (size_t i = 0; i < e->slots.size(); ++i) { e->a_done[i].store(0); e->a_bytes[i].store(0); }
  e->a_packed = false;
  e->a_last_slot = -1;
  for (auto& sl : e->slots)
    if (!sl.h_ids) sl.h_ids.reset(new int32_t[(size_t)e->cfg.max_batch]());
  e->a_committed.assign(e->slots.size(), 0);
  e->a_closing.store(false);
  return EP_OK;
}

// Shutdown of the arrival-ordered ring: producers sleeping on a full ring and a consumer sleeping
// on an incomplete batch return EP_ERR_WOULD_BLOCK, and so does every later alloc that would have
// to wait. For owners that must join their producer threads (ep_ygo_pool_destroy); the ring is
// usable again after the next ep_ygo_async_begin.
int ep_ygo_async_end(ep_ygo_encoder* e) {
  if (!e) return fail(EP_ERR_INVALID, "null encoder");
  std::lock_guard<std::mutex> lk(e->mu);
  e->a_closing.store(true);
  e->cv_space.notify_all();
  e->cv_ready.notify_all();
  return EP_OK;
}

int ep_ygo_alloc_record(ep_ygo_encoder* e, void** record, uint64_t* ticket) {
  if (!e || !record || !ticket) return fail(EP_ERR_INVALID, "null argument");
  const int B = e->async_B;
  if (!B) return fail(EP_ERR_INVALID, "call ep_ygo_async_begin first");
  const uint64_t depth = e->slots.size();
  const uint64_t pos = e->a_alloc.fetch_add(1, std::memory_order_relaxed);
  const uint64_t bi = pos / (uint64_t)B;
  // the slot of batch bi held batch bi - depth, and the views of the batch received last are still
  // the consumer's: the slot is free once batch bi - depth + 1 has been received as well. (The
  // reference throws out_of_range when its queue overflows, state_buffer.h:111; producers wait here.)
  if (bi + 1 >= e->a_recv.load(std::memory_order_acquire) + depth) {
    const auto t0 = std::chrono::steady_clock::now();
    {
      std::unique_lock<std::mutex> lk(e->mu);
      e->cv_space.wait(lk, [&] { return e->a_closing.load() || bi + 1 < e->a_recv.load(std::memory_order_acquire) + depth; });
    }
    if (e->a_closing.load()) return fail(EP_ERR_WOULD_BLOCK, "the arrival-ordered ring was closed (ep_ygo_async_end) while it was full");
    e->a_space_wait_ns.fetch_add(
        (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
        std::memory_order_relaxed);
  }
  *record = e->slots[bi % depth].h_rec + (pos % (uint64_t)B) * (uint64_t)e->lay.stride;
  *ticket = pos;
  return EP_OK;
}

int ep_ygo_record_done(ep_ygo_encoder* e, uint64_t ticket) {
  if (!e) return fail(EP_ERR_INVALID, "null encoder");
  const int B = e->async_B;
  if (!B) return fail(EP_ERR_INVALID, "call ep_ygo_async_begin first");
  const uint64_t depth = e->slots.size();
  const uint64_t bi = ticket / (uint64_t)B;
  const int si = (int)(bi % depth);
  if (!e->a_packed) {  // (packed rows: put_arrival has the record at hand)
    const uint64_t row = ticket % (uint64_t)B;
    e->slots[si].h_ids[row] = reinterpret_cast<const ygo_header*>(e->slots[si].h_rec + row * (uint64_t)e->lay.stride)->env_id;
  }
  // acq_rel: the thread that sees the B-th completion also sees every record of the batch
  if (e->a_done[si].fetch_add(1, std::memory_order_acq_rel) + 1 != B) return EP_OK;
  e->a_done[si].store(0, std::memory_order_relaxed);
  std::lock_guard<std::mutex> lk(e->mu);
  EP_ENTER_DEVICE(e->device);
  Slot& s = e->slots[si];
  s.n = B;
  hipStream_t st = (bi & 1) ? e->copy_stream : e->stream;
  int rc;
  if (e->a_packed) {
    // packed batch: the offset table and exactly the bytes the records use go to the device in
    // ONE copy-engine transfer (large PCIe reads: 50 GB/s, against 35 GB/s for the kernel's own
    // 64-byte reads of a pinned slot and 3328 B per record for a fixed-stride copy)
    st = e->stream;
    const size_t used = e->a_bytes[si].exchange(0, std::memory_order_relaxed);
    static const size_t in_place_below =
        getenv("EP_YGO_PACKED_IN_PLACE_BYTES") ? (size_t)atoll(getenv("EP_YGO_PACKED_IN_PLACE_BYTES")) : (size_t)512 << 10;
    if (e->zero_copy && used <= in_place_below) {
      // up to about 300 records: the kernel reads the offset table and the records from the pinned slot
      // itself - one launch instead of copy + event + wait + launch (33 us of calls), and the bytes
      // start to move with the kernel. Measured through an async pool's recv / send loop: 57 -> 28 us
      // per batch of 4, 58 -> 39 us per batch of 32, 77 -> 42 us per batch of 256. At config 5's 512
      // records per batch the two ways are within the noise of each other (the workers are the
      // limit); from there on the copy engine's bigger PCIe reads (50 GB/s against 35) are kept.
      void* dev_tab = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dev_tab, s.h_tab, 0));
      rc = launch(e, (const char*)dev_tab + e->tab_bytes, B, si, /*lean_fetch=*/true, st, (const uint32_t*)dev_tab);
    } else {
      hipStream_t cs = (bi & 1) ? e->copy_stream2 : e->copy_stream;
      HIP_TRY(hipMemcpyAsync(s.d_tab, s.h_tab, e->tab_bytes + used, hipMemcpyHostToDevice, cs));
      HIP_TRY(hipEventRecord(s.copied, cs));
      HIP_TRY(hipStreamWaitEvent(e->stream, s.copied, 0));
      rc = launch(e, s.d_rec, B, si, false, st, (const uint32_t*)s.d_tab);
    }
  } else if (e->zero_copy) {
    void* dev_view = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dev_view, s.h_rec, 0));
    rc = launch(e, (const char*)dev_view, B, si, /*lean_fetch=*/true, st);
  } else {
    st = e->stream;
    HIP_TRY(hipMemcpyAsync(s.d_rec, s.h_rec, (size_t)B * e->lay.stride, hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(hipEventRecord(s.copied, e->copy_stream));
    HIP_TRY(hipStreamWaitEvent(e->stream, s.copied, 0));
    rc = launch(e, s.d_rec, B, si);
  }
  if (rc) return rc;
  HIP_TRY(hipEventRecord(s.done, st));
  e->a_committed[si] = bi + 1;
  e->cv_ready.notify_all();
  return EP_OK;
}

int ep_ygo_recv(ep_ygo_encoder* e, int32_t mem_out, ep_batch* out) {
  if (!e || !out) return fail(EP_ERR_INVALID, "null argument");
  if (epc::bad_mem(mem_out)) return fail(EP_ERR_INVALID, "mem_out = %d is neither EP_MEM_HOST nor EP_MEM_DEVICE", mem_out);
  if (!e->async_B) return fail(EP_ERR_INVALID, "call ep_ygo_async_begin first");
  std::unique_lock<std::mutex> lk(e->mu);
  const uint64_t depth = e->slots.size();
  const uint64_t bi = e->a_recv.load(std::memory_order_relaxed);
  const int si = (int)(bi % depth);
  auto ready = [&] { return e->a_committed[si] == bi + 1 || e->a_closing.load(); };
  if (e->a_committed[si] != bi + 1 && e->a_closing.load()) return fail(EP_ERR_WOULD_BLOCK, "the arrival-ordered ring was closed");
  if (!ready()) {
    if (e->a_recv_timeout_ms == 0) return fail(EP_ERR_WOULD_BLOCK, "batch %llu is not complete yet", (unsigned long long)bi);
    if (e->a_recv_timeout_ms < 0) e->cv_ready.wait(lk, ready);
    else if (!e->cv_ready.wait_for(lk, std::chrono::milliseconds(e->a_recv_timeout_ms), ready))
      return fail(EP_ERR_WOULD_BLOCK, "batch %llu is not complete yet", (unsigned long long)bi);
    if (e->a_committed[si] != bi + 1) return fail(EP_ERR_WOULD_BLOCK, "the arrival-ordered ring was closed");
  }
  lk.unlock();
  EP_ENTER_DEVICE(e->device);
  HIP_TRY(hipEventSynchronize(e->slots[si].done));
  lk.lock();
  e->collect_idx = bi;  // (describe() numbers the batch with it)
  int rc = describe(e, si, mem_out, out);
  e->collect_idx = e->submit_idx = 0;
  if (rc) return rc;
  e->a_committed[si] = 0;
  e->a_last_slot = si;
  const bool bad = e->h_err[si] != 0;
  e->h_err[si] = 0;
  e->a_recv.store(bi + 1, std::memory_order_release);
  e->cv_space.notify_all();
  if (bad)
    return fail(EP_ERR_ACTION_CHECK,
                "a record hit a condition on which the reference throws (unknown card code / attribute / race / "
                "position / phase, a spec that is not on the field, or more rows than max_cards)");
  return EP_OK;
}

int ep_ygo_last_env_ids(ep_ygo_encoder* e, const int32_t** ids, int32_t* n) {
  if (!e || !ids || !n) return fail(EP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->async_B) return fail(EP_ERR_INVALID, "arrival-ordered mode only (ep_ygo_async_begin): in submit / collect mode the caller placed the rows itself");
  if (e->a_last_slot < 0) return fail(EP_ERR_WOULD_BLOCK, "no batch has been received yet");
  // (the slot of the batch received last is not refilled before the next ep_ygo_recv: ep_ygo_alloc_record)
  *ids = e->slots[(size_t)e->a_last_slot].h_ids.get();
  *n = e->async_B;
  return EP_OK;
}

int64_t ep_ygo_record_used_bytes(const ep_ygo_config* cfg, const void* record) {
  if (check_config(cfg) || !record) return -1;
  const Layout lay = make_layout(*cfg);
  const ygo_header* h = (const ygo_header*)record;
  if (h->num_options == 0) return lay.off_cards;  // WriteState's early-out reads the header (and the ring travels with it)
  int listed = 0;
  for (int li = 0; li < 7; ++li) listed += h->loc_count[0][li];
  for (int li = 2; li < 6; ++li) listed += h->loc_count[1][li];
  if (h->revealed_any) listed += h->loc_count[1][1];
  const int n_opt = std::min<int>(h->num_options, cfg->max_options);
  const int64_t used = (int64_t)lay.off_cards + (int64_t)listed * (int64_t)sizeof(ygo_card) +
                       (int64_t)n_opt * ygk::option_bytes(cfg->max_multi_select);
  return std::min<int64_t>((used + 15) & ~15ll, lay.stride);
}

// alloc + copy + done for one finished record, packed: rows in arrival order, bytes back to back.
static int put_arrival(ep_ygo_encoder* e, const void* record, int64_t used) {
  void* row = nullptr;
  uint64_t ticket = 0;
  int rc = ep_ygo_alloc_record(e, &row, &ticket);  // (the row pointer is not used in packed mode)
  if (rc) return rc;
  const int B = e->async_B;
  const uint64_t bi = ticket / (uint64_t)B;
  const int si = (int)(bi % e->slots.size());
  const uint32_t off = e->a_bytes[si].fetch_add((uint32_t)used, std::memory_order_relaxed);
  Slot& s = e->slots[si];
  memcpy(s.h_rec + off, record, (size_t)used);
  reinterpret_cast<uint32_t*>(s.h_tab)[ticket % (uint64_t)B] = off;
  s.h_ids[ticket % (uint64_t)B] = reinterpret_cast<const ygo_header*>(record)->env_id;
  return ep_ygo_record_done(e, ticket);
}

int ep_ygo_set_packed(ep_ygo_encoder* e, int32_t enabled) {
  if (!e) return fail(EP_ERR_INVALID, "null encoder");
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->async_B) return fail(EP_ERR_INVALID, "call ep_ygo_async_begin first");
  if (e->a_alloc.load() != e->a_recv.load() * (uint64_t)e->async_B) return fail(EP_ERR_INVALID, "records are outstanding");
  if ((uint64_t)e->lay.stride * (uint64_t)e->async_B >= (1ull << 32)) return fail(EP_ERR_TOO_LARGE, "a packed slot must stay below 4 GiB");
  e->a_packed = enabled != 0;
  return EP_OK;
}

int ep_ygo_put_arrival(ep_ygo_encoder* e, const void* record) {
  if (!e || !record) return fail(EP_ERR_INVALID, "null argument");
  if (!e->a_packed) return fail(EP_ERR_INVALID, "call ep_ygo_set_packed(enc, 1) first");
  const int64_t used = ep_ygo_record_used_bytes(&e->cfg, record);
  if (used < 0) return fail(EP_ERR_INVALID, "bad record");
  return put_arrival(e, record, used);
}

int ep_ygo_emit_arrival(ep_ygo_encoder* e, const ygo_raw_state* in) {
  if (!e || !in) return fail(EP_ERR_INVALID, "null argument");
  if (!e->a_packed) return fail(EP_ERR_INVALID, "call ep_ygo_set_packed(enc, 1) first");
  static thread_local std::vector<uint8_t> scratch;
  if (scratch.size() < (size_t)e->lay.stride) scratch.resize((size_t)e->lay.stride);
  int rc = ep_ygo_emit_record(&e->cfg, in, scratch.data());
  if (rc) return rc;
  return put_arrival(e, scratch.data(), ep_ygo_record_used_bytes(&e->cfg, scratch.data()));
}

int ep_ygo_ring_stats(ep_ygo_encoder* e, double* producer_wait_seconds, uint64_t* records, uint64_t* batches_received) {
  if (!e) return fail(EP_ERR_INVALID, "null encoder");
  if (producer_wait_seconds) *producer_wait_seconds = 1e-9 * (double)e->a_space_wait_ns.load();
  if (records) *records = e->a_alloc.load();
  if (batches_received) *batches_received = e->a_recv.load();
  return EP_OK;
}

int ep_ygo_set_recv_timeout(ep_ygo_encoder* e, int64_t milliseconds) {
  if (!e) return fail(EP_ERR_INVALID, "null encoder");
  std::lock_guard<std::mutex> lk(e->mu);
  e->a_recv_timeout_ms = milliseconds;
  return EP_OK;
}

int ep_ygo_sync(ep_ygo_encoder* e) {
  if (!e) return fail(EP_ERR_INVALID, "null encoder");
  EP_ENTER_DEVICE(e->device);
  HIP_TRY(hipStreamSynchronize(e->copy_stream));
  HIP_TRY(hipStreamSynchronize(e->copy_stream2));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return EP_OK;
}

void* ep_ygo_stream(ep_ygo_encoder* e) { return e ? (void*)e->stream : nullptr; }

int ep_ygo_set_zero_copy(ep_ygo_encoder* e, int32_t enabled) {
  if (!e) return fail(EP_ERR_INVALID, "null encoder");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->submit_idx != e->collect_idx) return fail(EP_ERR_INVALID, "batches in flight: collect them first");
  e->zero_copy = enabled != 0;
  return EP_OK;
}

int ep_ygo_set_timing(ep_ygo_encoder* e, int32_t enabled) {
  if (!e) return fail(EP_ERR_INVALID, "null encoder");
  e->timing = enabled != 0;
  return EP_OK;
}

int ep_ygo_kernel_time_us(ep_ygo_encoder* e, double* total_us, int64_t* launches, int32_t reset) {
  if (!e) return fail(EP_ERR_INVALID, "null encoder");
  std::lock_guard<std::mutex> lk(e->mu);
  EP_ENTER_DEVICE(e->device);
  for (auto& ab : e->timed) {
    HIP_TRY(hipEventSynchronize(ab.second));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ab.first, ab.second));
    e->timed_total_us += (double)ms * 1e3;
    e->timed_launches++;
    (void)hipEventDestroy(ab.first);
    (void)hipEventDestroy(ab.second);
  }
  e->timed.clear();
  if (total_us) *total_us = e->timed_total_us;
  if (launches) *launches = e->timed_launches;
  if (reset) { e->timed_total_us = 0; e->timed_launches = 0; }
  return EP_OK;
}

}  // extern "C"

// ---- ygopro-core constants (include/envpool_amd_ygo_consts.h) ----------------------------------
extern "C" {

void ep_ygo_default_constants(ep_ygo_core_constants* o) {
  if (!o) return;
  const int32_t loc[8] = {EP_YGO_LOCATION_DECK, EP_YGO_LOCATION_HAND, EP_YGO_LOCATION_MZONE, EP_YGO_LOCATION_SZONE,
                          EP_YGO_LOCATION_GRAVE, EP_YGO_LOCATION_REMOVED, EP_YGO_LOCATION_EXTRA, EP_YGO_LOCATION_OVERLAY};
  const int32_t pos[8] = {EP_YGO_POS_FACEUP_ATTACK, EP_YGO_POS_FACEDOWN_ATTACK, EP_YGO_POS_FACEUP_DEFENSE, EP_YGO_POS_FACEDOWN_DEFENSE,
                          EP_YGO_POS_FACEUP, EP_YGO_POS_FACEDOWN, EP_YGO_POS_ATTACK, EP_YGO_POS_DEFENSE};
  const uint32_t type[EP_YGO_NUM_TYPES] = {
      EP_YGO_TYPE_MONSTER, EP_YGO_TYPE_SPELL, EP_YGO_TYPE_TRAP, EP_YGO_TYPE_NORMAL, EP_YGO_TYPE_EFFECT, EP_YGO_TYPE_FUSION, EP_YGO_TYPE_RITUAL,
      EP_YGO_TYPE_TRAPMONSTER, EP_YGO_TYPE_SPIRIT, EP_YGO_TYPE_UNION, EP_YGO_TYPE_DUAL, EP_YGO_TYPE_TUNER, EP_YGO_TYPE_SYNCHRO, EP_YGO_TYPE_TOKEN,
      EP_YGO_TYPE_QUICKPLAY, EP_YGO_TYPE_CONTINUOUS, EP_YGO_TYPE_EQUIP, EP_YGO_TYPE_FIELD, EP_YGO_TYPE_COUNTER, EP_YGO_TYPE_FLIP, EP_YGO_TYPE_TOON,
      EP_YGO_TYPE_XYZ, EP_YGO_TYPE_PENDULUM, EP_YGO_TYPE_SPSUMMON, EP_YGO_TYPE_LINK};
  const uint32_t attr[EP_YGO_NUM_ATTRIBUTES] = {EP_YGO_ATTRIBUTE_EARTH, EP_YGO_ATTRIBUTE_WATER, EP_YGO_ATTRIBUTE_FIRE, EP_YGO_ATTRIBUTE_WIND,
                                                EP_YGO_ATTRIBUTE_LIGHT, EP_YGO_ATTRIBUTE_DARK, EP_YGO_ATTRIBUTE_DEVINE};
  const int32_t phase[EP_YGO_NUM_PHASES] = {EP_YGO_PHASE_DRAW, EP_YGO_PHASE_STANDBY, EP_YGO_PHASE_MAIN1, EP_YGO_PHASE_BATTLE_START,
                                            EP_YGO_PHASE_BATTLE_STEP, EP_YGO_PHASE_DAMAGE, EP_YGO_PHASE_DAMAGE_CAL, EP_YGO_PHASE_BATTLE,
                                            EP_YGO_PHASE_MAIN2, EP_YGO_PHASE_END};
  const int32_t msg[EP_YGO_NUM_MSGS] = {
      EP_YGO_MSG_SELECT_IDLECMD, EP_YGO_MSG_SELECT_CHAIN, EP_YGO_MSG_SELECT_CARD, EP_YGO_MSG_SELECT_TRIBUTE, EP_YGO_MSG_SELECT_POSITION,
      EP_YGO_MSG_SELECT_EFFECTYN, EP_YGO_MSG_SELECT_YESNO, EP_YGO_MSG_SELECT_BATTLECMD, EP_YGO_MSG_SELECT_UNSELECT_CARD,
      EP_YGO_MSG_SELECT_OPTION, EP_YGO_MSG_SELECT_PLACE, EP_YGO_MSG_SELECT_SUM, EP_YGO_MSG_SELECT_DISFIELD, EP_YGO_MSG_ANNOUNCE_ATTRIB};
  memcpy(o->location, loc, sizeof(loc));
  memcpy(o->pos, pos, sizeof(pos));
  memcpy(o->type, type, sizeof(type));
  memcpy(o->attribute, attr, sizeof(attr));
  for (int i = 0; i < EP_YGO_NUM_RACES; ++i) o->race[i] = EP_YGO_RACE(i);
  memcpy(o->phase, phase, sizeof(phase));
  memcpy(o->msg, msg, sizeof(msg));
  o->query_equip_card = EP_YGO_QUERY_EQUIP_CARD;
  o->len_empty = EP_YGO_LEN_EMPTY;
}

int ep_ygo_check_constants(const ep_ygo_core_constants* t) {
  if (!t) return fail(EP_ERR_INVALID, "null argument");
  ep_ygo_core_constants m;
  ep_ygo_default_constants(&m);
  static const char* const kLoc[8] = {"LOCATION_DECK", "LOCATION_HAND", "LOCATION_MZONE", "LOCATION_SZONE", "LOCATION_GRAVE",
                                      "LOCATION_REMOVED", "LOCATION_EXTRA", "LOCATION_OVERLAY"};
  static const char* const kPos[8] = {"POS_FACEUP_ATTACK", "POS_FACEDOWN_ATTACK", "POS_FACEUP_DEFENSE", "POS_FACEDOWN_DEFENSE", "POS_FACEUP",
                                      "POS_FACEDOWN", "POS_ATTACK", "POS_DEFENSE"};
  static const char* const kType[EP_YGO_NUM_TYPES] = {
      "TYPE_MONSTER", "TYPE_SPELL", "TYPE_TRAP", "TYPE_NORMAL", "TYPE_EFFECT", "TYPE_FUSION", "TYPE_RITUAL", "TYPE_TRAPMONSTER", "TYPE_SPIRIT",
      "TYPE_UNION", "TYPE_DUAL", "TYPE_TUNER", "TYPE_SYNCHRO", "TYPE_TOKEN", "TYPE_QUICKPLAY", "TYPE_CONTINUOUS", "TYPE_EQUIP", "TYPE_FIELD",
      "TYPE_COUNTER", "TYPE_FLIP", "TYPE_TOON", "TYPE_XYZ", "TYPE_PENDULUM", "TYPE_SPSUMMON", "TYPE_LINK"};
  static const char* const kAttr[EP_YGO_NUM_ATTRIBUTES] = {"ATTRIBUTE_EARTH", "ATTRIBUTE_WATER", "ATTRIBUTE_FIRE", "ATTRIBUTE_WIND",
                                                           "ATTRIBUTE_LIGHT", "ATTRIBUTE_DARK", "ATTRIBUTE_DEVINE"};
  static const char* const kRace[EP_YGO_NUM_RACES] = {
      "RACE_WARRIOR", "RACE_SPELLCASTER", "RACE_FAIRY", "RACE_FIEND", "RACE_ZOMBIE", "RACE_MACHINE", "RACE_AQUA", "RACE_PYRO", "RACE_ROCK",
      "RACE_WINDBEAST", "RACE_PLANT", "RACE_INSECT", "RACE_THUNDER", "RACE_DRAGON", "RACE_BEAST", "RACE_BEASTWARRIOR", "RACE_DINOSAUR",
      "RACE_FISH", "RACE_SEASERPENT", "RACE_REPTILE", "RACE_PSYCHO", "RACE_DEVINE", "RACE_CREATORGOD", "RACE_WYRM", "RACE_CYBERSE", "RACE_ILLUSION"};
  static const char* const kPhase[EP_YGO_NUM_PHASES] = {"PHASE_DRAW", "PHASE_STANDBY", "PHASE_MAIN1", "PHASE_BATTLE_START", "PHASE_BATTLE_STEP",
                                                        "PHASE_DAMAGE", "PHASE_DAMAGE_CAL", "PHASE_BATTLE", "PHASE_MAIN2", "PHASE_END"};
  static const char* const kMsg[EP_YGO_NUM_MSGS] = {
      "MSG_SELECT_IDLECMD", "MSG_SELECT_CHAIN", "MSG_SELECT_CARD", "MSG_SELECT_TRIBUTE", "MSG_SELECT_POSITION", "MSG_SELECT_EFFECTYN",
      "MSG_SELECT_YESNO", "MSG_SELECT_BATTLECMD", "MSG_SELECT_UNSELECT_CARD", "MSG_SELECT_OPTION", "MSG_SELECT_PLACE", "MSG_SELECT_SUM",
      "MSG_SELECT_DISFIELD", "MSG_ANNOUNCE_ATTRIB"};
  auto differ = [](const char* name, long long mine, long long theirs) {
    return fail(EP_ERR_INVALID,
                "ygopro-core constant %s: this library was built for 0x%llx (%lld), the caller's common.h has 0x%llx (%lld) - "
                "observations would carry wrong ids; rebuild with include/envpool_amd_ygo_consts.h corrected",
                name, mine, mine, theirs, theirs);
  };
  for (int i = 0; i < 8; ++i) if (m.location[i] != t->location[i]) return differ(kLoc[i], m.location[i], t->location[i]);
  for (int i = 0; i < 8; ++i) if (m.pos[i] != t->pos[i]) return differ(kPos[i], m.pos[i], t->pos[i]);
  for (int i = 0; i < EP_YGO_NUM_TYPES; ++i) if (m.type[i] != t->type[i]) return differ(kType[i], m.type[i], t->type[i]);
  for (int i = 0; i < EP_YGO_NUM_ATTRIBUTES; ++i) if (m.attribute[i] != t->attribute[i]) return differ(kAttr[i], m.attribute[i], t->attribute[i]);
  for (int i = 0; i < EP_YGO_NUM_RACES; ++i) if (m.race[i] != t->race[i]) return differ(kRace[i], m.race[i], t->race[i]);
  for (int i = 0; i < EP_YGO_NUM_PHASES; ++i) if (m.phase[i] != t->phase[i]) return differ(kPhase[i], m.phase[i], t->phase[i]);
  for (int i = 0; i < EP_YGO_NUM_MSGS; ++i) if (m.msg[i] != t->msg[i]) return differ(kMsg[i], m.msg[i], t->msg[i]);
  if (m.query_equip_card != t->query_equip_card) return differ("QUERY_EQUIP_CARD", m.query_equip_card, t->query_equip_card);
  if (m.len_empty != t->len_empty) return differ("LEN_EMPTY", m.len_empty, t->len_empty);
  return EP_OK;
}

}  // extern "C"

// =========================================================================================
// Host adapter: reference data forms -> POD record. No GPU involved.
// =========================================================================================
namespace {

constexpr uint32_t kQueryEquipCard = EP_YGO_QUERY_EQUIP_CARD;  // (include/envpool_amd_ygo_consts.h)
constexpr uint32_t kLenEmpty = EP_YGO_LEN_EMPTY;
constexpr uint8_t kLocationOverlay = EP_YGO_LOCATION_OVERLAY;

inline uint32_t ld32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}

// Where the parsed cards of one record go: straight into the record's card area, which holds
// `cap` of them. Cards past the capacity are counted, not kept (the caller refuses the state).
struct CardSink {
  ygo_card* out;
  int n, cap;
  void push(const ygo_card& c) {
    if (n < cap) out[n] = c;
    ++n;
  }
};

// get_cards_in_location (ygopro.h:2060-2137) without the static-table copy: that lookup happens
// on the device. One bounds check per fixed-size run of words, not one per word: this parser is
// 70 % of the emitter's time.
int parse_location(const uint8_t* b, int len, CardSink* out) {
  auto truncated = [] { return fail(EP_ERR_INVALID, "truncated query_field_card buffer"); };
  int p = 0;
  while (p < len) {
    if (p + 4 > len) return truncated();
    if (ld32(b + p) == kLenEmpty) { p += 4; continue; }
    // length | flags code info level rank attack defense
    if (p + 32 > len) return truncated();
    const uint32_t f = ld32(b + p + 4);
    ygo_card c;
    memset(&c, 0, sizeof(c));
    c.code = ld32(b + p + 8);
    c.controller = b[p + 12];
    c.location = b[p + 13];
    c.sequence = b[p + 14];
    c.position = b[p + 15];
    c.level = b[p + 16];   // (low bytes of little-endian words)
    c.rank = b[p + 20];
    c.attack = (uint16_t)(ld32(b + p + 24) & 0xffff);
    c.defense = (uint16_t)(ld32(b + p + 28) & 0xffff);
    p += 32;
    if (f & kQueryEquipCard) p += 4;
    if (p + 4 > len) return truncated();
    const uint32_t n_xyz = ld32(b + p);
    p += 4;
    if (n_xyz > (uint32_t)(len - p) / 4) return truncated();
    for (uint32_t i = 0; i < n_xyz; ++i) {
      ygo_card m;
      memset(&m, 0, sizeof(m));
      m.code = ld32(b + p + 4 * (int)i);
      m.controller = c.controller;
      m.location = c.location | kLocationOverlay;
      m.sequence = c.sequence;
      m.position = (uint8_t)i;
      m.is_material = 1;
      out->push(m);
    }
    p += 4 * (int)n_xyz;
    if (p + 4 > len) return truncated();
    const uint32_t n_counters = ld32(b + p);
    p += 4;
    if (n_counters > (uint32_t)(len - p) / 4) return truncated();
    p += 4 * (int)n_counters;
    // lscale rscale link link_marker
    if (p + 16 > len) return truncated();
    c.link = b[p + 8];
    const uint32_t link_marker = ld32(b + p + 12);
    if (link_marker > 0) c.defense = (uint16_t)(link_marker & 0xffff);
    p += 16;
    out->push(c);
  }
  return EP_OK;
}

// spec string ("oh3", "m2", "x1", "12", "m3a") -> key; mirrors spec_to_ls (:367-400) but keeps
// the presence of the overlay suffix, which the string comparison of spec2index distinguishes.
bool spec_key(std::string_view s, uint32_t* key) {
  size_t i = 0;
  uint32_t opp = 0;
  if (i < s.size() && s[i] == 'o') { opp = 1; ++i; }
  if (i >= s.size()) return false;
  uint32_t loc;
  switch (s[i]) {
    case 'h': loc = EP_YGO_LOCATION_HAND; ++i; break;
    case 'm': loc = EP_YGO_LOCATION_MZONE; ++i; break;
    case 's': loc = EP_YGO_LOCATION_SZONE; ++i; break;
    case 'g': loc = EP_YGO_LOCATION_GRAVE; ++i; break;
    case 'r': loc = EP_YGO_LOCATION_REMOVED; ++i; break;
    case 'x': loc = EP_YGO_LOCATION_EXTRA; ++i; break;
    default:
      if (!isdigit((unsigned char)s[i])) return false;
      loc = EP_YGO_LOCATION_DECK;
  }
  size_t j = i;
  uint32_t v = 0;
  while (j < s.size() && isdigit((unsigned char)s[j])) { v = v * 10 + (uint32_t)(s[j] - '0'); ++j; }
  if (j == i || v == 0 || v > 256) return false;
  uint32_t suf = 0;
  if (j < s.size()) {
    if (j + 1 != s.size() || s[j] < 'a' || s[j] > 'z') return false;
    suf = (uint32_t)(s[j] - 'a') + 1;
  }
  *key = opp | (loc << 8) | ((v - 1) << 16) | (suf << 24);
  return true;
}

int list_id(const char* list, char c) {
  const char* p = c ? strchr(list, c) : nullptr;
  return p ? (int)(p - list) + 1 : -1;
}

int place_id(std::string_view s) {  // cmd_place2id (:680-686)
  size_t i = 0;
  int base = 0;
  if (!s.empty() && s[0] == 'o') { base = 15; i = 1; }
  if (i + 2 != s.size()) return -1;
  const int d = s[i + 1] - '0';
  if (s[i] == 'm' && d >= 1 && d <= 7) return base + d;
  if (s[i] == 's' && d >= 1 && d <= 8) return base + 7 + d;
  return -1;
}

int msg_id(int msg) {  // msg2id (:660-669)
  static const int msgs[EP_YGO_NUM_MSGS] = {
      EP_YGO_MSG_SELECT_IDLECMD, EP_YGO_MSG_SELECT_CHAIN, EP_YGO_MSG_SELECT_CARD, EP_YGO_MSG_SELECT_TRIBUTE, EP_YGO_MSG_SELECT_POSITION,
      EP_YGO_MSG_SELECT_EFFECTYN, EP_YGO_MSG_SELECT_YESNO, EP_YGO_MSG_SELECT_BATTLECMD, EP_YGO_MSG_SELECT_UNSELECT_CARD,
      EP_YGO_MSG_SELECT_OPTION, EP_YGO_MSG_SELECT_PLACE, EP_YGO_MSG_SELECT_SUM, EP_YGO_MSG_SELECT_DISFIELD, EP_YGO_MSG_ANNOUNCE_ATTRIB};
  for (int i = 0; i < 14; ++i)
    if (msgs[i] == msg) return i + 1;
  return -1;
}

int pos_id_host(int p) {
  static const int keys[9] = {0x0, 0x1, 0x2, 0x3, 0x4, 0x5, 0x8, 0xa, 0xc};
  for (int i = 0; i < 9; ++i)
    if (keys[i] == p) return i;
  return -1;
}

enum { F_MSG = 0, F_ACT, F_YESNO, F_PHASE, F_CANCEL, F_POSITION, F_OPTION, F_PLACE, F_ATTRIB };

// The spec strings one option references: views into the option string (no allocation on the
// per-record path). More than the array holds are counted, not kept.
struct SpecList {
  std::string_view v[EP_YGO_MAX_SPECS + 1];
  size_t n = 0;
  void push_back(std::string_view s) {
    if (n <= EP_YGO_MAX_SPECS) v[n] = s;
    ++n;
  }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  std::string_view operator[](size_t i) const { return v[i]; }
};

// The grammar of _set_obs_action (:1674-1765). `specs` receives the referenced spec strings.
// `option_c` is NUL-terminated.
int parse_option(int msg, const char* option_c, uint8_t feats[9], SpecList* specs) {
  const std::string_view option(option_c);
  memset(feats, 0, 9);
  const int mid = msg_id(msg);
  if (mid < 0) return fail(EP_ERR_INVALID, "Unsupported message %d", msg);
  feats[F_MSG] = (uint8_t)mid;
  if (option.empty()) return fail(EP_ERR_INVALID, "empty option string");
  auto need = [&](int id, const char* what) -> int {
    return id < 0 ? fail(EP_ERR_INVALID, "option '%s': unknown %s", option_c, what) : EP_OK;
  };
  auto strip_effect = [](std::string_view* spec) -> uint8_t {
    if (!spec->empty() && isalpha((unsigned char)spec->back())) {
      uint8_t off = (uint8_t)(spec->back() - 'a');
      spec->remove_suffix(1);
      return off;
    }
    return 0;
  };
  int rc;
  switch (msg) {
    case EP_YGO_MSG_SELECT_IDLECMD: {
      if (option == "b" || option == "e") {
        feats[F_PHASE] = (uint8_t)list_id("bme", option[0]);
      } else {
        const char act = option[0];
        std::string_view spec = option.size() >= 2 ? option.substr(2) : std::string_view();
        uint8_t off = 0;
        if (act == 'v') off = strip_effect(&spec);
        if ((rc = need(list_id("trcsmav", act), "act"))) return rc;
        feats[F_ACT] = (uint8_t)(list_id("trcsmav", act) + off);
        specs->push_back(spec);
      }
      break;
    }
    case EP_YGO_MSG_SELECT_CHAIN: {
      if (option[0] == 'c') {
        feats[F_CANCEL] = 1;
      } else {
        std::string_view spec = option;
        const uint8_t off = strip_effect(&spec);
        feats[F_ACT] = (uint8_t)(7 + off);  // 'v'
        specs->push_back(spec);
      }
      break;
    }
    case EP_YGO_MSG_SELECT_CARD: case EP_YGO_MSG_SELECT_TRIBUTE: case EP_YGO_MSG_SELECT_SUM: {  // space separated specs
      size_t start = 0;
      while (start < option.size()) {
        const size_t idx = option.find(' ', start);
        if (idx == std::string_view::npos) { specs->push_back(option.substr(start)); break; }
        specs->push_back(option.substr(start, idx - start));
        start = idx + 1;
      }
      break;
    }
    case EP_YGO_MSG_SELECT_UNSELECT_CARD: {
      if (option[0] == 'f') feats[F_CANCEL] = 2;
      else specs->push_back(option);
      break;
    }
    case EP_YGO_MSG_SELECT_POSITION: {
      const int sh = option[0] - '1';
      const int pid = (sh >= 0 && sh < 8) ? pos_id_host(1 << sh) : -1;
      if ((rc = need(pid, "position"))) return rc;
      feats[F_POSITION] = (uint8_t)pid;
      break;
    }
    case EP_YGO_MSG_SELECT_EFFECTYN: {
      specs->push_back(option.size() >= 2 ? option.substr(2) : std::string_view());
      if ((rc = need(list_id("yn", option[0]), "yes/no"))) return rc;
      feats[F_YESNO] = (uint8_t)list_id("yn", option[0]);
      break;
    }
    case EP_YGO_MSG_SELECT_YESNO: {
      if ((rc = need(list_id("yn", option[0]), "yes/no"))) return rc;
      feats[F_YESNO] = (uint8_t)list_id("yn", option[0]);
      break;
    }
    case EP_YGO_MSG_SELECT_BATTLECMD: {
      if (option == "m" || option == "e") {
        feats[F_PHASE] = (uint8_t)list_id("bme", option[0]);
      } else {
        if ((rc = need(list_id("trcsmav", option[0]), "act"))) return rc;
        feats[F_ACT] = (uint8_t)list_id("trcsmav", option[0]);
        specs->push_back(option.size() >= 2 ? option.substr(2) : std::string_view());
      }
      break;
    }
    case EP_YGO_MSG_SELECT_OPTION:
      feats[F_OPTION] = (uint8_t)(option[0] - '0');
      break;
    case EP_YGO_MSG_SELECT_PLACE: case EP_YGO_MSG_SELECT_DISFIELD: {
      if ((rc = need(place_id(option), "place"))) return rc;
      feats[F_PLACE] = (uint8_t)place_id(option);
      break;
    }
    case EP_YGO_MSG_ANNOUNCE_ATTRIB: {  // attribute2id.at(1 << (c - '1'))
      const int sh = option[0] - '1';
      if (sh < 0 || sh > 6) return fail(EP_ERR_INVALID, "option '%s': unknown attribute", option_c);
      feats[F_ATTRIB] = (uint8_t)(sh + 1);
      break;
    }
    default:
      return fail(EP_ERR_INVALID, "Unsupported message %d", msg);
  }
  return EP_OK;
}

}  // namespace

extern "C" {

}  // extern "C"

static int emit_record_impl(const ep_ygo_config* cfg, const ygo_raw_state* in, void* record_out);

extern "C" {

// A state that cannot be parsed must not leave a half-written record (counts without their cards)
// where a caller may encode it: the record becomes the state's empty observation (no options: the
// early-out of WriteState), and the parse error is what the call returns.
int ep_ygo_emit_record(const ep_ygo_config* cfg, const ygo_raw_state* in, void* record_out) {
  int rc = check_config(cfg);
  if (rc) return rc;
  if (!in || !record_out) return fail(EP_ERR_INVALID, "null argument");
  rc = emit_record_impl(cfg, in, record_out);
  if (rc == EP_OK) return rc;
  const std::string why = epc::g_last_error;
  ygo_raw_state empty = *in;
  empty.n_options = 0;
  if (emit_record_impl(cfg, &empty, record_out) != EP_OK) {  // (to_play or ha_p out of range)
    memset(record_out, 0, (size_t)make_layout(*cfg).stride);
    ygo_header* h = (ygo_header*)record_out;
    h->env_id = in->env_id;
    h->elapsed_step = in->elapsed_step;
    h->done = in->done ? 1 : 0;
  }
  return fail(rc, "%s", why.c_str());
}

}  // extern "C"

static int emit_record_impl(const ep_ygo_config* cfg, const ygo_raw_state* in, void* record_out) {
  int rc;
  const Layout lay = make_layout(*cfg);
  uint8_t* rec = (uint8_t*)record_out;
  memset(rec, 0, (size_t)lay.stride);
  ygo_header* h = (ygo_header*)rec;
  if (in->to_play < 0 || in->to_play > 1) return fail(EP_ERR_INVALID, "to_play must be 0 or 1");
  h->to_play = (uint8_t)in->to_play;
  h->turn_player = (uint8_t)in->turn_player;
  h->msg = (uint8_t)in->msg;
  h->num_options = (uint8_t)std::min(in->n_options, 255);
  h->is_selfplay = in->is_selfplay ? 1 : 0;
  h->revealed_any = in->n_revealed != 0 ? 1 : 0;
  h->ha_p = (uint8_t)in->ha_p[in->to_play];
  h->done = in->done ? 1 : 0;
  h->env_id = in->env_id;
  h->elapsed_step = in->elapsed_step;
  h->lp[0] = in->lp[0];
  h->lp[1] = in->lp[1];
  h->current_phase = in->current_phase;
  h->win_reason = in->win_reason;
  h->reward = in->reward;
  if (in->ha_p[in->to_play] < 0 || in->ha_p[in->to_play] >= cfg->n_history_actions)
    return fail(EP_ERR_INVALID, "ha_p outside the ring");
  // revealed_: only entries that name an opponent hand card can matter (:1541-1545)
  for (int i = 0; i < in->n_revealed; ++i) {
    uint32_t key;
    if (in->revealed[i] && spec_key(in->revealed[i], &key) && (key & 0xff) == 1 && ((key >> 8) & 0xff) == EP_YGO_LOCATION_HAND &&
        (key >> 24) == 0) {
      const uint32_t seq = (key >> 16) & 0xff;
      if (seq < 64) h->revealed_hand_mask |= 1ull << seq;
    }
  }
  if (in->history[in->to_play])
    memcpy(rec + lay.off_hist, in->history[in->to_play], (size_t)cfg->n_history_actions * lay.nf);
  if (in->n_options == 0) return EP_OK;  // WriteState's early-out reads nothing else (:1837-1841)

  CardSink cards{(ygo_card*)(rec + lay.off_cards), 0, 2 * cfg->max_cards};  // parsed in place
  for (int pi = 0; pi < 2; ++pi) {
    const int player = (in->to_play + pi) % 2;
    for (int li = 0; li < 7; ++li) {
      bool hidden = (li == 0 || li == 1 || li == 6);
      if (li == 1 && in->n_revealed != 0) hidden = false;
      if (pi == 1 && hidden) {
        const int cnt = in->field_count[player][li];
        if (cnt < 0 || cnt > 255) return fail(EP_ERR_INVALID, "field count %d out of range", cnt);
        h->loc_count[pi][li] = (uint8_t)cnt;
      } else {
        const int before = cards.n;
        if (in->field_buf[player][li] && in->field_len[player][li] > 0)
          if ((rc = parse_location(in->field_buf[player][li], in->field_len[player][li], &cards))) return rc;
        const int cnt = cards.n - before;
        if (cnt > 255) return fail(EP_ERR_INVALID, "location holds %d cards", cnt);
        h->loc_count[pi][li] = (uint8_t)cnt;
      }
    }
  }
  if (cards.n > cards.cap)
    return fail(EP_ERR_INVALID, "%d listed cards exceed 2*max_cards = %d (the reference overruns obs:cards_ here)",
                cards.n, cards.cap);
  const int n_opt = std::min(in->n_options, cfg->max_options);
  const size_t opt_bytes = (size_t)ygk::option_bytes(cfg->max_multi_select);
  const size_t off_opts = (size_t)lay.off_cards + (size_t)cards.n * sizeof(ygo_card);  // right behind the listed cards
  for (int i = 0; i < n_opt; ++i) {
    ygo_option_head* oh = (ygo_option_head*)(rec + off_opts + (size_t)i * opt_bytes);
    uint32_t* keys = (uint32_t*)(oh + 1);  // spec references (ygo_kernels.hpp spec_ref)
    SpecList specs;
    if (!in->options[i]) return fail(EP_ERR_INVALID, "null option string");
    if ((rc = parse_option(in->msg, in->options[i], oh->feats, &specs))) return rc;
    if ((int)specs.size() > cfg->max_multi_select)
      return fail(EP_ERR_INVALID, "option '%s' references %zu specs, more than max_multi_select", in->options[i], specs.size());
    oh->n_specs = (uint8_t)specs.size();
    for (size_t j = 0; j < specs.size(); ++j) {
      // a malformed spec can never be a key of spec2index: give it a key no row carries, so the
      // device reports the same "throw" the reference's .at() would
      uint32_t key;
      keys[j] = spec_key(specs[j], &key) ? ygk::spec_ref(key, cfg->max_cards) : ygk::kSpecNone;
    }
  }
  return EP_OK;
}

extern "C" {

int ep_ygo_emit_records(const ep_ygo_config* cfg, const ygo_raw_state* in, int32_t n, void* records_out,
                        int32_t* status) {
  int rc = check_config(cfg);
  if (rc) return rc;
  const int64_t stride = make_layout(*cfg).stride;
  int first = EP_OK;
  for (int i = 0; i < n; ++i) {
    rc = ep_ygo_emit_record(cfg, in + i, (char*)records_out + (int64_t)i * stride);
    if (status) status[i] = rc;
    if (rc && !first) first = rc;
  }
  return first;
}

int ep_ygo_history_push(const ep_ygo_config* cfg, uint8_t* ring, int32_t* ha_p, int32_t msg, const char* option,
                        const uint16_t* card_ids, int32_t n_card_ids) {
  int rc = check_config(cfg);
  if (rc) return rc;
  if (!ring || !ha_p || !option) return fail(EP_ERR_INVALID, "null argument");
  const int nf = 9 + 2 * cfg->max_multi_select;
  uint8_t feats[9];
  SpecList specs;
  if ((rc = parse_option(msg, option, feats, &specs))) return rc;
  // with an empty spec2index the spec columns carry card ids (:1625): CARD/TRIBUTE/SUM write one
  // per card id (:1711-1714), every other spec-bearing option writes card_ids[0]
  int n_write = 0;
  if (msg == 15 || msg == 20 || msg == 23) n_write = n_card_ids;
  else if (!specs.empty()) n_write = 1;
  if (n_write > n_card_ids) return fail(EP_ERR_INVALID, "option '%s' needs a card id (the reference reads card_ids[0] out of bounds)", option);
  if (n_write > cfg->max_multi_select) return fail(EP_ERR_INVALID, "more card ids than max_multi_select");
  int p = *ha_p - 1;
  if (p < 0) p = cfg->n_history_actions - 1;
  *ha_p = p;
  uint8_t* row = ring + (size_t)p * nf;
  for (int j = 0; j < n_write; ++j) {
    row[2 * j] = (uint8_t)(card_ids[j] >> 8);
    row[2 * j + 1] = (uint8_t)(card_ids[j] & 0xff);
  }
  // only the columns the message's branch assigns are touched; the rest keep their old bytes
  const int off = 2 * cfg->max_multi_select;
  row[off + F_MSG] = feats[F_MSG];
  auto put = [&](int f) { row[off + f] = feats[f]; };
  switch (msg) {
    case EP_YGO_MSG_SELECT_IDLECMD: if (option[0] && (strcmp(option, "b") == 0 || strcmp(option, "e") == 0)) put(F_PHASE); else put(F_ACT); break;
    case EP_YGO_MSG_SELECT_CHAIN: if (option[0] == 'c') put(F_CANCEL); else put(F_ACT); break;
    case EP_YGO_MSG_SELECT_UNSELECT_CARD: if (option[0] == 'f') put(F_CANCEL); break;
    case EP_YGO_MSG_SELECT_POSITION: put(F_POSITION); break;
    case EP_YGO_MSG_SELECT_EFFECTYN: put(F_YESNO); break;
    case EP_YGO_MSG_SELECT_YESNO: put(F_YESNO); break;
    case EP_YGO_MSG_SELECT_BATTLECMD: if (strcmp(option, "m") == 0 || strcmp(option, "e") == 0) put(F_PHASE); else put(F_ACT); break;
    case EP_YGO_MSG_SELECT_OPTION: put(F_OPTION); break;
    case EP_YGO_MSG_SELECT_PLACE: case EP_YGO_MSG_SELECT_DISFIELD: put(F_PLACE); break;
    case EP_YGO_MSG_ANNOUNCE_ATTRIB: put(F_ATTRIB); break;
    default: break;
  }
  return EP_OK;
}

}  // extern "C"
