// ygo_kernels.hpp — gfx950 kernel for the YGOPro observation-encode tail.
//
// Restates, per env, YGOProEnv::WriteState and its callees
// (envpool2/ygopro/ygopro.h:1506-1893; tables :531-686; float_transform :704-710) plus
// Env::Allocate's common fields (envpool2/core/env.h:196-222), consuming the POD record of
// include/envpool_amd_ygo.h instead of live engine queries.
//
// Mapping to the hardware: one wave64 encodes one env, and waves never talk to each other (no
// workgroup barrier anywhere). A wave builds its env's rows in LDS images that start out zeroed
// (the reference's zero-filled StateBuffer, fused). A single env's 5850-byte obs:cards_ block
// is only 2-byte aligned in the output tensor, so every image sits in LDS at the SAME offset
// modulo 16 as its destination in HBM: the bulk of the block then leaves the CU as aligned
// 16-byte-per-lane stores, with at most 15 single bytes at either end. Card codes resolve
// through a device-resident open-addressing hash (L2-resident), the options' specs through a
// per-env direct-mapped table in LDS (the reference's per-step spec2index hash map), and
// h_card_ids is derived from the image exactly as :1863-1879 reads it back from obs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/envpool_amd_ygo.h"

namespace ygk {

#ifndef YGO_WAVES
#define YGO_WAVES 4
#endif
constexpr int kGroup = YGO_WAVES;       // envs (= waves) per workgroup; any value works
constexpr int kThreads = 64 * kGroup;   // one wave64 per env

struct __align__(16) HashEntry {  // 16 bytes: one 16-byte load per probe
  uint32_t code;
  uint32_t type;     // TYPE_* bits (27 used); bit 31 = entry valid
  uint16_t card_id;
  uint8_t level;
  uint8_t attr_id;   // attribute2id value, 0xff if the reference's .at() would throw
  uint8_t race_id;   // race2id value, 0xff likewise
  uint8_t pad[3];
};
constexpr uint32_t kEntryValid = 0x80000000u;
struct HashAux {  // same slot as the entry; only overlay materials need it (static atk/def)
  int32_t attack, defense;
};

__host__ __device__ inline uint32_t hash_code(uint32_t code) { return code * 2654435761u; }

struct Outs {
  int32_t* env_id; int32_t* penv; int32_t* elapsed; uint8_t* done; float* reward; float* discount;
  int32_t* step_type; uint8_t* trunc; uint8_t* cards; uint8_t* global; uint8_t* actions; uint8_t* h_actions;
  int32_t* num_options; int32_t* to_play; int32_t* is_selfplay; int32_t* win_reason; uint16_t* h_card_ids;
};

struct Args {
  const uint8_t* records;
  int64_t stride;
  int32_t n;
  const HashEntry* table;
  const HashAux* table_aux;
  uint32_t table_mask;  // capacity - 1
  int32_t max_cards, max_options, n_hist, mms, nf, max_episode_steps;
  int32_t off_hist, off_cards, off_opts;  // byte offsets inside a record
  Outs out;
  uint32_t* err;  // host-mapped error word of this batch
  unsigned long long* stamps;  // diagnostic builds only (-DYGO_STAMPS): cycles per phase
};

__host__ __device__ inline int al16(int x) { return (x + 15) & ~15; }
// per-wave LDS arena (byte offsets inside it, all 16-byte aligned). Each image region has 16
// spare bytes so that the image can start at its destination's address modulo 16.
struct Carve {
  int cards, actions, hact, global, hcids, spec, rowkey, wave_stride, total, spec_stride, rowkey_stride;
};
// direct-mapped spec table of one env: [2 players][7 locations][max_cards sequences] plus
// [2][8 monster zones][8 overlay letters], one byte each (1-based row index, 0 = absent)
__host__ __device__ inline int spec_table_bytes(int max_cards) { return al16(14 * max_cards + 128); }
__host__ __device__ inline Carve carve(int max_cards, int max_options, int n_hist, int mms) {
  const int nf = 9 + 2 * mms, R = 2 * max_cards;
  Carve c;
  int o = 0;
  c.cards = o;   o += al16(R * 39) + 16;
  c.actions = o; o += al16(max_options * nf) + 16;
  c.hact = o;    o += al16(n_hist * nf) + 16;
  c.hcids = o;   o += al16(max_options * mms * 2) + 16;
  c.global = o;  o += 16;
  c.spec_stride = spec_table_bytes(max_cards);
  c.spec = o;    o += c.spec_stride;
  c.rowkey_stride = al16(R * 4 + R);  // key of listed card k (u32), then its row (u8)
  c.rowkey = o;  o += c.rowkey_stride;
  c.wave_stride = o;
  c.total = o * kGroup;
  return c;
}
// staging of one env's ygo_card / ygo_option arrays inside its own (not yet used) card image
__host__ __device__ inline int card_chunks(int max_cards) { return (2 * max_cards * (int)sizeof(ygo_card) + 15) >> 4; }
__host__ __device__ inline int opt_chunks(int max_options) { return (max_options * (int)sizeof(ygo_option) + 15) >> 4; }
__host__ __device__ inline bool stage_fits(int max_cards, int max_options) {
  return (card_chunks(max_cards) + opt_chunks(max_options)) * 16 + 15 <= 2 * max_cards * 39;
}

__device__ __forceinline__ void ft(int x, uint8_t& hi, uint8_t& lo) {  // float_transform :704-710
  x = x % 65536;
  hi = (uint8_t)(x >> 8);
  lo = (uint8_t)(x & 0xff);
}

// location2id (:531-542): DECK 1 .. EXTRA 7; 0 = the reference's .at() throws
__device__ __forceinline__ uint8_t loc_id(uint32_t loc) {
  return (loc != 0 && (loc & (loc - 1)) == 0 && loc <= 0x40) ? (uint8_t)(__ffs(loc)) : 0;
}
// position2id (:546-559): keys {0,1,2,3,4,5,8,0xa,0xc} -> 0..8; 0xff = throws
__device__ __forceinline__ uint8_t pos_id(uint32_t p) {
  switch (p) {
    case 0x0: return 0; case 0x1: return 1; case 0x2: return 2; case 0x3: return 3; case 0x4: return 4;
    case 0x5: return 5; case 0x8: return 6; case 0xa: return 7; case 0xc: return 8; default: return 0xff;
  }
}
__device__ __forceinline__ uint32_t type_bit(int j) {  // ascending TYPE_* keys (:607-633)
  // 0x1,0x2,0x4, 0x10..0x4000 (bits 4..14), 0x10000..0x4000000 (bits 16..26)
  return j < 3 ? (1u << j) : (j < 14 ? (1u << (j + 1)) : (1u << (j + 2)));
}

// Slot of a spec key in the direct-mapped table, or -1 when the key is outside it (then the
// exact linear scan is used): plain specs by (opponent, location, sequence < max_cards), overlay
// specs by (opponent, monster zone < 8, letter < 8).
__device__ __forceinline__ int spec_slot(uint32_t key, int max_cards) {
  const uint32_t opp = key & 0xff, loc7 = (key >> 8) & 0xff, seq = (key >> 16) & 0xff, suf = key >> 24;
  if (opp > 1 || loc7 == 0 || (loc7 & (loc7 - 1)) != 0 || loc7 > 0x40) return -1;
  if (suf == 0) {
    if (seq >= (uint32_t)max_cards) return -1;
    return (int)((opp * 7 + (uint32_t)__ffs(loc7) - 1) * (uint32_t)max_cards + seq);
  }
  if (loc7 != 0x04 || seq >= 8 || suf > 8) return -1;
  return 14 * max_cards + (int)((opp * 8 + seq) * 8 + (suf - 1));
}

#ifndef YGO_SKIP
#define YGO_SKIP 0  // diagnostic builds: leave phases out to price them (results are then wrong)
#endif
#ifdef YGO_STAMPS
#define YGO_STAMP(k)                                                                          \
  do {                                                                                        \
    unsigned long long _t = __builtin_amdgcn_s_memtime();                                     \
    if (lane == 0 && (blockIdx.x & 63) == 0) atomicAdd(&a.stamps[k], _t - _t_prev);                                     \
    _t_prev = __builtin_amdgcn_s_memtime();                                                   \
  } while (0)
#else
#define YGO_STAMP(k) do {} while (0)
#endif

// LDS traffic of one wave is executed in issue order; the fences keep the compiler from moving
// accesses of different lanes' data across the hand-over points.
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// zero the LDS bytes [lo, hi) with one wave: 16-byte stores for the aligned middle
__device__ __forceinline__ void wave_zero(unsigned char* lds, int lo, int hi, int lane) {
  const int mlo = min((lo + 15) & ~15, hi), mhi = max(hi & ~15, mlo);
  if (lane < mlo - lo) lds[lo + lane] = 0;
  if (lane < hi - mhi) lds[mhi + lane] = 0;
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (int q = (mlo >> 4) + lane; q < (mhi >> 4); q += 64) reinterpret_cast<uint4*>(lds)[q] = z;
}

// copy the LDS bytes [lo, lo+bytes) to g with one wave. When g and lo agree modulo 16 the bulk
// goes as aligned 16-byte stores; otherwise (odd-aligned caller buffers) byte by byte.
__device__ __forceinline__ void wave_flush(uint8_t* g, const unsigned char* lds, int lo, int bytes, int lane) {
  if ((((uintptr_t)g ^ (uintptr_t)lo) & 15) == 0) {
    const int head = min((16 - (lo & 15)) & 15, bytes);
    if (lane < head) g[lane] = lds[lo + lane];
    const int nvec = (bytes - head) >> 4;
    const uint4* src = reinterpret_cast<const uint4*>(lds + lo + head);
    uint4* dst = reinterpret_cast<uint4*>(g + head);
    for (int q = lane; q < nvec; q += 64) dst[q] = src[q];
    const int done = head + (nvec << 4);
    if (lane < bytes - done) g[done + lane] = lds[lo + done + lane];
  } else {
    for (int b = lane; b < bytes; b += 64) g[b] = lds[lo + b];
  }
}

__device__ __forceinline__ uint32_t lane_word(uint32_t v, int src_lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane);
}

__global__ __launch_bounds__(kThreads, 4) void ygo_encode(Args a) {
  extern __shared__ __align__(16) unsigned char lds[];
  const Carve cv = carve(a.max_cards, a.max_options, a.n_hist, a.mms);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#ifdef YGO_STAMPS
  unsigned long long _t_prev = __builtin_amdgcn_s_memtime();
#endif
  const int R = 2 * a.max_cards, nf = a.nf;
  const int64_t e0 = (int64_t)blockIdx.x * kGroup;
  const int gvalid = (int)min((int64_t)kGroup, (int64_t)a.n - e0);
  const int64_t e = e0 + wave;
  bool bad = false;

  if (wave < gvalid) {  // wave-uniform: one wave64 encodes one env, with no cross-wave traffic
    const uint8_t* rec = a.records + e * a.stride;
    const int arena = wave * cv.wave_stride;
    // global destinations of this env's images, and LDS offsets congruent to them modulo 16
    uint8_t* g_cards = a.out.cards + e * (int64_t)(R * 39);
    uint8_t* g_actions = a.out.actions + e * (int64_t)(a.max_options * nf);
    uint8_t* g_hact = a.out.h_actions + e * (int64_t)(a.n_hist * nf);
    uint8_t* g_hcids = reinterpret_cast<uint8_t*>(a.out.h_card_ids + e * (int64_t)(a.max_options * a.mms));
    const int img_lo = arena + cv.cards + (int)((uintptr_t)g_cards & 15), img_hi = img_lo + R * 39;
    const int act_lo = arena + cv.actions + (int)((uintptr_t)g_actions & 15);
    const int hact_lo = arena + cv.hact + (int)((uintptr_t)g_hact & 15);
    const int hcids_lo = arena + cv.hcids + (int)((uintptr_t)g_hcids & 14);  // keep uint16 alignment
    uint8_t* my_cards = lds + img_lo;
    uint8_t* my_actions = lds + act_lo;
    uint8_t* my_hact = lds + hact_lo;
    uint8_t* my_global = lds + arena + cv.global;
    uint16_t* my_hcids = reinterpret_cast<uint16_t*>(lds + hcids_lo);
    uint32_t* my_rowkey = reinterpret_cast<uint32_t*>(lds + arena + cv.rowkey);
    uint8_t* my_krow = reinterpret_cast<uint8_t*>(my_rowkey + R);
    uint8_t* my_spec = lds + arena + cv.spec;

    // ---- every global read of this env is issued up front as 16-byte-per-lane coalesced loads
    // (the record layout is fixed, so no address depends on the header): header, the ygo_card
    // array, the ygo_option array, the history ring ----
    const bool staged = stage_fits(a.max_cards, a.max_options);
    const int n_cchunks = card_chunks(a.max_cards), n_ochunks = opt_chunks(a.max_options);
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    uint4 hq = z4, cq[5] = {z4, z4, z4, z4, z4}, oq[3] = {z4, z4, z4};
    if (lane < 4) hq = reinterpret_cast<const uint4*>(rec)[lane];
    if (staged) {
#pragma unroll
      for (int c = 0; c < 5; ++c)
        if (lane + 64 * c < n_cchunks) cq[c] = reinterpret_cast<const uint4*>(rec + a.off_cards)[lane + 64 * c];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (lane + 64 * c < n_ochunks) oq[c] = reinterpret_cast<const uint4*>(rec + a.off_opts)[lane + 64 * c];
    }
    const int hist_bytes = a.n_hist * nf;
    uint32_t hw[2] = {0, 0};  // ring bytes [4*lane, 4*lane+4) and [256+4*lane, ...)
    {
      const uint32_t* ring_w = reinterpret_cast<const uint32_t*>(rec + a.off_hist);
      if (4 * lane < hist_bytes) hw[0] = ring_w[lane];
      if (256 + 4 * lane < hist_bytes) hw[1] = ring_w[64 + lane];
    }

    // ---- while they fly: the env's small images start as zeros (the reference's fresh
    // zero-filled StateBuffer, array.h:77-81) ----
    wave_zero(lds, act_lo, act_lo + a.max_options * nf, lane);
    wave_zero(lds, hact_lo, hact_lo + hist_bytes, lane);
    wave_zero(lds, hcids_lo, hcids_lo + a.max_options * a.mms * 2, lane);
    wave_zero(lds, arena + cv.spec, arena + cv.spec + cv.spec_stride, lane);
    if (lane < 2) reinterpret_cast<uint32_t*>(my_global)[lane] = 0u;

    // ---- header: lanes 0..3 hold it; every field becomes a wave-uniform scalar ----
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[4 * i + 0] = lane_word(hq.x, i);
      w[4 * i + 1] = lane_word(hq.y, i);
      w[4 * i + 2] = lane_word(hq.z, i);
      w[4 * i + 3] = lane_word(hq.w, i);
    }
    const int to_play = w[0] & 0xff, turn_player = (w[0] >> 8) & 0xff;
    int n_opt = w[0] >> 24;
    const int is_selfplay = w[1] & 0xff, ha_p = (w[1] >> 16) & 0xff;
    const bool revealed_any = ((w[1] >> 8) & 0xff) != 0, done = (w[1] >> 24) != 0;
    const int env_id = (int)w[2], elapsed = (int)w[3];
    const int lp_me = (int)(to_play == 0 ? w[4] : w[5]), lp_op = (int)(to_play == 0 ? w[5] : w[4]);
    const uint32_t phase = w[6];
    const uint64_t hand_mask = (uint64_t)w[10] | ((uint64_t)w[11] << 32);
    const bool no_options = n_opt == 0;                // early-out of WriteState (:1837-1841)
    if (n_opt > a.max_options) n_opt = a.max_options;  // options_.resize (:1849-1851)

    // ---- segment table: 2 players x 7 locations (:1510-1519), all scalar ----
    int seg_row[14], seg_card[14], seg_cnt[14], rows_pi[2], n_cards = 0;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      int row = pi * a.max_cards;
#pragma unroll
      for (int li = 0; li < 7; ++li) {
        const int s = pi * 7 + li;
        const int cnt = (int)((w[12 + (s >> 2)] >> (8 * (s & 3))) & 0xff);
        bool hidden = (li == 0 || li == 1 || li == 6);
        if (li == 1 && revealed_any) hidden = false;  // :1522-1525
        const bool count_only = pi == 1 && hidden;
        seg_row[s] = row;
        seg_card[s] = count_only ? 0x7fffffff : n_cards;  // count-only: never matches a card index
        seg_cnt[s] = cnt;
        row += cnt;
        if (!count_only) n_cards += cnt;
      }
      // the reference has no bound; more rows than max_cards is a buffer overrun there
      if (row > (pi + 1) * a.max_cards) bad = true;
      rows_pi[pi] = row - pi * a.max_cards;
    }
    const int n_listed = min(n_cards, R);
    const bool overflow = rows_pi[0] > a.max_cards || rows_pi[1] > a.max_cards;
    YGO_STAMP(0);

    if (no_options) {
      wave_zero(lds, img_lo, img_hi, lane);
      if (lane == 0) my_global[7] = 1;  // :1839
    } else {
      // ---- hand the card / option arrays to their owning lanes: lane l owns cards l, l+64,
      // l+128, l+192 and option l. The env's own card image region is the staging area. ----
      ygo_card cin[4];
      ygo_option mine;
      mine.n_specs = 0;
      if (staged) {
        uint8_t* stage = lds + ((img_lo + 15) & ~15);
#pragma unroll
        for (int c = 0; c < 5; ++c)
          if (lane + 64 * c < n_cchunks) reinterpret_cast<uint4*>(stage)[lane + 64 * c] = cq[c];
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if (lane + 64 * c < n_ochunks) reinterpret_cast<uint4*>(stage + n_cchunks * 16)[lane + 64 * c] = oq[c];
        wave_fence();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          cin[it].code = 0;
          if (lane + 64 * it < n_listed) cin[it] = reinterpret_cast<const ygo_card*>(stage)[lane + 64 * it];
        }
        if (lane < n_opt) mine = reinterpret_cast<const ygo_option*>(stage + n_cchunks * 16)[lane];
      } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          cin[it].code = 0;
          if (lane + 64 * it < n_listed) cin[it] = reinterpret_cast<const ygo_card*>(rec + a.off_cards)[lane + 64 * it];
        }
        if (lane < n_opt) mine = reinterpret_cast<const ygo_option*>(rec + a.off_opts)[lane];
      }
      // ---- static data: cards_.at(code) / card_ids_.at(code). The first probes of a lane's
      // cards are in flight together, and the image is zeroed underneath them. ----
      HashEntry te[4];
      uint32_t slot0[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        slot0[it] = (hash_code(cin[it].code) >> 7) & a.table_mask;
        te[it].type = 0;
        te[it].code = 0;
        if (lane + 64 * it < n_listed && !(YGO_SKIP & (32 | 64))) te[it] = a.table[slot0[it]];
        if (YGO_SKIP & 64) {  // priced without the table read: a made-up known card
          te[it].code = cin[it].code;
          te[it].type = kEntryValid | 1u | (1u << (4 + (cin[it].code & 7)));
          te[it].card_id = (uint16_t)cin[it].code;
          te[it].level = 4; te[it].attr_id = 1; te[it].race_id = 1;
        }
      }
      wave_fence();  // staging area consumed
      wave_zero(lds, img_lo, img_hi, lane);
      wave_fence();
    YGO_STAMP(1);
#pragma unroll
      for (int it = 0; it < 4; ++it) {  // collisions (load factor <= 0.5): keep probing, rarely
        if ((te[it].type & kEntryValid) && te[it].code != cin[it].code) {
          uint32_t slot = slot0[it];
          HashEntry found;
          found.type = 0;
          found.code = 0;
          for (uint32_t probe = 0; probe < a.table_mask; ++probe) {
            slot = (slot + 1) & a.table_mask;
            const HashEntry cand = a.table[slot];
            if (!(cand.type & kEntryValid)) break;
            if (cand.code == cin[it].code) { found = cand; break; }
          }
          te[it] = found;
          slot0[it] = slot;
        }
      }

      YGO_STAMP(2);
      // ---- obs:cards_ : one lane per LISTED card (:1508-1603). Only non-zero bytes are
      // written: the image starts zeroed. Every listed row also enters the env's direct-mapped
      // spec table (the reference's per-step spec2index hash map, :1549). ----
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int k = lane + 64 * it;
        if (k < n_listed && !(YGO_SKIP & 1)) {
          // segment of card k: the last listed segment starting at or before k
          int s = 0, s_row = seg_row[0], s_card = seg_card[0], s_cnt = seg_cnt[0];
#pragma unroll
          for (int ss = 1; ss < 14; ++ss)
            if (k >= seg_card[ss]) { s = ss; s_row = seg_row[ss]; s_card = seg_card[ss]; s_cnt = seg_cnt[ss]; }
          const int pi = s >= 7 ? 1 : 0, li = s - 7 * pi;
          const int r = s_row + (k - s_card);
          uint32_t key = 0xffffffffu;
          if (!overflow || s_row + s_cnt <= (pi + 1) * a.max_cards) {
            const ygo_card c = cin[it];
            const HashEntry t = te[it];
            const bool known = (t.type & kEntryValid) && t.code == c.code;
            uint8_t* row = my_cards + r * 39;
            const uint32_t loc7 = c.location & 0x7f;
            const bool overlay = (c.location & 0x80) != 0;
            bool hide = false;
            if (pi == 1) {
              hide = (c.position & 0xa) != 0;  // POS_FACEDOWN (:1540)
              if (li == 1 && c.sequence < 64 && ((hand_mask >> c.sequence) & 1ull)) hide = false;
            }
            if (overlay) hide = false;  // :1562-1564
            if (!known) { bad = true; hide = true; }
            const uint8_t lid = loc_id(loc7);
            if (lid == 0) bad = true;
            row[2] = lid;
            row[3] = (loc7 == 0x04 || loc7 == 0x08 || loc7 == 0x10) ? (uint8_t)(c.sequence + 1) : (uint8_t)0;  // :1573-1578
            row[4] = (c.controller != to_play) ? 1 : 0;
            const uint8_t pid = pos_id(c.position);
            if (!overlay && pid == 0xff) bad = true;
            row[5] = overlay ? (uint8_t)5 : (pid == 0xff ? (uint8_t)0 : pid);  // overlay: position2id.at(POS_FACEUP)
            row[6] = overlay ? 1 : 0;
            if (!hide) {
              row[0] = (uint8_t)(t.card_id >> 8);
              row[1] = (uint8_t)(t.card_id & 0xff);
              if (t.attr_id == 0xff || t.race_id == 0xff) bad = true;
              row[7] = t.attr_id == 0xff ? 0 : t.attr_id;
              row[8] = t.race_id == 0xff ? 0 : t.race_id;
              int level = t.level, atk = c.attack, def = c.defense;
              if (c.is_material) {  // an overlay material shows its static values (c_get_card copy, :2109-2114)
                const HashAux x = a.table_aux[slot0[it]];
                atk = x.attack;
                def = x.defense;
              } else {  // dynamic overrides of get_cards_in_location (:2090-2099, :2126-2133)
                if (c.level > 0) level = c.level;
                if (c.rank > 0) level = c.rank;
                if (c.link > 0) level = c.link;
              }
              row[9] = (uint8_t)level;
              ft(atk, row[10], row[11]);
              ft(def, row[12], row[13]);
              // type_to_ids (:635-642): one byte per TYPE_* key in ascending order; set bits only
              uint32_t tb = t.type & 0x07ff7ff7u;
              while (tb) {
                const int b = __ffs(tb) - 1;
                tb &= tb - 1;
                row[14 + (b < 3 ? b : (b < 15 ? b - 1 : b - 2))] = 1;
              }
            }
            key = (uint32_t)pi | (loc7 << 8) | ((uint32_t)c.sequence << 16) | ((overlay ? (uint32_t)c.position + 1u : 0u) << 24);
            const int ti = spec_slot(key, a.max_cards);
            if (ti >= 0) my_spec[ti] = (uint8_t)(r + 1);
            my_krow[k] = (uint8_t)(r + 1);  // row of listed card k
          }
          my_rowkey[k] = key;
        }
      }
      // hidden opponent locations: placeholder rows {col2 = location id, col4 = 1} (:1526-1532)
#pragma unroll
      for (int li = 0; li < 7; li += (li == 1 ? 5 : 1)) {  // DECK, HAND, EXTRA
        const int s = 7 + li;
        if (seg_card[s] == 0x7fffffff && seg_row[s] + seg_cnt[s] <= 2 * a.max_cards)
          for (int q = lane; q < seg_cnt[s]; q += 64) {
            uint8_t* row = my_cards + (seg_row[s] + q) * 39;
            row[2] = (uint8_t)(li + 1);
            row[4] = 1;
          }
      }
      wave_fence();
      YGO_STAMP(3);
      // two rows with one spec (never in a consistent duel state): the table holds an arbitrary
      // one of them, the reference the LAST one; detect it and take the exact scan instead
      bool dup = false;
      for (int k = lane; k < n_listed && !(YGO_SKIP & 2); k += 64) {
        const uint32_t key = my_rowkey[k];
        const int ti = key == 0xffffffffu ? -1 : spec_slot(key, a.max_cards);
        if (ti >= 0 && my_spec[ti] != my_krow[k]) dup = true;
      }
      dup = __any(dup);
      // ---- obs:global_ (:1605-1620) ----
      if (lane == 0) {
        ft(lp_me, my_global[0], my_global[1]);
        ft(lp_op, my_global[2], my_global[3]);
        if (phase == 0 || (phase & (phase - 1)) != 0 || phase > 0x200) bad = true;
        else my_global[4] = (uint8_t)(__ffs(phase) - 1);  // phase2id: 0..9
        my_global[5] = to_play == 0 ? 1 : 0;
        my_global[6] = to_play == turn_player ? 1 : 0;
      }
      // ---- obs:h_actions_ : ring rotated so that the newest entry is row 0 (:1881-1892) ----
      if (!(YGO_SKIP & 4)) {
        const int shift = ha_p * nf;
#pragma unroll
        for (int ww = 0; ww < 2; ++ww) {
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) {
            const int src = 256 * ww + 4 * lane + bb;  // ring byte held by this lane
            const uint8_t v = (uint8_t)(hw[ww] >> (8 * bb));
            if (src < hist_bytes && v) {
              int dst = src - shift;
              if (dst < 0) dst += hist_bytes;
              my_hact[dst] = v;
            }
          }
        }
        for (int src = 512 + lane; src < hist_bytes; src += 64) {  // rings longer than 512 bytes
          int dst = src - shift;
          if (dst < 0) dst += hist_bytes;
          my_hact[dst] = (rec + a.off_hist)[src];
        }
      }
      YGO_STAMP(4);
      // ---- obs:actions_ (:1674-1765) + h_card_ids (:1863-1879): lane i owns option i ----
      if (lane < n_opt && !(YGO_SKIP & 8)) {
        uint8_t* arow = my_actions + lane * nf;
        const int ns = min((int)mine.n_specs, a.mms);
        bool open = true;
#pragma unroll
        for (int j = 0; j < EP_YGO_MAX_SPECS; ++j) {
          if (j < ns) {
            const uint32_t key = mine.spec_key[j];
            const int ti = spec_slot(key, a.max_cards);
            int idx = 0;
            if (ti >= 0 && !dup) {
              idx = my_spec[ti];
            } else {
              for (int k = 0; k < n_listed; ++k)
                if (my_rowkey[k] == key) idx = max(idx, (int)my_krow[k]);  // last row wins (:1549)
            }
            if (idx == 0) bad = true;  // spec2index.at(spec) throws
            arow[2 * j] = (uint8_t)(idx >> 8);
            arow[2 * j + 1] = (uint8_t)(idx & 0xff);
            // read back from the images, literally: low byte only (Q7), stop at the first zero
            const int spec_index = idx & 0xff;
            if (spec_index == 0) open = false;
            if (open) {
              const uint16_t c1 = my_cards[(spec_index - 1) * 39 + 0], c2 = my_cards[(spec_index - 1) * 39 + 1];
              my_hcids[lane * a.mms + j] = (uint16_t)((c1 << 8) + c2);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k)
          if (mine.feats[k]) arow[2 * a.mms + k] = mine.feats[k];
      }
    }
    YGO_STAMP(5);
    // ---- scalars + Env::Allocate common fields (env.h:196-222) ----
    if (lane == 0) {
      a.out.env_id[e] = env_id;
      a.out.penv[e] = env_id;
      a.out.elapsed[e] = elapsed;
      a.out.done[e] = done;
      a.out.reward[e] = __uint_as_float(w[8]);
      a.out.discount[e] = done ? 0.f : 1.f;
      a.out.step_type[e] = elapsed == 0 ? 0 : (done ? 2 : 1);
      a.out.trunc[e] = done && (elapsed >= a.max_episode_steps);
      a.out.num_options[e] = no_options ? 1 : n_opt;
      a.out.to_play[e] = to_play;
      a.out.is_selfplay[e] = is_selfplay;
      a.out.win_reason[e] = (int)w[7];
    }
    if (__any(bad) && lane == 0) *a.err = 1u;
    wave_fence();

    YGO_STAMP(6);
    // ---- the env's images leave the CU: aligned 16-byte-per-lane stores for the bulk ----
    if (!(YGO_SKIP & 16)) wave_flush(g_cards, lds, img_lo, R * 39, lane);
    wave_flush(g_actions, lds, act_lo, a.max_options * nf, lane);
    wave_flush(g_hact, lds, hact_lo, hist_bytes, lane);
    wave_flush(g_hcids, lds, hcids_lo, a.max_options * a.mms * 2, lane);
    if (lane < 8) a.out.global[e * 8 + lane] = my_global[lane];
    YGO_STAMP(7);
  }
}

}  // namespace ygk
