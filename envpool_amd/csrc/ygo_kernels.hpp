// ygo_kernels.hpp — gfx950 kernel for the YGOPro observation-encode tail.
//
// Restates, per env, YGOProEnv::WriteState and its callees
// (envpool2/ygopro/ygopro.h:1506-1893; tables :531-686; float_transform :704-710) plus
// Env::Allocate's common fields (envpool2/core/env.h:196-222), consuming the POD record of
// include/envpool_amd_ygo.h instead of live engine queries.
//
// Mapping to the hardware: one 512-thread workgroup encodes EIGHT consecutive envs, one wave64
// per env. Eight, because 8 * (2*max_cards*39) bytes is always a multiple of 16: the group's
// obs:cards_ block is then a 16-byte-aligned contiguous span of the output tensor and leaves
// the CU as full 16-byte-per-lane stores, although a single env's 5850-byte block is only
// 2-byte aligned. Each wave builds its env's rows in an LDS image that starts out zeroed (the
// reference's zero-filled StateBuffer, fused), resolves card codes through a device-resident
// open-addressing hash (L2-resident), resolves the options' specs against the per-duel row
// keys held in LDS (the reference's per-step spec2index hash map), and derives h_card_ids from
// the image exactly as :1863-1879 reads them back from obs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/envpool_amd_ygo.h"

namespace ygk {

constexpr int kGroup = 8;     // envs per workgroup
constexpr int kThreads = 512; // one wave64 per env

struct __align__(16) HashEntry {  // 32 bytes
  uint32_t code;
  uint32_t type;
  int32_t attack;
  int32_t defense;
  uint16_t card_id;
  uint8_t level;
  uint8_t attr_id;  // attribute2id value, 0xff if the reference's .at() would throw
  uint8_t race_id;  // race2id value, 0xff likewise
  uint8_t valid;
  uint16_t pad0;
  uint32_t pad1[2];
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
  uint32_t table_mask;  // capacity - 1
  int32_t max_cards, max_options, n_hist, mms, nf, max_episode_steps;
  int32_t off_hist, off_cards, off_opts;  // byte offsets inside a record
  Outs out;
  uint32_t* err;  // host-mapped error word of this batch
};

// per-workgroup LDS carve (bytes), all 16-byte aligned
struct Carve {
  int cards, actions, hact, global, hcids, rowkey, hdr, seg, total, zero_bytes;
};
__host__ __device__ inline int al16(int x) { return (x + 15) & ~15; }
__host__ __device__ inline Carve carve(int max_cards, int max_options, int n_hist, int mms) {
  const int nf = 9 + 2 * mms, R = 2 * max_cards;
  Carve c;
  int o = 0;
  c.cards = o;   o += al16(kGroup * R * 39);
  c.actions = o; o += al16(kGroup * max_options * nf);
  c.global = o;  o += al16(kGroup * 8);
  c.hcids = o;   o += al16(kGroup * max_options * mms * 2);
  c.hact = o;    o += al16(kGroup * n_hist * nf);
  c.zero_bytes = o;  // everything above starts as zeros
  c.rowkey = o;  o += al16(kGroup * R * 4);
  c.hdr = o;     o += kGroup * 64;
  c.seg = o;     o += kGroup * 64;  // per env: rowstart[14], cardstart[14], count[14], listed mask, nrows[2]
  c.total = o;
  return c;
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

__global__ __launch_bounds__(kThreads) void ygo_encode(Args a) {
  extern __shared__ __align__(16) unsigned char lds[];
  const Carve cv = carve(a.max_cards, a.max_options, a.n_hist, a.mms);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int R = 2 * a.max_cards, nf = a.nf;
  const int64_t e0 = (int64_t)blockIdx.x * kGroup;
  const int gvalid = (int)min((int64_t)kGroup, (int64_t)a.n - e0);
  const int64_t e = e0 + wave;
  const bool live = wave < gvalid;

  // ---- zero the images (fresh zero-filled StateBuffer, array.h:77-81) ----
  for (int q = tid; q < cv.zero_bytes / 16; q += kThreads) reinterpret_cast<uint4*>(lds)[q] = make_uint4(0, 0, 0, 0);

  const uint8_t* rec = a.records + e * a.stride;
  uint8_t* my_cards = lds + cv.cards + wave * R * 39;
  uint8_t* my_actions = lds + cv.actions + wave * a.max_options * nf;
  uint8_t* my_hact = lds + cv.hact + wave * a.n_hist * nf;
  uint8_t* my_global = lds + cv.global + wave * 8;
  uint16_t* my_hcids = reinterpret_cast<uint16_t*>(lds + cv.hcids) + wave * a.max_options * a.mms;
  uint32_t* my_rowkey = reinterpret_cast<uint32_t*>(lds + cv.rowkey) + wave * R;
  uint32_t* my_hdr = reinterpret_cast<uint32_t*>(lds + cv.hdr) + wave * 16;
  int32_t* my_seg = reinterpret_cast<int32_t*>(lds + cv.seg) + wave * 16;
  uint8_t* my_segb = reinterpret_cast<uint8_t*>(my_seg);  // rowstart[14] | cardstart[14] as u8 pairs below

  if (live && lane < 16) my_hdr[lane] = reinterpret_cast<const uint32_t*>(rec)[lane];
  __syncthreads();  // zeroing + headers visible

  bool bad = false;
  int n_opt = 0;
  const ygo_header* h = reinterpret_cast<const ygo_header*>(my_hdr);
  if (live) {
    const int to_play = h->to_play;
    n_opt = h->num_options;
    const bool no_options = n_opt == 0;  // early-out of WriteState (:1837-1841)
    if (n_opt > a.max_options) n_opt = a.max_options;  // options_.resize (:1849-1851)

    // ---- segment table: 2 players x 7 locations (:1510-1519) ----
    // my_segb[s] = first row, my_segb[16+s] = first listed-card index (u16 in my_seg[8+..]) ...
    uint16_t* seg_row = reinterpret_cast<uint16_t*>(my_seg);        // [14] first row of segment
    uint16_t* seg_card = reinterpret_cast<uint16_t*>(my_seg) + 14;  // [14] first ygo_card of segment, 0xffff = count-only
    if (lane == 0) {
      int card = 0;
      for (int pi = 0; pi < 2; ++pi) {
        int row = pi * a.max_cards;
        for (int li = 0; li < 7; ++li) {
          const int s = pi * 7 + li;
          bool hidden = (li == 0 || li == 1 || li == 6);
          if (li == 1 && h->revealed_any) hidden = false;  // :1522-1525
          const bool count_only = pi == 1 && hidden;
          seg_row[s] = (uint16_t)row;
          seg_card[s] = count_only ? (uint16_t)0xffff : (uint16_t)card;
          row += h->loc_count[pi][li];
          if (!count_only) card += h->loc_count[pi][li];
        }
        // the reference has no bound; more rows than max_cards is a buffer overrun there
        if (row > (pi + 1) * a.max_cards) bad = true;
      }
    }
    (void)my_segb;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    if (!no_options) {
      // ---- obs:cards_ : one lane per row (:1508-1603) ----
      const ygo_card* cards_in = reinterpret_cast<const ygo_card*>(rec + a.off_cards);
      for (int r = lane; r < R; r += 64) {
        const int pi = r >= a.max_cards ? 1 : 0;
        // find the segment of row r
        int s = -1, idx = 0;
#pragma unroll
        for (int li = 0; li < 7; ++li) {
          const int ss = pi * 7 + li;
          const int r0 = seg_row[ss], cnt = h->loc_count[pi][li];
          if (r >= r0 && r < r0 + cnt && r0 + cnt <= (pi + 1) * a.max_cards) { s = ss; idx = r - r0; }
        }
        uint32_t key = 0xffffffffu;
        if (s >= 0) {
          const int li = s - pi * 7;
          uint8_t* row = my_cards + r * 39;
          if (seg_card[s] == 0xffff) {
            // hidden opponent location: placeholder rows (:1526-1532)
            row[2] = (uint8_t)(li + 1);
            row[4] = 1;
          } else {
            const ygo_card c = cards_in[seg_card[s] + idx];
            const uint32_t loc7 = c.location & 0x7f;
            const bool overlay = (c.location & 0x80) != 0;
            bool hide = false;
            if (pi == 1) {
              hide = (c.position & 0xa) != 0;  // POS_FACEDOWN (:1540)
              if (li == 1 && c.sequence < 64 && ((h->revealed_hand_mask >> c.sequence) & 1ull)) hide = false;
            }
            if (overlay) hide = false;  // :1562-1564
            // static data: cards_.at(code) / card_ids_.at(code)
            HashEntry te;
            te.valid = 0;
            {
              uint32_t slot = (hash_code(c.code) >> 7) & a.table_mask;
              for (uint32_t probe = 0; probe <= a.table_mask; ++probe) {
                const HashEntry cand = a.table[slot];
                if (!cand.valid) break;
                if (cand.code == c.code) { te = cand; break; }
                slot = (slot + 1) & a.table_mask;
              }
            }
            if (!te.valid) { bad = true; te.type = 0; te.attack = 0; te.defense = 0; te.card_id = 0; te.level = 0; te.attr_id = 0; te.race_id = 0; }
            const uint8_t lid = loc_id(loc7);
            if (lid == 0) bad = true;
            row[2] = lid;
            row[3] = (loc7 == 0x04 || loc7 == 0x08 || loc7 == 0x10) ? (uint8_t)(c.sequence + 1) : 0;  // :1573-1578
            row[4] = (c.controller != to_play) ? 1 : 0;
            if (overlay) {
              row[5] = 5;  // position2id.at(POS_FACEUP)
              row[6] = 1;
            } else {
              const uint8_t pid = pos_id(c.position);
              if (pid == 0xff) bad = true;
              row[5] = pid == 0xff ? 0 : pid;
            }
            if (!hide) {
              row[0] = (uint8_t)(te.card_id >> 8);
              row[1] = (uint8_t)(te.card_id & 0xff);
              if (te.attr_id == 0xff || te.race_id == 0xff) bad = true;
              row[7] = te.attr_id == 0xff ? 0 : te.attr_id;
              row[8] = te.race_id == 0xff ? 0 : te.race_id;
              int level = te.level, atk = te.attack, def = te.defense;
              if (!c.is_material) {  // dynamic overrides of get_cards_in_location (:2090-2099, :2126-2133)
                if (c.level > 0) level = c.level;
                if (c.rank > 0) level = c.rank;
                if (c.link > 0) level = c.link;
                atk = c.attack;
                def = c.defense;
              }
              row[9] = (uint8_t)level;
              ft(atk, row[10], row[11]);
              ft(def, row[12], row[13]);
#pragma unroll
              for (int j = 0; j < 25; ++j) row[14 + j] = (te.type & type_bit(j)) ? 1 : 0;
            }
            key = (uint32_t)pi | (loc7 << 8) | ((uint32_t)c.sequence << 16) | ((overlay ? (uint32_t)c.position + 1u : 0u) << 24);
          }
        }
        my_rowkey[r] = key;
      }
      // ---- obs:global_ (:1605-1620) ----
      if (lane == 0) {
        const int me = to_play, op = 1 - to_play;
        ft(h->lp[me], my_global[0], my_global[1]);
        ft(h->lp[op], my_global[2], my_global[3]);
        const uint32_t ph = (uint32_t)h->current_phase;
        if (ph == 0 || (ph & (ph - 1)) != 0 || ph > 0x200) bad = true;
        else my_global[4] = (uint8_t)(__ffs(ph) - 1);  // phase2id: 0..9
        my_global[5] = me == 0 ? 1 : 0;
        my_global[6] = me == h->turn_player ? 1 : 0;
      }
      // ---- obs:h_actions_ : ring rotated so that the newest entry is row 0 (:1881-1892) ----
      {
        const uint8_t* ring = rec + a.off_hist;
        const int total = a.n_hist * nf, shift = h->ha_p * nf;
        for (int b = lane; b < total; b += 64) {
          int src = b + shift;
          if (src >= total) src -= total;
          my_hact[b] = ring[src];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- obs:actions_ (:1674-1765): spec -> 1-based row index, last writer wins (:1549) ----
      const ygo_option* opts = reinterpret_cast<const ygo_option*>(rec + a.off_opts);
      for (int p = lane; p < n_opt * a.mms; p += 64) {
        const int i = p / a.mms, j = p - i * a.mms;
        if (j < opts[i].n_specs) {
          const uint32_t key = opts[i].spec_key[j];
          int idx = 0;
          for (int r = 0; r < R; ++r)
            if (my_rowkey[r] == key) idx = r + 1;
          if (idx == 0) bad = true;  // spec2index.at(spec) throws
          my_actions[i * nf + 2 * j] = (uint8_t)(idx >> 8);
          my_actions[i * nf + 2 * j + 1] = (uint8_t)(idx & 0xff);
        }
      }
      for (int p = lane; p < n_opt * 9; p += 64) {
        const int i = p / 9, k = p - i * 9;
        my_actions[i * nf + 2 * a.mms + k] = opts[i].feats[k];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- h_card_ids read back from the images, literally (:1863-1879) ----
      if (lane < n_opt) {
        for (int j = 0; j < a.mms; ++j) {
          const uint8_t spec_index = my_actions[lane * nf + 2 * j + 1];  // low byte only (quirk Q7)
          if (spec_index == 0) break;
          const uint16_t c1 = my_cards[(spec_index - 1) * 39 + 0], c2 = my_cards[(spec_index - 1) * 39 + 1];
          my_hcids[lane * a.mms + j] = (uint16_t)((c1 << 8) + c2);
        }
      }
    } else if (lane == 0) {
      my_global[7] = 1;  // :1839
    }
    // ---- scalars + Env::Allocate common fields (env.h:196-222) ----
    if (lane == 0) {
      const bool done = h->done != 0;
      a.out.env_id[e] = h->env_id;
      a.out.penv[e] = h->env_id;
      a.out.elapsed[e] = h->elapsed_step;
      a.out.done[e] = done;
      a.out.reward[e] = h->reward;
      a.out.discount[e] = done ? 0.f : 1.f;
      a.out.step_type[e] = h->elapsed_step == 0 ? 0 : (done ? 2 : 1);
      a.out.trunc[e] = done && (h->elapsed_step >= a.max_episode_steps);
      a.out.num_options[e] = no_options ? 1 : n_opt;
      a.out.to_play[e] = to_play;
      a.out.is_selfplay[e] = h->is_selfplay;
      a.out.win_reason[e] = h->win_reason;
    }
  }
  if (__any(bad) && lane == 0) *a.err = 1u;
  __syncthreads();

  // ---- write the group's images: 16-byte-per-lane coalesced stores ----
  auto flush = [&](uint8_t* gbase, const uint8_t* lbase, int64_t per_env) {
    const int64_t bytes = per_env * gvalid;
    uint8_t* g = gbase + e0 * per_env;  // 16-byte aligned whenever 8*per_env is
    if (((uintptr_t)g & 15) == 0) {
      const int64_t nvec = bytes >> 4;
      for (int64_t q = tid; q < nvec; q += kThreads)
        reinterpret_cast<uint4*>(g)[q] = reinterpret_cast<const uint4*>(lbase)[q];
      for (int64_t b = (nvec << 4) + tid; b < bytes; b += kThreads) g[b] = lbase[b];
    } else {
      for (int64_t b = tid; b < bytes; b += kThreads) g[b] = lbase[b];
    }
  };
  flush(a.out.cards, lds + cv.cards, (int64_t)R * 39);
  flush(a.out.actions, lds + cv.actions, (int64_t)a.max_options * nf);
  flush(a.out.h_actions, lds + cv.hact, (int64_t)a.n_hist * nf);
  flush(a.out.global, lds + cv.global, 8);
  flush(reinterpret_cast<uint8_t*>(a.out.h_card_ids), lds + cv.hcids, (int64_t)a.max_options * a.mms * 2);
}

}  // namespace ygk
