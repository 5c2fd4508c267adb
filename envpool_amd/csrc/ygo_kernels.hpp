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
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/envpool_amd_ygo.h"
#include "../../include/envpool_amd_ygo_consts.h"  // every ygopro-core value assumed below

namespace ygk {

#ifndef YGO_WAVES
#define YGO_WAVES 4
#endif
constexpr int kGroup = YGO_WAVES;       // envs (= waves) per workgroup; any value works
constexpr int kThreads = 64 * kGroup;   // one wave64 per env

struct __align__(16) HashEntry {  // 16 bytes: one 16-byte load per probe
  uint32_t code;
  uint32_t type;     // TYPE_* flags compacted to bits 0..24 in ascending key order (:607-633); bit 31 = entry valid
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

// Two-choice (cuckoo) hashing: a code lives in slot_a or slot_b, both known from the code alone,
// so a lookup is two independent loads and never a dependent chain (with linear probing almost
// every wave had some lane in a collision chain, each step a full L2 round trip for all 64).
__host__ __device__ inline uint32_t slot_a(uint32_t code, uint32_t mask) { return ((code * 2654435761u) >> 7) & mask; }
__host__ __device__ inline uint32_t slot_b(uint32_t code, uint32_t mul_b, uint32_t mask) { return (((code ^ (code >> 15)) * mul_b) >> 9) & mask; }

struct Outs {
  int32_t* env_id; int32_t* penv; int32_t* elapsed; uint8_t* done; float* reward; float* discount;
  int32_t* step_type; uint8_t* trunc; uint8_t* cards; uint8_t* global; uint8_t* actions; uint8_t* h_actions;
  int32_t* num_options; int32_t* to_play; int32_t* is_selfplay; int32_t* win_reason; uint16_t* h_card_ids;
  uint8_t* action_mask;  // additive: [n, max_options] bool, arange(max_options) < info:num_options
};

__host__ __device__ inline int al16(int x) { return (x + 15) & ~15; }
// one option of a record: ygo_option_head + max_multi_select spec keys
__host__ __device__ inline int option_bytes(int mms) { return (int)sizeof(ygo_option_head) + 4 * mms; }
// per-wave LDS arena (byte offsets inside it, all 16-byte aligned). Each image region has 16
// spare bytes so that the image can start at its destination's address modulo 16.
struct Carve {
  int cards, actions, hact, global, hcids, spec, stage, wave_stride, total, spec_stride;
  int zero_bytes;                               // [cards, spec end) starts every env as zeros
  int off_hist, off_cards, rec_bytes;           // record layout (include/envpool_amd_ygo.h); options at off_cards + 16 * n_cards
  int stage_bytes;                              // landing area of the record in LDS (whole 1 KiB LDS-DMA instructions per span)
};
// direct-mapped spec table of one env: [2 players][7 locations][max_cards sequences] plus
// [2][8 monster zones][8 overlay letters], one byte each (1-based row index, 0 = absent)
#ifdef YGO_FAKE_SPEC  // diagnostic builds: a tiny (wrong) spec table, to price a smaller LDS arena
__host__ __device__ inline int spec_table_bytes(int) { return YGO_FAKE_SPEC; }
#define YGO_SPEC_IX(x) ((x) & (YGO_FAKE_SPEC - 1))
#else
__host__ __device__ inline int spec_table_bytes(int max_cards) { return al16(14 * max_cards + 128); }
#define YGO_SPEC_IX(x) (x)
#endif
__host__ __device__ inline Carve carve(int max_cards, int max_options, int n_hist, int mms) {
  const int nf = 9 + 2 * mms, R = 2 * max_cards;
  Carve c;
  c.off_hist = 64;
  // the card array comes LAST: only its used prefix is fetched
  c.off_cards = c.off_hist + al16(n_hist * nf);
  c.rec_bytes = al16(c.off_cards + R * (int)sizeof(ygo_card) + max_options * option_bytes(mms));
  // every LDS-DMA instruction writes a whole KiB (lanes past the end of a span repeat its last
  // piece): the landing area reaches to where the last instruction of either span can write
  c.stage_bytes = max((c.rec_bytes + 1023) & ~1023, c.off_hist + ((c.rec_bytes - c.off_hist + 1023) & ~1023));
  int o = 0;
  c.cards = o;   o += max(al16(R * 39) + 16, c.stage_bytes);  // doubles as the record's landing area
  c.actions = o; o += al16(max_options * nf) + 16;
  c.hact = o;    o += al16(n_hist * nf) + 16;
  c.hcids = o;   o += al16(max_options * mms * 2) + 16;
  c.global = o;  // (obs:global_ is built in scalar registers: no LDS image)
  c.spec_stride = spec_table_bytes(max_cards);
  c.spec = o;    o += c.spec_stride;
  c.zero_bytes = o;
  c.stage = c.cards;  // the record lands (LDS-DMA) where the card image is built afterwards
#ifdef YGO_LDS_PAD
  o += YGO_LDS_PAD;  // diagnostic builds: lower the occupancy to see how throughput follows it
#endif
  c.wave_stride = o;
  c.total = o * kGroup;
  return c;
}

struct Args {
  const uint8_t* records;
  const uint32_t* offsets;  // packed records: byte offset of record i from `records` (16-byte aligned); null: i * stride
  int64_t stride;
  int32_t n;     // rows [row0, n) of the batch are this launch's
  int32_t row0;
  const HashEntry* table;
  const HashAux* table_aux;
  uint32_t table_mask;  // capacity - 1
  uint32_t table_mul_b; // multiplier of the second hash function (chosen by the host build)
  int32_t max_cards, max_options, n_hist, mms, nf, max_episode_steps;
  int32_t lean_fetch;  // 1: records in pinned host memory, fetch schedule for PCIe (ygo_encode)
  Carve cv;  // LDS arena and record layout, computed once on the host
  Outs out;
  uint32_t* err;  // host-mapped error word of this batch
  unsigned long long* stamps;  // diagnostic builds only (-DYGO_STAMPS): cycles per phase
};

__device__ __forceinline__ void ft(int x, uint8_t& hi, uint8_t& lo) {  // float_transform :704-710
  x = x % 65536;
  hi = (uint8_t)(x >> 8);
  lo = (uint8_t)(x & 0xff);
}

// location2id (:531-542): DECK 1 .. EXTRA 7; 0 = the reference's .at() throws
__device__ __forceinline__ uint8_t loc_id(uint32_t loc) {
  return (loc != 0 && (loc & (loc - 1)) == 0 && loc <= EP_YGO_LOCATION_EXTRA) ? (uint8_t)(__ffs(loc)) : 0;
}
// position2id (:546-559): keys {0,1,2,3,4,5,8,0xa,0xc} -> 0..8; 0xff = throws
__device__ __forceinline__ uint8_t pos_id(uint32_t p) {
  switch (p) {
    case 0x0: return 0;  // POS_NONE (:544): overlay materials
    case EP_YGO_POS_FACEUP_ATTACK: return 1; case EP_YGO_POS_FACEDOWN_ATTACK: return 2; case EP_YGO_POS_ATTACK: return 3;
    case EP_YGO_POS_FACEUP_DEFENSE: return 4; case EP_YGO_POS_FACEUP: return 5; case EP_YGO_POS_FACEDOWN_DEFENSE: return 6;
    case EP_YGO_POS_FACEDOWN: return 7; case EP_YGO_POS_DEFENSE: return 8; default: return 0xff;
  }
}
__host__ __device__ inline uint32_t type_bit(int j) {  // ascending TYPE_* keys (:607-633)
  // 0x1,0x2,0x4, 0x10..0x4000 (bits 4..14), 0x10000..0x4000000 (bits 16..26)
  return j < 3 ? (1u << j) : (j < 14 ? (1u << (j + 1)) : (1u << (j + 2)));
}
static_assert(EP_YGO_TYPE_MONSTER == 0x1u && EP_YGO_TYPE_TRAP == 0x4u && EP_YGO_TYPE_NORMAL == 0x10u && EP_YGO_TYPE_TOKEN == 0x4000u &&
              EP_YGO_TYPE_QUICKPLAY == 0x10000u && EP_YGO_TYPE_LINK == 0x4000000u && EP_YGO_NUM_TYPES == 25,
              "type_bit(j) is the j-th TYPE_* key in ascending order");

// Slot of a spec key in the direct-mapped table, or -1 when the key is outside it (then the
// exact linear scan is used): plain specs by (opponent, location, sequence < max_cards), overlay
// specs by (opponent, monster zone < 8, letter < 8).
__host__ __device__ __forceinline__ int spec_slot(uint32_t key, int max_cards) {
  // (bitwise &,| on purpose: && and || become branches around exec-mask updates)
  const uint32_t opp = key & 0xff, loc7 = (key >> 8) & 0xff, seq = (key >> 16) & 0xff, suf = key >> 24;
  const bool loc_ok = (opp <= 1) & (loc7 != 0) & ((loc7 & (loc7 - 1)) == 0) & (loc7 <= EP_YGO_LOCATION_EXTRA);
  const bool plain_ok = loc_ok & (suf == 0) & (seq < (uint32_t)max_cards);
  const bool over_ok = loc_ok & (suf != 0) & (loc7 == EP_YGO_LOCATION_MZONE) & (seq < 8) & (suf <= 8);
  const int ti_plain = (int)((opp * 7 + (uint32_t)__builtin_ffs((int)loc7) - 1) * (uint32_t)max_cards + seq);
  const int ti_over = 14 * max_cards + (int)((opp * 8 + seq) * 8 + (suf - 1));
  return plain_ok ? ti_plain : (over_ok ? ti_over : -1);
}
__host__ __device__ __forceinline__ int spec_slots(int max_cards) { return 14 * max_cards + 128; }

// What a record stores per referenced spec ("spec reference"), and what the device derives for
// every card row to compare against it in the exact search: the slot in the direct-mapped table
// when the spec has one (an injective code of such specs), else bit 31 | the spec itself,
// opponent | location << 1 | sequence << 8 | suffix << 16. kSpecNone matches no row.
constexpr uint32_t kSpecNone = 0xfffffffeu;
__host__ __device__ __forceinline__ uint32_t spec_ref(uint32_t key, int max_cards) {
  const int ti = spec_slot(key, max_cards);
  if (ti >= 0) return (uint32_t)ti;
  const uint32_t opp = key & 0xff, loc = (key >> 8) & 0xff, seq = (key >> 16) & 0xff, suf = key >> 24;
  if (opp > 1 || loc > 0x7f) return kSpecNone;
  return 0x80000000u | opp | (loc << 1) | (seq << 8) | (suf << 16);
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
#define YGO_STAMP_END(k) do { const int lane = lane0; YGO_STAMP(k); } while (0)

// LDS traffic of one wave is executed in issue order; the fences keep the compiler from moving
// accesses of different lanes' data across the hand-over points.
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// zero the 16-byte aligned LDS span [lo, lo+bytes) with one wave
__device__ __forceinline__ void wave_zero16(unsigned char* lds, int lo, int bytes, int lane) {
  const uint4 z = make_uint4(0, 0, 0, 0);
  uint4* p = reinterpret_cast<uint4*>(lds + lo) + lane;
  const int nq = bytes >> 4, full = nq >> 6;  // whole-wave iterations need no per-lane bound check
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
  for (int i = 0; i < full; ++i) p[64 * i] = z;
  if (lane < nq - 64 * full) p[64 * full] = z;
}

// One LDS-DMA instruction: every lane copies 16 bytes from its `gsrc` to lds_base + lane*16 (1 KiB
// per wave instruction, no VGPR round trip). M0 carries the wave-uniform LDS base and is
// compiler-reserved, so it is saved and restored inside the statement
// (cdna_hip_programming.md section 5.7). The compiler does not see these loads: the waits for
// them are the explicit vm_wait below.
#ifdef YGO_NT_LOADS
#define YGO_NT_LOAD " nt"
#else
#define YGO_NT_LOAD ""
#endif
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" YGO_NT_LOAD "\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); }

// copy the LDS bytes [lo, lo+bytes) to g with one wave. When g and lo agree modulo 16 the bulk
// goes as aligned 16-byte stores; otherwise (odd-aligned caller buffers) byte by byte.
__device__ __forceinline__ void wave_flush(uint8_t* g, const unsigned char* lds, int lo, int bytes, int lane) {
  if ((((uintptr_t)g ^ (uintptr_t)lo) & 15) == 0) {
    const int head = min((16 - (lo & 15)) & 15, bytes);
    if (lane < head) g[lane] = lds[lo + lane];
    const int nvec = (bytes - head) >> 4, full = nvec >> 6;
    const unsigned char* src = lds + lo + head;
    uint8_t* dst = g + head;  // wave-uniform base; the lane part stays a 32-bit offset
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto piece = [&](uint32_t off) {  // written once, never re-read by this kernel: non-temporal
      const uint4 v = *reinterpret_cast<const uint4*>(src + off);
      u32x4 raw;
      __builtin_memcpy(&raw, &v, 16);
#ifndef YGO_NO_NT_STORES
      __builtin_nontemporal_store(raw, reinterpret_cast<u32x4*>(dst + off));
#else
      *reinterpret_cast<u32x4*>(dst + off) = raw;
#endif
    };
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int i = 0; i < full; ++i) piece((uint32_t)(lane + 64 * i) << 4);
    if (lane < nvec - 64 * full) piece((uint32_t)(lane + 64 * full) << 4);
    const int done = head + (nvec << 4);
    if (lane < bytes - done) g[done + lane] = lds[lo + done + lane];
  } else {
#pragma unroll 1
    for (int b = lane; b < bytes; b += 64) g[b] = lds[lo + b];
  }
}

__device__ __forceinline__ uint32_t lane_word(uint32_t v, int src_lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane);
}

// The kernel arguments, re-read from the kernarg segment at the point of use (scalar loads) and
// laundered so that the compiler cannot merge the read with the copy it made at kernel entry:
// 17 output pointers held live across the whole kernel cost ~60 spilled SGPRs.
typedef const __attribute__((address_space(4))) Args* KArgs;
__device__ __forceinline__ KArgs late_args() {
  KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

#ifndef YGO_OCC
#define YGO_OCC 5  // waves per SIMD the register budget is sized for (the LDS arena allows 20 per CU)
#endif
__global__ __launch_bounds__(kThreads, YGO_OCC) void ygo_encode(Args a) {
  extern __shared__ __align__(16) unsigned char lds[];
  const Carve& cv = a.cv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, and the compiler knows it
#ifdef YGO_STAMPS
  unsigned long long _t_prev = __builtin_amdgcn_s_memtime();
#endif
  const int R = 2 * a.max_cards, nf = a.nf;
  const int hist_bytes = a.n_hist * nf;
  const int64_t e = (int64_t)a.row0 + (int64_t)blockIdx.x * kGroup + wave;
  if (e >= a.n) return;  // wave-uniform; no workgroup barrier anywhere in this kernel

  const int arena = wave * cv.wave_stride;
  uint8_t* const my_spec = lds + arena + cv.spec;
  const uint8_t* const stage = lds + arena + cv.stage;
  const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)stage);
  // Only the used part of a record is fetched, by LDS-DMA in 16-byte pieces (lanes past the last
  // piece of a span re-read it: same cache line, no extra traffic). Two schedules, chosen by the
  // host per launch:
  //   records in HBM: the first KiB at once, in parallel with the header's scalar loads (below),
  //     the rest up to the last listed card once the header has told how many there are;
  //   records in pinned host memory (the kernel reads them over PCIe itself, a.lean_fetch): nothing
  //     before the header; then exactly the history ring with the used options, and the listed
  //     cards, each span to its own place in the landing area - every byte saved is PCIe time.
  const uint8_t* rec_at = a.records + e * a.stride;
  if (a.offsets) {  // packed batch: records back to back, in the order their producers finished
    typedef const __attribute__((address_space(4))) uint32_t* KOff;
    rec_at = a.records + ((KOff)(uintptr_t)a.offsets)[e];  // e is wave-uniform: a scalar load
  }
  const uint8_t* const rec = rec_at;
  auto fetch_span = [&](int first_piece, int n_pieces, int c_lo) {  // pieces [first, first + n) -> stage + 16*first
    for (int c = c_lo; 64 * c < n_pieces; ++c) {
      const int piece = first_piece + min(lane + 64 * c, n_pieces - 1);
      dma16(rec + (int64_t)piece * 16, stage_lds + 16u * (uint32_t)first_piece + 1024u * (uint32_t)c);
    }
  };
  if (!a.lean_fetch) fetch_span(0, 64, 0);

  // the images of one env leave the CU: aligned 16-byte-per-lane stores for the bulk
  auto flush_env = [&](int64_t env, int lane) {
    KArgs ka = late_args();
    const int act_bytes = a.max_options * nf, cid_bytes = a.max_options * a.mms * 2;
    uint8_t* g_actions = ka->out.actions + env * (int64_t)act_bytes;
    uint8_t* g_hact = ka->out.h_actions + env * (int64_t)hist_bytes;
    uint8_t* g_hcids = reinterpret_cast<uint8_t*>(ka->out.h_card_ids) + env * (int64_t)cid_bytes;
    const int pa = act_bytes >> 4, ph = hist_bytes >> 4, pc = cid_bytes >> 4;
    const uint32_t misfit = (uint32_t)(act_bytes | hist_bytes | cid_bytes);
    if (YGO_SKIP & 1024) {
    } else if ((misfit & 15u) == 0 && pa + ph + pc <= 64) {
      // all three images are whole 16-byte pieces at 16-byte aligned addresses (the default
      // shapes): one load and one store per lane moves all of them
      const int q = lane;
      const bool in_a = q < pa, in_h = q < pa + ph;
      const int piece = in_a ? q : (in_h ? q - pa : q - pa - ph);
      const int src = (in_a ? cv.actions : (in_h ? cv.hact : cv.hcids)) + 16 * piece;
      uint8_t* dst = (in_a ? g_actions : (in_h ? g_hact : g_hcids)) + 16 * piece;
      if (q < pa + ph + pc) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint4 v = *reinterpret_cast<const uint4*>(lds + arena + src);
        u32x4 raw;
        __builtin_memcpy(&raw, &v, 16);
#ifndef YGO_NO_NT_STORES
        __builtin_nontemporal_store(raw, reinterpret_cast<u32x4*>(dst));
#else
        *reinterpret_cast<u32x4*>(dst) = raw;
#endif
      }
    } else {
      wave_flush(g_actions, lds, arena + cv.actions + (int)((uintptr_t)g_actions & 15), act_bytes, lane);
      wave_flush(g_hact, lds, arena + cv.hact + (int)((uintptr_t)g_hact & 15), hist_bytes, lane);
      wave_flush(g_hcids, lds, arena + cv.hcids + (int)((uintptr_t)g_hcids & 14), cid_bytes, lane);
    }
  };

  // Per env, in this order:
  //   A  the header arrives by scalar loads; (records in HBM: the first KiB of the record is on
  //      its way into LDS by LDS-DMA meanwhile, no VGPR round trip, into the region where the card
  //      image is built later;) the small images and the spec table are zeroed (the reference's
  //      fresh zero-filled StateBuffer, array.h:77-81)
  //   B  header -> scalars: segment walk, obs:global_; the rest of the record's used prefix is
  //      fetched; then, per instantiation (1, 2 or 4 cards per lane): packed card / option words
  //      -> lanes, history rows rotated into their image, hash probes issued
  //   D  the landing area becomes the zeroed card image
  //   E  placeholder rows, option feature bytes
  //   F  probes settled
  //   C  scalar outputs and obs:global_ (after F: VMEM retires in issue order)
  //   H  rows (the card image leaves the CU right behind them), options, then the small images
  {
    bool bad_u = false;   // raised in wave-uniform code
    uint32_t badv = 0;    // raised per lane (a VGPR flag: no exec-mask bookkeeping around it)
    // LDS offsets congruent to the images' destinations modulo 16 (the output arrays themselves
    // are 256-byte aligned: they are carved from the library's own slab, ygo.hip make_outs)
    const uint32_t e4 = (uint32_t)e & 15u;
    uint8_t* const my_cards = lds + arena + cv.cards + (int)((e4 * (uint32_t)(R * 39)) & 15u);
    uint8_t* const my_actions = lds + arena + cv.actions + (int)((e4 * (uint32_t)(a.max_options * nf)) & 15u);
    uint8_t* const my_hact = lds + arena + cv.hact + (int)((e4 * (uint32_t)hist_bytes) & 15u);
    uint16_t* const my_hcids =
        reinterpret_cast<uint16_t*>(lds + arena + cv.hcids + (int)((e4 * (uint32_t)(a.max_options * a.mms * 2)) & 14u));

    if (!(YGO_SKIP & 512)) wave_zero16(lds, arena + cv.actions, cv.zero_bytes - cv.actions, lane);
    // ---- B: header, every field a wave-uniform scalar. It comes by scalar loads straight from
    // the record (the address is wave-uniform), not from the landing area: the segment walk
    // below runs while the DMA is still in flight ----
    uint32_t w[16];
    {
      typedef const __attribute__((address_space(4))) uint32_t* KWords;
      KWords hdr = (KWords)(uintptr_t)rec;
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i] = hdr[i];
    }
    const int to_play = w[0] & 0xff;
    int n_opt = w[0] >> 24;
    const int ha_p = (w[1] >> 16) & 0xff;
    const bool revealed_any = ((w[1] >> 8) & 0xff) != 0;
    const uint64_t hand_mask = (uint64_t)w[10] | ((uint64_t)w[11] << 32);
    const bool no_options = n_opt == 0;                // early-out of WriteState (:1837-1841)
    if (n_opt > a.max_options) n_opt = a.max_options;  // options_.resize (:1849-1851)

    // segments: 2 players x 7 locations (:1510-1519), walked with scalar counters only. The listed
    // cards of player pi=0 sit in rows equal to their card index; those of pi=1 are all shifted by
    // ONE constant (rows of the hidden DECK / HAND in front of them; the hidden EXTRA comes last),
    // so a lane needs no per-segment search: pi = (k >= n0), r = k + (pi ? shift1 : 0). A segment
    // that would overrun max_cards rows (the reference has no bound: a buffer overrun there) and
    // every later one of that player are left out and the batch is flagged: k < fit_end[pi].
    // [hand1_lo, hand1_hi) is the opponent's listed HAND (the reveal rule, :1541-1545).
    int n_cards = 0, n0 = 0, shift1 = 0, fit_end0 = 0, fit_end1 = 0, hand1_lo = 0, hand1_hi = 0;
    int ph_row[3] = {0, 0, 0}, ph_cnt[3] = {0, 0, 0};
    if (!no_options) {
      bool over = false;
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        int row = pi * a.max_cards;
        if (pi == 1) n0 = n_cards;
#pragma unroll
        for (int li = 0; li < 7; ++li) {
          const int s = pi * 7 + li;
          const int cnt = (int)((w[12 + (s >> 2)] >> (8 * (s & 3))) & 0xff);
          bool hidden = (li == 0 || li == 1 || li == 6);
          if (li == 1 && revealed_any) hidden = false;  // :1522-1525
          if (pi == 1 && hidden) {
            const int ph = li == 0 ? 0 : (li == 1 ? 1 : 2);
            ph_row[ph] = row;
            ph_cnt[ph] = cnt;
          } else {
            if (pi == 1 && li == 1) { hand1_lo = n_cards; hand1_hi = n_cards + cnt; }
            if (pi == 1 && li == 2) shift1 = row - n_cards;  // the same for every listed pi=1 segment
            n_cards += cnt;
          }
          row += cnt;
        }
        if (row > (pi + 1) * a.max_cards) over = true;
      }
      fit_end0 = n0;
      fit_end1 = n_cards;
      if (over) {  // some player has more rows than max_cards: find where each stops fitting
        bad_u = true;
        int nc = 0;
        fit_end0 = 0;
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
          int row = pi * a.max_cards;
          if (pi == 1) fit_end1 = nc;
#pragma unroll
          for (int li = 0; li < 7; ++li) {
            const int s = pi * 7 + li;
            const int cnt = (int)((w[12 + (s >> 2)] >> (8 * (s & 3))) & 0xff);
            bool hidden = (li == 0 || li == 1 || li == 6);
            if (li == 1 && revealed_any) hidden = false;
            const bool fits = row + cnt <= (pi + 1) * a.max_cards;
            if (pi == 1 && hidden) {
              if (!fits) ph_cnt[li == 0 ? 0 : (li == 1 ? 1 : 2)] = 0;
            } else {
              if (fits) { if (pi == 0) fit_end0 = nc + cnt; else fit_end1 = nc + cnt; }
              nc += cnt;
            }
            row += cnt;
          }
        }
      }
    }
    const int n_listed = min(n_cards, R);
    // the used part of a record is contiguous: header | history ring | listed cards | used options
    const int off_opts = cv.off_cards + n_listed * (int)sizeof(ygo_card);
    if (!no_options) {
      const int used_end = min((off_opts + n_opt * option_bytes(a.mms) + 15) >> 4, cv.rec_bytes >> 4);
      if (!a.lean_fetch) fetch_span(0, used_end, 1);
      else fetch_span(cv.off_hist >> 4, used_end - (cv.off_hist >> 4), 0);
    }

    // obs:global_ (:1605-1620), eight bytes built as two wave-uniform words: lp of to_play and of
    // the opponent (float_transform: high byte, low byte), phase id, "I am player 0", "it is my
    // turn"; an env without options shows only byte 7 = 1 (:1839)
    uint32_t glob0 = 0, glob1 = 1u << 24;
    if (!no_options) {
      const uint32_t turn_player = (w[0] >> 8) & 0xff, phase = w[6];
      const uint32_t me = to_play == 0 ? w[4] : w[5], op = to_play == 0 ? w[5] : w[4];
      glob0 = ((me >> 8) & 0xff) | ((me & 0xff) << 8) | (((op >> 8) & 0xff) << 16) | ((op & 0xff) << 24);
      const bool phase_ok = phase != 0 && (phase & (phase - 1)) == 0 && phase <= EP_YGO_PHASE_END;  // phase2id.at()
      if (!phase_ok) bad_u = true;
      glob1 = (phase_ok ? (uint32_t)(__ffs(phase) - 1) : 0u) | (to_play == 0 ? 1u << 8 : 0u) |
              ((uint32_t)to_play == turn_player ? 1u << 16 : 0u);
    }

    // Everything from here on comes in three instantiations by the number of listed cards: NIT = 1
    // (at most 64: one per lane), NIT = 2 (at most 128: a mid-game state lists 60-90, the own
    // deck included), NIT = 4 otherwise (lane l owns cards l, l+64, l+128, l+192). The choice is
    // wave-uniform.
    auto tail = [&](auto nit_c) {
      constexpr int NIT = decltype(nit_c)::value;
      // the card / option arrays go to their owning lanes: lane l owns cards l, l+64, l+128, l+192
      // and option l; the history ring is held four bytes per lane. They stay PACKED (raw dwords)
      // until the code that needs a field: unpacked early they cost 60 more VGPRs.
      uint32_t cw[NIT][4];   // ygo_card: code | controller,location,sequence,position | level,rank,link,is_material | attack,defense
      uint32_t mw[3 + EP_YGO_MAX_SPECS];  // option: feats[0..8], n_specs, pad | spec_key[max_multi_select]
      // (read unconditionally, index clamped: a lane beyond the listed cards / the options holds
      // words it never uses, and its hash probe coalesces with the last listed card's)
      // ---- A: the landing area is complete ----
      vm_wait<0>();
      wave_fence();
      const int last_card = max(n_listed, 1) - 1;  // lanes without a card re-read the last listed one
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const uint4 c4 = reinterpret_cast<const uint4*>(stage + cv.off_cards)[min(lane + 64 * it, last_card)];  // one 16-byte read
        cw[it][0] = c4.x; cw[it][1] = c4.y; cw[it][2] = c4.z; cw[it][3] = c4.w;
      }
      {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(stage + off_opts) + min(lane, a.max_options - 1) * (3 + a.mms);
#pragma unroll
        for (int d = 0; d < 3 + EP_YGO_MAX_SPECS; ++d) mw[d] = src[d];  // (past 3 + mms: in-bounds bytes nobody uses)
      }
      // obs:h_actions_ is the ring rotated so that the newest entry is row 0 (:1881-1892): image
      // row i = ring row (i + ha_p) mod n_history_actions. One lane per row, copied now (the
      // landing area is about to be zeroed) in 8-byte pieces at byte granularity; the last piece
      // ends with the row and may overlap the one before it (rows are 11..25 bytes).
      if (!no_options && !(YGO_SKIP & 4)) {
        const uint8_t* ring = stage + cv.off_hist;
        const int hp = ha_p < a.n_hist ? ha_p : 0;
        for (int i = lane; i < a.n_hist; i += 64) {
          int srow = i + hp;
          srow -= srow >= a.n_hist ? a.n_hist : 0;
          const uint8_t* sp = ring + srow * nf;
          uint8_t* dp = my_hact + i * nf;
          uint64_t t0, t1 = 0, t2 = 0, tl;
          __builtin_memcpy(&t0, sp, 8);
          if (nf >= 16) __builtin_memcpy(&t1, sp + 8, 8);
          if (nf >= 24) __builtin_memcpy(&t2, sp + 16, 8);
          __builtin_memcpy(&tl, sp + nf - 8, 8);
          __builtin_memcpy(dp, &t0, 8);
          if (nf >= 16) __builtin_memcpy(dp + 8, &t1, 8);
          if (nf >= 24) __builtin_memcpy(dp + 16, &t2, 8);
          __builtin_memcpy(dp + nf - 8, &tl, 8);
        }
      }
      // static data: cards_.at(code) / card_ids_.at(code). The first probes of a lane's cards are
      // in flight together. (Unconditional loads: a lane without a card probes the slot of code 0,
      // in bounds, and ignores the answer - a conditional load would make the compiler wait for
      // each in turn.) An overlay material shows its static attack / defense (c_get_card copy,
      // :2109-2114): they replace the (absent) dynamic ones in the lane's copy.
      uint4 tq[NIT], tq2[NIT];  // HashEntry: code | type | card_id, level, attr_id | race_id (both candidate slots)
      const char* const table_b = reinterpret_cast<const char*>(a.table);  // (32-bit byte offsets: the table is < 4 GiB)
      auto probe = [&](int it) {
        if (!(YGO_SKIP & (32 | 64))) {
          tq[it] = *reinterpret_cast<const uint4*>(table_b + (slot_a(cw[it][0], a.table_mask) << 4));
          tq2[it] = *reinterpret_cast<const uint4*>(table_b + (slot_b(cw[it][0], a.table_mul_b, a.table_mask) << 4));
        }
        if (YGO_SKIP & 64)  // priced without the table read: a made-up known card
          tq[it] = tq2[it] = make_uint4(cw[it][0], kEntryValid | 1u | (1u << (4 + (cw[it][0] & 7))), (cw[it][0] & 0xffff) | (4u << 16) | (1u << 24), 1u);
        if (YGO_SKIP & 32) tq[it] = tq2[it] = make_uint4(0, 0, 0, 0);
      };
#pragma unroll
      for (int it = 0; it < NIT; ++it) probe(it);
      YGO_STAMP(0);

      YGO_STAMP(1);
      // ---- D ----
      wave_fence();
      if (!(YGO_SKIP & 128)) wave_zero16(lds, arena + cv.cards, cv.actions, lane);
      wave_fence();

      // ---- E: hidden opponent locations get placeholder rows {col2 = location id, col4 = 1}
      // (:1526-1532); obs:global_ (:1605-1620) ----
      // DECK and HAND placeholders are adjacent rows (the hidden HAND follows the DECK directly);
      // the EXTRA ones come after the listed segments
      if (!(YGO_SKIP & 256)) {
        const int c0 = ph_cnt[0], c01 = ph_cnt[0] + ph_cnt[1], call = c01 + ph_cnt[2];
        for (int q = lane; q < call; q += 64) {
          uint8_t* prow = my_cards + (q < c01 ? ph_row[0] + q : ph_row[2] + (q - c01)) * 39;
          prow[2] = (uint8_t)(q < c0 ? 1 : (q < c01 ? 2 : 7));
          prow[4] = 1;
        }
      }
      // (work that needs no card data goes here, under the latency of the hash probes)
      // the nine spec-independent feature bytes of each option (columns 2*mms .. 2*mms+8)
      if (lane < n_opt && !(YGO_SKIP & 8)) {  // (the image is zero: writing zeros changes nothing)
        uint8_t* frow = my_actions + lane * nf + 2 * a.mms;
        __builtin_memcpy(frow, &mw[0], 8);
        frow[8] = (uint8_t)mw[2];
      }
      YGO_STAMP(2);

      // ---- F: the entry is the candidate whose code matches (neither: the card is unknown). An
      // overlay material shows its static attack / defense (c_get_card copy, :2109-2114), fetched
      // from the parallel table now that the slot is known: they replace the (absent) dynamic ones.
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const bool hit_a = ((tq[it].y & kEntryValid) != 0) & (tq[it].x == cw[it][0]);
        if (!hit_a) tq[it] = tq2[it];
        if ((cw[it][2] >> 24) != 0 && lane + 64 * it < n_listed && !(YGO_SKIP & (32 | 64))) {
          const uint32_t slot = hit_a ? slot_a(cw[it][0], a.table_mask) : slot_b(cw[it][0], a.table_mul_b, a.table_mask);
          const uint2 x = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.table_aux) + (slot << 3));
          cw[it][3] = (x.x & 0xffff) | (x.y << 16);
        }
      }
      // every table read has returned (make the compiler settle them here: a wait for one of them
      // after the prefetch below would also wait for the prefetch)
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        asm volatile("" : "+v"(tq[it].x), "+v"(tq[it].y), "+v"(tq[it].z), "+v"(tq[it].w), "+v"(cw[it][3]) : : "memory");
      YGO_STAMP(3);

      // ---- C: scalars + Env::Allocate common fields (env.h:196-222). Issued only now: VMEM retires
      // in issue order, so stores in front of the probe wait above would have put their
      // acknowledgement latency into it ----
      if (lane == 0 && !(YGO_SKIP & 2048)) {
        KArgs ka = late_args();
        const bool done = (w[1] >> 24) != 0;
        const int elapsed = (int)w[3];
        // (one 32-bit lane offset per element width, added to the wave-uniform array bases by the
        // store instruction itself: e < 2^29)
        uint32_t ev = (uint32_t)e;
        asm volatile("" : "+v"(ev));
        const uint32_t o4 = ev << 2, o8 = ev << 3;
        auto put = [](auto* base, uint32_t off, auto v) {
          *reinterpret_cast<decltype(v)*>(reinterpret_cast<char*>(base) + off) = v;
        };
        put(ka->out.env_id, o4, (int)w[2]);
        put(ka->out.penv, o4, (int)w[2]);
        put(ka->out.elapsed, o4, elapsed);
        put(ka->out.done, ev, (uint8_t)done);
        put(ka->out.reward, o4, __uint_as_float(w[8]));
        put(ka->out.discount, o4, done ? 0.f : 1.f);
        put(ka->out.step_type, o4, elapsed == 0 ? 0 : (done ? 2 : 1));
        put(ka->out.trunc, ev, (uint8_t)(done && (elapsed >= ka->max_episode_steps)));
        put(ka->out.num_options, o4, no_options ? 1 : n_opt);
        put(ka->out.to_play, o4, to_play);
        put(ka->out.is_selfplay, o4, (int)(w[1] & 0xff));
        put(ka->out.win_reason, o4, (int)w[7]);
        put(ka->out.global, o8, make_uint2(glob0, glob1));
      }
      // the legal-action mask the reference leaves implicit (row i of obs:actions_ is an option iff
      // i < info:num_options): one byte per option slot, lanes 0..max_options-1
      if (!(YGO_SKIP & 2048)) {
        KArgs ka = late_args();
        const int shown = no_options ? 1 : n_opt;
        uint8_t* const mrow = ka->out.action_mask + (int64_t)e * ka->max_options;
        for (int k = lane; k < ka->max_options; k += 64) mrow[k] = (uint8_t)(k < shown);
      }


      // ---- H ----
      // ---- obs:cards_ : one lane per LISTED card (:1508-1603), written branch-free: the 14 leading
      // bytes of a row are computed with selects (hidden fields as zeros, which is what the zeroed
      // image holds anyway) and leave as three dwords and a short; only the type flags loop.
      // Every listed row also enters the env's direct-mapped spec table (the reference's per-step
      // spec2index hash map, :1549). ----
      int ti_of[NIT];
      uint32_t row1_of[NIT], key_of[NIT];  // spec-table slot, 1-based row and spec key of the lane's cards
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        // one card at a time, unpacked here and no earlier
        asm volatile("" : "+v"(cw[it][0]), "+v"(cw[it][1]), "+v"(cw[it][2]), "+v"(tq[it].z) : : "memory");
        const int k = lane + 64 * it;
        ti_of[it] = -1;
        row1_of[it] = 0;
        key_of[it] = kSpecNone;
        if (k < n_listed && !(YGO_SKIP & 1)) {
          const uint32_t pi = k >= n0 ? 1u : 0u;
          if (k < (pi ? fit_end1 : fit_end0)) {
            const uint32_t c1 = cw[it][1], c2 = cw[it][2];
            const uint32_t ctrl = c1 & 0xff, loc = (c1 >> 8) & 0xff, seq = (c1 >> 16) & 0xff, pos = c1 >> 24;
            const bool in_hand1 = (k >= hand1_lo) & (k < hand1_hi);
            const int r = k + (pi ? shift1 : 0);
            const uint32_t loc7 = loc & 0x7f, overlay = loc >> 7;
            // (bitwise &,| on purpose: && and || become branches around exec-mask updates)
            const bool loc_ok = (loc7 != 0) & ((loc7 & (loc7 - 1)) == 0) & (loc7 <= EP_YGO_LOCATION_EXTRA);
            const uint32_t lid = loc_ok ? (uint32_t)__ffs(loc7) : 0u;  // location2id
            // position2id (:546-559): keys {0,1,2,3,4,5,8,0xa,0xc} -> 0..8, one nibble per key; 0xf = throws
            const uint32_t pid = pos < 16 ? (uint32_t)(0xFFF8F7F6FF543210ull >> (4 * (pos & 15))) & 0xf : 0xfu;
            const bool known = ((tq[it].y & kEntryValid) != 0) & (tq[it].x == cw[it][0]);
            const bool revealed = in_hand1 & (seq < 64) & (((hand_mask >> (seq & 63)) & 1ull) != 0);  // :1541-1545
            bool hide = (pi != 0) & ((pos & EP_YGO_POS_FACEDOWN) != 0) & !revealed;                                   // POS_FACEDOWN (:1540)
            hide = (hide & (overlay == 0)) | !known;                                                  // :1562-1564
            const uint32_t cid = tq[it].z & 0xffff, tlevel = (tq[it].z >> 16) & 0xff, attr = tq[it].z >> 24, race = tq[it].w & 0xff;
            badv |= (!known | !loc_ok | ((overlay == 0) & (pid == 0xf)) | (!hide & ((attr == 0xff) | (race == 0xff)))) ? 1u : 0u;
            const uint32_t lv = c2 & 0xff, rk = (c2 >> 8) & 0xff, lk = (c2 >> 16) & 0xff;
            uint32_t level = lv ? lv : tlevel;  // dynamic overrides of get_cards_in_location (:2090-2099, :2126-2133)
            level = rk ? rk : level;
            level = lk ? lk : level;
            level = (c2 >> 24) ? tlevel : level;  // an overlay material shows its static level
            // float_transform (:704-710): x % 65536 keeps the low 16 bits of x, sign included
            const uint32_t atk = cw[it][3] & 0xffff, def = cw[it][3] >> 16;
            const uint32_t show = hide ? 0u : 0xffffffffu;
            const uint32_t b3 = ((loc7 == EP_YGO_LOCATION_MZONE) | (loc7 == EP_YGO_LOCATION_SZONE) | (loc7 == EP_YGO_LOCATION_GRAVE)) ? seq + 1 : 0u;  // :1573-1578
            const uint32_t b5 = overlay ? 5u : (pid == 0xf ? 0u : pid);  // overlay: position2id.at(POS_FACEUP)
            const uint32_t w0 = (((cid >> 8) | ((cid & 0xff) << 8)) & show) | (lid << 16) | ((b3 & 0xff) << 24);
            const uint32_t w1 = (ctrl != (uint32_t)to_play ? 1u : 0u) | (b5 << 8) | (overlay << 16) | (((attr == 0xff ? 0u : attr) << 24) & show);
            const uint32_t w2 = ((race == 0xff ? 0u : race) | (level << 8) | ((atk >> 8) << 16) | ((atk & 0xff) << 24)) & show;
            const uint16_t w3 = (uint16_t)(((def >> 8) | ((def & 0xff) << 8)) & show);
            uint8_t* row = my_cards + r * 39;
            __builtin_memcpy(row, &w0, 4);
            __builtin_memcpy(row + 4, &w1, 4);
            __builtin_memcpy(row + 8, &w2, 4);
            __builtin_memcpy(row + 12, &w3, 2);
            // type_to_ids (:635-642): one byte per TYPE_* key in ascending order (the table holds the
            // flags already compacted to bits 0..24); set bits only
            uint32_t tb = tq[it].y & 0x01ffffffu & show;
            while (tb) {
              const int b = __ffs(tb) - 1;
              tb &= tb - 1;
              row[14 + b] = 1;
            }
            const uint32_t suf = overlay ? (pos + 1) & 0xff : 0u;
            const int ti_plain = (int)((pi * 7 + lid - 1) * (uint32_t)a.max_cards + seq);
            const int ti_over = 14 * a.max_cards + (int)((pi * 8 + seq) * 8 + (suf - 1));
            const bool plain_ok = loc_ok & (suf == 0) & (seq < (uint32_t)a.max_cards);
            const bool over_ok = loc_ok & (suf != 0) & (loc7 == EP_YGO_LOCATION_MZONE) & (seq < 8) & (suf <= 8);
            const int ti = plain_ok ? ti_plain : (over_ok ? ti_over : -1);
            if (ti >= 0) my_spec[YGO_SPEC_IX(ti)] = (uint8_t)(r + 1);
            ti_of[it] = ti;
            row1_of[it] = (uint32_t)(r + 1) & 0xff;
            key_of[it] = ti >= 0 ? (uint32_t)ti : 0x80000000u | pi | (loc7 << 1) | (seq << 8) | (suf << 16);
          }
        }
      }
      wave_fence();
      // obs:cards_ is complete: its 5.7 KB leave the CU now, under the option phase (which only
      // reads the image), instead of at the very end of the wave's life
      if (!(YGO_SKIP & 16)) {
        uint8_t* g_cards = late_args()->out.cards + e * (int64_t)(R * 39);
        wave_flush(g_cards, lds, arena + cv.cards + (int)((uintptr_t)g_cards & 15), R * 39, lane);
      }
      YGO_STAMP(4);
      // two rows with one spec (never in a consistent duel state): the table holds an arbitrary
      // one of them, the reference the LAST one; detect it and take the exact scan instead
      bool dup = false;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (ti_of[it] >= 0 && !(YGO_SKIP & 2) && my_spec[YGO_SPEC_IX(ti_of[it])] != row1_of[it]) dup = true;
      dup = __any(dup);
      // ---- obs:actions_ (:1674-1765) + h_card_ids (:1863-1879): lane i owns option i ----
#pragma unroll
      for (int d = 0; d < 3 + EP_YGO_MAX_SPECS; ++d) asm volatile("" : "+v"(mw[d]));
      if (!(YGO_SKIP & 8)) {
        uint8_t* arow = my_actions + lane * nf;
        const int ns = lane < n_opt ? min((int)((mw[2] >> 8) & 0xff), a.mms) : 0;
        bool open = true;
#pragma unroll
        for (int j = 0; j < EP_YGO_MAX_SPECS; ++j) {
          if (__any(j < ns)) {  // wave-uniform: the exact scan below is a whole-wave affair
            const bool act = j < ns;
            const uint32_t key = mw[3 + j];  // spec reference: a table slot, or bit 31 | the spec
            const bool fast = act & (key < (uint32_t)spec_slots(a.max_cards)) & !dup;
            int idx = 0;
            if (fast) idx = my_spec[YGO_SPEC_IX(key)];
            // A key outside the direct-mapped table, or an env with two rows under one spec: the
            // exact search, last row wins (:1549). Rare; done by the whole wave for one option at a
            // time - every lane compares the key with the cards it owns, then a wave-wide maximum.
            uint64_t need = __ballot(act & !fast);
            while (need) {
              const int src = __ffsll((unsigned long long)need) - 1;
              need &= need - 1;
              const uint32_t want = (uint32_t)__builtin_amdgcn_readlane((int)key, src);
              int best = 0;
#pragma unroll
              for (int it = 0; it < NIT; ++it)
                if (key_of[it] == want) best = max(best, (int)row1_of[it]);
#pragma unroll
              for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off));
              if (lane == src) idx = best;
            }
            if (act) {
              badv |= idx == 0 ? 1u : 0u;  // spec2index.at(spec) throws
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
        }
      }
      YGO_STAMP(5);
      if ((bad_u || __any(badv != 0)) && lane == 0) *late_args()->err = 1u;
      wave_fence();
      flush_env(e, lane);
    };
    if (n_listed <= 64) tail(std::integral_constant<int, 1>{});
    else if (n_listed <= 128) tail(std::integral_constant<int, 2>{});
    else tail(std::integral_constant<int, 4>{});
    YGO_STAMP(6);
  }
}

}  // namespace ygk
