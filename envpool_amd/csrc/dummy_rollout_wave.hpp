// dummy_rollout_wave.hpp — the fused lockstep loop, tuned for gfx950.
//
// Same arithmetic as dummy_kernels.hpp (see the reference citations there); what changes is
// the schedule:
//   * one wave64 per workgroup, 64 envs per wave, env state in registers for all T steps;
//     waves never meet at a barrier, so each of a CU's four SIMDs runs its own stream;
//   * the action rows of step t+D are requested D steps ahead into a register queue, so the
//     wave always has D steps of loads in flight and never waits for its own stores
//     (s_waitcnt vmcnt is in-order: nothing in the loop may force vmcnt(0));
//   * a step's 64 result rows are staged in 1.75 KiB of wave-private LDS (no barrier: LDS
//     operations of one wave execute in order) and leave as FIVE 16-byte-per-lane store
//     instructions: two carry all ten non-obs keys (each 16-lane group owns one key's
//     256-byte tile), three carry the 2560-byte obs:raw tile, zeros included;
//   * rare work (episode statistics, error flags) is kept in registers and flushed after
//     the loop, so the loop body has no data-dependent memory traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dummy_kernels.hpp"

namespace epk {

constexpr int kWave = 64;
// wave-private LDS image of one step's results
constexpr int kLdsEnvId = 0;      // int32[64]
constexpr int kLdsElapsed = 256;  // int32[64]
constexpr int kLdsStepType = 512; // int32[64]
constexpr int kLdsDiscount = 768; // float[64]
constexpr int kLdsSR = 1024;      // int2[64]  (state, raw1)
constexpr int kLdsDone = 1536;    // uint8[64]
constexpr int kLdsTrunc = 1600;   // uint8[64]
constexpr int kLdsZero = 1664;    // 16 zero bytes
constexpr int kLdsBytes = 1792;

// Per-lane, loop-invariant description of the five store instructions.
struct LaneMap {
  uint32_t dst1, src1;  // store 1: byte offset inside the slot / LDS byte address
  uint32_t dst2, src2;  // store 2 (lanes 0..59)
  uint32_t rdst[3];     // obs:raw chunk offsets inside the slot
  uint32_t rsrc[3];     // LDS address of the (state, raw1) pair feeding the chunk
  uint32_t rkind[3];    // 0: zeros, 1: pair in .xy, 2: pair in .zw
};

__device__ __forceinline__ LaneMap make_lane_map(const RingLayout& ring, int lane, int64_t row_base) {
  LaneMap m;
  const int g = lane >> 4, j = lane & 15;
  // store 1: info:env_id | info:players.env_id | elapsed_step | step_type
  {
    const uint32_t koff = g == 0 ? ring.off[0] : g == 1 ? ring.off[1] : g == 2 ? ring.off[2] : ring.off[6];
    const int lsrc = g <= 1 ? kLdsEnvId : g == 2 ? kLdsElapsed : kLdsStepType;
    m.dst1 = koff + (uint32_t)(row_base * 4) + j * 16;
    m.src1 = lsrc + j * 16;
  }
  // store 2: discount | reward (+0.0f) | info:players.id (0) | done, trunc, players.done
  if (g < 3) {
    const uint32_t koff = g == 0 ? ring.off[5] : g == 1 ? ring.off[4] : ring.off[10];
    m.dst2 = koff + (uint32_t)(row_base * 4) + j * 16;
    m.src2 = g == 0 ? kLdsDiscount + j * 16 : kLdsZero;
  } else {
    const int b = j >> 2, jj = j & 3;  // b: 0 done, 1 trunc, 2 players.done, 3 idle
    const uint32_t koff = b == 1 ? ring.off[7] : b == 2 ? ring.off[9] : ring.off[3];
    m.dst2 = koff + (uint32_t)row_base + jj * 16;
    m.src2 = (b == 1 ? kLdsTrunc : kLdsDone) + jj * 16;
  }
  // obs:raw (S = 10): the wave's tile is 160 16-byte chunks; five chunks cover two rows:
  //   chunk 5k   = {state[2k], raw1[2k], 0, 0}      chunk 5k+2 = {0, 0, state[2k+1], raw1[2k+1]}
  //   chunks 5k+1, 5k+3, 5k+4 = zeros
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int q = lane + 64 * c;
    int k = q / 5, r = q - 5 * k;
    m.rdst[c] = ring.off[8] + (uint32_t)(row_base * 40) + q * 16;
    m.rkind[c] = r == 0 ? 1u : (r == 2 ? 2u : 0u);
    int row = r == 0 ? 2 * k : 2 * k + 1;
    m.rsrc[c] = kLdsSR + (row & 63) * 8;
  }
  return m;
}

__device__ __forceinline__ void lds_fence_wave() {
  // order this wave's LDS writes before its LDS reads (and vice versa); no other wave shares
  // the region, so no s_barrier is needed.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int D>
__global__ __launch_bounds__(kWave) void dummy_rollout_wave(RolloutArgs a) {
  __shared__ __align__(16) unsigned char lds[kLdsBytes];
  const int lane = threadIdx.x;
  const int64_t row_base = (int64_t)blockIdx.x * kWave;
  const int64_t i = row_base + lane;  // < N: the host only takes this path when N % 64 == 0
  const int T = a.T;

  if (lane < 4) reinterpret_cast<uint32_t*>(lds + kLdsZero)[lane] = 0u;
  const LaneMap lm = make_lane_map(a.ring, lane, row_base);

  const int32_t eid = a.env_id[i];
  const int32_t local = eid - a.pc.env_off;
  const bool in_range = local >= 0 && local < a.pc.N;
  EnvRec rec{0, 0};
  if (in_range) rec = a.env[local];
  const bool dup = in_range && stamp_seen(a.stamp, local, a.stamp_val);
  const int32_t L = episode_len(a.pc.seed, eid);
  const int32_t max_steps = a.pc.max_episode_steps;

  // register queue of prefetched action rows
  int32_t q_penv[D];
  double2 q_la[D][3];
  auto fetch = [&](int d, int t) {
    const int tc = t < T ? t : T - 1;  // harmless re-read of the last step past the end
    q_penv[d] = a.players_env_id[(int64_t)tc * a.penv_stride + i];
    const double2* la = reinterpret_cast<const double2*>(a.list_action + ((int64_t)tc * a.la_stride + i) * 6);
    q_la[d][0] = la[0];
    q_la[d][1] = la[1];
    q_la[d][2] = la[2];
  };
#pragma unroll
  for (int d = 0; d < D; ++d) fetch(d, d);

  // rare-path accumulators, flushed after the loop (first_bad: the first step whose self-check failed)
  int32_t first_bad = -1;
  int32_t fin_len = 0, fin_cnt = 0;

  char* const ring = a.ring.ring;
  const uint64_t stride = a.ring.stride;
  int slot = a.slot0;

  for (int t0 = 0; t0 < T; t0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int t = t0 + d;  // < T: the host picks D so that T % D == 0
      {
        // consume the queue entry completely (compares only) BEFORE refilling it, so the new
        // loads can land in the same registers and no copy (= wait) is needed at the back edge
        const bool act_own = q_penv[d] == eid;
        const double x0 = q_la[d][0].x;
        const bool ok = (x0 == q_la[d][0].y) & (x0 == q_la[d][1].x) & (x0 == q_la[d][1].y) &
                        (x0 == q_la[d][2].x) & (x0 == q_la[d][2].y);
        int flags = (ok ? 1 : 0) | (act_own ? 2 : 0);
        // pin the compares above the refill: the empty asm needs `flags` and, with its memory
        // clobber, no load may be hoisted over it
        asm volatile("" : "+v"(flags) : : "memory");
        fetch(d, t + D);  // keep D steps of loads in flight
        const bool will_reset = rec.state >= L;
        const bool bad = ((flags & 1) == 0) & (!will_reset) & in_range;
        first_bad = (bad & (first_bad < 0)) ? t : first_bad;
        StepOut o = step_env(rec, eid, L, false, (flags >> 1) & 1, max_steps);
        if (!in_range) { o.elapsed = 0; o.step_type = 0; o.state = 0; o.raw1 = 0; o.done = 0; o.trunc = 0; }
        if (o.done) { fin_len = o.elapsed; fin_cnt += 1; }
        // stage the wave's 64 rows
        reinterpret_cast<int32_t*>(lds + kLdsEnvId)[lane] = eid;
        reinterpret_cast<int32_t*>(lds + kLdsElapsed)[lane] = o.elapsed;
        reinterpret_cast<int32_t*>(lds + kLdsStepType)[lane] = o.step_type;
        reinterpret_cast<float*>(lds + kLdsDiscount)[lane] = o.done ? 0.f : 1.f;
        reinterpret_cast<int2*>(lds + kLdsSR)[lane] = make_int2(o.state, o.raw1);
        lds[kLdsDone + lane] = o.done;
        lds[kLdsTrunc + lane] = o.trunc;
        lds_fence_wave();
        char* sb = ring + (uint64_t)slot * stride;
        const uint4 s1 = *reinterpret_cast<const uint4*>(lds + lm.src1);
        const uint4 s2 = *reinterpret_cast<const uint4*>(lds + lm.src2);
        int2 p[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = *reinterpret_cast<const int2*>(lds + lm.rsrc[c]);
        lds_fence_wave();  // the next step's staging writes must not pass these reads
        *reinterpret_cast<uint4*>(sb + lm.dst1) = s1;
        if (lane < 60) *reinterpret_cast<uint4*>(sb + lm.dst2) = s2;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          int4 v;
          v.x = lm.rkind[c] == 1u ? p[c].x : 0;
          v.y = lm.rkind[c] == 1u ? p[c].y : 0;
          v.z = lm.rkind[c] == 2u ? p[c].x : 0;
          v.w = lm.rkind[c] == 2u ? p[c].y : 0;
          if (c < 2 || lane < 32) *reinterpret_cast<int4*>(sb + lm.rdst[c]) = v;
        }
        slot = (slot + 1 == a.ring.depth) ? 0 : slot + 1;
      }
    }
  }
  if (in_range) {
    a.env[local] = rec;
    if (fin_cnt) {
      a.stats.fin_return[local] = 0.f;  // single player: every reward is float(-0) = +0.0f
      a.stats.fin_length[local] = fin_len;
      a.stats.ep_count[local] += fin_cnt;
    }
  } else {
    rollout_err(a, 0)->env_id_range = 1u;
  }
  if (dup) rollout_err(a, 0)->dup_env = 1u;
  if (first_bad >= 0) rollout_err(a, first_bad)->action_check = 1u;  // reported with the step that failed
}

// ---------------------------------------------------------------------------------------
// dummy_rollout_dma<D>: the same wave-autonomous loop, but the action rows travel
// HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into a D-deep
// wave-private ring, D steps ahead of their use. Nothing the compiler can see is in flight
// across iterations, so it inserts no vmcnt waits of its own; the one wait per step is the
// hand-counted s_waitcnt below. VMEM operations retire in issue order, so "all but the N
// youngest are done" is exact:
//   per step the wave issues G = 4 LDS-DMA loads (3 x 1 KiB list_action + 256 B
//   players.env_id) and S = 5 stores; the loads of step t were issued in step t-D, and
//   younger than them are D*S stores + (D-1)*G loads  =>  vmcnt(D*S + (D-1)*G).
//   The first D steps follow the prologue (D*G loads, no stores): step d waits for
//   vmcnt((D-1-d)*G + d*(G+S)).
// ---------------------------------------------------------------------------------------
constexpr int kDmaG = 4, kDmaS = 5;
constexpr int kRingLa = 3072, kRingSlot = 3328;  // 64 x 48 B list_action + 64 x 4 B players.env_id

template <int N>
__device__ __forceinline__ void vm_wait() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// One LDS-DMA instruction: every lane copies BYTES from its `gsrc` to lds_base + lane*BYTES.
// M0 carries the wave-uniform LDS base and is compiler-reserved, so it is saved and restored
// inside the statement (cdna_hip_programming.md §5.7).
// Non-temporal policy on the once-read action rows (LDS-DMA loads) and on the once-written
// StateBuffer rows: measured 188 -> 179 us per 128-step launch at 65 536 envs. The build flags
// EP_ROLLOUT_NO_NT_LOADS / EP_ROLLOUT_NO_NT_STORES switch either half back to the default policy.
#ifndef EP_ROLLOUT_NO_NT_LOADS
#define EP_NT_SUFFIX " nt"
#else
#define EP_NT_SUFFIX ""
#endif
template <typename V>
__device__ __forceinline__ void ring_store(V* dst, V v) {
#ifndef EP_ROLLOUT_NO_NT_STORES
  static_assert(sizeof(V) == 16, "16-byte ring stores only");
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 raw;
  __builtin_memcpy(&raw, &v, 16);
  __builtin_nontemporal_store(raw, reinterpret_cast<u32x4*>(dst));
#else
  *dst = v;
#endif
}
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" EP_NT_SUFFIX "\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" EP_NT_SUFFIX "\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}

template <int D>
__global__ __launch_bounds__(kWave) void dummy_rollout_dma(RolloutArgs a) {
  __shared__ __align__(16) unsigned char lds[kLdsBytes + D * kRingSlot];
  const int lane = threadIdx.x;
  const int64_t row_base = (int64_t)blockIdx.x * kWave;
  const int64_t i = row_base + lane;
  const int T = a.T;
  unsigned char* const ringq = lds + kLdsBytes;
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ringq);

  if (lane < 4) reinterpret_cast<uint32_t*>(lds + kLdsZero)[lane] = 0u;
  const LaneMap lm = make_lane_map(a.ring, lane, row_base);

  int32_t eid_ld = a.env_id[i];
  // make every compiler-visible load land NOW: once the DMA prologue is issued, a compiler
  // wait for a still-pending ordinary load would be a vmcnt(0) that drains the whole ring
  asm volatile("" : "+v"(eid_ld) : : "memory");
  const int32_t eid = eid_ld;
  const int32_t local = eid - a.pc.env_off;
  const bool in_range = local >= 0 && local < a.pc.N;
  EnvRec rec{0, 0};
  if (in_range) rec = a.env[local];
  uint32_t dup_w = (in_range && stamp_seen(a.stamp, local, a.stamp_val)) ? 1u : 0u;
  asm volatile("" : "+v"(rec.state), "+v"(rec.cur_step), "+v"(dup_w) : : "memory");
  const bool dup = dup_w != 0;
  const int32_t L = episode_len(a.pc.seed, eid);
  const int32_t max_steps = a.pc.max_episode_steps;

  // lane's source addresses inside a step's action block
  const char* const la_lane = reinterpret_cast<const char*>(a.list_action) + row_base * 48 + lane * 16;
  const char* const pe_lane = reinterpret_cast<const char*>(a.players_env_id) + row_base * 4 + lane * 4;
  const int64_t la_step = a.la_stride * 48, pe_step = a.penv_stride * 4;

  auto fetch = [&](int d, int t) {
    const int tc = t < T ? t : T - 1;  // past the end: re-read the last step (never consumed)
    const char* g = la_lane + (int64_t)tc * la_step;
    const uint32_t l = ring_lds + (uint32_t)d * kRingSlot;
    dma16(g, l);
    dma16(g + 1024, l + 1024);
    dma16(g + 2048, l + 2048);
    dma4(pe_lane + (int64_t)tc * pe_step, l + kRingLa);
  };

  int32_t first_bad = -1;  // the first step whose self-check failed for this lane
  int32_t fin_len = 0, fin_cnt = 0;
  char* const ring = a.ring.ring;
  const uint64_t stride = a.ring.stride;
  int slot = a.slot0;

  // nothing older than the prologue may still be counted by the hand-counted waits below
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
  for (int d = 0; d < D; ++d) fetch(d, d);

  auto step = [&](auto wait_tag, int d, int t) {
    constexpr int WAIT = decltype(wait_tag)::value;
    vm_wait<WAIT>();
    const unsigned char* q = ringq + d * kRingSlot;
    const double2* la = reinterpret_cast<const double2*>(q + lane * 48);
    const double2 v0 = la[0], v1 = la[1], v2 = la[2];
    const int32_t penv = reinterpret_cast<const int32_t*>(q + kRingLa)[lane];
    const bool ok = (v0.x == v0.y) & (v0.x == v1.x) & (v0.x == v1.y) & (v0.x == v2.x) & (v0.x == v2.y);
    int flags = (ok ? 1 : 0) | (penv == eid ? 2 : 0);
    // the ring slot is dead once `flags` exists; only then may the DMA for step t+D overwrite it
    asm volatile("" : "+v"(flags) : : "memory");
    fetch(d, t + D);
    const bool will_reset = rec.state >= L;
    const bool bad = ((flags & 1) == 0) & (!will_reset) & in_range;
    first_bad = (bad & (first_bad < 0)) ? t : first_bad;
    StepOut o = step_env(rec, eid, L, false, (flags >> 1) & 1, max_steps);
    if (!in_range) { o.elapsed = 0; o.step_type = 0; o.state = 0; o.raw1 = 0; o.done = 0; o.trunc = 0; }
    if (o.done) { fin_len = o.elapsed; fin_cnt += 1; }
    reinterpret_cast<int32_t*>(lds + kLdsEnvId)[lane] = eid;
    reinterpret_cast<int32_t*>(lds + kLdsElapsed)[lane] = o.elapsed;
    reinterpret_cast<int32_t*>(lds + kLdsStepType)[lane] = o.step_type;
    reinterpret_cast<float*>(lds + kLdsDiscount)[lane] = o.done ? 0.f : 1.f;
    reinterpret_cast<int2*>(lds + kLdsSR)[lane] = make_int2(o.state, o.raw1);
    lds[kLdsDone + lane] = o.done;
    lds[kLdsTrunc + lane] = o.trunc;
    lds_fence_wave();
    char* sb = ring + (uint64_t)slot * stride;
    const uint4 s1 = *reinterpret_cast<const uint4*>(lds + lm.src1);
    const uint4 s2 = *reinterpret_cast<const uint4*>(lds + lm.src2);
    int2 p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = *reinterpret_cast<const int2*>(lds + lm.rsrc[c]);
    lds_fence_wave();
    ring_store(reinterpret_cast<uint4*>(sb + lm.dst1), s1);
    if (lane < 60) ring_store(reinterpret_cast<uint4*>(sb + lm.dst2), s2);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int4 v;
      v.x = lm.rkind[c] == 1u ? p[c].x : 0;
      v.y = lm.rkind[c] == 1u ? p[c].y : 0;
      v.z = lm.rkind[c] == 2u ? p[c].x : 0;
      v.w = lm.rkind[c] == 2u ? p[c].y : 0;
      if (c < 2 || lane < 32) ring_store(reinterpret_cast<int4*>(sb + lm.rdst[c]), v);
    }
    slot = (slot + 1 == a.ring.depth) ? 0 : slot + 1;
  };

  // first group: behind the prologue
  {
    auto first = [&](auto dc) {
      constexpr int d = decltype(dc)::value;
      if (d < T) step(std::integral_constant<int, (D - 1 - d) * kDmaG + d * (kDmaG + kDmaS)>{}, d, d);
    };
    first(std::integral_constant<int, 0>{});
    if constexpr (D > 1) first(std::integral_constant<int, 1>{});
    if constexpr (D > 2) first(std::integral_constant<int, 2>{});
    if constexpr (D > 3) first(std::integral_constant<int, 3>{});
    if constexpr (D > 4) first(std::integral_constant<int, 4>{});
    if constexpr (D > 5) first(std::integral_constant<int, 5>{});
  }
  // steady state
  constexpr int kSteady = D * kDmaS + (D - 1) * kDmaG;
  for (int t0 = D; t0 < T; t0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (t0 + d < T) step(std::integral_constant<int, kSteady>{}, d, t0 + d);
    }
  }
  // LDS-DMA still in flight past the end (the clamped re-reads) must land before the LDS is released
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
  if (in_range) {
    a.env[local] = rec;
    if (fin_cnt) {
      a.stats.fin_return[local] = 0.f;
      a.stats.fin_length[local] = fin_len;
      a.stats.ep_count[local] += fin_cnt;
    }
  } else {
    rollout_err(a, 0)->env_id_range = 1u;
  }
  if (dup) rollout_err(a, 0)->dup_env = 1u;
  if (first_bad >= 0) rollout_err(a, first_bad)->action_check = 1u;  // reported with the step that failed
}

}  // namespace epk
