// dummy_mp_kernels.hpp — max_num_players > 1: the ragged player dimension of the StateBuffer
// on the device.
//
//   ParseAction multi-player branch   envpool2/core/env.h:124-159
//   num_players = state % (P-1) + 1   envpool2/dummy/dummy_envpool.h:139-140, 175-176
//   (player_offset, shared_offset)    envpool2/core/state_buffer.h:85-88 — the reference bumps
//       two packed counters with one fetch_add per env; here the shared offset is known on the
//       host and the player offset is an exclusive prefix sum over the segment's num_players
//       (wave64 shuffles + one LDS pass), added to the slot's running player count.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dummy_kernels.hpp"

namespace epk {

// players.env_id histogram: how many action rows each env owns in this send (env.h:128-132).
__global__ void mp_count_rows(const int32_t* __restrict__ penv, int np_rows, int32_t env_off, int32_t N,
                              int32_t* __restrict__ counts, int32_t delta_or_zero) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= np_rows) return;
  int32_t local = penv[j] - env_off;
  if (local < 0 || local >= N) return;  // rows of envs this pool does not own match nothing
  if (delta_or_zero) atomicAdd(&counts[local], 1);
  else counts[local] = 0;  // second pass: leave the histogram clean for the next send
}

struct MpItem {  // per-item scratch carried from the step kernel to the player-row writer
  int32_t eid, state, raw1, np;
  int32_t done;
};

struct MpStepArgs {
  const int32_t* env_id;
  const double* list_action;
  const int32_t* counts;
  EnvRec* env;
  StatPtrs stats;
  SlotPtrs slot;
  ErrWords* err;
  PoolConst pc;
  MpItem* items;
  int32_t n, row0, force_reset, P;
};

// Step + shared-row keys (rows follow allocation order: order == -1 whenever P > 1,
// async_envpool.h:69,95).
__global__ __launch_bounds__(256) void mp_step(MpStepArgs a) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int32_t eid = a.env_id[i];
  const int32_t local = eid - a.pc.env_off;
  MpItem it{eid, 0, 0, 1, 0};
  StepOut o;
  o.env_id = eid; o.elapsed = 0; o.step_type = 0; o.state = 0; o.raw1 = 0; o.done = 0; o.trunc = 0;
  if (local >= 0 && local < a.pc.N) {
    EnvRec rec = a.env[local];
    const int32_t L = episode_len(a.pc.seed, eid);
    int32_t action_num = 0;
    if (!a.force_reset && !(rec.state >= L)) {
      action_num = a.counts[local];
      const double* la = a.list_action + (int64_t)i * 6;
      bool ok = true;
#pragma unroll
      for (int k = 1; k < 6; ++k) ok &= (la[0] == la[k]);
      if (!ok) a.err->action_check = 1u;
    }
    o = step_env(rec, eid, L, a.force_reset != 0, action_num, a.pc.max_episode_steps);
    a.env[local] = rec;
    it.np = rec.state % (a.P - 1) + 1;
    // return of the env = sum of its players' rewards, reward[k] = float(-k)
    float r = -0.5f * (float)(it.np - 1) * (float)it.np;
    float acc = (o.elapsed == 0) ? 0.f : a.stats.ep_return[local];
    acc += r;
    a.stats.ep_return[local] = acc;
    if (o.done) {
      a.stats.fin_return[local] = acc;
      a.stats.fin_length[local] = o.elapsed;
      a.stats.ep_count[local] += 1;
    }
  } else {
    a.err->env_id_range = 1u;
  }
  it.state = o.state; it.raw1 = o.raw1; it.done = o.done;
  a.items[i] = it;
  const int64_t r = (int64_t)a.row0 + i;
  a.slot.env_id[r] = o.env_id;
  a.slot.elapsed[r] = o.elapsed;
  a.slot.step_type[r] = o.step_type;
  a.slot.done[r] = o.done;
  a.slot.trunc[r] = o.trunc;
}

// Exclusive prefix sum of items[].np over one segment (single workgroup, wave64 shuffles),
// plus the hand-over of the slot's running player count.
__global__ __launch_bounds__(1024) void mp_scan(const MpItem* __restrict__ items, int n,
                                                int32_t* __restrict__ off, uint32_t* slot_players,
                                                int32_t* seg_base) {
  __shared__ int32_t wave_tot[16];
  __shared__ int32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + tid;
    int v = i < n ? items[i].np : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wprefix = 0;
    for (int w = 0; w < wave; ++w) wprefix += wave_tot[w];
    int c = carry;
    if (i < n) off[i] = c + wprefix + incl - v;
    __syncthreads();
    if (tid == 1023) carry = c + wprefix + incl;
    __syncthreads();
  }
  if (tid == 0) {
    uint32_t b = *slot_players;
    *seg_base = (int32_t)b;
    *slot_players = b + (uint32_t)carry;
  }
}

struct MpWriteArgs {
  const MpItem* items;
  const int32_t* off;
  const int32_t* seg_base;
  SlotPtrs slot;
  int32_t n, S;
};

// Player-row keys. One thread per item; an item owns np <= P consecutive rows.
__global__ __launch_bounds__(256) void mp_write_players(MpWriteArgs a) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const MpItem it = a.items[i];
  const int64_t r0 = (int64_t)(*a.seg_base) + a.off[i];
  for (int k = 0; k < it.np; ++k) {
    const int64_t r = r0 + k;
    a.slot.penv[r] = it.eid;                              // env.h:210-213
    a.slot.pid[r] = k;                                    // dummy_envpool.h:148/:200
    a.slot.pdone[r] = (uint8_t)it.done;                   // :149/:201
    a.slot.reward[r] = (float)(-k);                       // :152/:204
    a.slot.discount[r] = (k == 0 && !it.done) ? 1.f : 0.f;  // env.h:202 hits element 0 only
    int32_t* row = a.slot.raw + r * a.S;
    row[0] = it.state;
    row[1] = it.raw1;
    for (int c = 2; c < a.S; ++c) row[c] = 0;
  }
}

}  // namespace epk
