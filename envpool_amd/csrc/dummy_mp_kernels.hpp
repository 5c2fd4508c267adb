// dummy_mp_kernels.hpp — max_num_players > 1: the ragged player dimension of the StateBuffer
// on the device.
//
//   ParseAction multi-player branch   envpool2/core/env.h:124-159
//   num_players = state % (P-1) + 1   envpool2/dummy/dummy_envpool.h:139-140, 175-176
//   (player_offset, shared_offset)    envpool2/core/state_buffer.h:85-88 — the reference bumps
//       two packed counters with one fetch_add per env; here the shared offset is known on the
//       host and the player offset is an exclusive prefix sum over the items' num_players.
//
// Two launches per send:
//   mp_count_rows   players.env_id histogram (how many action rows each env owns, env.h:128-132)
//                   into epoch-tagged 64-bit counters, so nothing has to be cleared between sends;
//   mp_step_fused   256 items per workgroup: env step + shared-row keys, a workgroup scan of
//                   num_players, a single-pass decoupled look-back across workgroups (256 tiles
//                   per hop; tile ids are blockIdx when the whole launch fits on the device at
//                   once - every tile a tile waits for is then resident or will be without anyone
//                   else finishing first - and come from a ticket otherwise, so that a tile only
//                   ever waits for tiles that already run; the ticket costs ~3 us at 256 tiles), then
//                   the workgroup writes its contiguous block of player rows, one thread per row
//                   (one owner search per row, 16- and 8-byte stores for the obs:raw row). The last tile
//                   leaves the slot's player total in a host-mapped word: recv needs no copy.
// The scratch (counters, look-back granules, ticket) comes in as an argument: a pool brings its
// own allocation, ep_dummy_send_noalloc the library's static array.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dummy_kernels.hpp"

namespace epk {

typedef __attribute__((address_space(1))) unsigned long long gu64;
#define EP_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int kMpTile = 256;
constexpr uint32_t kMpFlagAgg = 1u, kMpFlagIncl = 2u;  // look-back granule: {epoch:32 | flag:2 | value:30}
constexpr uint32_t kMpValueMask = 0x3fffffffu;
constexpr unsigned kMpSpinLimit = 1u << 22;  // look-back polls before a tile gives up (and flags the batch)

struct MpScratch {
  unsigned long long* counts;  // [N] {epoch:32 | rows:32}
  unsigned long long* tiles;   // [ceil(max items per launch / 256)] look-back granules
  uint32_t* ticket;            // dynamic tile ids: tile = atomicAdd(ticket, 1) - ticket_base
  uint32_t* barrier;           // folded histogram: workgroups that have counted their rows, ever (never goes back); may be null
};

// One row of the histogram (the caller passes local = -1 for lanes past the end). All 64 lanes of
// the wave must call it together.
// Returns what the second atomic returned (0 for lanes that added nothing): a caller that must know
// its atomics have been PERFORMED before it goes on waits for that value; one that ignores it gets
// fire-and-forget atomics.
__device__ __forceinline__ unsigned long long mp_count_row(int32_t local, int32_t N, int lane, unsigned long long* counts, uint32_t epoch) {
  const bool valid = local >= 0 && local < N;  // rows of envs this pool does not own match nothing
  // rows of one env usually sit next to each other (they are the previous batch's
  // info:players.env_id): the first lane of a run adds the whole run, the others add nothing.
  // Two atomics on one address from neighbouring lanes serialise in the L2; at two rows per env
  // the histogram took 9 us instead of 2.7.
  const int32_t prev = __shfl_up(local, 1, 64);
  const bool head = valid && (lane == 0 || prev != local);
  const unsigned long long bounds = __ballot(head || !valid);
  if (!head) return 0ull;
  const unsigned long long above = lane == 63 ? 0ull : bounds & ~((2ull << lane) - 1ull);
  const int end = above ? __builtin_ctzll(above) : 64;
  // two single-shot atomics, no retry loop: the max moves a counter of an older send to
  // {epoch, 0} and leaves one of this send alone; then the rows are added. Epochs only grow between
  // clears of the array (the host clears it when its 32-bit epoch wraps), and any number of rows
  // may name the same env without the threads starving each other.
  unsigned long long* c = counts + local;
  atomicMax(c, (unsigned long long)epoch << 32);
  return atomicAdd(c, (unsigned long long)(end - lane));
}

__global__ void mp_count_rows(const int32_t* __restrict__ penv, int np_rows, int32_t env_off, int32_t N,
                              unsigned long long* counts, uint32_t epoch) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  int32_t local = -1;
  if (j < np_rows) local = penv[j] - env_off;
  (void)mp_count_row(local, N, threadIdx.x & 63, counts, epoch);
}

struct MpFusedArgs {
  const int32_t* env_id;          // [n] items of this launch
  const int32_t* players_env_id;  // [n] single-player only (row i belongs to item i); else unused
  const double* list_action;      // [n,6]; null on a forced reset, or when the lanes were compared on the host (la_ok)
  const uint32_t* la_ok;          // bit la_bit0 + i: the six list_action lanes of item i are equal (host mode); else null
  MpScratch sc;
  uint32_t count_epoch;   // epoch of the histogram of this send
  uint32_t tile_epoch;    // epoch of this launch's look-back granules (never 0)
  uint32_t ticket_base;
  EnvRec* env;
  StatPtrs stats;         // members may be null (no episode statistics kept)
  SlotPtrs slot;
  ErrWords* err;
  uint32_t* stamp;        // see StepArgs; null = no duplicate check
  uint32_t stamp_val;
  PoolConst pc;
  uint32_t* slot_players;  // device: player rows in the slot so far (read by tile 0 unless first_segment, written by the last tile)
  uint32_t* host_players;  // host-visible copy written by the last tile (may be null)
  int32_t n, row0, force_reset, P, first_segment, num_tiles;
  int32_t la_bit0;
  int32_t static_tiles;  // tile id = blockIdx.x (every tile of the launch fits on the device at once)
  // The histogram folded into this launch (fold_rows > 0; needs static_tiles and a launch that is the
  // WHOLE send): every workgroup counts its share of the send's action rows, then waits until all
  // num_tiles workgroups of this launch have (barrier word >= fold_base + num_tiles), and only then
  // reads its envs' counts. A grid-wide wait: every workgroup must be resident, which static_tiles
  // with room to spare guarantees (dummy_mp_host.inl).
  const int32_t* fold_penv;
  int32_t fold_rows;
  uint32_t fold_base;
};

__device__ __forceinline__ int mp_find_item(const int32_t* excl, int r) {
  // largest j in [0,256) with excl[j] <= r (valid items have np >= 1, so it owns row r)
  int lo = 0, hi = kMpTile;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int mid = (lo + hi) >> 1;
    if (excl[mid] <= r) lo = mid; else hi = mid;
  }
  return lo;
}

// DEV: a multi-player step from device memory with its histogram in a launch of its own (P > 1, not
// a reset, list_action rows on the device, duplicate stamps on, episode statistics kept, no fold) as
// a compile-time case, so that everything an item needs from memory is requested in ONE straight
// line after env_id (see dummy_step_sp in dummy_kernels.hpp: as run-time branches these were six
// dependent round trips).
template <bool DEV>
__global__ __launch_bounds__(kMpTile) void mp_step_fused(MpFusedArgs a) {
  __shared__ int32_t s_excl[kMpTile + 1];
  __shared__ int32_t s_eid[kMpTile], s_state[kMpTile], s_raw1[kMpTile];
  __shared__ uint8_t s_done[kMpTile];
  __shared__ int32_t s_wave_tot[kMpTile / 64];
  __shared__ uint32_t s_tile, s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_tile = a.static_tiles ? blockIdx.x : atomicAdd(a.sc.ticket, 1u) - a.ticket_base;
  __syncthreads();
  const int tile = (int)s_tile;
  const int64_t i = (int64_t)tile * kMpTile + tid;
  const bool valid = tile < a.num_tiles && i < a.n;

  // ---- what does not depend on the histogram: the item's env, its record, its list_action lanes --
  int32_t eid = 0, local = -1;
  bool in_range = false, la_equal = true;
  EnvRec rec;
  rec.state = 0; rec.cur_step = 0;
  float ep_ret = 0.f;
  unsigned long long dev_count = 0ull;
  uint32_t old_stamp = 0u;  // (stamps are never 0)
  if (valid) {
    eid = a.env_id[i];
    local = eid - a.pc.env_off;
    in_range = local >= 0 && local < a.pc.N;
    if (in_range) {
      rec = a.env[local];
      if constexpr (DEV) {
        ep_ret = a.stats.ep_return[local];
        dev_count = a.sc.counts[local];
        const double2* la = reinterpret_cast<const double2*>(a.list_action + i * 6);
        const double2 v0 = la[0], v1 = la[1], v2 = la[2];
        // issued last, looked at last (returns come back in issue order)
        old_stamp = atomicExch(&a.stamp[local], a.stamp_val);
        la_equal = (v0.x == v0.y) & (v0.x == v1.x) & (v0.x == v1.y) & (v0.x == v2.x) & (v0.x == v2.y);
      } else if (!a.force_reset) {
        if (a.list_action) {
          const double2* la = reinterpret_cast<const double2*>(a.list_action + i * 6);
          const double2 v0 = la[0], v1 = la[1], v2 = la[2];
          la_equal = (v0.x == v0.y) & (v0.x == v1.x) & (v0.x == v1.y) & (v0.x == v2.x) & (v0.x == v2.y);
        } else {
          const int64_t j = (int64_t)a.la_bit0 + i;
          la_equal = ((a.la_ok[j >> 5] >> (j & 31)) & 1u) != 0;
        }
      }
    }
  }

  // ---- the histogram, folded in (see MpFusedArgs): count, then wait for every workgroup's count ----
  if (!DEV && a.fold_rows > 0) {
    // No fences anywhere in this phase: an agent-scope release / acquire on this chip writes back /
    // invalidates the XCD's whole L2 (megabytes of the previous batch's rows: the fused kernel went from
    // 12 to 42 us with __threadfence() here). Everything that crosses workgroups is an agent-scope
    // atomic, performed past the per-XCD L2s: the counters, the arrival word, and the loads of both.
    unsigned seen = 0;
    for (int base = blockIdx.x * kMpTile; base < a.fold_rows; base += gridDim.x * kMpTile) {
      const int j = base + tid;
      int32_t row_local = -1;
      if (j < a.fold_rows) row_local = a.fold_penv[j] - a.pc.env_off;
      const unsigned long long old = mp_count_row(row_local, a.pc.N, lane, a.sc.counts, a.count_epoch);
      seen ^= (unsigned)old ^ (unsigned)(old >> 32);
    }
    asm volatile("" ::"v"(seen));  // the atomics' return values have arrived: they have been performed ...
    __syncthreads();               // ... and so have the whole workgroup's
    if (tid == 0) {
      __hip_atomic_fetch_add(a.sc.barrier, 1u, EP_RLX_AGENT);
      const uint64_t t0 = wall_clock64();
      unsigned polls = 0;
      while (__hip_atomic_load(a.sc.barrier, EP_RLX_AGENT) - a.fold_base < (uint32_t)a.num_tiles) {
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 1023u) == 0u && wall_clock64() - t0 > 5000000ull) {  // 50 ms at 100 MHz: cannot happen with every
          a.err->env_id_range = 1u;                                        // workgroup resident; never hang the GPU on a bug
          break;
        }
      }
    }
    __syncthreads();
  }

  // ---- env step + shared-row keys -------------------------------------------------------
  int np = 0;
  if (valid) {
    StepOut o;
    o.env_id = eid; o.elapsed = 0; o.step_type = 0; o.state = 0; o.raw1 = 0; o.done = 0; o.trunc = 0;
    np = 1;
    if (in_range) {
      const int32_t L = episode_len(a.pc.seed, eid);
      int32_t action_num = 0;
      if (DEV) {
        if (!(rec.state >= L)) {
          action_num = (uint32_t)(dev_count >> 32) == a.count_epoch ? (int32_t)(uint32_t)dev_count : 0;
          if (!la_equal) a.err->action_check = 1u;
        }
      } else if (!a.force_reset && !(rec.state >= L)) {
        if (a.P <= 1) {
          action_num = a.players_env_id[i] == eid ? 1 : 0;  // env.h:115-123 + dummy_envpool.h:179-184
        } else {
          // (folded: the counts were written by other workgroups of THIS launch - an agent-scope load, past this XCD's L2)
          const unsigned long long c = a.fold_rows > 0 ? __hip_atomic_load((gu64*)a.sc.counts + local, EP_RLX_AGENT) : a.sc.counts[local];
          action_num = (uint32_t)(c >> 32) == a.count_epoch ? (int32_t)(uint32_t)c : 0;
        }
        if (!la_equal) a.err->action_check = 1u;
      }
      if (!DEV && !a.force_reset && a.stamp && stamp_seen(a.stamp, local, a.stamp_val)) a.err->dup_env = 1u;
      o = step_env(rec, eid, L, DEV ? false : a.force_reset != 0, action_num, a.pc.max_episode_steps);
      a.env[local] = rec;
      np = a.P <= 1 ? 1 : rec.state % (a.P - 1) + 1;
      if (DEV || a.stats.ep_return) {
        // return of the env = sum of its players' rewards, reward[k] = float(-k)
        const float r = -0.5f * (float)(np - 1) * (float)np;
        float acc = (o.elapsed == 0) ? 0.f : (DEV ? ep_ret : a.stats.ep_return[local]);
        acc += r;
        a.stats.ep_return[local] = acc;
        if (o.done) {
          a.stats.fin_return[local] = acc;
          a.stats.fin_length[local] = o.elapsed;
          atomicAdd(&a.stats.ep_count[local], 1);  // (fire and forget: no load in this branch)
        }
      }
    } else {
      a.err->env_id_range = 1u;
    }
    s_eid[tid] = eid;
    s_state[tid] = o.state;
    s_raw1[tid] = o.raw1;
    s_done[tid] = o.done;
    const int64_t r = (int64_t)a.row0 + i;
    a.slot.env_id[r] = o.env_id;
    a.slot.elapsed[r] = o.elapsed;
    a.slot.step_type[r] = o.step_type;
    a.slot.done[r] = o.done;
    a.slot.trunc[r] = o.trunc;
  }

  // ---- workgroup exclusive scan of np -----------------------------------------------------
  int incl = np;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) s_wave_tot[wave] = incl;
  __syncthreads();
  int wprefix = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kMpTile / 64; ++w) {
    const int t = s_wave_tot[w];
    wprefix += w < wave ? t : 0;
    total += t;
  }
  s_excl[tid] = wprefix + incl - np;
  if (tid == 0) s_excl[kMpTile] = total;

  // ---- decoupled look-back (wave 0): player rows of all earlier tiles of this launch ---------
  if (wave == 0 && tile < a.num_tiles) {
    gu64* tiles = (gu64*)a.sc.tiles;
    const unsigned long long tag = (unsigned long long)a.tile_epoch << 32;
    uint32_t base = 0;
    if (tile == 0) {
      base = a.first_segment ? 0u : *a.slot_players;  // rows earlier sends left in this batch
      if (lane == 0)
        __hip_atomic_store(tiles, tag | ((unsigned long long)kMpFlagIncl << 30) | (base + (uint32_t)total), EP_RLX_AGENT);
    } else {
      if (lane == 0)
        __hip_atomic_store(tiles + tile, tag | ((unsigned long long)kMpFlagAgg << 30) | (uint32_t)total, EP_RLX_AGENT);
      int look = tile - 1;  // nearest tile not accounted for yet
      unsigned spins = 0;
      for (;;) {
        // one hop covers the 256 nearest tiles: lane l looks at tiles look-4l .. look-4l-3 (four
        // independent loads), walks them nearest first and stops at the first inclusive value
        // (state 1) or at the first tile that has not published yet (state 2)
        unsigned long long g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = look - 4 * lane - q;
          g[q] = t >= 0 ? __hip_atomic_load(tiles + t, EP_RLX_AGENT) : 0ull;
        }
        uint32_t part = 0;
        int state = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = look - 4 * lane - q;
          if (state == 0 && t >= 0) {
            const bool fresh = (uint32_t)(g[q] >> 32) == a.tile_epoch;
            const uint32_t flag = fresh ? (uint32_t)(g[q] >> 30) & 3u : 0u;
            if (flag == 0u) {
              state = 2;
            } else {
              part += (uint32_t)g[q] & kMpValueMask;
              if (flag == kMpFlagIncl) state = 1;
            }
          }
        }
        const unsigned long long m_incl = __ballot(state == 1);
        const unsigned long long m_wait = __ballot(state == 2);
        const int first_incl = m_incl ? __builtin_ctzll(m_incl) : 64;
        const int first_wait = m_wait ? __builtin_ctzll(m_wait) : 64;
        if (first_wait < first_incl) {  // a tile in front of the nearest inclusive one has not published yet
          if (++spins > kMpSpinLimit) {  // cannot happen with ticket order; never hang the GPU on a bug
            if (lane == 0) a.err->env_id_range = 1u;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        if (lane > first_incl) part = 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        base += part;
        if (first_incl < 64 || look < 256) break;
        look -= 256;
      }
      if (lane == 0)
        __hip_atomic_store(tiles + tile, tag | ((unsigned long long)kMpFlagIncl << 30) | (base + (uint32_t)total), EP_RLX_AGENT);
    }
    if (lane == 0) {
      s_base = base;
      if (tile == a.num_tiles - 1) {
        *a.slot_players = base + (uint32_t)total;
        if (a.host_players) *a.host_players = base + (uint32_t)total;
      }
    }
  }
  __syncthreads();
  if (tile >= a.num_tiles) return;
  const int64_t R0 = (int64_t)s_base;

  // ---- player rows: the tile's `total` rows are contiguous from R0; one thread per row -----------
  const int S = a.pc.S;
  for (int r = tid; r < total; r += kMpTile) {
    const int j = mp_find_item(s_excl, r);
    const int k = r - s_excl[j];
    const int64_t R = R0 + r;
    const uint8_t done = s_done[j];
    a.slot.penv[R] = s_eid[j];                               // env.h:210-213
    a.slot.pid[R] = k;                                       // dummy_envpool.h:148/:200
    a.slot.pdone[R] = done;                                  // :149/:201
    a.slot.reward[R] = (float)(-k);                          // :152/:204
    a.slot.discount[R] = (k == 0 && !done) ? 1.f : 0.f;      // env.h:202 hits element 0 only
    // obs:raw row {state, raw1, 0, ...} (:150-151/:202-203; the rest stays at the buffer's zero)
    int32_t* const row = a.slot.raw + R * S;
    const int32_t v0 = s_state[j], v1 = s_raw1[j];
    if (S == 10) {
      // 40-byte rows from a 16-byte aligned base: even rows start on 16 bytes, odd rows 8 past
      if ((R & 1) == 0) {
        *reinterpret_cast<int4*>(row) = make_int4(v0, v1, 0, 0);
        *reinterpret_cast<int4*>(row + 4) = make_int4(0, 0, 0, 0);
        *reinterpret_cast<int2*>(row + 8) = make_int2(0, 0);
      } else {
        *reinterpret_cast<int2*>(row) = make_int2(v0, v1);
        *reinterpret_cast<int4*>(row + 2) = make_int4(0, 0, 0, 0);
        *reinterpret_cast<int4*>(row + 6) = make_int4(0, 0, 0, 0);
      }
    } else if ((S & 1) == 0) {
      int2* const row2 = reinterpret_cast<int2*>(row);
      row2[0] = make_int2(v0, v1);
      for (int c = 1; c < (S >> 1); ++c) row2[c] = make_int2(0, 0);
    } else {
      row[0] = v0;
      row[1] = v1;
      for (int c = 2; c < S; ++c) row[c] = 0;
    }
  }
  if (DEV && in_range && old_stamp == a.stamp_val) a.err->dup_env = 1u;  // this launch had stepped the env already
}

}  // namespace epk
