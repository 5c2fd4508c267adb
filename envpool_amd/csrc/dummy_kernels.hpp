// dummy_kernels.hpp — gfx950 kernels for the Dummy-v0 lockstep reset/step +
// StateBuffer observation-assembly path.
//
// What is computed (one thread = one env-step) follows, bit for bit:
//   worker loop           envpool2/core/async_envpool.h:116-129 (auto-reset one step after done)
//   Env::PreProcess       envpool2/core/env.h:181-189
//   DummyEnv::Reset/Step  envpool2/dummy/dummy_envpool.h:137-164, 173-210, 215
//   Env::Allocate         envpool2/core/env.h:196-222 (the 8 common fields)
//   StateBuffer::Allocate envpool2/core/state_buffer.h:79-112 (row assignment; here the row is
//                         computed by the host mirror of the counters and passed as row0)
// How it is computed is MI355X-first: every StateBuffer key is its own SoA array in the
// ring slot; a 256-env workgroup stages its results in LDS and writes each key's tile with
// 16-byte-per-lane coalesced stores (a 4-byte key's 256-row tile is ONE 1-KiB wave
// instruction); obs:raw rows are written whole, INCLUDING the zero columns, so the
// reference's per-batch zero-fill of a fresh StateBuffer (state_buffer_queue.h:84-94) is
// fused into the producing store and never touches HBM a second time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace epk {

constexpr int kBlock = 256;  // envs per workgroup (4 wave64)

// Device pointers of one StateBuffer slot (all keys SoA, each 256-B aligned).
struct SlotPtrs {
  int32_t* env_id;       // info:env_id          [B]
  int32_t* penv;         // info:players.env_id  [B*P]
  int32_t* elapsed;      // elapsed_step         [B]
  uint8_t* done;         // done                 [B]
  float* reward;         // reward               [B*P]
  float* discount;       // discount             [B*P]
  int32_t* step_type;    // step_type            [B]
  uint8_t* trunc;        // trunc                [B]
  int32_t* raw;          // obs:raw              [B*P, S]
  uint8_t* pdone;        // info:players.done    [B*P]
  int32_t* pid;          // info:players.id      [B*P]
};

// Byte offsets of the keys inside a slot; slot s lives at ring + s*stride.
struct RingLayout {
  char* ring;
  uint64_t stride;
  uint32_t off[11];
  int32_t depth;
};

__device__ __forceinline__ SlotPtrs slot_ptrs(const RingLayout& r, int slot) {
  char* b = r.ring + (uint64_t)slot * r.stride;
  SlotPtrs s;
  s.env_id = (int32_t*)(b + r.off[0]);
  s.penv = (int32_t*)(b + r.off[1]);
  s.elapsed = (int32_t*)(b + r.off[2]);
  s.done = (uint8_t*)(b + r.off[3]);
  s.reward = (float*)(b + r.off[4]);
  s.discount = (float*)(b + r.off[5]);
  s.step_type = (int32_t*)(b + r.off[6]);
  s.trunc = (uint8_t*)(b + r.off[7]);
  s.raw = (int32_t*)(b + r.off[8]);
  s.pdone = (uint8_t*)(b + r.off[9]);
  s.pid = (int32_t*)(b + r.off[10]);
  return s;
}

// Device-resident per-env state (the whole DummyEnv + Env bookkeeping: 8 bytes).
struct EnvRec {
  int32_t state;     // DummyEnv::state_         dummy_envpool.h:118
  int32_t cur_step;  // Env::current_step_{-1}   env.h:71
};

// Episode statistics (new; SURVEY.md §5 "Metrics"): touched only when an episode ends or a
// reward is non-zero, so the single-player hot path (reward == +0.0f always) pays nothing.
struct StatPtrs {
  float* ep_return;     // running return of the current episode
  float* fin_return;    // return of the last finished episode
  int32_t* fin_length;  // elapsed_step at its done
  int32_t* ep_count;    // finished episodes
};

// Error words live in host-mapped pinned memory: written (rarely) by plain stores, read by the
// host after the slot's ready event.
struct ErrWords {
  uint32_t action_check;  // list_action lanes differ (dummy_envpool.h:189-192 CHECK_EQ)
  uint32_t env_id_range;  // env_id outside this pool's shard
  uint32_t dup_env;       // one launch named an env twice: its two items raced (device-memory sends only;
                          // host-memory sends are split on the host so that such items run in order)
  uint32_t pad;
};

// "Written exactly once" check of a device-memory send: every env carries the stamp
// of the last launch that stepped it; an item that finds its own launch's stamp already there is
// the second item of that env in one launch. One L2 atomic per item, no other ordering needed.
__device__ __forceinline__ bool stamp_seen(uint32_t* stamp, int32_t local, uint32_t val) {
  return atomicExch(&stamp[local], val) == val;
}

struct PoolConst {
  int32_t N;        // local envs
  int32_t env_off;  // global id of local env 0
  int32_t seed;
  int32_t max_episode_steps;
  int32_t S;        // state_num
};

// Result of one env-step, staged through LDS.
struct StepOut {
  int32_t env_id, elapsed, step_type, state, raw1;
  uint8_t done, trunc;
};

// env.h:94 + dummy_envpool.h:126-128: seed_ = max(1, seed + GLOBAL env_id) (wrapping add).
__device__ __forceinline__ int32_t episode_len(int32_t seed, int32_t eid) {
  int32_t s = (int32_t)((uint32_t)seed + (uint32_t)eid);
  return s < 1 ? 1 : s;
}

// One env-step on registers. `rec` is updated in place.
__device__ __forceinline__ StepOut step_env(EnvRec& rec, int32_t eid, int32_t L, bool force_reset,
                                            int32_t action_num_if_step, int32_t max_steps) {
  bool reset = force_reset || (rec.state >= L);  // async_envpool.h:125
  if (reset) {
    rec.cur_step = 0;  // env.h:184-185
    rec.state = 0;     // dummy_envpool.h:138
  } else {
    rec.cur_step += 1;  // env.h:187
    rec.state += 1;     // dummy_envpool.h:174
  }
  bool done = rec.state >= L;  // dummy_envpool.h:215
  StepOut o;
  o.env_id = eid;
  o.elapsed = rec.cur_step;
  o.step_type = rec.cur_step == 0 ? 0 : (done ? 2 : 1);       // env.h:206
  o.done = done;
  o.trunc = done && (rec.cur_step >= max_steps);              // env.h:207
  o.state = rec.state;                                        // obs:raw[.,0]
  o.raw1 = reset ? 0 : action_num_if_step;                    // dummy_envpool.h:151 / :203
  return o;
}

// ---------------------------------------------------------------------------------------
// LDS-staged SoA tile writer for one workgroup's `nvalid` consecutive rows starting at
// global row `row_base` of slot `sp`. Single player: player rows == shared rows.
// ---------------------------------------------------------------------------------------
struct __align__(16) Stage {
  int32_t env_id[kBlock];
  int32_t elapsed[kBlock];
  int32_t step_type[kBlock];
  int32_t state[kBlock];
  int32_t raw1[kBlock];
  float discount[kBlock];
  uint8_t done[kBlock];
  uint8_t trunc[kBlock];
};

template <typename T>
__device__ __forceinline__ void store_tile_vec(T* __restrict__ gdst, const T* __restrict__ lsrc,
                                               int nvalid, int tid) {
  // gdst is 16-B aligned when row_base*sizeof(T) is; the caller guarantees that for the fast
  // path and otherwise uses store_tile_scalar.
  constexpr int per = 16 / sizeof(T);
  int nvec = nvalid / per;
  if (tid < nvec) {
    reinterpret_cast<uint4*>(gdst)[tid] = reinterpret_cast<const uint4*>(lsrc)[tid];
  }
  int tail0 = nvec * per;
  if (tid < nvalid - tail0) gdst[tail0 + tid] = lsrc[tail0 + tid];
}

template <typename T>
__device__ __forceinline__ void store_tile_scalar(T* __restrict__ gdst, const T* __restrict__ lsrc,
                                                  int nvalid, int tid) {
  if (tid < nvalid) gdst[tid] = lsrc[tid];
}

// obs:raw tile: nvalid rows x S ints, row r = {state[r], raw1[r], 0, ...}.
template <int S_CT>
__device__ __forceinline__ void store_raw_tile(int32_t* __restrict__ gdst, const Stage& st,
                                               int nvalid, int S_rt, int tid, bool aligned16) {
  const int S = S_CT > 0 ? S_CT : S_rt;
  const int total = nvalid * S;
  if (aligned16) {
    const int nvec = total >> 2;
    for (int q = tid; q < nvec; q += kBlock) {
      int e = q << 2;
      int r = e / S;
      int c = e - r * S;
      int v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = c == 0 ? st.state[r] : (c == 1 ? st.raw1[r] : 0);
        if (++c == S) { c = 0; ++r; }
      }
      reinterpret_cast<int4*>(gdst)[q] = make_int4(v[0], v[1], v[2], v[3]);
    }
    for (int e = (nvec << 2) + tid; e < total; e += kBlock) {
      int r = e / S;
      int c = e - r * S;
      gdst[e] = c == 0 ? st.state[r] : (c == 1 ? st.raw1[r] : 0);
    }
  } else {
    for (int e = tid; e < total; e += kBlock) {
      int r = e / S;
      int c = e - r * S;
      gdst[e] = c == 0 ? st.state[r] : (c == 1 ? st.raw1[r] : 0);
    }
  }
}

template <int S_CT>
__device__ __forceinline__ void write_tiles_sp(const SlotPtrs& sp, int64_t row_base, int nvalid,
                                               int S_rt, const Stage& st, int tid, int host_keys = 0) {
  // Four waves share the eleven key tiles; the branch is wave-uniform.
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const bool al4 = (row_base & 3) == 0;   // 4-byte keys 16-B aligned
  const bool al16 = (row_base & 15) == 0; // 1-byte keys 16-B aligned
  // 4-byte keys: 256 rows = 1 KiB = one 16-B/lane wave instruction each.
  auto put4 = [&](auto* g, const auto* l) {
    if (al4) store_tile_vec(g + row_base, l, nvalid, lane);
    else for (int t = lane; t < nvalid; t += 64) g[row_base + t] = l[t];
  };
  auto put1 = [&](uint8_t* g, const uint8_t* l) {
    if (al16) store_tile_vec(g + row_base, l, nvalid, lane);
    else for (int t = lane; t < nvalid; t += 64) g[row_base + t] = l[t];
  };
  if (wave == 0) {
    if (!(host_keys & 2)) {  // (host_keys bit 1: the sender copies the ids it sent into these two itself)
      put4(sp.env_id, st.env_id);
      put4(sp.penv, st.env_id);  // env.h:210-213: info:players.env_id[0] = env_id
    }
    put1(sp.done, st.done);
  } else if (wave == 1) {
    put4(sp.elapsed, st.elapsed);
    put4(sp.step_type, st.step_type);
    put1(sp.trunc, st.trunc);
  } else if (wave == 2) {
    put4(sp.discount, st.discount);
    put1(sp.pdone, st.done);  // dummy_envpool.h:149/:201 players.done = IsDone()
  }
  // constant-valued keys and obs:raw are spread over all four waves. (host_keys bit 0: the rows go to host
  // memory over PCIe and the host zeroes these two arrays itself: 8 of 72 bytes per env less to cross.)
  if (!(host_keys & 1)) {
    // reward[0] = float(-0) = +0.0f, players.id[0] = 0 (dummy_envpool.h:148,152 / :200,204)
    const uint4 z = make_uint4(0, 0, 0, 0);
    uint32_t* rw = reinterpret_cast<uint32_t*>(sp.reward) + row_base;
    uint32_t* pi = reinterpret_cast<uint32_t*>(sp.pid) + row_base;
    if (al4) {
      int nvec = nvalid >> 2;
      if (wave == 3) {
        if (lane < nvec) reinterpret_cast<uint4*>(rw)[lane] = z;
        if (lane < nvalid - (nvec << 2)) rw[(nvec << 2) + lane] = 0u;
      } else if (wave == 2) {
        if (lane < nvec) reinterpret_cast<uint4*>(pi)[lane] = z;
        if (lane < nvalid - (nvec << 2)) pi[(nvec << 2) + lane] = 0u;
      }
    } else {
      if (wave == 3) for (int t = lane; t < nvalid; t += 64) rw[t] = 0u;
      if (wave == 2) for (int t = lane; t < nvalid; t += 64) pi[t] = 0u;
    }
  }
  const int S = S_CT > 0 ? S_CT : S_rt;
  const bool raw_al = ((row_base * S) & 3) == 0;
  store_raw_tile<S_CT>(sp.raw + row_base * S, st, nvalid, S_rt, tid, raw_al);
}

// ---------------------------------------------------------------------------------------
// Single-player step / reset of `n` envs (one AsyncEnvPool::SendImpl or ::Reset call, or the
// part of it that lands in one ring slot). Item i -> output row row0 + i.
// ---------------------------------------------------------------------------------------
struct StepArgs {
  const int32_t* env_id;          // [n] global ids (action key 0)
  const int32_t* players_env_id;  // [n] (action key 1; single player => row i belongs to item i)
  const double* list_action;      // [n,6] (action key 2); null: the lanes were compared on the host, see la_ok
  const uint32_t* la_ok;          // bit la_bit0 + i: the six list_action lanes of item i are equal (host mode)
  EnvRec* env;
  StatPtrs stats;
  SlotPtrs slot;
  ErrWords* err;
  uint32_t* stamp;     // [N] per-env launch stamps, or null (host-validated send: no check needed)
  uint32_t stamp_val;  // this launch's stamp (never 0)
  PoolConst pc;
  int32_t n;
  int32_t row0;
  int32_t force_reset;
  int32_t la_bit0;
  uint8_t* term;  // [B] or null: gymnasium's terminated = done & ~trunc of each row, beside the slot (ep_pool_terminated_of)
  int32_t host_keys;  // rows that go to host memory: bit 0 - reward / info:players.id (always zero), bit 1 - info:env_id /
                      // info:players.env_id (the ids that were sent) are written by the sending thread, not over PCIe
};

// DEV: the device-memory step (not a reset, list_action rows on the device, duplicate stamps on):
// those three are compile-time facts there, so that the loads below are ONE straight line of code.
// With them as run-time (scalar) branches the compiler cannot keep the loads of different blocks in
// flight together and waits at every merge: five dependent round trips, 4.4 us for 65 536 envs.
template <int S_CT, bool DEV>
__global__ __launch_bounds__(kBlock) void dummy_step_sp(StepArgs a) {
  __shared__ Stage st;
  const int tid = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * kBlock + tid;
  const int blk_valid = min((int64_t)kBlock, (int64_t)a.n - (int64_t)blockIdx.x * kBlock);
  uint32_t old_stamp = 0u;  // (stamps are never 0: "not seen")
  bool check_stamp = false;
  if (i < a.n) {
    const int32_t eid = a.env_id[i];
    const int32_t local = eid - a.pc.env_off;
    StepOut o;
    if (local >= 0 && local < a.pc.N) {
      // Everything the step needs from memory is requested HERE, together, one round trip after
      // env_id: the env record, the episode return so far (needed only by an env that finishes, but
      // a load inside that branch is a round trip of its own for the whole wave), the action rows and
      // the duplicate stamp's exchange. The compiler had them as five dependent round trips (record;
      // actions; return; stamp, waited for on the spot; episode count), 4.4 us for 65 536 envs.
      EnvRec rec = a.env[local];
      const float ep_ret = a.stats.ep_return[local];
      int32_t action_num = 0;
      bool ok = true;
      if constexpr (DEV) {
        const int32_t penv = a.players_env_id[i];
        const double2* la = reinterpret_cast<const double2*>(a.list_action + i * 6);
        const double2 v0 = la[0], v1 = la[1], v2 = la[2];
        // issued last, looked at last (end of the kernel): returns come back in issue order, so nothing
        // that is needed early may be requested behind this exchange
        old_stamp = atomicExch(&a.stamp[local], a.stamp_val);
        check_stamp = true;
        ok = (v0.x == v0.y) & (v0.x == v1.x) & (v0.x == v1.y) & (v0.x == v2.x) & (v0.x == v2.y);
        action_num = (penv == eid) ? 1 : 0;
      } else if (!a.force_reset) {
        // ParseAction single-player branch (env.h:115-123) + dummy_envpool.h:179-184
        const int32_t penv = a.players_env_id[i];
        // dummy_envpool.h:187-192: the six lanes must be equal
        if (a.list_action) {
          const double2* la = reinterpret_cast<const double2*>(a.list_action + i * 6);
          double2 v0 = la[0], v1 = la[1], v2 = la[2];
          ok = (v0.x == v0.y) & (v0.x == v1.x) & (v0.x == v1.y) & (v0.x == v2.x) & (v0.x == v2.y);
        } else {
          const int64_t j = (int64_t)a.la_bit0 + i;
          ok = ((a.la_ok[j >> 5] >> (j & 31)) & 1u) != 0;
        }
        // (a forced reset of an env named twice writes the same record twice: harmless, not flagged)
        if (a.stamp) {  // the answer is looked at when the rows are on their way (end of the kernel)
          old_stamp = atomicExch(&a.stamp[local], a.stamp_val);
          check_stamp = true;
        }
        action_num = (penv == eid) ? 1 : 0;
      }
      const int32_t L = episode_len(a.pc.seed, eid);
      // the reference only reaches the check on the Step branch, not on an auto-reset
      if (!ok && !(rec.state >= L)) a.err->action_check = 1u;
      o = step_env(rec, eid, L, DEV ? false : a.force_reset != 0, action_num, a.pc.max_episode_steps);
      a.env[local] = rec;
      if (o.done) {  // episode statistics (rare path): stores and a fire-and-forget add, no load
        a.stats.fin_return[local] = ep_ret;
        a.stats.fin_length[local] = o.elapsed;
        atomicAdd(&a.stats.ep_count[local], 1);
      }
      if (o.elapsed == 0) a.stats.ep_return[local] = 0.f;
    } else {
      a.err->env_id_range = 1u;
      o.env_id = eid; o.elapsed = 0; o.step_type = 0; o.state = 0; o.raw1 = 0; o.done = 0; o.trunc = 0;
    }
    st.env_id[tid] = o.env_id;
    st.elapsed[tid] = o.elapsed;
    st.step_type[tid] = o.step_type;
    st.state[tid] = o.state;
    st.raw1[tid] = o.raw1;
    st.discount[tid] = o.done ? 0.f : 1.f;  // env.h:202
    st.done[tid] = o.done;
    st.trunc[tid] = o.trunc;
    if (a.term) a.term[(int64_t)a.row0 + i] = (uint8_t)(o.done & (o.trunc ^ 1));  // (64 lanes, 64 consecutive bytes)
  }
  __syncthreads();
  write_tiles_sp<S_CT>(a.slot, (int64_t)a.row0 + (int64_t)blockIdx.x * kBlock, blk_valid, a.pc.S, st, tid, DEV ? 0 : a.host_keys);
  if (check_stamp && old_stamp == a.stamp_val) a.err->dup_env = 1u;  // this launch had stepped the env already
}

// ---------------------------------------------------------------------------------------
// Fused lockstep loop: T sync-mode steps of all N local envs in one launch. Step t reads
// its action rows at step stride `*_stride` (in rows) and writes ring slot (slot0+t)%depth.
// Env state stays in registers across the T steps.
// ---------------------------------------------------------------------------------------
struct RolloutArgs {
  const int32_t* env_id;          // [N]     the same env at a given position every step
  const int32_t* players_env_id;  // [T?][N]
  const double* list_action;      // [T?][N,6]
  int64_t penv_stride, la_stride;  // rows between steps (0 = same every step)
  EnvRec* env;
  StatPtrs stats;
  RingLayout ring;
  ErrWords* err;       // error words, one per batch in flight: step t reports into err[(err0 + t) % err_depth]
  int32_t err0, err_depth;
  uint32_t* stamp;     // [N] per-env launch stamps (checked once per launch: env_id is the same every step)
  uint32_t stamp_val;
  PoolConst pc;
  int32_t slot0;
  int32_t T;
};

__device__ __forceinline__ ErrWords* rollout_err(const RolloutArgs& a, int t) {
  int w = a.err0 + t;
  w -= (w >= a.err_depth) ? a.err_depth : 0;  // T <= ring depth <= err_depth
  return a.err + w;
}

template <int S_CT>
__global__ __launch_bounds__(kBlock) void dummy_rollout_sp(RolloutArgs a) {
  __shared__ Stage st[2];
  const int tid = threadIdx.x;
  const int N = a.pc.N;
  const int64_t i = (int64_t)blockIdx.x * kBlock + tid;
  const int blk_valid = min((int64_t)kBlock, (int64_t)N - (int64_t)blockIdx.x * kBlock);
  const bool valid = i < N;

  // prefetch step 0's action
  int32_t eid = 0, penv = 0;
  double2 v0 = make_double2(0, 0), v1 = v0, v2 = v0;
  if (valid) {
    eid = a.env_id[i];
    penv = a.players_env_id[i];
    const double2* la = reinterpret_cast<const double2*>(a.list_action + i * 6);
    v0 = la[0]; v1 = la[1]; v2 = la[2];
  }
  const int32_t local = eid - a.pc.env_off;
  const bool in_range = valid && local >= 0 && local < N;
  EnvRec rec{0, 0};
  if (in_range) rec = a.env[local];
  if (in_range && stamp_seen(a.stamp, local, a.stamp_val)) rollout_err(a, 0)->dup_env = 1u;
  const int32_t L = episode_len(a.pc.seed, eid);

  for (int t = 0; t < a.T; ++t) {
    // issue step t+1's loads before doing step t's work
    int32_t n_penv = penv;
    double2 n0 = v0, n1 = v1, n2 = v2;
    if (valid && t + 1 < a.T) {
      n_penv = a.players_env_id[(int64_t)(t + 1) * a.penv_stride + i];
      const double2* la =
          reinterpret_cast<const double2*>(a.list_action + ((int64_t)(t + 1) * a.la_stride + i) * 6);
      n0 = la[0]; n1 = la[1]; n2 = la[2];
    }
    Stage& s = st[t & 1];
    if (valid) {
      StepOut o;
      if (in_range) {
        bool will_reset = rec.state >= L;
        bool ok = (v0.x == v0.y) & (v0.x == v1.x) & (v0.x == v1.y) & (v0.x == v2.x) & (v0.x == v2.y);
        if (!ok && !will_reset) rollout_err(a, t)->action_check = 1u;
        o = step_env(rec, eid, L, false, (penv == eid) ? 1 : 0, a.pc.max_episode_steps);
        if (o.done) {
          a.stats.fin_return[local] = a.stats.ep_return[local];
          a.stats.fin_length[local] = o.elapsed;
          a.stats.ep_count[local] += 1;
        }
        if (o.elapsed == 0) a.stats.ep_return[local] = 0.f;
      } else {
        rollout_err(a, t)->env_id_range = 1u;
        o.env_id = eid; o.elapsed = 0; o.step_type = 0; o.state = 0; o.raw1 = 0; o.done = 0; o.trunc = 0;
      }
      s.env_id[tid] = o.env_id;
      s.elapsed[tid] = o.elapsed;
      s.step_type[tid] = o.step_type;
      s.state[tid] = o.state;
      s.raw1[tid] = o.raw1;
      s.discount[tid] = o.done ? 0.f : 1.f;
      s.done[tid] = o.done;
      s.trunc[tid] = o.trunc;
    }
    __syncthreads();  // one barrier per step: st[] is double-buffered
    int slot = a.slot0 + t;
    slot -= (slot >= a.ring.depth) ? a.ring.depth : 0;
    SlotPtrs sp = slot_ptrs(a.ring, slot);
    write_tiles_sp<S_CT>(sp, (int64_t)blockIdx.x * kBlock, blk_valid, a.pc.S, s, tid);
    // rotate the prefetched action in
    penv = n_penv; v0 = n0; v1 = n1; v2 = n2;
  }
  if (in_range) a.env[local] = rec;
}

// One consistent copy of the per-env episode statistics, taken between two launches on the
// step stream (feeds the optional RCCL all-gather; never on the step kernels' critical path).
__global__ void snapshot_stats(StatPtrs st, int n, float* __restrict__ fin_return,
                               int32_t* __restrict__ fin_length, int32_t* __restrict__ ep_count) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (fin_return) fin_return[i] = st.fin_return[i];
  if (fin_length) fin_length[i] = st.fin_length[i];
  if (ep_count) ep_count[i] = st.ep_count[i];
}

}  // namespace epk
