// dummy_server.hpp — the resident step server behind ep_pool_step.
//
// One launch per step costs 6.9 us on this platform before the kernel does anything (empty
// kernel + completion event, tools/launch_floor.cpp): more than the 4.5 us step kernel itself.
// A resident kernel that waits for a doorbell answers without any launch: the host writes the
// step number into the doorbell and spins on the words the workgroups write back.
//
// Round 3 (tools/doorbell_floor.cpp, profiles/r03/doorbell_floor.txt): of round 2's 6.4 us per
// 65 536-env step, 2.2 us were the wake-up and the join themselves. Wake-up: the doorbell used to
// be host-mapped memory that ONE workgroup polled over PCIe and forwarded through a device flag
// (two dependent hops); it is now a word of fine-grained DEVICE memory that the host writes through
// the PCIe BAR (a posted write + sfence) and that EVERY workgroup polls locally - no leader, no
// forwarding hop (256 workgroups polling host memory directly is no alternative: 30-100 us). Join:
// the workgroups used to be counted in 8 shard counters and then in a second-level counter whose
// last arriver told the host; now the last arriver of each SHARD writes its own word of one
// host-mapped line and the host waits for the 8 words. Where device memory is not host-accessible
// (no large BAR) the round-2 wake-up is kept (ServerArgs::bell_dev == nullptr).
//
// What is computed per row is dummy_kernels.hpp's step_env (same reference citations). What
// differs is memory visibility: the kernel outlives many steps while other kernels (the policy
// that writes the next actions, the consumer of the rows) run between them, and the per-XCD L2s
// are not coherent. So every byte this kernel reads that somebody else may have written since
// its launch is read with sc1 (agent-scope) loads, and every byte it writes is stored sc1
// (write-through): action arrays, env records, statistics, StateBuffer rows. Completion is
// signalled after every wave's s_waitcnt vmcnt(0) and a workgroup barrier.
//
// When it leaves: on the quit bit, which the HOST sets - every other entry point of the pool, the
// pool's idle watchdog (pool.hip: a step cadence of g us keeps the server for about 10 g, at least
// 200 us and at most 50 ms, so that a caller's device-wide synchronisation is not held up for long)
// or ep_pool_destroy. The poll limit inside the kernel is only a last resort against a host that
// never comes back (about a second); in the direct wake-up each workgroup counts for itself, and a
// workgroup that gives up says so in ServerCtl::alive.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dummy_kernels.hpp"

namespace epk {

#define EP_RLX_AGENT_ __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define EP_RLX_SYS_ __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM

constexpr unsigned long long kServerQuit = 1ull << 32;  // doorbell / device flag: {step number : 32 | leave : 32}
constexpr int kServerShards = 8;       // fan-in counters (each on a 128-byte line of its own)
constexpr int kServerLineWords = 32;

struct ServerCtl {  // host-mapped; one 128-byte line per direction
  unsigned long long bell;  // host -> device (leader wake-up only): number of the last step requested (low word) | kServerQuit
  uint32_t pad0[30];
  uint32_t done[kServerShards];  // device -> host: per shard of workgroups, the number of the last step it finished
  uint32_t alive;   // device -> host: 0 once a workgroup has given up waiting (its poll limit ran out)
  uint32_t pad1[32 - kServerShards - 1];
};

struct ServerArgs {
  ServerCtl* ctl;                 // device alias of the control block
  const unsigned long long* bell_dev;  // fine-grained device memory written by the host through the BAR: the doorbell
                                  // every workgroup polls (same encoding as ServerCtl::bell); null: leader wake-up
  unsigned long long* dev_flag;   // device (leader wake-up): the step the leader has seen (low word) | kServerQuit: leave
  uint32_t* counters;             // device: kServerShards shard counters + the global one, a line each, zero at launch
  const int32_t* env_id;          // [N] action arrays, fixed for the life of this launch
  const int32_t* players_env_id;  // [N]
  const double* list_action;      // [N,6], 16-byte aligned
  EnvRec* env;
  StatPtrs stats;
  RingLayout ring;
  ErrWords* err;                  // device alias of the host-mapped error words
  int32_t err_depth;
  uint32_t* stamp;
  PoolConst pc;
  uint32_t first_step;            // the doorbell value of the first step this launch serves
  uint64_t bi0;                   // batch index of that step
  uint32_t idle_polls;
  int32_t tiles_per_wg;           // N / 256 tiles dealt round robin over the workgroups
  int32_t num_tiles;
  int32_t two_level_join;         // 1: shard counters, then one global counter whose last arriver writes done[0] (round 2's join)
  unsigned long long* trace;      // diagnostic (EP_SERVER_TRACE=1), else null: per workgroup 4 timestamps of its last
                                  // step (100 MHz wall clock): doorbell seen, rows staged, stores drained, counted in
  uint8_t* term;                  // [depth][term_stride]: terminated = done & ~trunc of every row, beside the ring (ep_pool_terminated_of)
  uint32_t term_stride;
};

__device__ __forceinline__ uint4 ld16_sc1(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16));
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, uint32_t off, uint4 v) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 16);
}

template <int S_CT>
__global__ __launch_bounds__(kBlock) void dummy_step_server(ServerArgs a) {
  __shared__ Stage st;
  __shared__ __align__(16) uint8_t s_term[kBlock];
  __shared__ uint32_t s_go;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int S = S_CT > 0 ? S_CT : a.pc.S;
  const uint32_t n_wg = gridDim.x;
  const uint32_t shard = blockIdx.x % kServerShards;
  const uint32_t shard_size = n_wg / kServerShards + (shard < n_wg % kServerShards ? 1u : 0u);
  const __amdgpu_buffer_rsrc_t la_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.list_action, 0, (int)((uint32_t)a.pc.N * 48u), 0x00020000);

  for (uint32_t k = 0;; ++k) {
    const uint32_t want = a.first_step + k;
    // ---- wait for the doorbell ---------------------------------------------------------
    if (tid == 0) {
      uint32_t go = 0, polls = 0;
      if (a.bell_dev) {  // direct: every workgroup reads the host's word in device memory
        for (;;) {
          const unsigned long long b = __hip_atomic_load(a.bell_dev, EP_RLX_SYS_);
          if (b & kServerQuit) break;
          if ((int32_t)((uint32_t)b - want) >= 0) { go = 1; break; }
          if (++polls > a.idle_polls) { __hip_atomic_store(&a.ctl->alive, 0u, EP_RLX_SYS_); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      } else if (blockIdx.x == 0) {
        for (;;) {
          const unsigned long long b = __hip_atomic_load(&a.ctl->bell, EP_RLX_SYS_);
          if (b & kServerQuit) break;
          if ((int32_t)((uint32_t)b - want) >= 0) { go = 1; break; }
          if (++polls > a.idle_polls) break;
          __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store(a.dev_flag, go ? (unsigned long long)want : kServerQuit, EP_RLX_AGENT_);
        if (!go) __hip_atomic_store(&a.ctl->alive, 0u, EP_RLX_SYS_);
      } else {
        for (;;) {
          const unsigned long long f = __hip_atomic_load(a.dev_flag, EP_RLX_AGENT_);
          if (f & kServerQuit) break;
          if ((int32_t)((uint32_t)f - want) >= 0) { go = 1; break; }
          if (++polls > 4u * a.idle_polls) break;  // (the leader is gone: never spin for ever)
          __builtin_amdgcn_s_sleep(1);
        }
      }
      s_go = go;
    }
    __syncthreads();
    if (!s_go) return;
    if (a.trace && tid == 0) a.trace[blockIdx.x * 4 + 0] = wall_clock64();

    // ---- the step ------------------------------------------------------------------------
    const uint64_t bi = a.bi0 + k;
    const int slot = (int)(bi % (uint64_t)a.ring.depth);
    ErrWords* const err = a.err + (int)(bi % (uint64_t)a.err_depth);
    char* const slot_base = a.ring.ring + (uint64_t)slot * a.ring.stride;
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)slot_base, 0, (int)(uint32_t)a.ring.stride, 0x00020000);
    const __amdgpu_buffer_rsrc_t term_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.term + (uint64_t)slot * a.term_stride), 0, (int)a.term_stride, 0x00020000);
    for (int j = 0; j < a.tiles_per_wg; ++j) {
      const int tile = (int)blockIdx.x + j * (int)n_wg;
      if (tile >= a.num_tiles) break;
      const int i = tile * kBlock + tid;
      const int32_t eid = __hip_atomic_load(a.env_id + i, EP_RLX_AGENT_);
      const int32_t penv = __hip_atomic_load(a.players_env_id + i, EP_RLX_AGENT_);
      const uint4 l0 = ld16_sc1(la_rsrc, (uint32_t)i * 48u), l1 = ld16_sc1(la_rsrc, (uint32_t)i * 48u + 16u),
                  l2 = ld16_sc1(la_rsrc, (uint32_t)i * 48u + 32u);
      // Speculation: in a lockstep step row i almost always names env i of this pool, so that env's
      // record is requested together with the action row instead of after env_id has arrived (one
      // memory round trip of the step's critical path instead of two: about 0.7 us of 3.4 on the
      // device side, tools/host_latency.cpp with EP_SERVER_TRACE=1). A row that names another env
      // takes the dependent load.
      const unsigned long long spec_rw = __hip_atomic_load(reinterpret_cast<unsigned long long*>(a.env + i), EP_RLX_AGENT_);
      const int32_t local = eid - a.pc.env_off;
      StepOut o;
      if (local >= 0 && local < a.pc.N) {
        unsigned long long* recp = reinterpret_cast<unsigned long long*>(a.env + local);
        unsigned long long rw = spec_rw;
        bool dup = false;
        if (local != i) {
          rw = __hip_atomic_load(recp, EP_RLX_AGENT_);
          // "written exactly once": rows that name their own position cannot collide with each other
          // and exchange no stamp; a row that names another env collides either with THAT env's own
          // row (if row `local` names `local`) or with another displaced row (the stamp exchange)
          dup = __hip_atomic_load(a.env_id + local, EP_RLX_AGENT_) == eid || stamp_seen(a.stamp, local, want | 0x40000000u);
        }
        EnvRec rec{(int32_t)(uint32_t)rw, (int32_t)(uint32_t)(rw >> 32)};
        const double2 v0 = __builtin_bit_cast(double2, l0), v1 = __builtin_bit_cast(double2, l1),
                      v2 = __builtin_bit_cast(double2, l2);
        const bool ok = (v0.x == v0.y) & (v0.x == v1.x) & (v0.x == v1.y) & (v0.x == v2.x) & (v0.x == v2.y);
        const int32_t L = episode_len(a.pc.seed, eid);
        if (!ok && !(rec.state >= L)) __hip_atomic_store(&err->action_check, 1u, EP_RLX_SYS_);
        if (dup) __hip_atomic_store(&err->dup_env, 1u, EP_RLX_SYS_);
        o = step_env(rec, eid, L, false, penv == eid ? 1 : 0, a.pc.max_episode_steps);
        __hip_atomic_store(recp, (unsigned long long)(uint32_t)rec.state | ((unsigned long long)(uint32_t)rec.cur_step << 32),
                           EP_RLX_AGENT_);
        if (o.done) {  // episode statistics (rare; single player: every reward is +0.0f)
          __hip_atomic_store(a.stats.fin_return + local, __hip_atomic_load(a.stats.ep_return + local, EP_RLX_AGENT_), EP_RLX_AGENT_);
          __hip_atomic_store(a.stats.fin_length + local, o.elapsed, EP_RLX_AGENT_);
          __hip_atomic_fetch_add(a.stats.ep_count + local, 1, EP_RLX_AGENT_);
        }
        if (o.elapsed == 0) __hip_atomic_store(a.stats.ep_return + local, 0.f, EP_RLX_AGENT_);
      } else {
        __hip_atomic_store(&err->env_id_range, 1u, EP_RLX_SYS_);
        o.env_id = eid; o.elapsed = 0; o.step_type = 0; o.state = 0; o.raw1 = 0; o.done = 0; o.trunc = 0;
      }
      st.env_id[tid] = o.env_id;
      st.elapsed[tid] = o.elapsed;
      st.step_type[tid] = o.step_type;
      st.state[tid] = o.state;
      st.raw1[tid] = o.raw1;
      st.discount[tid] = o.done ? 0.f : 1.f;
      st.done[tid] = o.done;
      st.trunc[tid] = o.trunc;
      s_term[tid] = (uint8_t)(o.done & (o.trunc ^ 1));
      __syncthreads();
      if (a.trace && tid == 0 && j == 0) a.trace[blockIdx.x * 4 + 1] = wall_clock64();
      // the tile's rows: 256 consecutive rows from row tile*256, every key 16-byte aligned there.
      // 4-byte keys: 1 KiB = one 16-byte store per lane of ONE wave; 1-byte keys: 16 lanes.
      const uint32_t r4 = (uint32_t)tile * kBlock * 4u, r1 = (uint32_t)tile * kBlock;
      auto put4 = [&](int key, const void* l) {
        st16_sc1(out_rsrc, a.ring.off[key] + r4 + (uint32_t)lane * 16u, reinterpret_cast<const uint4*>(l)[lane]);
      };
      auto put1 = [&](int key, const uint8_t* l) {
        if (lane < 16) st16_sc1(out_rsrc, a.ring.off[key] + r1 + (uint32_t)lane * 16u, reinterpret_cast<const uint4*>(l)[lane]);
      };
      const uint4 z = make_uint4(0, 0, 0, 0);
      if (wave == 0) {
        put4(0, st.env_id);
        put4(1, st.env_id);  // env.h:210-213: info:players.env_id[0] = env_id
        put1(3, st.done);
      } else if (wave == 1) {
        put4(2, st.elapsed);
        put4(6, st.step_type);
        put1(7, st.trunc);
      } else if (wave == 2) {
        put4(5, st.discount);
        put1(9, st.done);    // dummy_envpool.h:149/:201 players.done = IsDone()
        st16_sc1(out_rsrc, a.ring.off[10] + r4 + (uint32_t)lane * 16u, z);  // players.id[0] = 0
      } else {
        st16_sc1(out_rsrc, a.ring.off[4] + r4 + (uint32_t)lane * 16u, z);   // reward[0] = float(-0) = +0.0f
        if (lane < 16) st16_sc1(term_rsrc, r1 + (uint32_t)lane * 16u, reinterpret_cast<const uint4*>(s_term)[lane]);
      }
      // obs:raw: 256 rows x S ints, row r = {state, raw1, 0, ...}; 256*S*4 bytes is a multiple of 16
      const int nvec = kBlock * S / 4;
      const uint32_t raw0 = a.ring.off[8] + (uint32_t)tile * kBlock * (uint32_t)S * 4u;
      for (int q = tid; q < nvec; q += kBlock) {
        int e = q << 2;
        int r = e / S;
        int c = e - r * S;
        uint32_t v[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          v[x] = (uint32_t)(c == 0 ? st.state[r] : (c == 1 ? st.raw1[r] : 0));
          if (++c == S) { c = 0; ++r; }
        }
        st16_sc1(out_rsrc, raw0 + (uint32_t)q * 16u, make_uint4(v[0], v[1], v[2], v[3]));
      }
      __syncthreads();  // the stage is reused by the next tile
    }
    // ---- completion: every wave's stores have left, then one lane counts the workgroup in; the
    // last workgroup of a shard tells the host (which waits for every shard's word) ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (a.trace) a.trace[blockIdx.x * 4 + 2] = wall_clock64();
      const uint32_t old = __hip_atomic_fetch_add(a.counters + shard * kServerLineWords, 1u, EP_RLX_AGENT_);
      if (old + 1 == (k + 1) * shard_size) {
        if (a.two_level_join) {  // one word for the host: the last shard to complete writes it
          const uint32_t n_shards = n_wg < (uint32_t)kServerShards ? n_wg : (uint32_t)kServerShards;
          const uint32_t g = __hip_atomic_fetch_add(a.counters + kServerShards * kServerLineWords, 1u, EP_RLX_AGENT_);
          if (g + 1 == (k + 1) * n_shards) __hip_atomic_store(&a.ctl->done[0], want, EP_RLX_SYS_);
        } else {
          __hip_atomic_store(&a.ctl->done[shard], want, EP_RLX_SYS_);
        }
      }
      if (a.trace) a.trace[blockIdx.x * 4 + 3] = wall_clock64();
    }
  }
}

}  // namespace epk
