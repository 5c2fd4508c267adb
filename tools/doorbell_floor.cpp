// doorbell_floor.cpp — what the wake-up and the join around one resident-server step cost (GPU box,
// diagnostic; round 3's pricing of the per-step floor of ep_pool_step).
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/doorbell_floor.cpp -o tools/doorbell_floor
// A resident kernel of G workgroups x T threads serves R rounds: the host rings a doorbell, every
// workgroup copies its share of 9.4 MB (the bytes of one 65 536-env Dummy step; 0 = no work), the
// workgroups join, the host learns that the round is done. Variants:
//   wake   P0  workgroup 0 polls the doorbell in host-mapped memory and raises a device flag (sc1
//              store) that the others poll (sc1 loads)            [what dummy_step_server does]
//          P1  every workgroup polls the host-mapped doorbell itself
//          P2  the doorbell is a word of fine-grained DEVICE memory that the host writes through the
//              PCIe BAR (store + sfence: the mapping is write-combining); every workgroup polls it
//              locally (system-scope loads)
//          P3  that word polled by workgroup 0 only, which raises the device flag of P0
//          P4  P2 with 8 replicas of the word, 4 KiB apart (workgroup b polls replica b % 8)
//   join   C0  one device counter; the last arriver writes the host's done word
//          C1  8 shard counters (blockIdx % 8); each shard's last arriver writes ITS word of one
//              host line; the host waits for the 8 words   [no second-level join on the device]
//          C2  every workgroup writes a host word of its own; the host scans G words
// Every poll loop is bounded; on a host-side timeout the host opens all remaining rounds so that
// the kernel drains.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <immintrin.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double pct(std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; }

#define RLX_SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Args {
  unsigned* bell_host;   // device alias of the host-mapped doorbell
  unsigned* bell_dev;    // fine-grained device doorbell (P2)
  unsigned* done_host;   // device alias: [0] C0 word, [16..24) C1 shard words, [64..64+G) C2 words
  unsigned* dev_flag;    // device: P0 flag
  unsigned* counters;    // device: [0] C0 counter, [32*(1+s)] C1 shard counters
  const uint4* src;
  uint4* dst;
  unsigned rounds, max_polls;
  int wake, join, n16_per_wg;
};

__global__ __launch_bounds__(1024) void k_server(Args a) {
  __shared__ unsigned s_go;
  const unsigned G = gridDim.x, shard = blockIdx.x % 8u;
  const unsigned shard_size = G / 8u + (shard < G % 8u ? 1u : 0u);
  for (unsigned want = 1; want <= a.rounds; ++want) {
    if (threadIdx.x == 0) {
      unsigned polls = 0, go = 0;
      if (a.wake == 0) {
        if (blockIdx.x == 0) {
          for (;;) {
            if (__hip_atomic_load(a.bell_host, RLX_SYS) >= want) { go = 1; break; }
            if (++polls > a.max_polls) break;
            __builtin_amdgcn_s_sleep(1);
          }
          __hip_atomic_store(a.dev_flag, go ? want : 0xffffffffu, RLX_AGENT);
        } else {
          for (;;) {
            const unsigned f = __hip_atomic_load(a.dev_flag, RLX_AGENT);
            if (f == 0xffffffffu) break;
            if (f >= want) { go = 1; break; }
            if (++polls > 4u * a.max_polls) break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
      } else if (a.wake == 3) {
        if (blockIdx.x == 0) {
          for (;;) {
            if (__hip_atomic_load(a.bell_dev, RLX_SYS) >= want) { go = 1; break; }
            if (++polls > a.max_polls) break;
            __builtin_amdgcn_s_sleep(1);
          }
          __hip_atomic_store(a.dev_flag, go ? want : 0xffffffffu, RLX_AGENT);
        } else {
          for (;;) {
            const unsigned f = __hip_atomic_load(a.dev_flag, RLX_AGENT);
            if (f == 0xffffffffu) break;
            if (f >= want) { go = 1; break; }
            if (++polls > 4u * a.max_polls) break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
      } else {
        unsigned* bell = a.wake == 1 ? a.bell_host : a.wake == 4 ? a.bell_dev + 1024 * (blockIdx.x % 8u) : a.bell_dev;
        for (;;) {
          if (__hip_atomic_load(bell, RLX_SYS) >= want) { go = 1; break; }
          if (++polls > a.max_polls) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      s_go = go;
    }
    __syncthreads();
    if (!s_go) return;
    if (a.n16_per_wg) {
      const size_t base = (size_t)blockIdx.x * a.n16_per_wg;
      for (int i = threadIdx.x; i < a.n16_per_wg; i += blockDim.x) a.dst[base + i] = a.src[base + i];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (a.join == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.counters, 1u, RLX_AGENT);
        if (old == want * G - 1) __hip_atomic_store(a.done_host, want, RLX_SYS);
      } else if (a.join == 1) {
        const unsigned old = __hip_atomic_fetch_add(a.counters + 32 * (1 + shard), 1u, RLX_AGENT);
        if (old == want * shard_size - 1) __hip_atomic_store(a.done_host + 16 + shard, want, RLX_SYS);
      } else {
        __hip_atomic_store(a.done_host + 64 + blockIdx.x, want, RLX_SYS);
      }
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const bool try_bar = argc > 1 && !strcmp(argv[1], "bar");
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned* h = nullptr;
  CK(hipHostMalloc((void**)&h, 8192, hipHostMallocMapped));
  unsigned* d = nullptr;
  CK(hipHostGetDevicePointer((void**)&d, h, 0));
  volatile unsigned* bell = h;            // line 0
  volatile unsigned* done = h + 256;      // from byte 1024 on
  unsigned* dev = nullptr;
  CK(hipMalloc((void**)&dev, 8192));
  const size_t total = 4718592;  // 65 536 rows x 72 B copied: 9.4 MB of traffic, one 65 536-env Dummy step's
  uint4 *src = nullptr, *dst = nullptr;
  CK(hipMalloc((void**)&src, total));
  CK(hipMalloc((void**)&dst, total));
  CK(hipMemset(src, 1, total));
  unsigned* bar = nullptr;
  int large_bar = 0;
  CK(hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0));
  printf("large BAR: %d\n", large_bar);
  if (try_bar && large_bar) {
    CK(hipExtMallocWithFlags((void**)&bar, 32768, hipDeviceMallocFinegrained));
    CK(hipMemset(bar, 0, 32768));
    CK(hipDeviceSynchronize());
    printf("fine-grained device word at %p; writing it from the host ...\n", (void*)bar);
    fflush(stdout);
    *(volatile unsigned*)bar = 0;  // a machine without host access to device memory faults HERE (host side)
    printf("host write through the BAR went through\n");
    fflush(stdout);
  }
  const int R = 1500;
  struct Shape { int G, T; };
  const Shape shapes[] = {{256, 256}, {256, 512}, {128, 512}, {128, 1024}, {64, 1024}};
  auto ring = [&](int wake, unsigned v) {
    if (wake < 2) { *bell = v; return; }
    volatile unsigned* b = (volatile unsigned*)bar;
    if (wake == 4) for (int r = 0; r < 8; ++r) b[1024 * r] = v; else b[0] = v;
    _mm_sfence();
  };
  for (int wake = 0; wake < 5; ++wake) {
    if (wake == 1) continue;  // (every workgroup polling host memory: 8-100 us, round 3's first run)
    if (wake >= 2 && !bar) continue;
    for (int join = 0; join < 2; ++join)
      for (const Shape& sh : shapes)
        for (int work = 0; work < 2; ++work) {
          if (join == 2 && sh.G != 256 && sh.G != 64) continue;
          CK(hipMemset(dev, 0, 8192));
          memset(h, 0, 8192);
          if (bar) { ring(4, 0); ring(2, 0); }
          CK(hipDeviceSynchronize());
          Args a;
          a.bell_host = d;
          a.bell_dev = bar;
          a.done_host = d + 256;
          a.dev_flag = dev + 1024;
          a.counters = dev;
          a.src = src;
          a.dst = dst;
          a.rounds = R;
          a.max_polls = 400000;
          a.wake = wake;
          a.join = join;
          a.n16_per_wg = work ? (int)(total / 16 / sh.G) : 0;
          hipLaunchKernelGGL(k_server, dim3(sh.G), dim3(sh.T), 0, st, a);
          CK(hipGetLastError());
          std::vector<double> t;
          bool timed_out = false;
          for (int i = 1; i <= R && !timed_out; ++i) {
            const double t0 = now_us();
            ring(wake, (unsigned)i);
            const double limit = t0 + 3e5;
            for (;;) {
              bool ok;
              if (join == 0) ok = done[0] == (unsigned)i;
              else if (join == 1) {
                ok = true;
                const int ns = sh.G < 8 ? sh.G : 8;
                for (int s = 0; s < ns; ++s) ok &= done[16 + s] == (unsigned)i;
              } else {
                ok = true;
                for (int b = 0; b < sh.G; ++b) if (done[64 + b] != (unsigned)i) { ok = false; break; }
              }
              if (ok) break;
              if (now_us() > limit) { timed_out = true; break; }
            }
            t.push_back(now_us() - t0);
          }
          if (timed_out) ring(wake, (unsigned)R);
          CK(hipStreamSynchronize(st));
          printf("wake P%d join C%d  %3d wg x %4d thr  work %d: median %.2f us  p10 %.2f  p90 %.2f%s\n", wake, join, sh.G, sh.T,
                 work, pct(t, 0.5), pct(t, 0.1), pct(t, 0.9), timed_out ? "  TIMED OUT" : "");
          fflush(stdout);
        }
  }
  return 0;
}
