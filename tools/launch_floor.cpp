// launch_floor.cpp — what one lockstep step can cost at best on this box (GPU box, diagnostic).
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/launch_floor.cpp -o tools/launch_floor
// (a) empty kernel, completion event on its dispatch packet, hipEventSynchronize
// (b) empty kernel whose one wave writes a host-mapped word; the host spins on the word
// (c) resident kernel, one wave: polls a host-mapped doorbell, answers in a host-mapped word
// (d) resident kernel, 256 workgroups: workgroup 0 polls the doorbell and raises a device flag, all
//     workgroups see it (sc1 polls), add to a device counter; the last one answers
// Every resident kernel leaves by itself after a bounded number of polls.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

typedef __attribute__((address_space(1))) unsigned int gu32;
#define RLX_SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__global__ void k_empty() {}
__global__ void k_word(unsigned* done, unsigned v) {
  if (threadIdx.x == 0) __hip_atomic_store(done, v, RLX_SYS);
}
// (c): rounds of doorbell -> answer; leaves after `rounds` answers or `max_polls` idle polls
__global__ void k_pingpong(unsigned* bell, unsigned* done, unsigned rounds, unsigned max_polls) {
  if (threadIdx.x != 0) return;
  unsigned polls = 0;
  for (unsigned want = 1; want <= rounds;) {
    const unsigned b = __hip_atomic_load(bell, RLX_SYS);
    if (b >= want) { __hip_atomic_store(done, want, RLX_SYS); ++want; polls = 0; continue; }
    if (++polls > max_polls) break;
    __builtin_amdgcn_s_sleep(1);
  }
}
// (d)
__global__ __launch_bounds__(256) void k_server(unsigned* bell, unsigned* done, unsigned* dev_flag, unsigned* dev_count,
                                                unsigned rounds, unsigned max_polls, int work_bytes, const uint4* src, uint4* dst) {
  __shared__ unsigned s_go;
  for (unsigned want = 1; want <= rounds; ++want) {
    if (threadIdx.x == 0) {
      unsigned polls = 0, go = 0;
      if (blockIdx.x == 0) {
        for (;;) {
          if (__hip_atomic_load(bell, RLX_SYS) >= want) { go = 1; break; }
          if (++polls > max_polls) break;
          __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store(dev_flag, go ? want : 0xffffffffu, RLX_AGENT);
      } else {
        for (;;) {
          const unsigned f = __hip_atomic_load(dev_flag, RLX_AGENT);
          if (f == 0xffffffffu) break;
          if (f >= want) { go = 1; break; }
          if (++polls > 64u * max_polls) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      s_go = go;
    }
    __syncthreads();
    if (!s_go) return;
    if (work_bytes) {  // a stand-in for the step: each workgroup copies its tile (16-byte loads and stores)
      const int n16 = work_bytes / 16;
      for (int i = threadIdx.x; i < n16; i += 256) dst[(size_t)blockIdx.x * n16 + i] = src[(size_t)blockIdx.x * n16 + i];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(dev_count, 1u, RLX_AGENT);
      if (old == want * gridDim.x - 1) __hip_atomic_store(done, want, RLX_SYS);
    }
    __syncthreads();
  }
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned* h = nullptr;  // [0] bell, [16] done (separate lines)
  CK(hipHostMalloc((void**)&h, 4096, hipHostMallocMapped));
  unsigned* d = nullptr;
  CK(hipHostGetDevicePointer((void**)&d, h, 0));
  volatile unsigned* bell = h;
  volatile unsigned* done = h + 16;
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const int R = 2000;
  // warm up
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st);
  CK(hipStreamSynchronize(st));
  {
    std::vector<double> t;
    for (int i = 0; i < R; ++i) {
      const double a = now_us();
      hipExtLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, nullptr, ev, 0);
      CK(hipEventSynchronize(ev));
      t.push_back(now_us() - a);
    }
    printf("(a) empty kernel + event on the dispatch + hipEventSynchronize: %.2f us\n", med(t));
  }
  {
    std::vector<double> t, tl;
    for (int i = 1; i <= R; ++i) {
      const double a = now_us();
      hipLaunchKernelGGL(k_word, dim3(1), dim3(64), 0, st, d + 16, (unsigned)i);
      const double b = now_us();
      while (*done != (unsigned)i) {}
      t.push_back(now_us() - a);
      tl.push_back(b - a);
    }
    CK(hipStreamSynchronize(st));
    printf("(b) kernel writes a host-mapped word, host spins: %.2f us (launch call itself %.2f)\n", med(t), med(tl));
  }
  {
    // (b') the same kernel as a captured one-node hipGraph, replayed: what "a hipGraph of {step
    // kernel} per step" would cost (the kernel's argument is fixed at capture, so completion is seen
    // with hipStreamSynchronize)
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 100; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    std::vector<double> t, tl;
    for (int i = 0; i < R; ++i) {
      const double a = now_us();
      CK(hipGraphLaunch(ge, st));
      const double b = now_us();
      CK(hipStreamSynchronize(st));
      t.push_back(now_us() - a);
      tl.push_back(b - a);
    }
    printf("(b') one-kernel hipGraph replay + hipStreamSynchronize: %.2f us (hipGraphLaunch itself %.2f)\n", med(t), med(tl));
    std::vector<double> t2;
    for (int i = 0; i < R; ++i) {
      const double a = now_us();
      hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st);
      CK(hipStreamSynchronize(st));
      t2.push_back(now_us() - a);
    }
    printf("(b'') plain launch of the same kernel + hipStreamSynchronize: %.2f us\n", med(t2));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  for (int variant = 0; variant < 3; ++variant) {
    // (e) what hipDeviceSynchronize costs AFTER the launch's completion event has been waited for:
    // 0 = nothing in between, 1 = hipStreamQuery first, 2 = hipEventQuery first
    std::vector<double> t, tq;
    for (int i = 0; i < R; ++i) {
      hipExtLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, nullptr, ev, 0);
      CK(hipEventSynchronize(ev));
      const double q0 = now_us();
      if (variant == 1) (void)hipStreamQuery(st);
      if (variant == 2) (void)hipEventQuery(ev);
      const double a = now_us();
      CK(hipDeviceSynchronize());
      t.push_back(now_us() - a);
      tq.push_back(a - q0);
    }
    printf("(e%d) hipDeviceSynchronize after the completion event was waited for: %.2f us (query before it %.2f)\n", variant, med(t), med(tq));
  }
  {
    *bell = 0; *done = 0;
    hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(64), 0, st, d, d + 16, (unsigned)R, 2000000u);
    std::vector<double> t;
    for (int i = 1; i <= R; ++i) {
      const double a = now_us();
      *bell = (unsigned)i;
      while (*done != (unsigned)i) {}
      t.push_back(now_us() - a);
    }
    CK(hipStreamSynchronize(st));
    printf("(c) resident wave: doorbell -> answer round trip: %.2f us\n", med(t));
  }
  for (int work : {0, 18432, 36864}) {
    unsigned* dev = nullptr;
    CK(hipMalloc((void**)&dev, 4096));
    CK(hipMemset(dev, 0, 4096));
    uint4 *src = nullptr, *dst = nullptr;
    CK(hipMalloc((void**)&src, 256 * 36864));
    CK(hipMalloc((void**)&dst, 256 * 36864));
    CK(hipMemset(src, 1, 256 * 36864));
    *bell = 0; *done = 0;
    hipLaunchKernelGGL(k_server, dim3(256), dim3(256), 0, st, d, d + 16, dev, dev + 64, (unsigned)R, 300000u, work, src, dst);
    std::vector<double> t;
    for (int i = 1; i <= R; ++i) {
      const double a = now_us();
      *bell = (unsigned)i;
      double limit = a + 2e6;
      while (*done != (unsigned)i && now_us() < limit) {}
      t.push_back(now_us() - a);
    }
    CK(hipStreamSynchronize(st));
    printf("(d) resident 256 workgroups, %d bytes copied per workgroup per round: %.2f us\n", work, med(t));
    hipFree(dev); hipFree(src); hipFree(dst);
  }
  return 0;
}
