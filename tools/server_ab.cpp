// server_ab.cpp — A/B/A/B of the resident step server's wake-up and join variants in ONE process on
// one box (boxes and runs differ by more than the variants do): per variant a fresh 65 536-env pool,
// 500 untimed + 3000 timed ep_pool_steps, four rounds over all variants.
//   hipcc -O2 -std=c++17 -Iinclude tools/server_ab.cpp -Lenvpool_amd -lenvpool_amd -Wl,-rpath,'$ORIGIN/../envpool_amd' -o tools/server_ab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "envpool_amd.h"

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { int _rc = (x); if (_rc != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, _rc, ep_last_error()); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 65536;
  int32_t* d_ids; double* d_la;
  if (hipMalloc((void**)&d_ids, 4 * (size_t)N) != hipSuccess || hipMalloc((void**)&d_la, 48 * (size_t)N) != hipSuccess) return 1;
  {
    std::vector<int32_t> ids(N);
    for (int i = 0; i < N; ++i) ids[i] = i;
    (void)hipMemcpy(d_ids, ids.data(), 4 * (size_t)N, hipMemcpyHostToDevice);
    std::vector<double> la((size_t)N * 6, 1.5);
    (void)hipMemcpy(d_la, la.data(), 48 * (size_t)N, hipMemcpyHostToDevice);
  }
  const void* a1[5] = {d_ids, d_ids, d_la, d_ids, d_ids};
  int64_t dims[5] = {N, N, N, N, N};
  struct V { const char* join; const char* bar; const char* pause; const char* name; };
  const V vs[] = {{"1", "1", "0", "BAR doorbell, 8 host words"}, {"1", "1", "1", "BAR doorbell, 8 host words, PAUSE in the host spin"},
                  {"2", "1", "0", "BAR doorbell, two-level join"},
                  {"1", "0", "0", "leader wake-up, 8 host words"}, {"2", "0", "1", "leader, two-level join, PAUSE (round 2)"}};
  for (int round = 0; round < 4; ++round)
    for (const V& v : vs) {
      setenv("EP_SERVER_JOIN", v.join, 1);
      setenv("EP_SERVER_BAR", v.bar, 1);
      setenv("EP_SERVER_PAUSE", v.pause, 1);
      ep_dummy_config cfg{};
      cfg.num_envs = N; cfg.max_num_players = 1; cfg.thread_affinity_offset = -1; cfg.seed = 42;
      cfg.max_episode_steps = 2147483647; cfg.state_num = 10; cfg.action_num = 6; cfg.ring_depth = 4;
      ep_pool* p = nullptr;
      CK(ep_dummy_pool_create(&cfg, &p));
      ep_batch b;
      CK(ep_pool_reset(p, d_ids, N, EP_MEM_DEVICE));
      CK(ep_pool_recv(p, EP_MEM_DEVICE, &b));
      std::vector<double> t;
      for (int r = 0; r < 3500; ++r) {
        const double a0 = now_us();
        CK(ep_pool_step(p, 5, a1, dims, &b));
        if (r >= 500) t.push_back(now_us() - a0);
      }
      std::sort(t.begin(), t.end());
      uint64_t launches = 0, stops = 0, served = 0;
      CK(ep_pool_server_stats(p, &launches, &stops, &served, nullptr));
      printf("round %d  %-52s median %.2f  p10 %.2f  p90 %.2f us  (%llu launches, %llu served)\n", round, v.name, t[t.size() / 2],
             t[t.size() / 10], t[t.size() * 9 / 10], (unsigned long long)launches, (unsigned long long)served);
      if (getenv("EP_SERVER_TRACE")) {
        static uint64_t tr[4 * 1024];
        const int n = -ep_pool_server_trace(p, tr, 1024);
        if (n > 0) {
          uint64_t s0 = ~0ull, s1 = 0, d1 = 0, c1 = 0;
          double w01 = 0, w12 = 0, w23 = 0;
          for (int g = 0; g < n; ++g) {
            s0 = std::min(s0, tr[4 * g]); s1 = std::max(s1, tr[4 * g]);
            d1 = std::max(d1, tr[4 * g + 2]); c1 = std::max(c1, tr[4 * g + 3]);
            w01 += (double)(tr[4 * g + 1] - tr[4 * g]); w12 += (double)(tr[4 * g + 2] - tr[4 * g + 1]);
            w23 += (double)(tr[4 * g + 3] - tr[4 * g + 2]);
          }
          printf("         device side of the last step: doorbell seen over %.2f us; first seen -> last drained %.2f, -> last counted %.2f us;"
                 " per workgroup seen->staged %.2f, staged->drained %.2f, drained->counted %.2f us\n", (s1 - s0) * 0.01, (d1 - s0) * 0.01,
                 (c1 - s0) * 0.01, w01 / n * 0.01, w12 / n * 0.01, w23 / n * 0.01);
        }
      }
      fflush(stdout);
      ep_pool_destroy(p);
    }
  return 0;
}
