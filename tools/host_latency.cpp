// host_latency.cpp — where the host time of one lockstep step goes (GPU box, diagnostic).
//   hipcc -O2 -std=c++17 -Iinclude tools/host_latency.cpp -Lenvpool_amd -lenvpool_amd -Wl,-rpath,'$ORIGIN/../envpool_amd' -o tools/host_latency
// Prints per-phase host times of (a) a 20-step fused rollout bracketed by synchronisation, with
// and without kernel timing, (b) the blocking ep_pool_send + ep_pool_recv loop, (c) ep_pool_step.
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
#define CK(x)                                                            \
  do {                                                                   \
    int _rc = (x);                                                       \
    if (_rc != 0) {                                                      \
      fprintf(stderr, "%s -> %d: %s\n", #x, _rc, ep_last_error());       \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

static double median(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 65536, T = 20, Tbig = 128;
  ep_dummy_config cfg{};
  cfg.num_envs = N; cfg.max_num_players = 1; cfg.thread_affinity_offset = -1; cfg.seed = 42;
  cfg.max_episode_steps = 2147483647; cfg.state_num = 10; cfg.action_num = 6; cfg.ring_depth = 2 * Tbig;
  ep_pool* p = nullptr;
  CK(ep_dummy_pool_create(&cfg, &p));
  int32_t* d_ids; int32_t* d_penv; double* d_la;
  hipMalloc((void**)&d_ids, 4 * (size_t)N);
  hipMalloc((void**)&d_penv, 4 * (size_t)N * Tbig);
  hipMalloc((void**)&d_la, 48 * (size_t)N * Tbig);
  {
    std::vector<int32_t> ids(N);
    for (int i = 0; i < N; ++i) ids[i] = i;
    hipMemcpy(d_ids, ids.data(), 4 * (size_t)N, hipMemcpyHostToDevice);
    for (int t = 0; t < Tbig; ++t) hipMemcpy(d_penv + (size_t)t * N, ids.data(), 4 * (size_t)N, hipMemcpyHostToDevice);
    std::vector<double> la((size_t)N * 6, 1.5);
    for (int t = 0; t < Tbig; ++t) hipMemcpy(d_la + (size_t)t * N * 6, la.data(), 48 * (size_t)N, hipMemcpyHostToDevice);
  }
  const void* arrs[5] = {d_ids, d_penv, d_la, d_ids, d_ids};
  int64_t dims[5] = {N, N, N, N, N}, strides[5] = {0, N, N, 0, 0};
  std::vector<ep_batch> batches(Tbig);
  ep_batch b;
  CK(ep_pool_reset(p, d_ids, N, EP_MEM_DEVICE));
  CK(ep_pool_recv(p, EP_MEM_DEVICE, &b));
  // warm the clock
  for (int i = 0; i < 200; ++i) {
    CK(ep_pool_rollout(p, Tbig, 5, arrs, dims, strides));
    CK(ep_pool_recv_many(p, EP_MEM_DEVICE, Tbig, batches.data()));
  }
  CK(ep_pool_sync(p));
  for (int timing = 0; timing < 2; ++timing) {
    CK(ep_pool_set_timing(p, timing));
    std::vector<double> t_call, t_recv, t_sync, t_dev, t_all;
    for (int r = 0; r < 30; ++r) {
      hipDeviceSynchronize();
      const double a0 = now_us();
      CK(ep_pool_rollout(p, T, 5, arrs, dims, strides));
      const double a1 = now_us();
      CK(ep_pool_recv_many(p, EP_MEM_DEVICE, T, batches.data()));
      const double a2 = now_us();
      CK(ep_pool_sync(p));
      const double a3 = now_us();
      hipDeviceSynchronize();
      const double a4 = now_us();
      t_call.push_back(a1 - a0); t_recv.push_back(a2 - a1); t_sync.push_back(a3 - a2); t_dev.push_back(a4 - a3); t_all.push_back(a4 - a0);
    }
    double tot = 0; int64_t cnt = 0;
    CK(ep_pool_kernel_time_us(p, &tot, &cnt, 1));
    printf("rollout T=%d timing=%d: call %.1f  recv_many %.1f  pool_sync %.1f  device_sync %.1f  total %.1f us (medians of 30); kernel %.1f us\n",
           T, timing, median(t_call), median(t_recv), median(t_sync), median(t_dev), median(t_all), cnt ? tot / cnt : 0.0);
  }
  CK(ep_pool_set_timing(p, 0));
  // the same region in other orders: which waits the runtime makes cheap for each other
  for (int order = 0; order < 4; ++order) {
    std::vector<double> t_all, t_last;
    for (int r = 0; r < 40; ++r) {
      hipDeviceSynchronize();
      const double a0 = now_us();
      CK(ep_pool_rollout(p, T, 5, arrs, dims, strides));
      if (order == 0) {         // stream-ordered hand-out, then the device-wide wait (what bench.py times)
        CK(ep_pool_recv_many_on_stream(p, nullptr, T, batches.data()));
      } else if (order == 1) {  // ep_pool_sync (a stream synchronise) BEFORE receiving, then the device-wide wait
        CK(ep_pool_sync(p));
        CK(ep_pool_recv_many(p, EP_MEM_DEVICE, T, batches.data()));
      } else if (order == 2) {  // stream-ordered hand-out, the pool's stream synchronised, then the device-wide wait
        CK(ep_pool_recv_many_on_stream(p, nullptr, T, batches.data()));
        (void)hipStreamSynchronize((hipStream_t)ep_pool_stream(p));
      } else {                  // blocking hand-out (event wait), then the device-wide wait
        CK(ep_pool_recv_many(p, EP_MEM_DEVICE, T, batches.data()));
      }
      const double a1 = now_us();
      hipDeviceSynchronize();
      const double a2 = now_us();
      t_all.push_back(a2 - a0); t_last.push_back(a2 - a1);
      CK(ep_pool_sync(p));
    }
    static const char* names[4] = {"recv_many_on_stream + device sync", "ep_pool_sync + recv_many + device sync",
                                   "recv_many_on_stream + stream sync + device sync", "recv_many (event wait) + device sync"};
    printf("rollout T=%d region, %-50s: total %.1f us, of which the closing device sync %.1f us (medians of 40)\n", T, names[order],
           median(t_all), median(t_last));
  }
  // one launch per step, blocking
  {
    const void* a1[5] = {d_ids, d_ids, d_la, d_ids, d_ids};
    std::vector<double> t_send, t_recv;
    for (int r = 0; r < 2000; ++r) {
      const double a0 = now_us();
      CK(ep_pool_send(p, 5, a1, dims, EP_MEM_DEVICE));
      const double a1t = now_us();
      CK(ep_pool_recv(p, EP_MEM_DEVICE, &b));
      const double a2 = now_us();
      if (r >= 200) { t_send.push_back(a1t - a0); t_recv.push_back(a2 - a1t); }
    }
    printf("send+recv per step: send %.2f  recv %.2f  total %.2f us (medians of 1800)\n", median(t_send), median(t_recv),
           median(t_send) + median(t_recv));
  }
  {
    const void* a1[5] = {d_ids, d_ids, d_la, d_ids, d_ids};
    std::vector<double> t_step;
    for (int r = 0; r < 4000; ++r) {
      const double a0 = now_us();
      CK(ep_pool_step(p, 5, a1, dims, &b));
      const double a2 = now_us();
      if (r >= 400) t_step.push_back(a2 - a0);
    }
    std::sort(t_step.begin(), t_step.end());
    printf("ep_pool_step per step: median %.2f  p10 %.2f  p90 %.2f us\n", t_step[t_step.size() / 2], t_step[t_step.size() / 10],
           t_step[t_step.size() * 9 / 10]);
    uint64_t launches = 0, stops = 0, served = 0;
    int64_t limit = 0;
    CK(ep_pool_server_stats(p, &launches, &stops, &served, &limit));
    printf("  server: %llu launches, %llu idle stops, %llu steps served, idle limit %lld us\n", (unsigned long long)launches,
           (unsigned long long)stops, (unsigned long long)served, (long long)limit);
    if (getenv("EP_SERVER_TRACE")) {
      // where a step's time goes on the device: per workgroup seen -> staged -> drained -> counted
      for (int rep = 0; rep < 3; ++rep) {
        for (int r = 0; r < 50; ++r) CK(ep_pool_step(p, 5, a1, dims, &b));
        static uint64_t tr[4 * 1024];
        const int n = -ep_pool_server_trace(p, tr, 1024);
        if (n <= 0) break;
        uint64_t s0 = ~0ull, s1 = 0, d1 = 0, c1 = 0;
        double w01 = 0, w12 = 0, w23 = 0;
        for (int g = 0; g < n; ++g) {
          s0 = std::min(s0, tr[4 * g]); s1 = std::max(s1, tr[4 * g]);
          d1 = std::max(d1, tr[4 * g + 2]); c1 = std::max(c1, tr[4 * g + 3]);
          w01 += (double)(tr[4 * g + 1] - tr[4 * g]); w12 += (double)(tr[4 * g + 2] - tr[4 * g + 1]);
          w23 += (double)(tr[4 * g + 3] - tr[4 * g + 2]);
        }
        printf("  trace (%d workgroups, 10 ns ticks): doorbell seen over %.2f us; first seen -> last drained %.2f us, -> last counted %.2f us;"
               " per workgroup: seen->staged %.2f, staged->drained %.2f, drained->counted %.2f us\n", n, (s1 - s0) * 0.01,
               (d1 - s0) * 0.01, (c1 - s0) * 0.01, w01 / n * 0.01, w12 / n * 0.01, w23 / n * 0.01);
      }
    }
    // begin / end halves timed apart
    std::vector<double> t_b, t_e;
    for (int r = 0; r < 2000; ++r) {
      const double a0 = now_us();
      CK(ep_pool_step_begin(p, 5, a1, dims));
      const double a1t = now_us();
      CK(ep_pool_step_end(p, &b));
      const double a2 = now_us();
      t_b.push_back(a1t - a0); t_e.push_back(a2 - a1t);
    }
    printf("ep_pool_step_begin %.2f + ep_pool_step_end %.2f us (medians)\n", median(t_b), median(t_e));
    // the caller synchronises the whole device before every step (round 2: up to 50 ms each)
    std::vector<double> t_sync;
    for (int r = 0; r < 1200; ++r) {
      const double a0 = now_us();
      (void)hipDeviceSynchronize();
      CK(ep_pool_step(p, 5, a1, dims, &b));
      t_sync.push_back(now_us() - a0);
    }
    double sum = 0;
    for (double v : t_sync) sum += v;
    std::sort(t_sync.begin(), t_sync.end());
    CK(ep_pool_server_stats(p, &launches, &stops, &served, &limit));
    printf("hipDeviceSynchronize + ep_pool_step: mean %.2f  median %.2f  p99 %.2f  max %.2f us\n", sum / t_sync.size(),
           t_sync[t_sync.size() / 2], t_sync[t_sync.size() * 99 / 100], t_sync.back());
    printf("  server: %llu launches, %llu idle stops, %llu steps served\n", (unsigned long long)launches, (unsigned long long)stops,
           (unsigned long long)served);
  }
  {
    // the stream-ordered step (ep_pool_step_on_stream): host time per call with a consumer stream
    // that does nothing else, and the whole loop including one closing synchronise; then K = 4 pools
    // of a quarter the size stepped from ONE stream the way python/sharded.py does it (send to all,
    // then receive from each)
    const void* a1[5] = {d_ids, d_ids, d_la, d_ids, d_ids};
    hipStream_t user;
    (void)hipStreamCreateWithFlags(&user, hipStreamNonBlocking);
    std::vector<double> ts, tsend, trecv;
    CK(ep_pool_sync(p));
    for (int r = 0; r < 300; ++r) CK(ep_pool_step_on_stream(p, user, 5, a1, dims, &b));
    (void)hipStreamSynchronize(user);
    const double w0 = now_us();
    for (int r = 0; r < 2000; ++r) {
      const double a0 = now_us();
      CK(ep_pool_send_on_stream(p, user, 5, a1, dims));
      const double a1t = now_us();
      CK(ep_pool_recv_on_stream(p, user, &b));
      const double a2 = now_us();
      tsend.push_back(a1t - a0); trecv.push_back(a2 - a1t); ts.push_back(a2 - a0);
    }
    (void)hipStreamSynchronize(user);
    const double w1 = now_us();
    printf("ep_pool_step_on_stream per step: host %.2f us (send half %.2f, recv half %.2f; medians), %.2f us per step over 2000 steps incl. the closing synchronise\n",
           median(ts), median(tsend), median(trecv), (w1 - w0) / 2000);
    CK(ep_pool_poll_errors(p));
    const int K = 4;
    ep_pool* q[K];
    void* qa[K][5];
    int64_t qd[5];
    const int nq = N / K;
    for (int k = 0; k < K; ++k) {
      ep_dummy_config c2 = cfg;
      c2.num_envs = nq; c2.batch_size = nq; c2.env_id_offset = k * nq;
      CK(ep_dummy_pool_create(&c2, &q[k]));
      std::vector<int32_t> ids(nq);
      for (int i = 0; i < nq; ++i) ids[i] = k * nq + i;
      CK(ep_pool_reset(q[k], ids.data(), nq, EP_MEM_HOST));
      CK(ep_pool_recv(q[k], EP_MEM_DEVICE, &b));
      qa[k][0] = (char*)d_ids + 4 * (size_t)k * nq; qa[k][1] = qa[k][0];
      qa[k][2] = (char*)d_la + 48 * (size_t)k * nq; qa[k][3] = qa[k][0]; qa[k][4] = qa[k][0];
    }
    for (int i = 0; i < 5; ++i) qd[i] = nq;
    ep_batch bk[K];
    auto one = [&]() {
      for (int k = 0; k < K; ++k) CK(ep_pool_send_on_stream(q[k], user, 5, (const void* const*)qa[k], qd));
      for (int k = 0; k < K; ++k) CK(ep_pool_recv_on_stream(q[k], user, &bk[k]));
      return 0;
    };
    for (int r = 0; r < 300; ++r) if (one()) return 1;
    (void)hipStreamSynchronize(user);
    ts.clear();
    const double v0 = now_us();
    for (int r = 0; r < 2000; ++r) {
      const double a0 = now_us();
      if (one()) return 1;
      ts.push_back(now_us() - a0);
    }
    (void)hipStreamSynchronize(user);
    const double v1 = now_us();
    printf("%d pools of %d envs from one stream, send to all then receive from each: host %.2f us per step (median), %.2f us per step over 2000 steps incl. the closing synchronise\n",
           K, nq, median(ts), (v1 - v0) / 2000);
    // the same two loops with the stream ADOPTED by the pools (ep_pool_set_stream): no events at all
    CK(ep_pool_set_stream(p, user, 0));
    for (int r = 0; r < 300; ++r) CK(ep_pool_step_on_stream(p, user, 5, a1, dims, &b));
    (void)hipStreamSynchronize(user);
    ts.clear();
    const double x0 = now_us();
    for (int r = 0; r < 2000; ++r) {
      const double a0 = now_us();
      CK(ep_pool_step_on_stream(p, user, 5, a1, dims, &b));
      ts.push_back(now_us() - a0);
    }
    (void)hipStreamSynchronize(user);
    const double x1 = now_us();
    printf("  with the stream adopted: host %.2f us per step (median), %.2f us per step over 2000 steps incl. the closing synchronise\n", median(ts),
           (x1 - x0) / 2000);
    CK(ep_pool_set_stream(p, nullptr, 1));
    for (int k = 0; k < K; ++k) CK(ep_pool_set_stream(q[k], user, 0));
    for (int r = 0; r < 300; ++r) if (one()) return 1;
    (void)hipStreamSynchronize(user);
    ts.clear();
    const double y0 = now_us();
    for (int r = 0; r < 2000; ++r) {
      const double a0 = now_us();
      if (one()) return 1;
      ts.push_back(now_us() - a0);
    }
    (void)hipStreamSynchronize(user);
    const double y1 = now_us();
    printf("  %d pools with the stream adopted: host %.2f us per step (median), %.2f us per step over 2000 steps incl. the closing synchronise\n", K,
           median(ts), (y1 - y0) / 2000);
    for (int k = 0; k < K; ++k) ep_pool_destroy(q[k]);
    (void)hipStreamDestroy(user);
  }
  ep_pool_destroy(p);
  return 0;
}
