// hbm_mix_ceiling.cpp — what plain streaming kernels reach on this part at the read : write mixes
// of this library's two dominant kernels, to put their roofline fractions (against the 8 TB/s
// peak) next to a measured ceiling for the same mix:
//   dummy_rollout_dma<4>  reads 52 B and writes 71 B per env-step   (42 % read, 58 % write)
//   ygo_encode            reads ~0.9 KB and writes ~6.8 KB per record (11 % read, 89 % write)
// Each workgroup-strided thread moves 16-byte pieces: R loads and W stores per iteration from / to
// separate large arrays, nothing else; stores plain or non-temporal.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_mix_ceiling tools/hbm_mix_ceiling.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void k_mix(const v4* __restrict__ src, v4* __restrict__ dst, size_t iters) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  // lane-contiguous pieces: piece p of iteration `it` sits at (it * R + p) * stride + thread
  for (size_t it = 0; it < iters; ++it) {
    v4 v[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = __builtin_nontemporal_load(src + (it * R + r) * stride + i);
#pragma unroll
    for (int r = 0; r < R; ++r) acc += v[r];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      v4 out = acc;
      out.x += (float)w;
      if (NT) __builtin_nontemporal_store(out, dst + (it * W + w) * stride + i);
      else dst[(it * W + w) * stride + i] = out;
    }
  }
  if (W == 0 && acc.x == 123.456f) dst[0] = acc;  // keep the loads alive
}

template <int R, int W, bool NT>
double run(const v4* src, v4* dst, size_t total_pieces, int wgs, const char* label) {
  // pieces per thread-iteration = R + W; choose iterations so that about total_pieces move
  const size_t threads = (size_t)wgs * 256;
  const size_t iters = std::max<size_t>(1, total_pieces / ((size_t)(R + W) * threads));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int rep = 0; rep < 12; ++rep) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_mix<R, W, NT>), dim3(wgs), dim3(256), 0, 0, src, dst, iters);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float t;
    CK(hipEventElapsedTime(&t, a, b));
    if (rep >= 4) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double bytes = 16.0 * (double)(R + W) * (double)iters * (double)threads;
  const double tbs = bytes / (ms[ms.size() / 2] * 1e-3) / 1e12;
  printf("%-34s R:W %d:%d  %s stores, %5d workgroups: %.2f TB/s\n", label, R, W, NT ? "non-temporal" : "plain       ", wgs, tbs);
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  return tbs;
}

int main() {
  const size_t pieces = (size_t)3 << 26;  // 3 GiB of traffic per launch
  v4 *src = nullptr, *dst = nullptr;
  CK(hipMalloc((void**)&src, (pieces + 4096) * 16));
  CK(hipMalloc((void**)&dst, (pieces + 4096) * 16));
  CK(hipMemset(src, 1, (pieces + 4096) * 16));
  CK(hipMemset(dst, 0, (pieces + 4096) * 16));
  for (int wgs : {1024, 4096, 16384}) {
    run<1, 0, false>(src, dst, pieces, wgs, "read only");
    run<0, 1, false>(src, dst, pieces, wgs, "write only");
    run<0, 1, true>(src, dst, pieces, wgs, "write only");
    run<1, 1, false>(src, dst, pieces, wgs, "copy");
    run<1, 1, true>(src, dst, pieces, wgs, "copy");
    run<3, 4, true>(src, dst, pieces, wgs, "rollout mix (43 % read)");
    run<3, 4, false>(src, dst, pieces, wgs, "rollout mix (43 % read)");
    run<1, 8, true>(src, dst, pieces, wgs, "ygo_encode mix (11 % read)");
    run<1, 8, false>(src, dst, pieces, wgs, "ygo_encode mix (11 % read)");
  }
  return 0;
}
