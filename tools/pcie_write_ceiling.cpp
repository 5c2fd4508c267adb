// What can a kernel's stores into pinned host memory reach over PCIe, against the copy engine?
// (The numpy path's step kernels write their rows straight into a pinned block: DESIGN.md §3.3.)
// hipcc -O2 --offload-arch=gfx950 tools/pcie_write_ceiling.cpp -o /tmp/pcie_write_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NT>
__global__ __launch_bounds__(256) void fill16(uint4* out, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  typedef unsigned int v4 __attribute__((ext_vector_type(4)));
  v4 v = {(unsigned)i, 1u, 2u, 3u};
  v4* o = reinterpret_cast<v4*>(out);
  for (; i < n16; i += stride) {
    if (NT) __builtin_nontemporal_store(v, o + i);
    else o[i] = v;
  }
}
__global__ __launch_bounds__(256) void fill4(uint32_t* out, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n4; i += stride) out[i] = (uint32_t)i;
}

int main() {
  const size_t bytes_list[] = {1 << 20, 4718592, 16 << 20};
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (size_t bytes : bytes_list) {
    void *h = nullptr, *hd = nullptr, *d = nullptr;
    CK(hipHostMalloc(&h, bytes, hipHostMallocMapped));
    CK(hipHostGetDevicePointer(&hd, h, 0));
    CK(hipMalloc(&d, bytes));
    auto time = [&](auto&& fn, const char* name) {
      std::vector<float> ms;
      for (int it = 0; it < 30; ++it) {
        hipEventRecord(a, st);
        fn();
        hipEventRecord(b, st);
        hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      printf("  %-44s %8.1f us  %6.1f GB/s\n", name, ms[ms.size() / 2] * 1e3, bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
    };
    printf("%zu bytes\n", bytes);
    time([&] { hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st); }, "copy engine D2H");
    for (int grid : {64, 256, 1024, 4096}) {
      char name[96];
      snprintf(name, sizeof name, "kernel 16-B stores, %d workgroups", grid);
      time([&] { hipLaunchKernelGGL(fill16<0>, dim3(grid), dim3(256), 0, st, (uint4*)hd, bytes / 16); }, name);
      snprintf(name, sizeof name, "kernel 16-B non-temporal stores, %d wgs", grid);
      time([&] { hipLaunchKernelGGL(fill16<1>, dim3(grid), dim3(256), 0, st, (uint4*)hd, bytes / 16); }, name);
    }
    time([&] { hipLaunchKernelGGL(fill4, dim3(256), dim3(256), 0, st, (uint32_t*)hd, bytes / 4); }, "kernel 4-B stores, 256 workgroups");
    // four pieces back to back, as the host-mode send launches them
    time([&] { for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(fill16<0>, dim3(64), dim3(256), 0, st, (uint4*)((char*)hd + q * (bytes / 4 / 16 * 16)), bytes / 4 / 16); }, "4 kernels of a quarter each, 64 wgs");
    CK(hipHostFree(h)); CK(hipFree(d));
  }
  return 0;
}
