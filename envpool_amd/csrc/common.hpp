// common.hpp — small host-side helpers shared by the translation units of libenvpool_amd.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/envpool_amd.h"

namespace epc {

inline thread_local std::string g_last_error;

inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return ::epc::fail(EP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                                \
  } while (0)

// Every entry point works on ITS object's device and leaves the calling thread's current device as
// it found it: a process that drives several GPUs (python/sharded.py over K devices, or a caller with
// torch beside it, whose notion of "the current device" is HIP's) must not find it changed by a pool
// or encoder call.
struct DeviceScope {
  int prev = -1;
  bool changed = false;
  hipError_t enter(int dev) {
    if (changed) return hipSuccess;
    if (hipGetDevice(&prev) == hipSuccess && prev == dev) return hipSuccess;
    const hipError_t e = hipSetDevice(dev);
    changed = e == hipSuccess && prev >= 0;
    return e;
  }
  ~DeviceScope() {
    if (changed) (void)hipSetDevice(prev);
  }
};
#define EP_ENTER_DEVICE(dev)       \
  ::epc::DeviceScope _dev_scope;   \
  HIP_TRY(_dev_scope.enter(dev))

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

// A pool of pinned host blocks that outlives its owner if numpy still holds a batch (the
// reference hands out buffers that are never reused, py_envpool.h:42-46).
struct HostBlocks {
  std::mutex mu;
  std::vector<void*> free_list;
  size_t bytes = 0;
  bool alive = true;
};
struct HostBlockOwner {
  std::shared_ptr<HostBlocks> home;
  void* ptr;
};

// An argument that says where the caller's memory is must be one of the two kinds: anything else
// would be read as one of them and the pointers with it.
inline bool bad_mem(int32_t mem) { return mem != EP_MEM_HOST && mem != EP_MEM_DEVICE; }

inline int take_block(const std::shared_ptr<HostBlocks>& hb, void** out) {
  void* blk = nullptr;
  {
    std::lock_guard<std::mutex> bl(hb->mu);
    if (!hb->free_list.empty()) {
      blk = hb->free_list.back();
      hb->free_list.pop_back();
    }
  }
  if (!blk) HIP_TRY(hipHostMalloc(&blk, hb->bytes, hipHostMallocMapped));
  *out = blk;
  return EP_OK;
}

inline void give_block(const std::shared_ptr<HostBlocks>& hb, void* blk) {
  std::lock_guard<std::mutex> lk(hb->mu);
  if (hb->alive && hb->free_list.size() < 8) hb->free_list.push_back(blk);
  else (void)hipHostFree(blk);
}

inline void retire_blocks(const std::shared_ptr<HostBlocks>& hb) {
  std::lock_guard<std::mutex> lk(hb->mu);
  hb->alive = false;
  for (void* b : hb->free_list) (void)hipHostFree(b);
  hb->free_list.clear();
}

}  // namespace epc
