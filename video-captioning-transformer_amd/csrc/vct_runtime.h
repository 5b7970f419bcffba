// Launch layer shared by every kernel file: a launch either goes straight to the stream (eager) or, while a launch
// list is being recorded on this host thread, is appended to that list as a closure holding the kernel, its geometry
// and BY-VALUE copies of its arguments.  Replaying the list re-issues exactly those launches -- no planning, no
// descriptor marshalling, no Python -- with eager multi-stream semantics (each entry remembers its stream, cross-stream
// edges are recorded event record / wait pairs).  See vct_runtime.hip and include/vct_hip.h (vct_cmdlist_*).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <functional>

namespace vct {

struct CmdList;
extern thread_local CmdList* g_rec;                                           // non-null while this thread records
void rec_push(hipStream_t st, std::function<void(hipStream_t)>&& fn);         // append to g_rec
// Replayed closures cannot return a status to their original caller: the first non-zero code noted during a replay is
// kept (sticky) and becomes the return value of vct_cmdlist_replay -- a failed RCCL collective / event call is not silent.
void replay_note_error(int code);

template <typename K, typename... A>
inline void launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... args) {
  if (g_rec != nullptr) {
    rec_push(st, [=](hipStream_t s) { hipLaunchKernelGGL(kernel, grid, block, lds, s, args...); });
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
  }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per DEVICE: a process that drives several GPUs must opt in on each of them (a
// once-per-process flag left the > 64 KB launches of the second device failing).  One bit per device per call site.
struct DynLdsOptIn {
  std::atomic<unsigned long long> done[4] = {};   // up to 256 devices; host threads driving different devices may race here
  hipError_t ensure(const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 256 && ((done[dev >> 6].load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 256) done[dev >> 6].fetch_or(1ull << (dev & 63), std::memory_order_release);
    return e;
  }
};

inline hipError_t memset_async(void* p, int value, size_t bytes, hipStream_t st) {
  if (g_rec != nullptr) {
    rec_push(st, [=](hipStream_t s) { const hipError_t e = hipMemsetAsync(p, value, bytes, s); if (e != hipSuccess) replay_note_error((int)e); });
    return hipSuccess;
  }
  return hipMemsetAsync(p, value, bytes, st);
}

}  // namespace vct
