// mrs_tg_pool.h -- caching device allocator of the library (internal).
//
// The one-call host interface (mrs_tg_solve_batch, mrs_tg_find_trajectory, mrs_tg_optimize_paths) creates a plan and
// a dozen device buffers per call; hipMalloc / hipFree cost ~0.1 ms each and hipFree synchronises the device, which
// made a 25 us solve a 1.6 ms call.  Freed blocks are kept per device and handed out again (smallest cached block that
// fits and is at most 4x the request).  Contract: a block is returned to the pool only after the stream that used it
// has been synchronised (mrs_tg_solve_batch ends with a synchronise; mrs_tg_plan_destroy synchronises the context's
// stream first), so a cached block has no work in flight and may be reused on any stream.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace mrs_tg {

hipError_t pool_alloc_bytes(void** p, size_t bytes);
void pool_free(void* p);
// hipFree every cached block (all devices); live blocks are unaffected.  Called when the last context is destroyed.
void pool_release_cached();

template <typename T>
inline hipError_t pool_alloc(T** p, size_t bytes) {
  void* v = nullptr;
  const hipError_t e = pool_alloc_bytes(&v, bytes);
  *p = static_cast<T*>(v);
  return e;
}

}  // namespace mrs_tg
