// mrs_tg_pool.hip -- caching device allocator (see mrs_tg_pool.h) and the per-thread kernel timer of the launchers
#include "mrs_tg_pool.h"

#include "mrs_tg_launch.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

namespace mrs_tg {

namespace {

struct Live {
  size_t bytes;
  int device;
};

std::mutex g_mutex;
std::unordered_map<void*, Live> g_live;                    // handed out
std::map<int, std::multimap<size_t, void*>> g_cached;      // per device, by rounded size
size_t g_cached_bytes = 0;
constexpr size_t kMaxCachedBytes = (size_t)8 << 30;        // beyond this, blocks go straight back to the driver

size_t round_size(size_t bytes) {
  if (bytes < 256) return 256;
  if (bytes <= ((size_t)1 << 20)) {  // next power of two up to 1 MiB
    size_t r = 256;
    while (r < bytes) r <<= 1;
    return r;
  }
  const size_t mib = (size_t)1 << 20;
  return (bytes + mib - 1) / mib * mib;
}

}  // namespace

// MRS_TG_POOL_POISON=1 fills every block handed out with 0xFF bytes (NaN as a double, -1 as an int): a test-suite run
// under it shows any read of memory that this library did not write first.
static hipError_t poison(void* p, size_t bytes) {
  static const bool on = std::getenv("MRS_TG_POOL_POISON") != nullptr;
  if (!on) return hipSuccess;
  hipError_t e = hipMemset(p, 0xFF, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  return e;
}

hipError_t pool_alloc_bytes(void** p, size_t bytes) {
  *p = nullptr;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const size_t want = round_size(bytes);
  {
    std::lock_guard<std::mutex> lk(g_mutex);
    auto& bucket = g_cached[dev];
    auto it = bucket.lower_bound(want);
    if (it != bucket.end() && it->first <= 4 * want) {
      *p = it->second;
      g_live[*p] = Live{it->first, dev};
      g_cached_bytes -= it->first;
      const size_t got = it->first;
      bucket.erase(it);
      return poison(*p, got);
    }
  }
  e = hipMalloc(p, want);
  if (e != hipSuccess) {  // give cached memory back to the driver and retry once
    pool_release_cached();
    e = hipMalloc(p, want);
    if (e != hipSuccess) return e;
  }
  {
    std::lock_guard<std::mutex> lk(g_mutex);
    g_live[*p] = Live{want, dev};
  }
  return poison(*p, want);
}

void pool_free(void* p) {
  if (!p) return;
  std::unique_lock<std::mutex> lk(g_mutex);
  auto it = g_live.find(p);
  if (it == g_live.end()) {  // not ours (should not happen): hand it to the driver
    lk.unlock();
    (void)hipFree(p);
    return;
  }
  const Live info = it->second;
  g_live.erase(it);
  if (g_cached_bytes + info.bytes > kMaxCachedBytes) {
    lk.unlock();
    (void)hipFree(p);
    return;
  }
  g_cached[info.device].emplace(info.bytes, p);
  g_cached_bytes += info.bytes;
}

void pool_release_cached() {
  std::map<int, std::multimap<size_t, void*>> take;
  {
    std::lock_guard<std::mutex> lk(g_mutex);
    take.swap(g_cached);
    g_cached_bytes = 0;
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (auto& [dev, bucket] : take) {
    (void)hipSetDevice(dev);
    for (auto& [bytes, ptr] : bucket) (void)hipFree(ptr);
  }
  (void)hipSetDevice(cur);
}

// ---- per-dispatch timing (mrs_tg_launch.h): armed by the ABI layer for the kernel it wants timed, consumed by the
// launcher of that kernel; per thread, because contexts are driven from one thread each
static thread_local KernelTimer t_kernel_timer;
static thread_local bool t_shared_device = false;

void set_shared_device_hint(bool on) { t_shared_device = on; }
bool shared_device_hint() { return t_shared_device; }
static thread_local bool t_constrained_slots = false;
void set_constrained_slots_hint(bool on) { t_constrained_slots = on; }
bool constrained_slots_hint() { return t_constrained_slots; }
static thread_local bool t_moving_starts = false;
bool moving_starts_hint() { return t_moving_starts; }
void set_moving_starts_hint(bool on) { t_moving_starts = on; }
static thread_local bool t_dry_run = false;
bool dry_run() { return t_dry_run; }
void set_dry_run(bool on) { t_dry_run = on; }


void set_kernel_timer(hipEvent_t start, hipEvent_t stop) {
  t_kernel_timer.start = start;
  t_kernel_timer.stop = stop;
}

// ---- kernel trace (mrs_tg_kernel_trace): names are string literals of the launch macros, kept by pointer
static constexpr int kTraceCap = 32;
static thread_local const char* t_trace[kTraceCap];
static thread_local unsigned t_trace_n = 0;  // launches noted since the last reset

void note_kernel(const char* name) {
  t_trace[t_trace_n % kTraceCap] = name;
  ++t_trace_n;
}

void kernel_trace_reset() { t_trace_n = 0; }

int kernel_trace(const char** names_out, int capacity) {
  unsigned kept = t_trace_n < (unsigned)kTraceCap ? t_trace_n : (unsigned)kTraceCap;
  if (capacity >= 0 && kept > (unsigned)capacity) kept = (unsigned)capacity;  // a small array receives the NEWEST names
  const unsigned first = t_trace_n - kept;
  int n = 0;
  for (unsigned k = first; k < t_trace_n && n < capacity; ++k) names_out[n++] = t_trace[k % kTraceCap];
  return n;
}

KernelTimer take_kernel_timer() {
  const KernelTimer k = t_kernel_timer;
  t_kernel_timer = KernelTimer{};
  return k;
}

}  // namespace mrs_tg
