// Library-wide state: last error string, ABI version.
#include <stdarg.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "common.h"

namespace cpm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace cpm

CPM_EXPORT int cpm_abi_version(void) { return 1; }
CPM_EXPORT const char* cpm_last_error(void) { return cpm::g_err; }

// ---- stream ordering without framework objects ------------------------------------------------------------------
// `to` waits for everything queued on `from` so far.  Events come from a small ring: a wait refers to the record that
// preceded it, so re-recording a ring slot later does not disturb waits that are already queued.
namespace {
constexpr int EV_RING = 64, MAX_DEV = 16;
struct Ring {
  hipEvent_t ev[EV_RING];
  std::atomic<bool> ready{false};
  std::atomic<unsigned> next{0};
  std::mutex init;
};
Ring g_ring[MAX_DEV];      // one ring per device: autograd runs one backward thread per device, and an event belongs to
}  // namespace            // the device that was current when it was created

CPM_EXPORT int cpm_stream_fork(void* from, void* to) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) {
    cpm::set_error("cpm_stream_fork: no current device (or more than %d devices in one process)", MAX_DEV);
    return CPM_EINVAL;
  }
  Ring& r = g_ring[dev];
  if (!r.ready.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(r.init);
    if (!r.ready.load(std::memory_order_relaxed)) {
      for (int i = 0; i < EV_RING; ++i)
        if (hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming) != hipSuccess) return cpm::check_launch("event create");
      r.ready.store(true, std::memory_order_release);
    }
  }
  hipEvent_t ev = r.ev[r.next.fetch_add(1, std::memory_order_relaxed) % EV_RING];
  if (hipEventRecord(ev, (hipStream_t)from) != hipSuccess) return cpm::check_launch("event record");
  if (hipStreamWaitEvent((hipStream_t)to, ev, 0) != hipSuccess) return cpm::check_launch("stream wait");
  return CPM_OK;
}
