// Library-wide state: last error string, ABI version.
#include <stdarg.h>
#include <string.h>

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
constexpr int EV_RING = 64;
hipEvent_t g_ev[EV_RING];
bool g_ev_ready = false;
unsigned g_ev_next = 0;
}  // namespace

CPM_EXPORT int cpm_stream_fork(void* from, void* to) {
  if (!g_ev_ready) {
    for (int i = 0; i < EV_RING; ++i)
      if (hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming) != hipSuccess) return cpm::check_launch("event create");
    g_ev_ready = true;
  }
  hipEvent_t ev = g_ev[g_ev_next++ % EV_RING];
  if (hipEventRecord(ev, (hipStream_t)from) != hipSuccess) return cpm::check_launch("event record");
  if (hipStreamWaitEvent((hipStream_t)to, ev, 0) != hipSuccess) return cpm::check_launch("stream wait");
  return CPM_OK;
}
