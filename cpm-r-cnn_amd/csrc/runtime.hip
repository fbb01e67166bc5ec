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

// A stream whose kernels stay off `reserve_cus` of the chip's compute units (hipExtStreamCreateWithCUMask): the
// weight-gradient stream of a data-parallel run can leave those CUs to the RCCL kernels that all-reduce finished
// gradient chunks beside the backward pass (pet/utils/parallel.py, CPM_WGRAD_RESERVE_CUS).  The reserved units are
// taken evenly from the ends of the eight 32-bit words of the mask (one word per 32 CUs).  *out: a hipStream_t the
// caller owns (cpm_stream_destroy).
CPM_EXPORT int cpm_stream_create_cu_reserve(int reserve_cus, void** out) {
  CPM_REQUIRE(out, "null pointer");
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return cpm::check_launch("device properties");
  const int cus = prop.multiProcessorCount;
  CPM_REQUIRE(reserve_cus >= 0 && reserve_cus < cus, "reserve must leave at least one compute unit");
  const int words = (cus + 31) / 32;
  uint32_t mask[32];
  CPM_REQUIRE(words <= 32, "more than 1024 compute units");
  for (int w = 0; w < words; ++w) {
    const int bits = cus - 32 * w >= 32 ? 32 : cus - 32 * w;
    mask[w] = bits == 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
  }
  for (int i = 0; i < reserve_cus; ++i) {          // round-robin over the words, highest bits first
    const int w = i % words, b = 31 - i / words;
    mask[w] &= ~(1u << b);
  }
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask) != hipSuccess) return cpm::check_launch("stream with CU mask");
  *out = (void*)st;
  return CPM_OK;
}

CPM_EXPORT int cpm_stream_destroy(void* stream) {
  if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) return cpm::check_launch("stream destroy");
  return CPM_OK;
}

// RoI counts to the host without a copy command: the kernel stores the n counts and then a sequence number into PINNED
// host memory that the device has mapped (cpm_host_device_pointer), with system-scope release in between; the host
// polls the sequence word (pet/lib/ops/roi_lists.py: Counts).  A copy + event + hipEventSynchronize costs the host a
// blit kernel, an event and a wake-up per read (60-90 us from the counts' kernel to the host's next launch, three times
// per training step); the stores land a few microseconds behind the kernel.
namespace {
__global__ void publish_counts_kernel(const int32_t* __restrict__ src, int n, int32_t* __restrict__ dst, int32_t seq) {
  const int i = threadIdx.x;
  if (i < n) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (i == 0) {
    __threadfence_system();
    __hip_atomic_store(dst + n, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace

CPM_EXPORT int cpm_host_device_pointer(void* host_pinned, void** out) {
  CPM_REQUIRE(host_pinned && out, "null pointer");
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, host_pinned, 0) != hipSuccess || !d) return cpm::check_launch("hipHostGetDevicePointer");
  *out = d;
  return CPM_OK;
}

CPM_EXPORT int cpm_publish_counts(const int32_t* counts, int n, int32_t* host_mapped, int32_t seq, void* stream) {
  CPM_REQUIRE(counts && host_mapped && n >= 1 && n <= 255, "1..255 counts, non-null pointers");
  hipLaunchKernelGGL(publish_counts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, counts, n, host_mapped, seq);
  return cpm::check_launch("publish_counts");
}
