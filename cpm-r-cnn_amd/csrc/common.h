// Shared host/device helpers for libcpmrcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cpmrcnn_hip.h"

#define CPM_EXPORT extern "C" __attribute__((visibility("default")))

namespace cpm {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return CPM_ELAUNCH;
  }
  return CPM_OK;
}

#define CPM_REQUIRE(cond, msg)                         \
  do {                                                 \
    if (!(cond)) {                                     \
      cpm::set_error("%s: %s", __func__, msg);         \
      return CPM_EINVAL;                               \
    }                                                  \
  } while (0)

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

constexpr int WAVE = 64;

}  // namespace cpm
