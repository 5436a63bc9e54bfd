// Shared host-side helpers for libanorag_hip.so (error reporting, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "anorag.h"

namespace anr {

// thread-local last-error text behind anr_last_error()
std::string &last_error_ref();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define ANR_HIP(expr)                                                                            \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return anr::fail(ANR_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                       __LINE__);                                                                \
  } while (0)

#define ANR_TRY(expr)       \
  do {                      \
    int _r = (expr);        \
    if (_r != ANR_OK) return _r; \
  } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }

// RAII device guard: every entry point switches to the handle's device and restores the caller's.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int device_cu_count(int device);

}  // namespace anr
