// Error plumbing + trivial introspection entry points of the C ABI (include/anorag.h).
#include "common.hpp"

#include <mutex>
#include <set>
#include <utility>

namespace anr {

std::string &last_error_ref() {
  static thread_local std::string s;
  return s;
}

int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}

int device_cu_count(int device) {
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0)
    n = 256;  // MI355X
  return n;
}

int ensure_dynamic_lds(const void *kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void *, int>> done;  // (kernel, device)
  int dev = 0;
  ANR_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  if (done.count({kernel, dev})) return ANR_OK;
  ANR_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({kernel, dev});
  return ANR_OK;
}

}  // namespace anr

extern "C" {

const char *anr_last_error(void) { return anr::last_error_ref().c_str(); }

const char *anr_version(void) { return "anorag-hip 0.1 (gfx950)"; }

int anr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

}  // extern "C"
