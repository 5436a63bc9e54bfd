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

const char *anr_version(void) { return "anorag-hip 0.2 (gfx950)"; }

int anr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int anr_device_malloc(int32_t device, int64_t bytes, void **out) {
  if (!out || bytes < 0) return anr::fail(ANR_EINVAL, "bad argument");
  *out = nullptr;
  anr::DeviceGuard g(device);
  if (!g.ok) return anr::fail(ANR_EHIP, "hipSetDevice(%d) failed", device);
  ANR_HIP(hipMalloc(out, (size_t)(bytes > 0 ? bytes : 8)));
  return ANR_OK;
}

int anr_device_free(int32_t device, void *ptr) {
  if (!ptr) return ANR_OK;
  anr::DeviceGuard g(device);
  ANR_HIP(hipFree(ptr));
  return ANR_OK;
}

int anr_device_copy(int32_t device, void *dst, const void *src, int64_t bytes, int32_t kind) {
  if (bytes < 0 || (bytes > 0 && (!dst || !src))) return anr::fail(ANR_EINVAL, "bad argument");
  if (kind < 0 || kind > 2) return anr::fail(ANR_EINVAL, "kind must be 0 (host to device), 1 (device to host) or 2 (device to device)");
  if (bytes == 0) return ANR_OK;
  anr::DeviceGuard g(device);
  if (!g.ok) return anr::fail(ANR_EHIP, "hipSetDevice(%d) failed", device);
  static const hipMemcpyKind k[3] = {hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice};
  ANR_HIP(hipMemcpy(dst, src, (size_t)bytes, k[kind]));
  return ANR_OK;
}

}  // extern "C"
