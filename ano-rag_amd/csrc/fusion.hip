// Host side of anr_fuse_lists (include/anorag.h); the kernel lives in fusion_kernels.hpp.
#include "fusion_kernels.hpp"

using namespace anr;

extern "C" int anr_fuse_lists(int32_t device, int32_t method, int64_t nq, const int64_t *ids_host,
                              const double *scores_host, const int64_t *offs_host, const double *weights,
                              double rrf_k, int32_t pool, int64_t *out_ids, double *out_final, double *out_src,
                              int32_t *out_count) {
  if (nq < 0 || pool <= 0 || !offs_host || !weights || !out_ids || !out_final || !out_src || !out_count)
    return fail(ANR_EINVAL, "bad argument");
  if (method != 0 && method != 1) return fail(ANR_EINVAL, "method must be 0 (linear) or 1 (rrf)");
  if (nq == 0) return ANR_OK;
  const int64_t total = offs_host[nq * 5 - 1];
  for (int64_t q = 0; q < nq; ++q) {
    const int64_t *o = offs_host + q * 5;
    if (o[4] - o[0] > kFuseMax)
      return fail(ANR_EINVAL, "query %lld has %lld list entries; the fused kernel handles at most %d", (long long)q,
                  (long long)(o[4] - o[0]), kFuseMax);
    for (int s = 0; s < 4; ++s)
      if (o[s + 1] < o[s]) return fail(ANR_EINVAL, "offsets must be non-decreasing");
  }
  if (total > 0 && (!ids_host || !scores_host)) return fail(ANR_EINVAL, "null list pointers");
  DeviceGuard g(device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", device);
  int64_t *d_ids = nullptr, *d_offs = nullptr, *d_oid = nullptr;
  double *d_sc = nullptr, *d_of = nullptr, *d_os = nullptr;
  int *d_cnt = nullptr;
  int rc = ANR_OK;
  auto cleanup = [&]() {
    if (d_ids) (void)hipFree(d_ids);
    if (d_offs) (void)hipFree(d_offs);
    if (d_oid) (void)hipFree(d_oid);
    if (d_sc) (void)hipFree(d_sc);
    if (d_of) (void)hipFree(d_of);
    if (d_os) (void)hipFree(d_os);
    if (d_cnt) (void)hipFree(d_cnt);
  };
#define FUSE_HIP(x)                                                          \
  do {                                                                       \
    hipError_t _e = (x);                                                     \
    if (_e != hipSuccess) {                                                  \
      rc = fail(ANR_EHIP, "%s failed: %s", #x, hipGetErrorString(_e));      \
      cleanup();                                                             \
      return rc;                                                             \
    }                                                                        \
  } while (0)
  const size_t nt = (size_t)(total > 0 ? total : 1);
  FUSE_HIP(hipMalloc((void **)&d_ids, nt * 8));
  FUSE_HIP(hipMalloc((void **)&d_sc, nt * 8));
  FUSE_HIP(hipMalloc((void **)&d_offs, (size_t)nq * 5 * 8));
  FUSE_HIP(hipMalloc((void **)&d_oid, (size_t)nq * pool * 8));
  FUSE_HIP(hipMalloc((void **)&d_of, (size_t)nq * pool * 8));
  FUSE_HIP(hipMalloc((void **)&d_os, (size_t)nq * pool * 4 * 8));
  FUSE_HIP(hipMalloc((void **)&d_cnt, (size_t)nq * 4));
  if (total > 0) {
    FUSE_HIP(hipMemcpy(d_ids, ids_host, (size_t)total * 8, hipMemcpyHostToDevice));
    FUSE_HIP(hipMemcpy(d_sc, scores_host, (size_t)total * 8, hipMemcpyHostToDevice));
  }
  FUSE_HIP(hipMemcpy(d_offs, offs_host, (size_t)nq * 5 * 8, hipMemcpyHostToDevice));
  FuseParams p{};
  p.method = method;
  p.ids = d_ids;
  p.scores = d_sc;
  p.offs = d_offs;
  for (int s = 0; s < 4; ++s) p.w[s] = weights[s];
  p.rrf_k = rrf_k;
  p.pool = pool;
  p.out_ids = d_oid;
  p.out_final = d_of;
  p.out_src = d_os;
  p.out_count = d_cnt;
  FUSE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fuse<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)sizeof(FuseShared)));
  hipLaunchKernelGGL(k_fuse<false>, dim3((unsigned)nq), dim3(1024), sizeof(FuseShared), 0, p);
  FUSE_HIP(hipGetLastError());
  FUSE_HIP(hipMemcpy(out_ids, d_oid, (size_t)nq * pool * 8, hipMemcpyDeviceToHost));
  FUSE_HIP(hipMemcpy(out_final, d_of, (size_t)nq * pool * 8, hipMemcpyDeviceToHost));
  FUSE_HIP(hipMemcpy(out_src, d_os, (size_t)nq * pool * 4 * 8, hipMemcpyDeviceToHost));
  FUSE_HIP(hipMemcpy(out_count, d_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost));
#undef FUSE_HIP
  cleanup();
  return ANR_OK;
}
