// Host side of anr_fuse_lists (include/anorag.h); the kernel lives in fusion_kernels.hpp.
#include <cstring>

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
  if (device < 0 || device >= kFuseMaxDevices) return fail(ANR_EINVAL, "device %d out of range", device);
  DeviceGuard g(device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", device);
  // One device block and one pinned block, kept across calls; ONE upload [ids | scores | offsets], ONE download
  // [ids | final | per-source | count].  (Seven hipMalloc / hipFree pairs and seven synchronous pageable copies per
  // call were most of a single query's 480 us through HybridSearcher.fuse.)
  static FuseArena arenas[kFuseMaxDevices];
  FuseArena &ar = arenas[device];
  std::lock_guard<std::mutex> lock(ar.mu);
  Carve dc, hc;
  const size_t nt = (size_t)total;
  const size_t up_bytes = nt * 16 + (size_t)nq * 5 * 8;
  const size_t o_fin = (size_t)nq * pool * 8, o_src = 2 * o_fin, o_cnt = o_src + (size_t)nq * pool * 32,
               out_bytes = o_cnt + (size_t)nq * 4;
  const size_t d_up = dc.take(up_bytes), d_out = dc.take(out_bytes);
  const size_t h_up = hc.take(up_bytes), h_out = hc.take(out_bytes);
  ANR_TRY(ar.reserve(dc.off, hc.off));
  char *D = ar.dev, *H = ar.host;
  if (nt) {
    std::memcpy(H + h_up, ids_host, nt * 8);
    std::memcpy(H + h_up + nt * 8, scores_host, nt * 8);
  }
  std::memcpy(H + h_up + nt * 16, offs_host, (size_t)nq * 5 * 8);
  hipStream_t st = nullptr;
  ANR_HIP(hipMemcpyAsync(D + d_up, H + h_up, up_bytes, hipMemcpyHostToDevice, st));
  FuseParams p{};
  p.method = method;
  p.ids = reinterpret_cast<const int64_t *>(D + d_up);
  p.scores = reinterpret_cast<const double *>(D + d_up + nt * 8);
  p.offs = reinterpret_cast<const int64_t *>(D + d_up + nt * 16);
  for (int s = 0; s < 4; ++s) p.w[s] = weights[s];
  p.rrf_k = rrf_k;
  p.pool = pool;
  p.out_ids = reinterpret_cast<int64_t *>(D + d_out);
  p.out_final = reinterpret_cast<double *>(D + d_out + o_fin);
  p.out_src = reinterpret_cast<double *>(D + d_out + o_src);
  p.out_count = reinterpret_cast<int *>(D + d_out + o_cnt);
  ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_fuse<false>), (int)sizeof(FuseShared)));
  hipLaunchKernelGGL(k_fuse<false>, dim3((unsigned)nq), dim3(1024), sizeof(FuseShared), st, p);
  ANR_HIP(hipGetLastError());
  ANR_HIP(hipMemcpyAsync(H + h_out, D + d_out, out_bytes, hipMemcpyDeviceToHost, st));
  ANR_HIP(hipStreamSynchronize(st));
  std::memcpy(out_ids, H + h_out, o_fin);
  std::memcpy(out_final, H + h_out + o_fin, o_fin);
  std::memcpy(out_src, H + h_out + o_src, (size_t)nq * pool * 32);
  std::memcpy(out_count, H + h_out + o_cnt, (size_t)nq * 4);
  return ANR_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Candidate-level fusion of QueryProcessor: the scoring loops of _hybrid_search
// (query/query_processor.py:3703-3760: linear / rrf with 0-based ranks, guardrail multipliers) and of
// _enhanced_hybrid_search_v2 (:1104-1143), one workgroup per query, float64 in the reference's order of operations,
// followed by the stable descending sort the reference applies (:3762 / :1146).  Text matching (which candidate
// misses the must-have terms, how many boost entities / predicates it contains, the section / lexical penalties)
// stays host work and arrives as per-candidate flags, counts and multipliers.
// ------------------------------------------------------------------------------------------------------------
namespace anr {

struct CandParams {
  int mode;                 // 0 linear, 1 rrf, 2 enhanced v2
  const int64_t *offs;      // [nq + 1] candidate ranges
  const double *a;          // vector (dense) similarity per candidate
  const double *b;          // bm25 (sparse) score per candidate
  const int32_t *flags;     // bit 0: misses the must-have terms (modes 0/1) / does NOT satisfy them (mode 2)
  const int32_t *n_ent;     // modes 0: boost entities found in the candidate (vector_score *= 1.2 each)
  const int32_t *n_pred;    // mode 0: boost predicates found (bm25_score *= 1.3 each)
  const double *mult;       // mode 2: [4] per candidate: section, lexical, entity, predicate multipliers (1.0 = not applied)
  double wa, wb, rrf_k, noise;
  double *score;            // out, per candidate
  int32_t *order;           // out, per query range: candidate indices (relative to the range) best first, stable
};

__global__ __launch_bounds__(256) void k_fuse_candidates(CandParams p) {
  const int q = blockIdx.x, tid = threadIdx.x;
  const int64_t lo = p.offs[q];
  const int n = (int)(p.offs[q + 1] - lo);
  const double *a = p.a + lo, *b = p.b + lo;
  double *sc = p.score + lo;
  for (int i = tid; i < n; i += 256) {
    double f;
    const bool miss = p.flags && (p.flags[lo + i] & 1);
    if (p.mode == 0) {
      double v = a[i], s = b[i];
      if (miss) s *= 0.1;
      const int ne = p.n_ent ? p.n_ent[lo + i] : 0, np = p.n_pred ? p.n_pred[lo + i] : 0;
      for (int e = 0; e < ne; ++e) v *= 1.2;
      for (int e = 0; e < np; ++e) s *= 1.3;
      f = p.wa * v + p.wb * s;
    } else if (p.mode == 1) {
      // 0-based ranks of a stable descending sort (sorted(range(n), key=..., reverse=True), :3737-3740)
      int ra = 0, rb = 0;
      const double ai = a[i], bi = b[i];
      for (int j = 0; j < n; ++j) {
        ra += (a[j] > ai || (a[j] == ai && j < i)) ? 1 : 0;
        rb += (b[j] > bi || (b[j] == bi && j < i)) ? 1 : 0;
      }
      f = p.wa / (p.rrf_k + (double)ra) + p.wb / (p.rrf_k + (double)rb);
      if (miss) f *= 0.1;
    } else {
      const double *m = p.mult + (lo + i) * 4;
      f = 1.0 * a[i] + 0.6 * b[i];
      f *= m[0];
      f *= m[1];
      if (f < p.noise && miss) f = 0.0;
      f *= m[2];
      f *= m[3];
    }
    sc[i] = f;
  }
  __syncthreads();
  // stable descending order by rank counting (ties keep the candidate order, as list.sort(reverse=True) does)
  for (int i = tid; i < n; i += 256) {
    const double si = sc[i];
    int pos = 0;
    for (int j = 0; j < n; ++j) pos += (sc[j] > si || (sc[j] == si && j < i)) ? 1 : 0;
    p.order[lo + pos] = i;
  }
}

}  // namespace anr

extern "C" int anr_fuse_candidates(int32_t device, int32_t mode, int64_t nq, const int64_t *offs_host, const double *a_host,
                                   const double *b_host, const int32_t *flags_host, const int32_t *n_ent_host,
                                   const int32_t *n_pred_host, const double *mult_host, double wa, double wb,
                                   double rrf_k, double noise, double *score_host, int32_t *order_host) {
  if (nq < 0 || !offs_host || !score_host || !order_host) return fail(ANR_EINVAL, "bad argument");
  if (mode < 0 || mode > 2) return fail(ANR_EINVAL, "mode must be 0 (linear), 1 (rrf) or 2 (enhanced v2)");
  if (nq == 0) return ANR_OK;
  const int64_t total = offs_host[nq] - offs_host[0];
  if (offs_host[0] != 0) return fail(ANR_EINVAL, "offs[0] must be 0");
  for (int64_t q = 0; q < nq; ++q)
    if (offs_host[q + 1] < offs_host[q] || offs_host[q + 1] - offs_host[q] > (1 << 20))
      return fail(ANR_EINVAL, "query %lld: bad candidate range", (long long)q);
  if (total == 0) return ANR_OK;
  if (!a_host || !b_host) return fail(ANR_EINVAL, "null score arrays");
  if (mode == 2 && !mult_host) return fail(ANR_EINVAL, "mode 2 needs the multiplier array");
  DeviceGuard g(device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", device);
  if (device < 0 || device >= kFuseMaxDevices) return fail(ANR_EINVAL, "device %d out of range", device);
  // inputs [offs | a | b | mult | flags | n_ent | n_pred] then outputs [score | order], the same layout in one pinned
  // host block and one device block kept per device: one upload, one download (it was a hipMalloc / hipFree pair, up
  // to seven synchronous pageable uploads and two downloads per call — for ~100 candidates of one query)
  const size_t o_offs = 0, o_a = round_up((nq + 1) * 8, 16), o_b = o_a + total * 8, o_m = o_b + total * 8;
  const size_t o_f = o_m + (mode == 2 ? total * 32 : 0), o_e = o_f + total * 4, o_p = o_e + total * 4;
  const size_t in_bytes = round_up((int64_t)(o_p + total * 4), 16);
  const size_t o_s = in_bytes, o_o = o_s + total * 8, bytes = o_o + total * 4;
  static FuseArena arenas[kFuseMaxDevices];
  FuseArena &ar = arenas[device];
  std::lock_guard<std::mutex> lock(ar.mu);
  ANR_TRY(ar.reserve(bytes, bytes));
  unsigned char *d = reinterpret_cast<unsigned char *>(ar.dev), *hp = reinterpret_cast<unsigned char *>(ar.host);
  std::memcpy(hp + o_offs, offs_host, (nq + 1) * 8);
  std::memcpy(hp + o_a, a_host, total * 8);
  std::memcpy(hp + o_b, b_host, total * 8);
  if (mode == 2) std::memcpy(hp + o_m, mult_host, total * 32);
  if (flags_host) std::memcpy(hp + o_f, flags_host, total * 4);
  if (n_ent_host) std::memcpy(hp + o_e, n_ent_host, total * 4);
  if (n_pred_host) std::memcpy(hp + o_p, n_pred_host, total * 4);
  hipStream_t st = nullptr;
  hipError_t e = hipMemcpyAsync(d, hp, in_bytes, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    CandParams p{};
    p.mode = mode;
    p.offs = reinterpret_cast<const int64_t *>(d + o_offs);
    p.a = reinterpret_cast<const double *>(d + o_a);
    p.b = reinterpret_cast<const double *>(d + o_b);
    p.flags = flags_host ? reinterpret_cast<const int32_t *>(d + o_f) : nullptr;
    p.n_ent = n_ent_host ? reinterpret_cast<const int32_t *>(d + o_e) : nullptr;
    p.n_pred = n_pred_host ? reinterpret_cast<const int32_t *>(d + o_p) : nullptr;
    p.mult = mode == 2 ? reinterpret_cast<const double *>(d + o_m) : nullptr;
    p.wa = wa; p.wb = wb; p.rrf_k = rrf_k; p.noise = noise;
    p.score = reinterpret_cast<double *>(d + o_s);
    p.order = reinterpret_cast<int32_t *>(d + o_o);
    hipLaunchKernelGGL(k_fuse_candidates, dim3((unsigned)nq), dim3(256), 0, st, p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(hp + o_s, d + o_s, total * 12, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess) {
    std::memcpy(score_host, hp + o_s, total * 8);
    std::memcpy(order_host, hp + o_o, total * 4);
  }
  if (e != hipSuccess) return fail(ANR_EHIP, "candidate fusion failed: %s", hipGetErrorString(e));
  return ANR_OK;
}
