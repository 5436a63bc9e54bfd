// anr_fuse_rrf_long (include/anorag.h): weighted RRF (retrieval/hybrid_search.py:60-72) over lists of ANY length — the
// case anr_fuse_lists (4096 entries per query, LDS-resident) and anr_fuse_dense (one full-corpus source) leave: two or
// three long lists, e.g. a dense score and a bm25 score for every note.  The reference ranks each list by a stable
// descending sort; here that sort is the stable device radix sort of radix_sort.hip, per source and query, over the list IN ITS
// OWN ORDER (so equal scores keep their list positions, whatever the id numbering):
//   keys  = order-preserving 64-bit image of the score, values = the entries' ids
//   rank_s[id at sorted position r] = r + 1                         (ids outside the list keep rank 0 = absent)
//   final(i) = ((w_d/(k+r_d) + w_b/(k+r_b)) + w_g/(k+r_g)) + w_p*s_p   over the sources holding i, in that order (float64,
//              no contraction: bit-identical to the Python loop); ids held by no dense / bm25 / graph source are dropped
//   order : final descending, ties by the reference's ranks-dict insertion order = (first source holding the id, its rank
//           there) — two more stable sorts: by that tie key ascending, then by the final descending
// This is the completeness path of HybridSearcher.fuse (two or three lists longer than the LDS kernel holds, method rrf);
// it is correct first and costs a handful of N-wide sorts per query.  One long source goes through anr_fuse_dense, which
// counts the ranks it needs in a single streaming pass.

#include <algorithm>
#include <cstring>

#include "fusion_kernels.hpp"

namespace anr {

__device__ __forceinline__ unsigned long long ra_d2ord(double v) {  // monotone double -> u64, > 0 for every non-NaN
  if (v == 0.0) v = 0.0;                                            // -0.0 == +0.0
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__global__ void k_rl_keys(const double *scores, const int64_t *ids, int64_t len, unsigned long long *keys, unsigned *vals,
                          double *score_of) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= len) return;
  const double v = scores[p];
  keys[p] = v == v ? ra_d2ord(v) : 1ull;  // (a NaN score ranks below every number)
  vals[p] = (unsigned)ids[p];
  score_of[ids[p]] = v;                    // raw score by id, for the emitted entries
}

__global__ void k_rl_ranks(const unsigned *ids_sorted, int64_t len, unsigned *rank) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < len) rank[ids_sorted[r]] = (unsigned)(r + 1);
}

__global__ void k_rl_scatter(const double *scores, const int64_t *ids, int64_t len, double *score_of) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < len) score_of[ids[p]] = scores[p];
}

struct RaFuse {
  const unsigned *rank[3];   // [n] ranks of the ids in the source, 0 = absent; or nullptr
  const double *path;        // [n] path scores by id, NaN = absent; or nullptr
  double w[4];
  double rrf_k;
  int64_t n;
};

__device__ __forceinline__ bool ra_final(const RaFuse &p, int64_t i, double &f, unsigned long long &tie) {
  f = 0.0;
  bool any = false;
  tie = ~0ull;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const unsigned r = p.rank[s] ? p.rank[s][i] : 0u;
    if (r) {
      f += p.w[s] / (p.rrf_k + (double)r);
      if (!any) tie = ((unsigned long long)s << 40) | (unsigned long long)r;
      any = true;
    }
  }
  if (any && p.path) {
    const double pv = p.path[i];
    if (pv == pv) f += p.w[3] * pv;
  }
  return any;
}

__global__ void k_ra_fuse(RaFuse p, unsigned long long *fkey, unsigned long long *tiekey, unsigned *iota) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  double f;
  unsigned long long tie;
  const bool any = ra_final(p, i, f, tie);
  fkey[i] = any ? (f == f ? ra_d2ord(f) : 1ull) : 0ull;  // NaN finals sort below every number, above the dropped ids
  tiekey[i] = tie;
  iota[i] = (unsigned)i;
}

__global__ void k_ra_gather(const unsigned long long *fkey, const unsigned *ids, int64_t n, unsigned long long *out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) out[j] = fkey[ids[j]];
}

__global__ void k_ra_emit(RaFuse p, const unsigned long long *keys_sorted, const unsigned *ids_sorted, const double *const *arr,
                          int pool, int64_t *out_ids, double *out_final, double *out_src, int *out_count) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= pool) return;
  const double nan = __builtin_nan("");
  double *os = out_src + (int64_t)j * 4;
  os[0] = os[1] = os[2] = os[3] = nan;
  const bool live = j < p.n && keys_sorted[j] != 0ull;
  if (live) {
    const int64_t i = ids_sorted[j];
    double f;
    unsigned long long tie;
    (void)ra_final(p, i, f, tie);
    out_ids[j] = i;
    out_final[j] = f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (arr[s]) os[s] = arr[s][i];  // NaN where the id is absent from the source, as the list form reports it
  } else {
    out_ids[j] = -1;
    out_final[j] = 0.0;
  }
  // the count: live entries form a prefix (dropped ids sort last)
  const bool next_live = (j + 1 < pool) && (j + 1 < p.n) && keys_sorted[j + 1] != 0ull;
  if (live && !next_live) *out_count = j + 1;
  if (j == 0 && !live) *out_count = 0;
}

}  // namespace anr

using namespace anr;

extern "C" int anr_fuse_rrf_long(int32_t device, int64_t nq, const int64_t *ids_host, const double *scores_host,
                                 const int64_t *offs_host, int64_t n, const double *weights, double rrf_k, int32_t pool,
                                 int64_t *out_ids, double *out_final, double *out_src, int32_t *out_count) {
  if (nq < 0 || !offs_host || !weights || pool <= 0 || !out_ids || !out_final || !out_src || !out_count)
    return fail(ANR_EINVAL, "bad argument");
  if (n <= 0 || n > 0x7fffffffLL) return fail(ANR_EINVAL, "the id universe must hold 1 .. 2^31-1 ids");
  if (nq == 0) return ANR_OK;
  const int64_t total = offs_host[nq * 5 - 1];
  int64_t longest = 1;
  for (int64_t q = 0; q < nq; ++q) {
    const int64_t *o = offs_host + q * 5;
    for (int s = 0; s < 4; ++s) {
      if (o[s + 1] < o[s]) return fail(ANR_EINVAL, "offsets must be non-decreasing");
      longest = std::max(longest, o[s + 1] - o[s]);
    }
  }
  if (longest > n) return fail(ANR_EINVAL, "a list holds more entries than the id universe (ids must be unique inside a list)");
  if (total > 0 && (!ids_host || !scores_host)) return fail(ANR_EINVAL, "null list pointers");
  for (int64_t e = 0; e < total; ++e)
    if (ids_host[e] < 0 || ids_host[e] >= n) return fail(ANR_EINVAL, "id %lld outside the universe", (long long)ids_host[e]);
  if (device < 0 || device >= kFuseMaxDevices) return fail(ANR_EINVAL, "device %d out of range", device);
  DeviceGuard g(device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", device);
  static FuseArena arenas[kFuseMaxDevices];
  FuseArena &ar = arenas[device];
  std::lock_guard<std::mutex> lock(ar.mu);
  hipStream_t st = nullptr;
  const size_t temp_bytes = radix_sort_temp_bytes(n);
  Carve dc, hc;
  const size_t N = (size_t)n, T = (size_t)(total > 0 ? total : 1);
  const size_t d_ids = dc.take(8 * T), d_sc = dc.take(8 * T);
  const size_t d_A = dc.take(8 * N), d_B = dc.take(8 * N), d_E = dc.take(8 * N), d_C = dc.take(4 * N), d_D = dc.take(4 * N),
               d_F = dc.take(4 * N), d_R = dc.take(12 * N), d_S = dc.take(32 * N), d_T = dc.take(temp_bytes),
               d_ptr = dc.take(4 * sizeof(void *));
  const size_t out_bytes = (size_t)pool * 48 + 8;
  const size_t d_out = dc.take(out_bytes);
  const size_t h_in = hc.take(16 * T), h_out = hc.take(out_bytes);
  ANR_TRY(ar.reserve(dc.off, hc.off));
  char *D = ar.dev, *H = ar.host;
  if (total > 0) {
    std::memcpy(H + h_in, ids_host, (size_t)total * 8);
    std::memcpy(H + h_in + 8 * T, scores_host, (size_t)total * 8);
    ANR_HIP(hipMemcpyAsync(D + d_ids, H + h_in, (size_t)total * 8, hipMemcpyHostToDevice, st));
    ANR_HIP(hipMemcpyAsync(D + d_sc, H + h_in + 8 * T, (size_t)total * 8, hipMemcpyHostToDevice, st));
  }
  const int64_t *ids = reinterpret_cast<const int64_t *>(D + d_ids);
  const double *sc = reinterpret_cast<const double *>(D + d_sc);
  unsigned long long *A = reinterpret_cast<unsigned long long *>(D + d_A), *B = reinterpret_cast<unsigned long long *>(D + d_B),
                     *E = reinterpret_cast<unsigned long long *>(D + d_E);
  unsigned *C = reinterpret_cast<unsigned *>(D + d_C), *Dv = reinterpret_cast<unsigned *>(D + d_D),
           *F = reinterpret_cast<unsigned *>(D + d_F), *R = reinterpret_cast<unsigned *>(D + d_R);
  double *S = reinterpret_cast<double *>(D + d_S);
  void *temp = D + d_T;
  const unsigned grid = (unsigned)ceil_div(n, 256);
  for (int64_t q = 0; q < nq; ++q) {
    const int64_t *o = offs_host + q * 5;
    RaFuse fp{};
    const double *arr_q[4];
    for (int s = 0; s < 4; ++s) fp.w[s] = weights[s];
    fp.rrf_k = rrf_k;
    fp.n = n;
    ANR_HIP(hipMemsetAsync(R, 0, 12 * N, st));
    ANR_HIP(hipMemsetAsync(S, 0xff, 32 * N, st));  // all-ones doubles are NaN: "absent"
    for (int s = 0; s < 4; ++s) {
      const int64_t len = o[s + 1] - o[s];
      double *score_of = S + (size_t)s * N;
      arr_q[s] = len > 0 ? score_of : nullptr;
      if (len == 0) continue;
      const unsigned gl = (unsigned)ceil_div(len, 256);
      if (s == 3) {
        hipLaunchKernelGGL(k_rl_scatter, dim3(gl), dim3(256), 0, st, sc + o[s], ids + o[s], len, score_of);
        fp.path = score_of;
        continue;
      }
      unsigned *rank = R + (size_t)s * N;
      hipLaunchKernelGGL(k_rl_keys, dim3(gl), dim3(256), 0, st, sc + o[s], ids + o[s], len, A, C, score_of);
      ANR_TRY(radix_sort_pairs_u64(temp, A, B, C, Dv, len, true, st));
      hipLaunchKernelGGL(k_rl_ranks, dim3(gl), dim3(256), 0, st, Dv, len, rank);
      fp.rank[s] = rank;
    }
    // finals and tie keys of every id; order by (final desc, tie asc): stable sort by tie, then by final
    hipLaunchKernelGGL(k_ra_fuse, dim3(grid), dim3(256), 0, st, fp, A, B, C);
    ANR_TRY(radix_sort_pairs_u64(temp, B, E, C, Dv, n, false, st));                                  // ids by tie: Dv
    hipLaunchKernelGGL(k_ra_gather, dim3(grid), dim3(256), 0, st, A, Dv, n, B);                      // their finals: B
    ANR_TRY(radix_sort_pairs_u64(temp, B, E, Dv, F, n, true, st));                                   // final order: F
    ANR_HIP(hipMemcpyAsync(D + d_ptr, arr_q, sizeof arr_q, hipMemcpyHostToDevice, st));
    int64_t *o_ids = reinterpret_cast<int64_t *>(D + d_out);
    double *o_fin = reinterpret_cast<double *>(D + d_out + (size_t)pool * 8);
    double *o_src = reinterpret_cast<double *>(D + d_out + (size_t)pool * 16);
    int *o_cnt = reinterpret_cast<int *>(D + d_out + (size_t)pool * 48);
    hipLaunchKernelGGL(k_ra_emit, dim3((unsigned)ceil_div(pool, 256)), dim3(256), 0, st, fp, E, F,
                       reinterpret_cast<const double *const *>(D + d_ptr), pool, o_ids, o_fin, o_src, o_cnt);
    ANR_HIP(hipGetLastError());
    ANR_HIP(hipMemcpyAsync(H + h_out, D + d_out, out_bytes, hipMemcpyDeviceToHost, st));
    ANR_HIP(hipStreamSynchronize(st));  // (arr_q on the stack and the pinned block are reused by the next query)
    std::memcpy(out_ids + q * pool, H + h_out, (size_t)pool * 8);
    std::memcpy(out_final + q * pool, H + h_out + (size_t)pool * 8, (size_t)pool * 8);
    std::memcpy(out_src + q * pool * 4, H + h_out + (size_t)pool * 16, (size_t)pool * 32);
    std::memcpy(out_count + q, H + h_out + (size_t)pool * 48, 4);
  }
  return ANR_OK;
}
