// N-array score fusion behind anr_fuse_dense (include/anorag.h): HybridSearcher.fuse (reference
// retrieval/hybrid_search.py:34-103) when a source is the full-corpus score vector that
// utils/bm25_search.py:286-340 (bm25_scores) returns — one score per note, N of them — instead of a short list.
//
// BASELINE.json north_star: "linear/RRF fusion with the existing utils/bm25_search.py scores happens on-device as a
// fused elementwise+argk kernel".  An array source stands for the list [(0, a[0]), (1, a[1]), ... (N-1, a[N-1])]:
// every id present (NaN marks an absent id), list order = id order.  The other sources stay short (id, score) lists.
//
// Pipeline per batch of queries (everything float64, bit-identical finals):
//   k_fd_prep    unique ids of the short lists; K' = pool + that count; (rrf) their keys in the array, sorted
//   k_fd_max     linear: per-source maximum over the whole array (the reference max-normalises, :26-32)
//   k_fd_scan    THE kernel, HBM-bound: streams the arrays once, computes the fused value of every id from its array
//                sources ("elementwise"), and keeps the K' best per 4096-id chunk ("arg-k"): chunk 0 first (its K'-th
//                value is a strict threshold for all later ids — they rank below chunk 0's K' on ties), then all
//                other chunks, threshold-gated; a chunk that lets more than its list holds through selects its own
//                K' best in LDS (radix select) and raises a shared running threshold.  rrf with one array source
//                ranks by the raw array value and, in the same pass, counts for every short-list id how many array
//                entries beat it (its exact 1-based rank among all N) — so no sort of the N-vector is needed.
//   k_fd_build   per query: the K' best of all chunk lists, ordered; composes SHORT lists (candidates + short-list
//                ids, raw scores, whole-source maxima / ranks as overrides)
//   k_fuse<true> the reference arithmetic on those short lists (the kernel anr_fuse_lists uses; golden-pinned)
// An array source may also arrive in SPARSE form (its explicit entries, every other id 0.0 — a BM25 row): k_fs_sort +
// k_fs_stage then stand in for k_fd_max + k_fd_scan and nothing of length N is read (see "the sparse form" below).
// Exactness: an id outside the short lists has the same fused value in the stream as in the reference (all its
// terms come from arrays), so the best `pool` of those are among the stream's K' best; ids of the short lists are
// always candidates.  Ties: final desc, then the reference's order (rrf: ranks-dict insertion; linear: lower id,
// where the reference iterates a set).
#include <algorithm>
#include <type_traits>
#include <cmath>
#include <mutex>
#include <cstring>
#include <vector>

#include "fusion_kernels.hpp"

namespace anr {

constexpr int kFdChunk = 8192;      // ids per scan workgroup
constexpr int kFdThreads = 1024;
constexpr int kFdPer = kFdChunk / kFdThreads;
// the streaming scan runs TWO 512-thread workgroups per CU on 4096-id chunks (one fat workgroup per CU spent its time
// at its own barriers: its 16 waves arrive 4 us apart, and nothing overlapped the load latency of the next chunk)
constexpr int kFsChunk = 4096;
constexpr int kFsThreads = 512;
static_assert(kFsChunk / kFsThreads == kFdPer, "same ids per thread in both chunkings");
constexpr int kFdMaxSparse = 1024;  // unique short-list ids per query
constexpr int kFdMaxK = 2048;       // K' = pool + unique short-list ids

__device__ __forceinline__ unsigned long long d2ord(double v) {
  // monotone double -> u64; NaN -> 1 (below every number, above the "no threshold" value 0); -0.0 == +0.0
  if (v != v) return 1ull;
  if (v == 0.0) v = 0.0;
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord2d(unsigned long long o) {
  const unsigned long long u = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o;
  return __longlong_as_double((long long)u);
}

struct FdSrc {
  const void *arr;  // device [nq][len] or nullptr (short list / absent)
  int dtype;        // 0 float64, 1 float32
  int64_t len;      // ids >= len are absent from this source
  const double *known_max;  // device [nq_total] maxima supplied by the caller, or nullptr (k_fd_max computes them)
};

// An array source handed over in SPARSE form (anr_fuse_source.sparse_*): explicit (id, value) entries, every other id of
// the row is 0.0.  Kept out of FdSrc so that the streaming scan's code and registers do not change.
constexpr int kFsSpMax = 8192;      // explicit entries per row that k_fs_sort orders itself (one LDS bitonic network)
constexpr int kFsSpMaxBig = 65536;  // ... per row when the caller hands the rows over SORTED BY ID (anr_bm25_sparse_dev does)
struct FdSparse {
  int src;              // the source (0..2) given in this form, -1: none
  int cap;
  const unsigned *in_id;   // the caller's rows [nq_total][cap], any order — cap > kFsSpMax: ascending ids (indexed by q0 + q)
  const double *in_val;
  const int *in_cnt;
  unsigned *id;         // [nq][cap] this sub-batch's rows sorted by id (k_fs_sort), values beside them
  double *val;
  int *cnt;             // [nq]
  unsigned *err;        // one word, zeroed per sub-batch: bit 0 a row count < 0 (the producer's overflow mark), bit 1 a count
                        // above cap, bit 2 an id >= the array length, bit 3 an id listed twice, bit 4 a row of a call with
                        // cap > kFsSpMax that is not sorted by id — anr_fuse_dense returns ANR_EINVAL instead of fusing
                        // such a row as if its missing scores were 0.0
};

struct FdParams {
  int method;     // 0 linear, 1 rrf
  int r1_src;     // rrf: the one array source (0..2), ranked by raw value; -1 for linear
  int pool;
  FdSrc src[4];
  double w[4];
  double rrf_k;
  int64_t U;      // id universe of the stream: max array length
  int64_t q0;     // first query of this sub-batch in the array sources
  // short lists of the sub-batch (device copies of the caller's, same layout as anr_fuse_lists)
  const int64_t *l_ids;
  const double *l_sc;
  const int64_t *l_offs;  // [nq][5]
  // prep outputs
  int *kprime;                   // [nq]
  unsigned *su_id;               // [nq][kFdMaxSparse] unique short-list ids (ascending)
  int *su_n;                     // [nq]
  unsigned long long *sk_hi;     // [nq][kFdMaxSparse] rrf: keys of the short-list ids present in the array, descending
  unsigned *sk_id;
  int *sk_n;
  unsigned *H;                   // [nq][kFdMaxSparse + 1] rrf rank histogram
  unsigned long long *smax_ord;  // [nq][4] ordinal of the array maxima (0 = no entry)
  unsigned long long *tau0;      // [nq] strict threshold from chunk 0 (0 = none)
  unsigned long long *T;         // [nq] running (>=) threshold
  unsigned long long *c_hi;      // [nq][n_chunks][lcap] candidate keys
  unsigned *c_id;
  unsigned *c_cnt;               // [nq][n_chunks]
  int lcap, n_chunks;
  int chunk0, prefix;            // scan launch: first chunk; prefix = 1 for the chunk-0 launch
  // composed short lists for k_fuse<true>
  int64_t *o_ids;                // [nq][kFuseMax]
  double *o_sc;
  int *o_rank;
  int64_t *o_offs;               // [nq][5]
  double *o_smax;                // [nq][4]
  FdSparse sp;
  // linear, one array source: the barrier-free pass (k_fd_scan_free) flags the queries whose chunk lists overflowed;
  // the regular scan then runs for those alone
  unsigned *ovf;       // [nq + 1]: per query, and [nq] = any
  int ovf_any;         // index of the "any" word
  int only_flagged;    // regular scan: skip the items of queries that are not flagged
};

__device__ __forceinline__ bool fd_is_array(const FdParams &p, int s) { return p.src[s].arr != nullptr || s == p.sp.src; }

__device__ __forceinline__ bool fd_val(const FdSrc &s, int64_t q, int64_t i, double &v) {
  if (!s.arr || i >= s.len) return false;
  v = s.dtype == 0 ? reinterpret_cast<const double *>(s.arr)[q * s.len + i]
                   : (double)reinterpret_cast<const float *>(s.arr)[q * s.len + i];
  return v == v;
}
// the value source s holds for id i of the sub-batch's query q (dense array or sparse rows; false: absent)
__device__ __forceinline__ bool fd_val_any(const FdParams &p, int s, int q, int64_t i, double &v) {
  if (s != p.sp.src) return fd_val(p.src[s], p.q0 + q, i, v);
  if (i >= p.src[s].len) return false;
  const unsigned *ids = p.sp.id + (int64_t)q * p.sp.cap;
  int lo = 0, hi = p.sp.cnt[q];  // first entry with id >= i
  const int n = hi;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)ids[mid] < i) lo = mid + 1;
    else hi = mid;
  }
  v = 0.0;  // not listed: an implicit zero
  if (lo < n && (int64_t)ids[lo] == i) v = p.sp.val[(int64_t)q * p.sp.cap + lo];
  return v == v;
}

// One chunk's entries of a source for this thread: ids base + e * kFdThreads + tid.  The loads are UNCONDITIONAL (index
// clamped, the array / dtype tests are uniform and sit outside the element loop): a per-element `if (i < len) load`
// makes hipcc branch around every load and wait for each one before the next — eight dependent HBM round trips per
// thread, measured at 19 us per 64-KiB chunk (0.9 TB/s chip-wide).
template <int TH>
__device__ __forceinline__ void fd_load_chunk(const FdSrc &s, int64_t q, int64_t base, int tid, double (&v)[kFdPer],
                                              bool (&ok)[kFdPer]) {
  if (!s.arr) {
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) {
      v[e] = 0.0;
      ok[e] = false;
    }
    return;
  }
  const int64_t last = s.len - 1;
  if (s.dtype == 0) {
    const double *a = reinterpret_cast<const double *>(s.arr) + q * s.len;
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) {
      const int64_t i = base + e * TH + tid;
      v[e] = a[i < last ? i : last];
    }
  } else {
    const float *a = reinterpret_cast<const float *>(s.arr) + q * s.len;
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) {
      const int64_t i = base + e * TH + tid;
      v[e] = (double)a[i < last ? i : last];
    }
  }
#pragma unroll
  for (int e = 0; e < kFdPer; ++e) ok[e] = base + e * TH + tid < s.len && v[e] == v[e];
}

template <int CH>
struct FdSharedT {
  unsigned long long hi[CH];
  unsigned idx[CH];
  unsigned long long sk_hi[kFdMaxSparse];
  unsigned sk_id[kFdMaxSparse];
  unsigned H[kFdMaxSparse + 1];
  unsigned hist[256];
  unsigned long long red[2][16];
  unsigned long long bnd;  // select boundary
  unsigned cut;
  unsigned n, cnt, above, d, hd;
  int pbz[2];  // rrf: short-list keys beating a zero entry at the chunk's first id / the next chunk's
};
using FdShared = FdSharedT<kFdChunk>;   // prep / build


// The need-th largest key among the participating entries i < n (8-bit radix passes, common leading bytes skipped);
// on return *eq = entries equal to it, *need_eq = how many of those belong to the `need` largest.  All threads call.
template <typename SH, typename KeyFn, typename PartFn>
__device__ __forceinline__ unsigned long long fd_radix_kth(SH &sh, int n, unsigned need, int total_bits, KeyFn key, PartFn part,
                                           unsigned *eq, unsigned *need_eq) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  unsigned long long kmin = ~0ull, kmax = 0ull;
  unsigned np = 0;
  for (int i = tid; i < n; i += nthr)
    if (part(i)) {
      const unsigned long long k = key(i);
      kmin = k < kmin ? k : kmin;
      kmax = k > kmax ? k : kmax;
      ++np;
    }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o);
    kmin = a < kmin ? a : kmin;
    kmax = b > kmax ? b : kmax;
    np += __shfl_xor(np, o);
  }
  __syncthreads();
  if (lane == 0) {
    sh.red[0][wave] = kmin;
    sh.red[1][wave] = kmax;
    sh.hist[wave] = np;
  }
  __syncthreads();
  kmin = sh.red[0][0];
  kmax = sh.red[1][0];
  unsigned total = sh.hist[0];
  for (int w = 1; w < nthr / 64; ++w) {
    kmin = sh.red[0][w] < kmin ? sh.red[0][w] : kmin;
    kmax = sh.red[1][w] > kmax ? sh.red[1][w] : kmax;
    total += sh.hist[w];
  }
  __syncthreads();
  int bits = 0;
  while (bits < total_bits && (kmin >> (total_bits - 8 - bits)) == (kmax >> (total_bits - 8 - bits))) bits += 8;
  unsigned long long prefix = bits ? (kmax >> (total_bits - bits)) : 0ull;
  unsigned count_eq = total;  // entries matching the current prefix
  while (bits < total_bits) {
    for (int i = tid; i < 256; i += nthr) sh.hist[i] = 0;
    __syncthreads();
    const int sh_d = total_bits - 8 - bits;
    for (int i0 = 0; i0 < n; i0 += nthr) {
      const int i = i0 + tid;
      bool act = false;
      unsigned dg = 0;
      if (i < n && part(i)) {
        const unsigned long long k = key(i);
        if (bits == 0 || (k >> (total_bits - bits)) == prefix) {
          act = true;
          dg = (unsigned)(k >> sh_d) & 255u;
        }
      }
      // one wave-aggregated round (the common digit of a tie storm), the rest one atomic each
      const unsigned long long m = __ballot(act);
      if (m) {
        const int leader = __ffsll((long long)m) - 1;
        const unsigned dl = __shfl(dg, leader);
        const unsigned long long same = __ballot(act && dg == dl);
        if (lane == leader) atomicAdd(&sh.hist[dl], (unsigned)__popcll(same));
        if (act && dg != dl) atomicAdd(&sh.hist[dg], 1u);
      }
    }
    __syncthreads();
    if (wave == 0) {
      // lane l owns digits 255-4l .. 252-4l: the digit at which the count from the top reaches `need`
      const int dtop = 255 - 4 * lane;
      const unsigned h0 = sh.hist[dtop], h1 = sh.hist[dtop - 1], h2 = sh.hist[dtop - 2], h3 = sh.hist[dtop - 3];
      const unsigned own = h0 + h1 + h2 + h3;
      unsigned incl = own;
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      const unsigned excl = incl - own;
      if (excl < need && need <= incl) {
        unsigned c = excl;
        int d = dtop;
        unsigned hd = h0;
        if (c + h0 < need) { c += h0; d = dtop - 1; hd = h1;
          if (c + h1 < need) { c += h1; d = dtop - 2; hd = h2;
            if (c + h2 < need) { c += h2; d = dtop - 3; hd = h3; } } }
        sh.d = (unsigned)d;
        sh.above = c;
        sh.hd = hd;
      }
    }
    __syncthreads();
    need -= sh.above;
    prefix = (prefix << 8) | (unsigned long long)sh.d;
    count_eq = sh.hd;
    bits += 8;
    __syncthreads();
  }
  *eq = count_eq;
  *need_eq = need;
  return prefix;
}

// boundary of the K largest (hi desc, idx asc) of the n staged pairs (n > K): selected <=> hi > bnd || (hi == bnd &&
// idx <= cut).  All threads call; result in sh.bnd / sh.cut.
template <typename SH>
__device__ __forceinline__ void fd_select_boundary(SH &sh, int n, int K) {
  unsigned eq, need_eq;
  const unsigned long long B = fd_radix_kth(
      sh, n, (unsigned)K, 64, [&](int i) { return sh.hi[i]; }, [](int) { return true; }, &eq, &need_eq);
  unsigned cut = 0xffffffffu;
  if (need_eq < eq) {
    // more entries tie with the boundary value than are wanted: the need_eq smallest ids of them
    unsigned e2, n2;
    const unsigned long long r = fd_radix_kth(
        sh, n, need_eq, 32, [&](int i) { return (unsigned long long)(~sh.idx[i]); },
        [&](int i) { return sh.hi[i] == B; }, &e2, &n2);
    cut = ~(unsigned)r;
  }
  if (threadIdx.x == 0) {
    sh.bnd = B;
    sh.cut = cut;
  }
  __syncthreads();
}

// Chunk 0 of a mostly-zero vector: fewer than K positive entries, so the K-th best is 0.0 and the boundary is a cut
// through the ZEROS by id — which the radix selection above finds in ~12 passes over a 4000-fold tie (46 us per query
// on the chunk-0 launch).  Here: count the entries above / at zero and mark the zeros' ids in a bitmap (one pass), then
// the (K - positives)-th set bit is the cut.  false: the boundary is not at zero (nothing changed, use the selection).
// ids of the staged entries lie in [base, base + 8192).  All threads call; result in sh.bnd / sh.cut.
template <typename SH>
__device__ __forceinline__ bool fd_zero_boundary(SH &sh, int n, int K, int64_t base) {
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const unsigned long long kz = 0x8000000000000000ull;  // d2ord(0.0)
  for (int i = tid; i < 256; i += nthr) sh.hist[i] = 0;
  if (tid == 0) {
    sh.above = 0;
    sh.hd = 0;
  }
  __syncthreads();
  unsigned pos_n = 0, zero_n = 0;
  for (int i = tid; i < n; i += nthr) {
    const unsigned long long k = sh.hi[i];
    pos_n += k > kz ? 1u : 0u;
    if (k == kz) {
      ++zero_n;
      const unsigned r = (unsigned)((int64_t)sh.idx[i] - base);
      atomicOr(&sh.hist[r >> 5], 1u << (r & 31u));
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    pos_n += __shfl_xor(pos_n, o);
    zero_n += __shfl_xor(zero_n, o);
  }
  if (lane == 0) {
    if (pos_n) atomicAdd(&sh.above, pos_n);
    if (zero_n) atomicAdd(&sh.hd, zero_n);
  }
  __syncthreads();
  const unsigned P = sh.above, Z = sh.hd;
  if (!(P < (unsigned)K && (unsigned)K <= P + Z)) {
    __syncthreads();  // (the counters are scratch of the selection that follows)
    return false;
  }
  if (tid < 64) {  // the (K - P)-th zero in id order: lane l owns bitmap words 4l .. 4l + 3 (8192 ids)
    const unsigned need = (unsigned)K - P;
    unsigned w[4], c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[j] = sh.hist[4 * lane + j];
      c[j] = (unsigned)__popc(w[j]);
    }
    const unsigned own = c[0] + c[1] + c[2] + c[3];
    unsigned incl = own;
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    unsigned excl = incl - own;
    if (excl < need && need <= incl) {
      int j = 0;
      while (excl + c[j] < need) excl += c[j++];
      unsigned word = w[j];
      for (unsigned r = need - excl; r > 1; --r) word &= word - 1;  // drop the r - 1 lowest set bits
      sh.bnd = kz;
      sh.cut = (unsigned)(base + (int64_t)(4 * lane + j) * 32 + (__ffs(word) - 1));
    }
  }
  __syncthreads();
  return true;
}

// ---- prep: unique short-list ids, K', (rrf) their sorted keys in the array ---------------------------------
__global__ __launch_bounds__(kFdThreads) void k_fd_prep(FdParams p) {
  __shared__ unsigned long long key[2 * kFdMaxSparse];
  __shared__ unsigned ids[2 * kFdMaxSparse];
  __shared__ unsigned cnt;
  const int q = blockIdx.x, tid = threadIdx.x;
  const int64_t *off = p.l_offs + (int64_t)q * 5;
  const int m = (int)(off[4] - off[0]);  // <= kFdMaxSparse (host-checked)
  // sort the ids (with duplicates) ascending: bitonic over 1024 slots
  for (int i = tid; i < kFdMaxSparse; i += kFdThreads) ids[i] = i < m ? (unsigned)p.l_ids[off[0] + i] : 0xffffffffu;
  if (tid == 0) cnt = 0;
  __syncthreads();
  for (int k2 = 2; k2 <= kFdMaxSparse; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < kFdMaxSparse; i += kFdThreads) {
        const int x = i ^ j;
        if (x > i) {
          const unsigned a = ids[i], b = ids[x];
          if (((i & k2) == 0) ? (b < a) : (a < b)) {
            ids[i] = b;
            ids[x] = a;
          }
        }
      }
      __syncthreads();
    }
  // unique -> su_id (ascending order kept: compaction by prefix count)
  unsigned *su = p.su_id + (int64_t)q * kFdMaxSparse;
  for (int i = tid; i < m; i += kFdThreads) {
    const bool first = (i == 0) || ids[i] != ids[i - 1];
    ids[kFdMaxSparse + i] = first ? 1u : 0u;
  }
  __syncthreads();
  if (tid == 0) {
    unsigned c = 0;
    for (int i = 0; i < m; ++i)
      if (ids[kFdMaxSparse + i]) su[c++] = ids[i];
    cnt = c;
  }
  __syncthreads();
  const int mu = (int)cnt;
  if (tid == 0) {
    p.su_n[q] = mu;
    int kp = p.pool + mu;
    p.kprime[q] = kp < kFdMaxK ? kp : kFdMaxK;
  }
  if (p.method == 0 && tid < 3 && p.src[tid].arr && p.src[tid].known_max) {
    // the producer's maximum (NaN = the row has no entry): with every maximum known there is no k_fd_max launch at all
    const double m = p.src[tid].known_max[p.q0 + q];
    if (m == m) p.smax_ord[(int64_t)q * 4 + tid] = d2ord(m);
  }
  if (p.method != 1) return;
  // rrf: (key, id) of the unique ids present in the array source, descending by (value, lower id first)
  for (int i = tid; i < kFdMaxSparse; i += kFdThreads) {
    unsigned long long k = 0ull;
    unsigned id = 0xffffffffu;
    double v;
    if (i < mu && fd_val_any(p, p.r1_src, q, (int64_t)su[i], v)) {
      k = d2ord(v);
      id = su[i];
    }
    key[i] = k;  // 0 = absent, sorts last
    ids[i] = id;
  }
  __syncthreads();
  for (int k2 = 2; k2 <= kFdMaxSparse; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < kFdMaxSparse; i += kFdThreads) {
        const int x = i ^ j;
        if (x > i) {
          const unsigned long long ka = key[i], kb = key[x];
          const unsigned ia = ids[i], ib = ids[x];
          const bool b_first = kb > ka || (kb == ka && ib < ia);  // descending key, ascending id
          const bool a_first = ka > kb || (ka == kb && ia < ib);
          if (((i & k2) == 0) ? b_first : a_first) {
            key[i] = kb; key[x] = ka;
            ids[i] = ib; ids[x] = ia;
          }
        }
      }
      __syncthreads();
    }
  __shared__ int s_present;
  if (tid == 0) s_present = 0;
  __syncthreads();
  for (int i = tid; i < kFdMaxSparse; i += kFdThreads) {
    p.sk_hi[(int64_t)q * kFdMaxSparse + i] = key[i];
    p.sk_id[(int64_t)q * kFdMaxSparse + i] = ids[i];
    if (key[i] != 0ull) atomicAdd(&s_present, 1);
  }
  __syncthreads();
  if (tid == 0) p.sk_n[q] = s_present;
}

// ---- linear: per-source maximum over the arrays ----------------------------------------------------------
__global__ __launch_bounds__(kFdThreads) void k_fd_max(FdParams p) {
  // grid (x, query): workgroup x strides over the 8192-id chunks; maxima are taken as doubles (NaN never wins a `>`),
  // one d2ord and one atomic per workgroup and source
  __shared__ double s_best[kFdThreads / 64];
  __shared__ int s_any[kFdThreads / 64];
  const int q = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n8 = (p.U + kFdChunk - 1) / kFdChunk;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (!p.src[s].arr) continue;
    if (p.src[s].known_max) {  // the producer's maximum (NaN = the row has no entry)
      if (blockIdx.x == 0 && tid == 0) {
        const double m = p.src[s].known_max[p.q0 + q];
        if (m == m) p.smax_ord[(int64_t)q * 4 + s] = d2ord(m);
      }
      continue;
    }
    double best = 0.0;
    bool any = false;
    for (int64_t c = blockIdx.x; c < n8; c += gridDim.x) {
      double v[kFdPer];
      bool ok[kFdPer];
      fd_load_chunk<kFdThreads>(p.src[s], p.q0 + q, c * kFdChunk, tid, v, ok);
#pragma unroll
      for (int e = 0; e < kFdPer; ++e)
        if (ok[e]) {
          best = (!any || v[e] > best) ? v[e] : best;
          any = true;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double t = __shfl_xor(best, o);
      const bool ta = __shfl_xor((int)any, o) != 0;
      best = (ta && (!any || t > best)) ? t : best;
      any = any || ta;
    }
    __syncthreads();
    if (lane == 0) {
      s_best[wave] = best;
      s_any[wave] = any ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < kFdThreads / 64; ++w)
        if (s_any[w] && (!any || s_best[w] > best)) {
          best = s_best[w];
          any = true;
        }
      if (any) atomicMax(p.smax_ord + (int64_t)q * 4 + s, d2ord(best));
    }
  }
}

// ---- the scan ---------------------------------------------------------------------------------------------
// Persistent: two 512-thread workgroups per CU, each walking a contiguous range of (query, chunk) items; a query's
// short-list keys and its rank histogram stay in LDS across the consecutive chunks of that query and are flushed when
// the query changes.  The scan was issue- and latency-bound, not HBM-bound, so the per-chunk work is kept short and
// uniform (measured steps in DESIGN.md 5a):
//   * the NEXT chunk's values (and the running threshold) are loaded into registers before the current chunk is
//     processed; the chunk barrier waits for LDS only, so those loads stay in flight across it;
//   * ONE barrier per chunk in the steady state: the staging counter and the drain flag rotate through three slots
//     (slot c + 2 is cleared after the barrier of chunk c), and a second barrier is passed only when something was staged;
//   * values are compared as doubles; d2ord only for the rare entry that is kept or searched;
//   * rrf: zero entries (a BM25 vector is ~99.9 % zeros) are counted per wave without a search — the zero-valued
//     short-list ids inside the chunk are tracked by a pointer that advances with the chunks; the few non-zero entries of
//     a sparse wave are DEFERRED to an LDS list that all threads search together every ~100 chunks (a 7-round
//     dependent-LDS binary search inside the chunk made one wave late for the barrier every time); a wave with many
//     non-zero entries (a dense vector) searches in place.
constexpr int kFsWaves = kFsThreads / 64;
constexpr int kFsPendW = 96;     // deferred rank searches held in LDS, per wave (a private segment: no atomics)
constexpr int kFsPushMax = 8;    // a wave defers at most this many entries per chunk (more: it searches in place)
constexpr unsigned kFsDrainAt = kFsPendW - 3 * kFsPushMax;  // the drain is decided one chunk ahead, acted on one later
struct FsShared : FdSharedT<kFsChunk> {
  unsigned long long pk[kFsWaves * kFsPendW];
  unsigned pi[kFsWaves * kFsPendW];
  unsigned pw_n[kFsWaves];  // entries in each wave's segment (mirrors the wave's register copy)
  // four-slot rotations indexed by the chunk counter c % 4, so that ONE barrier per chunk orders everything:
  unsigned n4[4];      // staged entries of chunk c; slot (c + 2) % 4 is cleared after the barrier of chunk c
  unsigned drain4[4];  // "search the deferred lists after this chunk's barrier": raised for chunk c + 1 by a wave whose
                       // segment fills up during chunk c; slot (c + 3) % 4 is cleared after the barrier of chunk c
};

// the barrier-free rrf pass (k_fd_scan<1, DT, true>) stages nothing: the same fields without the 48 KiB staging arrays,
// so that four workgroups fit a CU instead of two
struct FsSharedFree : FdSharedT<1> {
  unsigned long long pk[kFsWaves * kFsPendW];
  unsigned pi[kFsWaves * kFsPendW];
  unsigned pw_n[kFsWaves];
  unsigned n4[4];
  unsigned drain4[4];
};
// (the barrier-free instantiations are launched without ensure_dynamic_lds: they must stay below the 64 KiB a kernel gets
// without asking — ADVICE r3)
static_assert(sizeof(FsSharedFree) <= 64 * 1024, "k_fd_scan<.., FREE>: FsSharedFree outgrew the default dynamic LDS limit");

// workgroup barrier that waits for this wave's LDS traffic only (global loads of the next chunk stay in flight)
__device__ __forceinline__ void fs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the chunk's values of one array: thread tid holds ids base + e * TH + tid.  Chunk base pointer in scalar registers,
// 32-bit per-thread offsets; only the array's last chunk clamps its index (64-bit index arithmetic and a clamp per
// load were a third of the scan's instructions).
template <int TH, int DT>
__device__ __forceinline__ void fd_load_raw(const FdSrc &s, int64_t q, int64_t base, int tid,
                                            unsigned long long (&v)[kFdPer]) {
  // RAW bits (an f32 array's value sits in the low word): converting here would make the prefetch wait for its data.
  // ONE code path (index clamped with a v_min, dtype a template parameter): with branches the loads landed in
  // temporaries that had to be copied at the join — a wait for the prefetched data right where it was issued.
  const int64_t room = s.len - base;  // ids of this array at or after base (may be <= 0: the array is shorter than U)
  const int lim = room >= TH * kFdPer ? TH * kFdPer - 1 : (room > 0 ? (int)room - 1 : 0);
  const int64_t row = q * s.len + (room > 0 ? base : s.len - 1);
  if (DT == 0) {
    const unsigned long long *a = reinterpret_cast<const unsigned long long *>(s.arr) + row;
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) {
      const int i = e * TH + tid;
      v[e] = a[i < lim ? i : lim];
    }
  } else {
    const unsigned *a = reinterpret_cast<const unsigned *>(s.arr) + row;
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) {
      const int i = e * TH + tid;
      v[e] = a[i < lim ? i : lim];
    }
  }
}
__device__ __forceinline__ double fd_raw_value(int dtype, unsigned long long bits) {
  return dtype == 0 ? __longlong_as_double((long long)bits) : (double)__uint_as_float((unsigned)bits);
}
// A per-lane copy of a (really uniform) index: a load through it stays in a vector register and is waited for where it
// is USED — the compiler moves a uniform load to scalar registers on the spot, which waits for it on the spot.
__device__ __forceinline__ int fd_per_lane(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

struct FsState {
  int cur_q, skn, kp;
  double smax[4];
  unsigned long long tau0;
  int z_hi;   // rrf: short-list keys at or above zero (the zero-valued ones end here)
  int zp, zc; // rrf: short-list keys beating a zero entry at the first id of chunk zc
  int par;    // chunk counter mod 4
  bool fast0; // linear: one array source, finite weights (see skip0 in fd_scan_chunk)
  bool flagged; // barrier-free rrf pass: this query's lists had already overflowed when this workgroup reached it
  bool dense_q; // barrier-free rrf pass: chunk 0's threshold is not 0.0 — the whole query is left to the regular scan
  unsigned pw; // deferred entries in this wave's segment
};

// one chunk; `raw` = the prefetched values of array source s0, T = the running threshold read with them
// FREE (rrf only): no staging and no chunk barrier — candidates go straight to the chunk's list in global memory (a
// list that would overflow flags the query for the regular scan, as in k_fd_scan_free), a wave searches its own deferred
// entries when its segment fills up, the running threshold is not consulted; the rank bookkeeping is unchanged.
template <int METHOD, int DT0, bool FREE, typename SH>
__device__ __forceinline__ void fd_scan_chunk(const FdParams &p, SH &sh, FsState &st, const int q, const int c,
                                              const int s0, const unsigned long long (&raw)[kFdPer], const unsigned long long T) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t base = (int64_t)c * kFsChunk;
  const int skn = st.skn;
  // the regular scan re-doing a flagged query after the barrier-free pass: the rank histogram may be complete already
  // (flag 1 = its lists overflowed DURING that pass, ranks done; flag 2 = left to the regular scan altogether)
  const bool ranks = METHOD == 1 && skn > 0 && !(p.only_flagged && !FREE && p.ovf[q] == 1u);
  const unsigned long long kzero = 0x8000000000000000ull;  // d2ord(0.0)
  // pb of one (key, id) pair = short-list keys (sorted descending, ties by id) that beat it: binary search, branch-free
  // (every LDS read unconditional)
  auto beaten_by = [&](unsigned long long k, int64_t i) -> int {
    int lo = 0, hi = skn;
    for (int step = skn; step > 0; step >>= 1) {  // ceil(log2(skn + 1)) rounds
      const int mid = (lo + hi) >> 1;
      const int m = mid < skn ? mid : skn - 1;
      const unsigned long long kh = sh.sk_hi[m];
      const unsigned kid = sh.sk_id[m];
      const bool open = lo < hi;
      const bool beats = kh > k || (kh == k && (int64_t)kid < i);
      lo = (open && beats) ? mid + 1 : lo;
      hi = (open && !beats) ? mid : hi;
    }
    return lo;
  };
  const unsigned long long Tu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(T >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)T);
  const bool has_t0 = st.tau0 != 0ull, has_T = !FREE && Tu != 0ull;
  const double t0d = has_t0 ? ord2d(st.tau0) : 0.0, Td = has_T ? ord2d(Tu) : 0.0;
  // do zero entries pass the thresholds?  (Almost never: chunk 0's K'-th best is >= 0 in a mostly-zero vector.)
  const bool zero_passes = (!has_t0 || 0.0 > t0d) && (!has_T || 0.0 >= Td);
  // linear with ONE array source and finite weights: a wave whose 8 x 64 raw entries are all +-0 (60 % of the waves of a
  // chunk at BM25 density) has nothing but fused values of exactly 0.0 — when zeros do not pass the thresholds it has no
  // candidate, and its share of the chunk is the barrier alone (the value arithmetic below was most of the kernel's
  // instructions: ~20 integer operations instead)
  // The same per entry ROW (64 ids, 94 % of them all-zero): its fused values are 0.0 without any arithmetic.
  bool skip0 = false;
  bool rowz[kFdPer];
#pragma unroll
  for (int e = 0; e < kFdPer; ++e) rowz[e] = false;
  if (METHOD == 0 && st.fast0 && !zero_passes) {
    skip0 = true;
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) {
      const bool nonzero = DT0 == 0 ? (raw[e] & 0x7fffffffffffffffull) != 0ull : ((unsigned)raw[e] & 0x7fffffffu) != 0u;
      rowz[e] = !__any(nonzero);
      skip0 = skip0 && rowz[e];
    }
  }
  // ---- values, in the double domain ----
  // Everything per entry is expressed as WAVE MASKS (a v_cmp writes its 64-lane result straight into a scalar pair):
  // per entry two to four compares, the rest is scalar; per-lane work only in the rare branches.  (Per-lane bit masks
  // and exec-masked `if`s cost ~450 instructions per wave and chunk, and the scan is issue-bound.)
  double f[kFdPer];
  auto ids_in_chunk = [&](const FdSrc &a) -> int {  // how many of the chunk's ids array a has
    const int64_t room = a.len - base;
    return room >= kFsChunk ? kFsChunk : (room > 0 ? (int)room : 0);
  };
  int nvalid = 0;           // METHOD 1: ids of the array in this chunk
  unsigned long long vb[kFdPer];  // METHOD 0: lanes whose entry e has a value in at least one source
  if (METHOD == 1) {
    nvalid = ids_in_chunk(p.src[s0]);
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) f[e] = fd_raw_value(DT0, raw[e]);
  } else if (!skip0) {
    // source after source, in the reference's order of operations (dense, bm25, graph, then path; an absent term
    // contributes w * 0.0 exactly as `normed[k].get(nid, 0.0)` does); a zero entry skips the f64 division (x / smax for x = +-0 is
    // +-0 with the sign of x * smax) — and a wave whose 64 entries are all zero skips it altogether
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) {
      f[e] = 0.0;
      vb[e] = 0ull;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (!p.src[s].arr) {
        if (s == 3) {
#pragma unroll
          for (int e = 0; e < kFdPer; ++e) f[e] = f[e] + p.w[3] * 0.0;
        }
        continue;
      }
      unsigned long long v[kFdPer];
      if (s == s0) {
#pragma unroll
        for (int e = 0; e < kFdPer; ++e) v[e] = raw[e];
      } else if (p.src[s].dtype == 0) {
        fd_load_raw<kFsThreads, 0>(p.src[s], p.q0 + q, base, tid, v);
      } else {
        fd_load_raw<kFsThreads, 1>(p.src[s], p.q0 + q, base, tid, v);
      }
      const double sm = s < 3 ? st.smax[s] : 0.0, w = p.w[s];
      const int nv = ids_in_chunk(p.src[s]);
      const int dt = s == s0 ? DT0 : p.src[s].dtype;
#pragma unroll
      for (int e = 0; e < kFdPer; ++e) {
        if (rowz[e]) continue;  // (uniform) f stays 0.0; the row is skipped below as well
        const double x = fd_raw_value(dt, v[e]);
        const bool ok = (nv == kFsChunk || e * kFsThreads + tid < nv) && x == x;
        vb[e] |= __ballot(ok);
        if (s < 3) {
          double r = sm < 0.0 ? -x : x;                       // the quotient of a zero entry
          if (__any(ok && x != 0.0)) r = x == 0.0 ? r : x / sm;  // (uniform branch)
          if (sm == 0.0) r = 0.0;
          const double t = f[e] + w * r;
          f[e] = ok ? t : f[e];
        } else {
          f[e] = f[e] + (ok ? w * x : w * 0.0);
        }
      }
    }
  }
  // lanes whose entry e exists (METHOD 1: in range and not NaN)
  auto valid_mask = [&](int e) -> unsigned long long {
    if (METHOD == 0) return vb[e];
    return nvalid == kFsChunk ? __ballot(f[e] == f[e]) : __ballot(e * kFsThreads + tid < nvalid && f[e] == f[e]);
  };
  // lanes whose entry e is non-zero or NaN.  rrf ranks by the raw array value, so the test runs on the raw bits (integer
  // operations instead of a float64 compare per entry); linear tests the fused value
  // (rrf: the eight masks are formed ONCE per chunk — the rank bookkeeping and the row loop below both read them)
  unsigned long long nzm1[kFdPer];
  if (METHOD == 1) {
#pragma unroll
    for (int e = 0; e < kFdPer; ++e)
      nzm1[e] = DT0 == 0 ? __ballot((raw[e] & 0x7fffffffffffffffull) != 0ull) : __ballot(((unsigned)raw[e] & 0x7fffffffu) != 0u);
  }
  auto nz_mask = [&](int e) -> unsigned long long {
    if (METHOD == 1) return nzm1[e];
    return __ballot(!(f[e] == 0.0));
  };
  // an all-zero row holds no NaN: its valid lanes are the ones inside the array
  auto range_mask = [&](int e) -> unsigned long long {
    if (METHOD == 0) return vb[e];
    return nvalid == kFsChunk ? ~0ull : __ballot(e * kFsThreads + tid < nvalid);
  };
  // lanes whose entry e passes both thresholds (for non-NaN doubles d2ord is strictly monotone with -0 == +0: exactly
  // these double comparisons)
  auto pass_mask = [&](int e) -> unsigned long long {
    unsigned long long m = valid_mask(e);
    if (has_t0) m &= __ballot(f[e] > t0d);
    if (has_T) m &= __ballot(f[e] >= Td);
    return m;
  };
  const int par = st.par;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  // ---- rrf bookkeeping for the zero entries ----
  // short-list keys [pb_z0, pb_z1) are exactly the zero-valued short-list ids that lie inside this chunk, in id order
  int pb_z0 = 0, nzid = 0;
  int64_t sid0 = 0;      // the first (usually only) such id
  unsigned cnt0 = 0, cnt1 = 0;  // zero entries of this wave with count pb_z0 / pb_z0 + 1
  unsigned wtotal = 0;   // upper bound of the wave's non-zero entries
  if (ranks) {
    if (c != st.zc) st.zp = beaten_by(kzero, base);  // first chunk of a run; afterwards the pointer just advances
    pb_z0 = st.zp;
    int pb_z1 = pb_z0;
    while (pb_z1 < st.z_hi && (int64_t)sh.sk_id[pb_z1] < base + kFsChunk) ++pb_z1;
    st.zp = pb_z1;
    st.zc = c + 1;
    nzid = pb_z1 - pb_z0;
    if (nzid > 0) sid0 = (int64_t)sh.sk_id[pb_z0];
    if (!FREE) {
#pragma unroll
      for (int e = 0; e < kFdPer; ++e) wtotal += (unsigned)__popcll(nz_mask(e));
    }
  }
  // a sparse wave defers its searches to its own LDS segment; the barrier-free pass searches in place (no workgroup waits
  // for the wave, and the pending lists' bookkeeping was a tenth of its instructions)
  const bool defer = !FREE && wtotal <= (unsigned)kFsPushMax;
  unsigned pend_free = 0;  // FREE: bit e = this lane's entry e wants its rank searched
  const int wstart = tid & ~63;
  // zero entries zb of entry row e: their rank counts, all in scalar registers unless two or more ids fall in the chunk
  auto add_zeros = [&](int e, unsigned long long zb) {
    if (nzid == 0) {
      cnt0 += (unsigned)__popcll(zb);
    } else if (nzid == 1) {
      const int64_t rel = sid0 - (base + e * kFsThreads + wstart);  // lanes <= rel hold ids <= sid0
      const unsigned long long below = rel < 0 ? 0ull : (rel >= 63 ? ~0ull : ((2ull << rel) - 1ull));
      cnt0 += (unsigned)__popcll(zb & below);
      cnt1 += (unsigned)__popcll(zb & ~below);
    } else {
      const int64_t i = base + e * kFsThreads + tid;
      unsigned pb = (unsigned)pb_z0;
      for (int j = 0; j < nzid; ++j) pb += ((int64_t)sh.sk_id[pb_z0 + j] < i) ? 1u : 0u;
      if ((zb >> lane) & 1ull) atomicAdd(&sh.H[pb], 1u);
    }
  };
  // ---- one pass over the entry rows: an all-zero row (94 % of them at BM25 density) costs three instructions ----
  if (!skip0)
#pragma unroll
  for (int e = 0; e < kFdPer; ++e) {
    const unsigned long long nzm = nz_mask(e);  // non-zero or NaN
    if (nzm == 0ull && !zero_passes) {
      if (ranks) add_zeros(e, range_mask(e));
      continue;
    }
    const unsigned long long ok = valid_mask(e);
    if (ranks) {
      add_zeros(e, ok & ~nzm);
      const unsigned long long nb = ok & nzm;
      if (FREE) pend_free |= (unsigned)((nb >> lane) & 1ull) << e;
      if (nb && defer) {
        if ((nb >> lane) & 1ull) {
          const unsigned pos = (unsigned)(tid >> 6) * kFsPendW + st.pw + (unsigned)__popcll(nb & lt_mask);
          sh.pk[pos] = d2ord(f[e]);
          sh.pi[pos] = (unsigned)(base + e * kFsThreads + tid);
        }
        st.pw += (unsigned)__popcll(nb);
      }
    }
    const unsigned long long pm = pass_mask(e);
    if (pm) {
      if constexpr (FREE) {
        if (st.flagged) continue;  // (uniform) the regular scan will stage this query's candidates
        const unsigned n = (unsigned)__popcll(pm);
        unsigned pos = 0;
        if (lane == 0) {
          pos = atomicAdd(p.c_cnt + (int64_t)q * p.n_chunks + c, n);
          if (pos + n > (unsigned)p.lcap) {
            p.ovf[q] = 1u;
            p.ovf[p.ovf_any] = 1u;
          }
        }
        pos = (unsigned)__builtin_amdgcn_readfirstlane((int)pos);
        if ((pm >> lane) & 1ull) {
          const unsigned mine = pos + (unsigned)__popcll(pm & lt_mask);
          if (mine < (unsigned)p.lcap) {
            const int64_t at = ((int64_t)q * p.n_chunks + c) * p.lcap + mine;
            p.c_hi[at] = d2ord(f[e]);
            p.c_id[at] = (unsigned)(base + e * kFsThreads + tid);
          }
        }
      } else {
        unsigned wbase = 0;
        if (lane == 0) wbase = atomicAdd(&sh.n4[par], (unsigned)__popcll(pm));
        wbase = (unsigned)__builtin_amdgcn_readfirstlane((int)wbase);
        if ((pm >> lane) & 1ull) {
          const unsigned pos = wbase + (unsigned)__popcll(pm & lt_mask);
          sh.hi[pos] = d2ord(f[e]);
          sh.idx[pos] = (unsigned)(base + e * kFsThreads + tid);
        }
      }
    }
  }
  if (ranks) {
    if (lane == 0) {
      if (cnt0) atomicAdd(&sh.H[pb_z0], cnt0);
      if (cnt1) atomicAdd(&sh.H[pb_z0 + 1], cnt1);
      if (defer && wtotal) {
        sh.pw_n[tid >> 6] = st.pw;
        if (!FREE && st.pw > kFsDrainAt) sh.drain4[(par + 1) & 3] = 1;  // acted on after the NEXT chunk's barrier
      }
    }
    if (FREE && defer && st.pw > kFsDrainAt) {  // the wave searches its own segment (its LDS writes above are complete
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // for the wave once the counter has drained)
      __builtin_amdgcn_wave_barrier();
      const int w0 = (tid >> 6) * kFsPendW;
      for (int i = lane; i < (int)st.pw; i += 64) atomicAdd(&sh.H[beaten_by(sh.pk[w0 + i], (int64_t)sh.pi[w0 + i])], 1u);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      st.pw = 0;
      if (lane == 0) sh.pw_n[tid >> 6] = 0;
    }
    if (!defer) {  // a dense wave: one search per round while any lane still has one
      unsigned pend = pend_free;
      if (!FREE) {
#pragma unroll
        for (int e = 0; e < kFdPer; ++e) {
          const unsigned long long nb = valid_mask(e) & nz_mask(e);
          pend |= (unsigned)((nb >> lane) & 1ull) << e;
        }
      }
      while (__any(pend != 0)) {
        const int e = pend ? __ffs(pend) - 1 : -1;
        double x = 0.0;
#pragma unroll
        for (int ee = 0; ee < kFdPer; ++ee)
          if (ee == e) x = f[ee];
        const int r = beaten_by(d2ord(x), base + (int64_t)e * kFsThreads + tid);
        if (e >= 0) atomicAdd(&sh.H[r], 1u);
        pend &= pend - 1;
      }
    }
  }
  if constexpr (FREE) return;
  fs_barrier();  // ---- the chunk barrier ----
  const int n = (int)sh.n4[par];
  const bool drain = METHOD == 1 && sh.drain4[par] != 0;
  if (drain) {  // (uniform: the flag was raised before this chunk's barrier at the latest)
    for (int w = 0; w < kFsWaves; ++w) {
      const int pn = (int)sh.pw_n[w];
      for (int i = tid; i < pn; i += kFsThreads)
        atomicAdd(&sh.H[beaten_by(sh.pk[w * kFsPendW + i], (int64_t)sh.pi[w * kFsPendW + i])], 1u);
    }
    fs_barrier();
    if (lane == 0) sh.pw_n[tid >> 6] = 0;
    st.pw = 0;
  }
  if (tid == 0) {
    sh.n4[(par + 2) & 3] = 0;
    sh.drain4[(par + 3) & 3] = 0;
  }
  st.par = (par + 1) & 3;
  if (n == 0) return;  // (c_cnt and tau0 were zeroed by the host)
  unsigned long long *lh = p.c_hi + ((int64_t)q * p.n_chunks + c) * p.lcap;
  unsigned *li = p.c_id + ((int64_t)q * p.n_chunks + c) * p.lcap;
  const int room = p.prefix ? st.kp : p.lcap;
  if (n <= room) {
    for (int i = tid; i < n; i += kFsThreads) {
      lh[i] = sh.hi[i];
      li[i] = sh.idx[i];
    }
    if (tid == 0) p.c_cnt[(int64_t)q * p.n_chunks + c] = (unsigned)n;  // (prefix: fewer than K' ids, tau0 stays 0)
    fs_barrier();  // the staging area is free again
    return;
  }
  if (tid == 0) sh.cnt = 0;
  if (!(p.prefix && fd_zero_boundary(sh, n, st.kp, base))) fd_select_boundary(sh, n, st.kp);
  const unsigned long long B = sh.bnd;
  const unsigned cut = sh.cut;
  for (int i0 = 0; i0 < n; i0 += kFsThreads) {
    const int i = i0 + tid;
    const bool sel = i < n && (sh.hi[i] > B || (sh.hi[i] == B && sh.idx[i] <= cut));
    const unsigned long long sm = __ballot(sel);
    if (sm) {
      unsigned wbase = 0;
      if (lane == 0) wbase = atomicAdd(&sh.cnt, (unsigned)__popcll(sm));
      wbase = __shfl(wbase, 0);
      if (sel) {
        const unsigned pos = wbase + (unsigned)__popcll(sm & ((1ull << lane) - 1ull));
        if (pos < (unsigned)p.lcap) {
          lh[pos] = sh.hi[i];
          li[pos] = sh.idx[i];
        }
      }
    }
  }
  if (tid == 0) {
    p.c_cnt[(int64_t)q * p.n_chunks + c] = (unsigned)st.kp;
    if (p.prefix) p.tau0[q] = B;          // later ids tie-break below chunk 0's K': strictly greater only
    else atomicMax(p.T + q, B);            // K' ids at or above B exist: a valid (>=) threshold for everyone
  }
  __syncthreads();
}

template <int METHOD, int DT0, bool FREE = false>  // DT0: dtype of the prefetched array source
__global__ __launch_bounds__(kFsThreads, 4) void k_fd_scan(FdParams p, int64_t n_items) {
  extern __shared__ unsigned char fd_smem[];
  using SH = typename std::conditional<FREE, FsSharedFree, FsShared>::type;
  SH &sh = *reinterpret_cast<SH *>(fd_smem);
  const int tid = threadIdx.x;
  const int per_q = p.prefix ? 1 : p.n_chunks - 1;  // items of one query in this launch
  const int64_t per_wg = (n_items + gridDim.x - 1) / gridDim.x;
  const int64_t it0 = (int64_t)blockIdx.x * per_wg, it1 = it0 + per_wg < n_items ? it0 + per_wg : n_items;
  // the array source whose next chunk is prefetched (rrf has exactly one; linear: the first present)
  const int s0 = METHOD == 1 ? p.r1_src : (p.src[0].arr ? 0 : p.src[1].arr ? 1 : p.src[2].arr ? 2 : 3);
  if (!FREE && p.only_flagged && p.ovf[p.ovf_any] == 0u) return;  // (uniform, before any barrier) nothing overflowed
  FsState st{};
  st.cur_q = -1;
  st.zc = -1;
  if (METHOD == 0) {
    int na = 0;
    for (int s = 0; s < 4; ++s) na += p.src[s].arr ? 1 : 0;
    const double w0 = p.w[s0], w3 = p.w[3];
    st.fast0 = na == 1 && w0 - w0 == 0.0 && w3 - w3 == 0.0;  // (x - x == 0 <=> x is finite)
  }
  if (tid == 0) {
    for (int i = 0; i < 4; ++i) sh.n4[i] = sh.drain4[i] = 0;
    for (int w = 0; w < kFsWaves; ++w) sh.pw_n[w] = 0;
  }
  __syncthreads();
  const unsigned long long kzero = 0x8000000000000000ull;
  auto finish_query = [&]() {  // deferred searches, then the finished query's rank histogram joins the global one
    if (st.cur_q >= 0 && METHOD == 1 && st.skn > 0) {
      const int skn = st.skn;
      for (int w = 0; w < kFsWaves; ++w) {
        const int pn = (int)sh.pw_n[w];
        for (int i = tid; i < pn; i += kFsThreads) {
          const unsigned long long k = sh.pk[w * kFsPendW + i];
          const int64_t id = (int64_t)sh.pi[w * kFsPendW + i];
          int lo = 0, hi = skn;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const unsigned long long kh = sh.sk_hi[mid];
            if (kh > k || (kh == k && (int64_t)sh.sk_id[mid] < id)) lo = mid + 1;
            else hi = mid;
          }
          atomicAdd(&sh.H[lo], 1u);
        }
      }
      __syncthreads();
      if ((tid & 63) == 0) sh.pw_n[tid >> 6] = 0;
      st.pw = 0;
      for (int i = 0; i < 4; ++i)
        if (tid == 0) sh.drain4[i] = 0;  // (a pending request is moot: the segments are empty)
      unsigned *Hq = p.H + (int64_t)st.cur_q * (kFdMaxSparse + 1);
      for (int i = tid; i <= skn; i += kFsThreads)
        if (sh.H[i]) atomicAdd(Hq + i, sh.H[i]);
    }
  };
  // (query, chunk) of an item advance incrementally — a 64-bit division per item was ~300 scalar instructions per chunk
  auto advance = [&](int &q, int &c) {
    if (p.prefix) ++q;
    else if (++c == p.n_chunks) {
      c = 1;
      ++q;
    }
  };
  auto load_T = [&](int q) -> unsigned long long {
    // a plain (cached) load: T only ever rises and any earlier value is still a valid threshold, so a stale line costs
    // a few extra candidates at worst — an agent-scope atomic load went past the L2 and took ~4 us per chunk
    // (relaxed agent-scope atomic: re-read every chunk, but — unlike a volatile load — not waited for on the spot; it
    // is consumed one chunk later, with the prefetched values)
    return (p.prefix || FREE) ? 0ull : __hip_atomic_load(p.T + fd_per_lane(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  unsigned long long nxt[kFdPer];
  unsigned long long T_nxt = 0ull;
  int q = 0, c = 0, q2 = 0, c2 = 0;
  bool dense2 = false;  // the query of the next item is not this pass's (FREE: left to the regular scan; re-do pass: not
                        // flagged) — its loads all go to one chunk (cache hits; no branch around the prefetch)
  if (it0 < it1) {
    q2 = (int)(it0 / per_q);
    c2 = p.prefix ? 0 : 1 + (int)(it0 % per_q);
    T_nxt = load_T(q2);
    if (FREE) dense2 = p.tau0[q2] != kzero;
    else if (p.only_flagged) dense2 = p.ovf[q2] == 0u;  // (the re-do pass: queries that are not flagged are not streamed)
    fd_load_raw<kFsThreads, DT0>(p.src[s0], p.q0 + q2, (int64_t)(dense2 ? 1 : c2) * kFsChunk, tid, nxt);
  }
  for (int64_t item = it0; item < it1; ++item) {
    q = q2;
    c = c2;
    unsigned long long raw[kFdPer];
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) raw[e] = nxt[e];
    const unsigned long long T = T_nxt;
    if (item + 1 < it1) {
      const int q_before = q2;
      advance(q2, c2);
      T_nxt = load_T(q2);
      if (FREE && q2 != q_before) dense2 = p.tau0[q2] != kzero;
      if (!FREE && p.only_flagged && q2 != q_before) dense2 = p.ovf[q2] == 0u;
      fd_load_raw<kFsThreads, DT0>(p.src[s0], p.q0 + q2, (int64_t)(dense2 ? 1 : c2) * kFsChunk, tid, nxt);
    }
    if (!FREE && p.only_flagged && p.ovf[q] == 0u) continue;  // (uniform) this query's lists came out of the barrier-free pass
    if (q != st.cur_q) {
      __syncthreads();
      finish_query();
      __syncthreads();
      st.cur_q = q;
      st.kp = p.kprime[q];
      if (METHOD == 0) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const unsigned long long o = p.smax_ord[(int64_t)q * 4 + s];
          st.smax[s] = o ? ord2d(o) : 0.0;
        }
      } else {
        st.skn = p.sk_n[q];
        for (int i = tid; i < st.skn; i += kFsThreads) {
          sh.sk_hi[i] = p.sk_hi[(int64_t)q * kFdMaxSparse + i];
          sh.sk_id[i] = p.sk_id[(int64_t)q * kFdMaxSparse + i];
        }
        for (int i = tid; i <= st.skn; i += kFsThreads) sh.H[i] = 0;
      }
      st.tau0 = p.prefix ? 0ull : p.tau0[q];
      if (FREE) {  // (see k_fd_scan_free: only threshold-0.0 queries are taken by the barrier-free pass)
        st.flagged = p.ovf[q] != 0u;  // (read once per query: a load per chunk would be waited for on the spot)
        st.dense_q = st.tau0 != kzero;
        if (st.dense_q && tid == 0) {
          p.ovf[q] = 2u;
          p.ovf[p.ovf_any] = 1u;
        }
      }
      __syncthreads();
      if (METHOD == 1) {  // where the zero-valued short-list keys end
        int lo = 0, hi = st.skn;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (sh.sk_hi[mid] >= kzero) lo = mid + 1;
          else hi = mid;
        }
        st.z_hi = lo;
        st.zc = -1;
      }
    }
    if (FREE) {
      if (st.dense_q) continue;  // (uniform) neither candidates nor ranks: the regular scan does both for this query
    }
    fd_scan_chunk<METHOD, DT0, FREE>(p, sh, st, q, c, s0, raw, T);
  }
  __syncthreads();
  finish_query();
}

// ---- linear, ONE array source: the barrier-free pass ---------------------------------------------------------------
// On a BM25 row (~0.1 % non-zero) almost nothing the scan above does per chunk is needed: chunk 0's K'-th best is 0.0, so
// a zero never passes, a chunk holds a handful of candidates, no list overflows and the running threshold never moves —
// but every chunk still pays the staging protocol's barrier (eight waves arriving microseconds apart).  Here the waves
// run free: a wave tests the raw bits of its eight entry rows, computes fused values only for rows with a non-zero entry,
// and appends what beats chunk 0's threshold straight to the chunk's list in global memory (one atomic per row that has
// a candidate).  No LDS, no barrier.  A list that would overflow (a dense vector, or no usable threshold) flags its QUERY;
// k_fd_redo_prep clears the flagged queries' counts and the regular scan runs for them alone (it returns at once when
// nothing is flagged).  Same fused-value arithmetic in the same order as fd_scan_chunk, so the keys of chunk 0 (regular
// scan) and of the other chunks compare consistently.
template <int DT0>
__global__ __launch_bounds__(kFsThreads) void k_fd_scan_free(FdParams p, int64_t n_items) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int per_q = p.n_chunks - 1;
  const int64_t per_wg = (n_items + gridDim.x - 1) / gridDim.x;
  const int64_t it0 = (int64_t)blockIdx.x * per_wg, it1 = it0 + per_wg < n_items ? it0 + per_wg : n_items;
  if (it0 >= it1) return;
  const int s0 = p.src[0].arr ? 0 : p.src[1].arr ? 1 : p.src[2].arr ? 2 : 3;
  const double w = p.w[s0], w3 = p.w[3];
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  unsigned long long nxt[kFdPer];
  int q2 = (int)(it0 / per_q), c2 = 1 + (int)(it0 % per_q);
  const unsigned long long kzero = 0x8000000000000000ull;
  // the query of the NEXT item is left to the regular scan: its loads all go to ONE chunk (cache hits, no stream) — a
  // branch around the prefetch would make hipcc wait for the loads at the join
  bool dense2 = p.tau0[q2] != kzero;
  fd_load_raw<kFsThreads, DT0>(p.src[s0], p.q0 + q2, (int64_t)(dense2 ? 1 : c2) * kFsChunk, tid, nxt);
  int cur_q = -1;
  double sm = 0.0, t0d = 0.0;
  bool has_t0 = false, zero_passes = true, dense_q = false;
  for (int64_t item = it0; item < it1; ++item) {
    const int q = q2, c = c2;
    unsigned long long raw[kFdPer];
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) raw[e] = nxt[e];
    if (item + 1 < it1) {
      if (++c2 == p.n_chunks) {
        c2 = 1;
        ++q2;
        dense2 = p.tau0[q2] != kzero;
      }
      fd_load_raw<kFsThreads, DT0>(p.src[s0], p.q0 + q2, (int64_t)(dense2 ? 1 : c2) * kFsChunk, tid, nxt);
    }
    if (q != cur_q) {
      cur_q = q;
      const unsigned long long o = s0 < 3 ? p.smax_ord[(int64_t)q * 4 + s0] : 0ull;
      sm = o ? ord2d(o) : 0.0;
      const unsigned long long tau0 = p.tau0[q];
      has_t0 = tau0 != 0ull;
      t0d = has_t0 ? ord2d(tau0) : 0.0;
      zero_passes = !has_t0 || 0.0 > t0d;
      // Only a query whose chunk 0 holds fewer than K' positive values (threshold exactly 0.0) is taken here: on a denser
      // vector the regular scan's RUNNING threshold is what keeps the candidate lists short (without it every chunk would
      // contribute ~K' candidates).  Such a query is flagged (2) for the regular scan at once.
      dense_q = tau0 != kzero;
      if (dense_q && tid == 0) {
        p.ovf[q] = 2u;
        p.ovf[p.ovf_any] = 1u;
      }
    }
    if (dense_q) continue;
    const int64_t base = (int64_t)c * kFsChunk;
    const int64_t room = p.src[s0].len - base;
    const int nv = room >= kFsChunk ? kFsChunk : (room > 0 ? (int)room : 0);
    unsigned *cnt = p.c_cnt + (int64_t)q * p.n_chunks + c;
    const int64_t at = ((int64_t)q * p.n_chunks + c) * p.lcap;
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) {
      const bool nonzero = DT0 == 0 ? (raw[e] & 0x7fffffffffffffffull) != 0ull : ((unsigned)raw[e] & 0x7fffffffu) != 0u;
      if (!__any(nonzero) && !zero_passes) continue;  // a row of zeros: fused values of exactly 0.0, below the threshold
      const double x = fd_raw_value(DT0, raw[e]);
      const bool ok = (nv == kFsChunk || e * kFsThreads + tid < nv) && x == x;
      double f = 0.0;
      if (s0 < 3) {
        double r = sm < 0.0 ? -x : x;
        if (__any(ok && x != 0.0)) r = x == 0.0 ? r : x / sm;
        if (sm == 0.0) r = 0.0;
        const double t = f + w * r;
        f = ok ? t : f;
        f = f + w3 * 0.0;
      } else {
        f = f + (ok ? w * x : w * 0.0);
      }
      const unsigned long long pm = __ballot(ok && (!has_t0 || f > t0d));
      if (pm == 0ull) continue;
      const unsigned n = (unsigned)__popcll(pm);
      unsigned pos = 0;
      if (lane == 0) {
        pos = atomicAdd(cnt, n);
        if (pos + n > (unsigned)p.lcap) {  // the regular scan takes this query over
          p.ovf[q] = 1u;
          p.ovf[p.ovf_any] = 1u;
        }
      }
      pos = (unsigned)__builtin_amdgcn_readfirstlane((int)pos);
      if ((pm >> lane) & 1ull) {
        const unsigned mine = pos + (unsigned)__popcll(pm & lt_mask);
        if (mine < (unsigned)p.lcap) {
          p.c_hi[at + mine] = d2ord(f);
          p.c_id[at + mine] = (unsigned)(base + e * kFsThreads + tid);
        }
      }
    }
  }
}

// the flagged queries start over: their chunk counts (chunk 0 stays: it came from the regular scan) back to zero
__global__ __launch_bounds__(256) void k_fd_redo_prep(FdParams p) {
  if (p.ovf[p.ovf_any] == 0u) return;
  const int q = blockIdx.x;
  if (p.ovf[q] == 0u) return;
  for (int c = 1 + threadIdx.x; c < p.n_chunks; c += 256) p.c_cnt[(int64_t)q * p.n_chunks + c] = 0u;
}

// ---- build: K' best of the chunk lists, ordered; compose the short lists k_fuse<true> consumes ------------------
__global__ __launch_bounds__(kFdThreads) void k_fd_build(FdParams p) {
  extern __shared__ unsigned char fd_smem[];
  FdShared &sh = *reinterpret_cast<FdShared *>(fd_smem);
  __shared__ unsigned s_off[6];
  __shared__ unsigned s_pref[kFdThreads + 1];
  __shared__ unsigned s_wsum[kFdThreads / 64];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int kp = p.kprime[q];
  int n = 0;
  auto reduce = [&]() {  // keep the kp best of the n staged pairs, compacted to the front
    if (n <= kp) return;
    fd_select_boundary(sh, n, kp);
    const unsigned long long B = sh.bnd;
    const unsigned cut = sh.cut;
    unsigned long long rh[kFdPer];
    unsigned ri[kFdPer];
    bool rs[kFdPer];
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) {
      const int i = e * kFdThreads + tid;
      rs[e] = i < n && (sh.hi[i] > B || (sh.hi[i] == B && sh.idx[i] <= cut));
      rh[e] = i < n ? sh.hi[i] : 0ull;
      ri[e] = i < n ? sh.idx[i] : 0u;
    }
    if (tid == 0) sh.cnt = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kFdPer; ++e) {
      const unsigned long long sm = __ballot(rs[e]);
      if (sm) {
        unsigned wbase = 0;
        if (lane == 0) wbase = atomicAdd(&sh.cnt, (unsigned)__popcll(sm));
        wbase = __shfl(wbase, 0);
        if (rs[e]) {
          const unsigned pos = wbase + (unsigned)__popcll(sm & ((1ull << lane) - 1ull));
          sh.hi[pos] = rh[e];
          sh.idx[pos] = ri[e];
        }
      }
    }
    __syncthreads();
    n = kp;
  };
  // gather the chunk lists, a tile of kFdThreads chunks at a time: counts -> prefix sums in LDS, then one thread per
  // ENTRY (its chunk found by bisection).  (One chunk after the other — a dependent count load, a handful of entries,
  // a barrier — took 0.75 us per chunk: 180 us at 244 chunks.)
  for (int tile0 = 0; tile0 < p.n_chunks; tile0 += kFdThreads) {
    const int nt = p.n_chunks - tile0 < kFdThreads ? p.n_chunks - tile0 : kFdThreads;
    const unsigned cnt = tid < nt ? p.c_cnt[(int64_t)q * p.n_chunks + tile0 + tid] : 0u;
    unsigned incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    __syncthreads();  // (s_pref / s_wsum of the previous tile are no longer read)
    if (lane == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < (tid >> 6); ++w) woff += s_wsum[w];
    s_pref[tid + 1] = woff + incl;
    if (tid == 0) s_pref[0] = 0;
    __syncthreads();
    int g0 = 0;
    while (g0 < nt) {
      // the longest run of chunks [g0, g1) whose entries still fit the staging area
      const unsigned limit = s_pref[g0] + (unsigned)(kFdChunk - n);
      int lo = g0, hi = nt;  // largest g1 with s_pref[g1] <= limit
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_pref[mid] <= limit) lo = mid;
        else hi = mid - 1;
      }
      const int g1 = lo;
      if (g1 == g0) {  // not even one more chunk fits: keep the best kp and go on (a list holds <= lcap entries)
        reduce();
        continue;
      }
      const unsigned first = s_pref[g0], total = s_pref[g1] - first;
      for (unsigned i = tid; i < total; i += kFdThreads) {
        const unsigned target = first + i;
        int a = g0, b = g1 - 1;  // the chunk j with s_pref[j] <= target < s_pref[j + 1]
        while (a < b) {
          const int mid = (a + b + 1) >> 1;
          if (s_pref[mid] <= target) a = mid;
          else b = mid - 1;
        }
        const int64_t at = ((int64_t)q * p.n_chunks + tile0 + a) * p.lcap + (target - s_pref[a]);
        sh.hi[n + i] = p.c_hi[at];
        sh.idx[n + i] = p.c_id[at];
      }
      n += (int)total;
      g0 = g1;
      __syncthreads();
    }
  }
  reduce();
  // order the n <= kFdMaxK candidates: (key desc, id asc) by rank counting into the upper half of the arrays
  unsigned long long *s_hi = sh.hi + kFdChunk / 2;
  unsigned *s_id = sh.idx + kFdChunk / 2;
  for (int i = tid; i < n; i += kFdThreads) {
    const unsigned long long k = sh.hi[i];
    const unsigned id = sh.idx[i];
    int r = 0;
    for (int j = 0; j < n; ++j) r += (sh.hi[j] > k || (sh.hi[j] == k && sh.idx[j] < id)) ? 1 : 0;
    s_hi[r] = k;
    s_id[r] = id;
  }
  __syncthreads();
  // rrf: exclusive prefix of the rank histogram -> rank of the j-th (0-based) sorted short-list key = sum H[0..j]
  const int mu = p.su_n[q];
  const unsigned *su = p.su_id + (int64_t)q * kFdMaxSparse;
  int skn = 0;
  if (p.method == 1) {
    skn = p.sk_n[q];
    const unsigned *Hq = p.H + (int64_t)q * (kFdMaxSparse + 1);
    for (int i = tid; i < skn; i += kFdThreads) {
      sh.sk_hi[i] = p.sk_hi[(int64_t)q * kFdMaxSparse + i];
      sh.sk_id[i] = p.sk_id[(int64_t)q * kFdMaxSparse + i];
    }
    if (tid == 0) {
      unsigned acc = 0;
      for (int j = 0; j < skn; ++j) {
        acc += Hq[j];
        sh.H[j] = acc;  // entries at or above key j, itself included == its 1-based rank
      }
    }
  }
  __syncthreads();
  // compose: per source either the caller's short list, or (array source) the candidates + the short-list ids
  int64_t *oi = p.o_ids + (int64_t)q * kFuseMax;
  double *os = p.o_sc + (int64_t)q * kFuseMax;
  int *orank = p.o_rank + (int64_t)q * kFuseMax;
  const int64_t *off = p.l_offs + (int64_t)q * 5;
  if (tid == 0) s_off[0] = 0;
  __syncthreads();
  for (int s = 0; s < 4; ++s) {
    const unsigned b0 = s_off[s];
    if (tid == 0) sh.cnt = 0;
    __syncthreads();
    if (!fd_is_array(p, s)) {
      const int m = (int)(off[s + 1] - off[s]);
      for (int i = tid; i < m; i += kFdThreads) {
        oi[b0 + i] = p.l_ids[off[s] + i];
        os[b0 + i] = p.l_sc[off[s] + i];
        orank[b0 + i] = 0;
      }
      if (tid == 0) sh.cnt = (unsigned)m;
    } else {
      const bool r1 = p.method == 1 && s == p.r1_src;
      // candidates (sorted position r -> rank r + 1 in the whole array)
      for (int r0 = 0; r0 < n; r0 += kFdThreads) {
        const int r = r0 + tid;
        double v = 0.0;
        const bool have = r < n && fd_val_any(p, s, q, (int64_t)s_id[r], v);
        const unsigned long long hm = __ballot(have);
        if (hm) {
          unsigned wbase = 0;
          if (lane == 0) wbase = atomicAdd(&sh.cnt, (unsigned)__popcll(hm));
          wbase = __shfl(wbase, 0);
          if (have) {
            const unsigned pos = b0 + wbase + (unsigned)__popcll(hm & ((1ull << lane) - 1ull));
            oi[pos] = (int64_t)s_id[r];
            os[pos] = v;
            orank[pos] = r1 ? r + 1 : 0;
          }
        }
      }
      // short-list ids that are not candidates
      for (int j0 = 0; j0 < mu; j0 += kFdThreads) {
        const int j = j0 + tid;
        double v = 0.0;
        bool have = j < mu && fd_val_any(p, s, q, (int64_t)su[j], v);
        int rank = 0;
        if (have) {
          const unsigned id = su[j];
          for (int r = 0; r < n; ++r)
            if (s_id[r] == id) {
              have = false;  // already listed as a candidate
              break;
            }
          if (have && r1) {
            const unsigned long long k = d2ord(v);
            int lo = 0, hi = skn;  // position of (k, id) among the sorted short-list keys
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              const bool before = sh.sk_hi[mid] > k || (sh.sk_hi[mid] == k && sh.sk_id[mid] < id);
              if (before) lo = mid + 1;
              else hi = mid;
            }
            rank = (int)sh.H[lo];
          }
        }
        const unsigned long long hm = __ballot(have);
        if (hm) {
          unsigned wbase = 0;
          if (lane == 0) wbase = atomicAdd(&sh.cnt, (unsigned)__popcll(hm));
          wbase = __shfl(wbase, 0);
          if (have) {
            const unsigned pos = b0 + wbase + (unsigned)__popcll(hm & ((1ull << lane) - 1ull));
            oi[pos] = (int64_t)su[j];
            os[pos] = v;
            orank[pos] = rank;
          }
        }
      }
    }
    __syncthreads();
    if (tid == 0) s_off[s + 1] = b0 + sh.cnt;
    __syncthreads();
  }
  if (tid < 5) p.o_offs[(int64_t)q * 5 + tid] = (int64_t)q * kFuseMax + s_off[tid];
  if (tid < 4) {
    double m = __builtin_nan("");
    if (p.method == 0 && tid < 3 && fd_is_array(p, tid)) {
      const unsigned long long o = p.smax_ord[(int64_t)q * 4 + tid];
      m = o ? ord2d(o) : -__builtin_inf();
    }
    p.o_smax[(int64_t)q * 4 + tid] = m;
  }
}

// ---- the sparse form of an array source ------------------------------------------------------------------------
// A BM25 row over N notes has a few thousand non-zero scores.  Given as (id, value) entries the stream over N ids is not
// needed at all; what k_fd_build consumes is produced from the entries:
//   k_fs_sort   the row's entries sorted by id (bitonic, LDS) -> the lookup table fd_val_any bisects; linear: the row's
//               maximum (an implicit zero takes part when the row has one)
//   k_fs_stage  candidates = every explicit entry + the K' lowest ids that are NOT listed (implicit zeros: after the
//               positive entries the best zero-valued ids are the lowest ones), keyed exactly as the scan keys them;
//               rrf: the rank histogram H — explicit non-zero entries by bisection of the short-list keys, the N - nnz
//               zeros by interval arithmetic between the zero-valued short-list ids (a zero entry with id i is beaten by
//               the z_lo positive short-list keys and by the zero-valued ones with a lower id).
// Same FdParams outputs as k_fd_max + k_fd_scan (smax_ord, c_hi / c_id / c_cnt, H), so k_fd_build and k_fuse<true> — and
// the bit-exactness argument at the top of this file — carry over unchanged.
// Bitonic sort of P2 (a power of two) 64-bit keys in LDS, ascending, by kFsSortThreads threads.  A 1024-thread network
// with a workgroup barrier after each of its ~70 passes spent 1.3 us per pass at the barrier (86 us per row of 2048);
// four waves without barriers were bound by the LDS latency of their eight pairs per pass (61 us).  Here each of the W
// active waves owns a contiguous segment of P2 / W keys: a pass whose stride stays inside a segment
// needs no workgroup barrier (a wave's LDS operations complete in order), only the few passes with a longer stride do.
constexpr int kFsSortThreads = 1024;
__device__ __forceinline__ void fs_bitonic_u64(unsigned long long *key, int P2) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int W = P2 / 128;  // active waves: segments of >= 128 keys (a short row: one wave, no barrier at all)
  W = W < 1 ? 1 : (W > kFsSortThreads / 64 ? kFsSortThreads / 64 : W);
  const int seg = P2 / W, half = seg >> 1;
  const bool active = wave < W;
  auto pass = [&](int k2, int j, int p0, int p1, int step) {
    constexpr int U = 4;  // pairs in flight per lane: with one wave per SIMD nothing else hides the LDS latency
    int pp = p0 + lane;
    for (; pp + (U - 1) * step < p1; pp += U * step) {
      int i[U];
      unsigned long long a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = pp + u * step;
        i[u] = ((q & ~(j - 1)) << 1) | (q & (j - 1));
        a[u] = key[i[u]];
        b[u] = key[i[u] + j];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool sw = ((i[u] & k2) == 0) ? (b[u] < a[u]) : (a[u] < b[u]);
        key[i[u]] = sw ? b[u] : a[u];
        key[i[u] + j] = sw ? a[u] : b[u];
      }
    }
    for (; pp < p1; pp += step) {
      const int i = ((pp & ~(j - 1)) << 1) | (pp & (j - 1)), x = i + j;
      const unsigned long long a = key[i], b = key[x];
      const bool up = (i & k2) == 0;
      if (up ? (b < a) : (a < b)) {
        key[i] = b;
        key[x] = a;
      }
    }
  };
  for (int k2 = 2; k2 <= P2; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      if (j < seg) {  // inside the wave's own segment
        if (active) pass(k2, j, wave * half, (wave + 1) * half, 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      } else {        // across segments (W > 1): every active wave takes its share of the pairs, barriers on both sides
        __syncthreads();
        if (active) pass(k2, j, wave * (P2 / 2 / W), (wave + 1) * (P2 / 2 / W), 64);
        __syncthreads();
      }
    }
}

// the row's count; linear: its maximum (an implicit zero takes part when the row does not list every id) -> smax_ord
__device__ __forceinline__ void fs_row_finish(const FdParams &p, int q, int nnz, double best, bool any, double *s_best,
                                              int *s_any) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) p.sp.cnt[q] = nnz;
  if (p.method != 0) return;
  for (int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_xor(best, o);
    const bool ta = __shfl_xor((int)any, o) != 0;
    best = (ta && (!any || t > best)) ? t : best;
    any = any || ta;
  }
  if (lane == 0) {
    s_best[wave] = best;
    s_any[wave] = any ? 1 : 0;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kFsSortThreads / 64; ++w)
      if (s_any[w] && (!any || s_best[w] > best)) {
        best = s_best[w];
        any = true;
      }
    if ((int64_t)nnz < p.src[p.sp.src].len) {  // an id that is not listed holds 0.0
      best = (!any || 0.0 > best) ? 0.0 : best;
      any = true;
    }
    p.smax_ord[(int64_t)q * 4 + p.sp.src] = any ? d2ord(best) : 0ull;
  }
}

__global__ __launch_bounds__(kFsSortThreads) void k_fs_sort(FdParams p, int P2max) {
  extern __shared__ unsigned char fd_smem[];
  unsigned long long *key = reinterpret_cast<unsigned long long *>(fd_smem);  // (id << 32) | position in the caller's row
  __shared__ double s_best[kFsSortThreads / 64];
  __shared__ int s_any[kFsSortThreads / 64];
  const int q = blockIdx.x, tid = threadIdx.x;
  const int cap = p.sp.cap;
  int nnz = p.sp.in_cnt[p.q0 + q];
  if (tid == 0 && (nnz < 0 || nnz > cap)) atomicOr(p.sp.err, nnz < 0 ? 1u : 2u);
  nnz = nnz < 0 ? 0 : (nnz > cap ? cap : nnz);
  int P2 = 2;
  while (P2 < nnz) P2 <<= 1;
  const unsigned *gi = p.sp.in_id + (p.q0 + q) * (int64_t)cap;
  const double *gv = p.sp.in_val + (p.q0 + q) * (int64_t)cap;
  for (int i = tid; i < P2; i += kFsSortThreads)
    key[i] = i < nnz ? ((unsigned long long)gi[i] << 32) | (unsigned)i : ~0ull;
  __syncthreads();
  fs_bitonic_u64(key, P2);
  __syncthreads();
  double best = 0.0;
  bool any = false;
  const unsigned alen = (unsigned)(p.src[p.sp.src].len > 0xffffffffll ? 0xffffffffll : p.src[p.sp.src].len);
  for (int i = tid; i < nnz; i += kFsSortThreads) {
    const unsigned long long k = key[i];
    const double v = gv[(unsigned)k];
    const unsigned id = (unsigned)(k >> 32);
    if (id >= alen) atomicOr(p.sp.err, 4u);
    if (i > 0 && (unsigned)(key[i - 1] >> 32) == id) atomicOr(p.sp.err, 8u);
    p.sp.id[(int64_t)q * cap + i] = (unsigned)(k >> 32);
    p.sp.val[(int64_t)q * cap + i] = v;
    if (v == v) {
      best = (!any || v > best) ? v : best;
      any = true;
    }
  }
  fs_row_finish(p, q, nnz, best, any, s_best, s_any);
}

// rows beyond the LDS sort arrive sorted by id: checked (strictly ascending, below the array length), copied, finished
__global__ __launch_bounds__(kFsSortThreads) void k_fs_take(FdParams p) {
  __shared__ double s_best[kFsSortThreads / 64];
  __shared__ int s_any[kFsSortThreads / 64];
  const int q = blockIdx.x, tid = threadIdx.x;
  const int cap = p.sp.cap;
  int nnz = p.sp.in_cnt[p.q0 + q];
  if (tid == 0 && (nnz < 0 || nnz > cap)) atomicOr(p.sp.err, nnz < 0 ? 1u : 2u);
  nnz = nnz < 0 ? 0 : (nnz > cap ? cap : nnz);
  const unsigned *gi = p.sp.in_id + (p.q0 + q) * (int64_t)cap;
  const double *gv = p.sp.in_val + (p.q0 + q) * (int64_t)cap;
  const unsigned alen = (unsigned)(p.src[p.sp.src].len > 0xffffffffll ? 0xffffffffll : p.src[p.sp.src].len);
  double best = 0.0;
  bool any = false;
  for (int i = tid; i < nnz; i += kFsSortThreads) {
    const unsigned id = gi[i];
    const double v = gv[i];
    if (id >= alen) atomicOr(p.sp.err, 4u);
    if (i > 0 && gi[i - 1] >= id) atomicOr(p.sp.err, gi[i - 1] == id ? 8u : 16u);
    p.sp.id[(int64_t)q * cap + i] = id;
    p.sp.val[(int64_t)q * cap + i] = v;
    if (v == v) {
      best = (!any || v > best) ? v : best;
      any = true;
    }
  }
  fs_row_finish(p, q, nnz, best, any, s_best, s_any);
}

struct FsStageShared {
  unsigned sid[kFsSpMax];
  unsigned long long sk_hi[kFdMaxSparse];
  unsigned sk_id[kFdMaxSparse];
  unsigned H[kFdMaxSparse + 1];
  unsigned nzc[kFdMaxSparse + 1];  // listed entries that are not zeros, per interval of the zero-valued short-list ids
  // candidate pre-selection (rows with at least K' entries above zero)
  unsigned hist[1024];
  unsigned long long kmin, kmax;
  unsigned npos, nout;
  int bsel;
};

// BIG (rows beyond kFsSpMax entries): the sorted ids are bisected where they lie, in global memory
template <bool BIG>
__global__ __launch_bounds__(kFdThreads) void k_fs_stage(FdParams p) {
  extern __shared__ unsigned char fd_smem[];
  FsStageShared &sh = *reinterpret_cast<FsStageShared *>(fd_smem);
  const int q = blockIdx.x, tid = threadIdx.x;
  const int s = p.sp.src, cap = p.sp.cap;
  const int nnz = p.sp.cnt[q], kp = p.kprime[q];
  const int64_t N = p.src[s].len;
  const unsigned *gid = p.sp.id + (int64_t)q * cap;
  const double *gval = p.sp.val + (int64_t)q * cap;
  const unsigned long long kzero = 0x8000000000000000ull;  // d2ord(0.0)
  if constexpr (!BIG)
    for (int i = tid; i < nnz; i += kFdThreads) sh.sid[i] = gid[i];
  auto sid = [&](int i) -> unsigned {
    if constexpr (BIG) return gid[i];
    else return sh.sid[i];
  };
  int skn = 0;
  if (p.method == 1) {
    skn = p.sk_n[q];
    for (int i = tid; i < skn; i += kFdThreads) {
      sh.sk_hi[i] = p.sk_hi[(int64_t)q * kFdMaxSparse + i];
      sh.sk_id[i] = p.sk_id[(int64_t)q * kFdMaxSparse + i];
    }
    for (int i = tid; i <= skn; i += kFdThreads) {
      sh.H[i] = 0;
      sh.nzc[i] = 0;
    }
  }
  __syncthreads();
  // the value an id contributes to the ordering of the ids outside the short lists: linear — the fused value in the
  // scan's order of operations (fd_scan_chunk; the absent sources add w * 0.0 only for the path term); rrf — the raw value
  double sm = 0.0;
  if (p.method == 0) {
    const unsigned long long o = p.smax_ord[(int64_t)q * 4 + s];
    sm = o ? ord2d(o) : 0.0;
  }
  auto key_of = [&](double x) -> unsigned long long {
    if (x != x) return 1ull;  // an absent id: below every number (k_fd_build drops it)
    if (p.method == 1) return d2ord(x);
    double r = x == 0.0 ? (sm < 0.0 ? -x : x) : x / sm;
    if (sm == 0.0) r = 0.0;
    double f = 0.0 + p.w[s] * r;
    f = f + p.w[3] * 0.0;
    return d2ord(f);
  };
  auto emit = [&](int pos, unsigned long long key, unsigned id) {
    const int64_t at = ((int64_t)q * p.n_chunks + pos / p.lcap) * p.lcap + pos % p.lcap;
    p.c_hi[at] = key;
    p.c_id[at] = id;
  };
  // Rows with at least K' entries above zero (a frequent-word BM25 row: 10 K - 60 K of them): the K' best are among those,
  // so neither the zeros nor most of the positives need to reach k_fd_build, which selects K' of whatever it is given (and
  // took 274 us per 200 rows of ~9.5 K candidates).  One histogram of 1024 equal key ranges between the smallest and the
  // largest positive key, the range in which the count from the top reaches K', and only the entries at or above it are
  // staged — a superset of the K' best, a few dozen more than K' at most times.
  const unsigned long long kz = key_of(0.0);
  if (tid == 0) {
    sh.kmin = ~0ull;
    sh.kmax = 0ull;
    sh.npos = 0;
    sh.nout = 0;
    sh.bsel = -1;
  }
  for (int i = tid; i < 1024; i += kFdThreads) sh.hist[i] = 0;
  __syncthreads();
  // (rows of a few thousand entries go to k_fd_build whole: it holds 8192 staged pairs without a selection round, and the
  // three passes below cost such a row more than they save — 0.047 -> 0.087 ms per 200 rows of ~1000 entries)
  const bool preselect = nnz > 4096;
  if (preselect) {
    unsigned long long lo = ~0ull, hi = 0ull;
    unsigned np = 0;
    for (int i = tid; i < nnz; i += kFdThreads) {
      const unsigned long long k = key_of(gval[i]);
      if (k > kz) {
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
        ++np;
      }
    }
    if (np) {
      atomicMin(&sh.kmin, lo);
      atomicMax(&sh.kmax, hi);
      atomicAdd(&sh.npos, np);
    }
  }
  __syncthreads();
  if (preselect && (int)sh.npos >= kp && kp > 0) {  // (uniform)
    const unsigned long long kmin = sh.kmin, span = sh.kmax - kmin;
    int shift = 0;
    while ((span >> shift) >= 1024ull) ++shift;
    for (int i = tid; i < nnz; i += kFdThreads) {
      const unsigned long long k = key_of(gval[i]);
      if (k > kz) atomicAdd(&sh.hist[(unsigned)((k - kmin) >> shift)], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned acc = 0;
      int b = 1023;
      for (; b > 0; --b) {
        acc += sh.hist[b];
        if (acc >= (unsigned)kp) break;
      }
      sh.bsel = b;  // (b == 0: the count is complete there at the latest — npos >= kp)
    }
    __syncthreads();
    const unsigned bsel = (unsigned)sh.bsel;
    const int lane = tid & 63;
    for (int i0 = 0; i0 < nnz; i0 += kFdThreads) {
      const int i = i0 + tid;
      unsigned long long k = 0ull;
      bool take = false;
      if (i < nnz) {
        k = key_of(gval[i]);
        take = k > kz && (unsigned)((k - kmin) >> shift) >= bsel;
      }
      const unsigned long long m = __ballot(take);
      if (m) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&sh.nout, (unsigned)__popcll(m));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (take) emit((int)(base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))), k, sid(i));
      }
    }
    __syncthreads();
    const int total = (int)sh.nout;
    for (int c = tid; c < p.n_chunks; c += kFdThreads) {
      const int left = total - c * p.lcap;
      p.c_cnt[(int64_t)q * p.n_chunks + c] = (unsigned)(left < 0 ? 0 : (left > p.lcap ? p.lcap : left));
    }
  } else {
  for (int i = tid; i < nnz; i += kFdThreads) emit(i, key_of(gval[i]), sid(i));
  // the j-th id that is not listed = j + (listed ids below it): the first i with sid[i] - i > j
  const int64_t missing = N - nnz;
  const int nz = (int)(missing < kp ? (missing > 0 ? missing : 0) : kp);
  for (int j = tid; j < nz; j += kFdThreads) {
    int lo = 0, hi = nnz;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)sid(mid) - mid > j) hi = mid;
      else lo = mid + 1;
    }
    emit(nnz + j, kz, (unsigned)(j + lo));
  }
  const int total = nnz + nz;
  for (int c = tid; c < p.n_chunks; c += kFdThreads) {
    const int left = total - c * p.lcap;
    p.c_cnt[(int64_t)q * p.n_chunks + c] = (unsigned)(left < 0 ? 0 : (left > p.lcap ? p.lcap : left));
  }
  }
  if (p.method != 1) return;
  // ---- rrf: how many of the N array entries beat each short-list key ----
  // sorted short-list keys: [0, z_lo) positive, [z_lo, z_hi) zero-valued (ids ascending), [z_hi, skn) negative
  int z_lo, z_hi;
  {
    int lo = 0, hi = skn;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sh.sk_hi[mid] > kzero) lo = mid + 1;
      else hi = mid;
    }
    z_lo = lo;
    hi = skn;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sh.sk_hi[mid] >= kzero) lo = mid + 1;
      else hi = mid;
    }
    z_hi = lo;
  }
  const int m = z_hi - z_lo;
  const unsigned *Z = sh.sk_id + z_lo;
  for (int i = tid; i < nnz; i += kFdThreads) {
    const double v = gval[i];
    const unsigned id = sid(i);
    if (v == 0.0) continue;  // counted with the zeros below
    int lo = 0, hi = m;      // zero-valued short-list ids below this id
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (Z[mid] < id) lo = mid + 1;
      else hi = mid;
    }
    atomicAdd(&sh.nzc[lo], 1u);
    if (v != v) continue;    // absent: no entry at all
    const unsigned long long k = d2ord(v);
    lo = 0;
    hi = skn;                // short-list keys that beat (k, id)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const bool beats = sh.sk_hi[mid] > k || (sh.sk_hi[mid] == k && sh.sk_id[mid] < id);
      if (beats) lo = mid + 1;
      else hi = mid;
    }
    atomicAdd(&sh.H[lo], 1u);
  }
  __syncthreads();
  // zero entries: the ids of interval t — [0, Z[0]], (Z[t-1], Z[t]], (Z[m-1], N - 1] — minus the listed non-zeros in it
  for (int t = tid; t <= m; t += kFdThreads) {
    int64_t span;
    if (m == 0) span = N;
    else if (t == 0) span = (int64_t)Z[0] + 1;
    else if (t < m) span = (int64_t)Z[t] - (int64_t)Z[t - 1];
    else span = N - 1 - (int64_t)Z[m - 1];
    const int64_t zeros = span - (int64_t)sh.nzc[t];
    sh.H[z_lo + t] += (unsigned)(zeros > 0 ? zeros : 0);
  }
  __syncthreads();
  for (int i = tid; i <= skn; i += kFdThreads) p.H[(int64_t)q * (kFdMaxSparse + 1) + i] = sh.H[i];
}

}  // namespace anr

using namespace anr;

namespace {
FuseArena g_fd_arena[kFuseMaxDevices];
void launch_scan(int method, unsigned grid, hipStream_t st, const FdParams &p, int64_t n_items) {
  // the prefetched source, as the kernel picks it
  const int s0 = method == 1 ? p.r1_src : (p.src[0].arr ? 0 : p.src[1].arr ? 1 : p.src[2].arr ? 2 : 3);
  const int sel = (method == 1 ? 2 : 0) + (p.src[s0].dtype ? 1 : 0);
  const dim3 g(grid), b(kFsThreads);
  switch (sel) {
    case 0: hipLaunchKernelGGL((k_fd_scan<0, 0>), g, b, sizeof(FsShared), st, p, n_items); break;
    case 1: hipLaunchKernelGGL((k_fd_scan<0, 1>), g, b, sizeof(FsShared), st, p, n_items); break;
    case 2: hipLaunchKernelGGL((k_fd_scan<1, 0>), g, b, sizeof(FsShared), st, p, n_items); break;
    default: hipLaunchKernelGGL((k_fd_scan<1, 1>), g, b, sizeof(FsShared), st, p, n_items); break;
  }
}
}  // namespace

extern "C" int anr_fuse_dense(int32_t device, int32_t method, int64_t nq, const anr_fuse_source *src,
                              const double *weights, double rrf_k, int32_t pool, int64_t *out_ids, double *out_final,
                              double *out_src, int32_t *out_count, anr_fuse_dense_stats *stats) {
  if (nq < 0 || !src || !weights || pool <= 0 || !out_ids || !out_final || !out_src || !out_count)
    return fail(ANR_EINVAL, "bad argument");
  if (method != 0 && method != 1) return fail(ANR_EINVAL, "method must be 0 (linear) or 1 (rrf)");
  if (pool > kFdMaxSparse) return fail(ANR_EINVAL, "pool must be <= %d", kFdMaxSparse);
  if (nq == 0) return ANR_OK;
  int n_arr = 0, r1 = -1, sp_src = -1, arr_src = -1;
  int64_t U = 0;
  for (int s = 0; s < 4; ++s) {
    if (src[s].sparse_ids_dev) {
      if (src[s].array_dev || src[s].list_offs) return fail(ANR_EINVAL, "source %d is given in two forms", s);
      if (s == 3 || sp_src >= 0) return fail(ANR_EINVAL, "one of dense / bm25 / graph may be given in sparse form");
      if (!src[s].sparse_scores_dev || !src[s].sparse_count_dev) return fail(ANR_EINVAL, "source %d: null sparse arrays", s);
      if (src[s].sparse_cap <= 0 || src[s].sparse_cap > kFsSpMaxBig)
        return fail(ANR_EINVAL, "source %d: sparse_cap must be in [1, %d]", s, kFsSpMaxBig);
      if (src[s].array_len <= 0 || src[s].array_len > 0xfffffff0ll) return fail(ANR_EINVAL, "source %d: bad array length", s);
      sp_src = s;
      ++n_arr;
      U = std::max<int64_t>(U, src[s].array_len);
      r1 = s;
    } else if (src[s].array_dev) {
      if (src[s].list_offs) return fail(ANR_EINVAL, "source %d is given both as an array and as lists", s);
      if (src[s].array_len <= 0 || src[s].array_len > 0xfffffff0ll) return fail(ANR_EINVAL, "source %d: bad array length", s);
      if (src[s].array_dtype != 0 && src[s].array_dtype != 1) return fail(ANR_EINVAL, "source %d: dtype must be 0 (f64) or 1 (f32)", s);
      ++n_arr;
      arr_src = s;
      U = std::max<int64_t>(U, src[s].array_len);
      if (s < 3) r1 = s;
      if (method == 1 && s == 3) return fail(ANR_EINVAL, "rrf: the path source must be a list");
    } else if (src[s].list_offs && (!src[s].list_ids || !src[s].list_scores) && src[s].list_offs[nq] > src[s].list_offs[0]) {
      return fail(ANR_EINVAL, "source %d: null list pointers", s);
    }
  }
  if (n_arr == 0) return fail(ANR_EINVAL, "no array source: use anr_fuse_lists");
  if (sp_src >= 0 && n_arr != 1)
    return fail(ANR_EINVAL, "a sparse source stands alone: the other sources must be short lists (%d arrays given)", n_arr);
  if (method == 1 && n_arr != 1)
    return fail(ANR_EINVAL, "rrf handles ONE array source (its ranks are counted in the stream); %d given", n_arr);
  if (method == 1) {
    // the streaming pass keeps the K' LARGEST raw values of the array source: the best finals only while its weight
    // is not negative (w / (rrf_k + rank) then grows with the rank, and the lowest-valued ids would win); and its tie
    // key packs the rank into 28 bits
    if (!(weights[r1] >= 0.0))
      return fail(ANR_EINVAL, "rrf: the array source's weight must be >= 0 (%g given): use anr_fuse_rrf_long", weights[r1]);
    if (src[r1].array_len >= (1ll << 28))
      return fail(ANR_EINVAL, "rrf: the array source holds %lld entries (at most 2^28 - 1): use anr_fuse_rrf_long",
                  (long long)src[r1].array_len);
  }
  // flatten the short lists of all queries into the anr_fuse_lists layout
  std::vector<int64_t> offs((size_t)nq * 5), lids;
  std::vector<double> lsc;
  for (int64_t q = 0; q < nq; ++q) {
    const int64_t start = (int64_t)lids.size();
    for (int s = 0; s < 4; ++s) {
      offs[q * 5 + s] = (int64_t)lids.size();
      if (!src[s].array_dev && !src[s].sparse_ids_dev && src[s].list_offs) {
        const int64_t a = src[s].list_offs[q], b = src[s].list_offs[q + 1];
        if (b < a) return fail(ANR_EINVAL, "source %d: offsets must be non-decreasing", s);
        for (int64_t e = a; e < b; ++e) {
          if (src[s].list_ids[e] < 0 || src[s].list_ids[e] > 0xfffffff0ll)
            return fail(ANR_EINVAL, "source %d: id %lld out of range", s, (long long)src[s].list_ids[e]);
          lids.push_back(src[s].list_ids[e]);
          lsc.push_back(src[s].list_scores[e]);
        }
      }
    }
    offs[q * 5 + 4] = (int64_t)lids.size();
    const int64_t m = (int64_t)lids.size() - start;
    if (m > kFdMaxSparse)
      return fail(ANR_EINVAL, "query %lld: %lld short-list entries beside the arrays (at most %d)", (long long)q,
                  (long long)m, kFdMaxSparse);
    if ((int64_t)n_arr * (pool + 2 * m) + m > kFuseMax)
      return fail(ANR_EINVAL, "query %lld: pool %d with %lld short-list entries and %d arrays exceeds the fused kernel's %d entries",
                  (long long)q, pool, (long long)m, n_arr, kFuseMax);
  }
  if (device < 0 || device >= kFuseMaxDevices) return fail(ANR_EINVAL, "device %d out of range", device);
  DeviceGuard g(device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", device);
  const int lcap = (int)std::max<int64_t>(256, round_up(std::min<int64_t>(kFdMaxK, pool + kFdMaxSparse), 64));
  const bool sparse = sp_src >= 0;
  const int sp_cap = sparse ? (int)src[sp_src].sparse_cap : 0;
  // (sparse: the "chunks" are just consecutive runs of the staged candidates — the entries and up to K' unlisted ids)
  const int n_chunks = sparse ? (int)ceil_div(sp_cap + kFdMaxK, lcap) : (int)ceil_div(U, kFsChunk);
  const bool sp_big = sp_cap > kFsSpMax;  // rows the caller sorted by id (k_fs_take checks)
  int sp_p2 = 2;
  while (sp_p2 < sp_cap && !sp_big) sp_p2 <<= 1;
  // query sub-batches so that the candidate lists stay below ~1 GiB
  const int64_t per_q = (int64_t)n_chunks * lcap * 12;
  const int64_t QB = std::max<int64_t>(1, std::min<int64_t>(nq, ((int64_t)1 << 30) / std::max<int64_t>(per_q, 1)));
  // ---- carve the arena ----
  Carve dc, hc;
  const size_t n_ent = lids.size();
  // uploaded once per call: [ids | scores | offsets of all queries]
  const size_t d_up = dc.take(n_ent * 16 + (size_t)nq * 5 * 8);
  // zeroed per sub-batch with one memset: [H | smax | T | tau0 | c_cnt]
  const size_t z_H = 0, z_smax = z_H + (((size_t)QB * (kFdMaxSparse + 1) * 4 + 7) & ~(size_t)7), z_T = z_smax + (size_t)QB * 32,
               z_tau = z_T + (size_t)QB * 8, z_cnt = z_tau + (size_t)QB * 8, z_ovf = z_cnt + (size_t)QB * n_chunks * 4,
               z_bytes = z_ovf + (size_t)(QB + 2) * 4;  // (+ the sparse-row error word)
  // (the 64-bit words behind the odd-sized histogram — 4100 bytes per query — must stay 8-byte aligned: k_fd_max and
  // the scan apply 64-bit atomics to them, and a 4-byte-aligned one raises a bus error)
  if ((z_smax | z_T | z_tau) & 7) return fail(ANR_EINTERNAL, "fuse_dense: misaligned 64-bit work area");
  const size_t d_zero = dc.take(z_bytes);
  const size_t d_kp = dc.take((size_t)QB * 4), d_su = dc.take((size_t)QB * kFdMaxSparse * 4), d_sun = dc.take((size_t)QB * 4),
               d_skh = dc.take((size_t)QB * kFdMaxSparse * 8), d_ski = dc.take((size_t)QB * kFdMaxSparse * 4),
               d_skn = dc.take((size_t)QB * 4), d_chi = dc.take((size_t)QB * n_chunks * lcap * 8),
               d_cid = dc.take((size_t)QB * n_chunks * lcap * 4), d_oi = dc.take((size_t)QB * kFuseMax * 8),
               d_os = dc.take((size_t)QB * kFuseMax * 8), d_or = dc.take((size_t)QB * kFuseMax * 4),
               d_oo = dc.take((size_t)QB * 5 * 8), d_om = dc.take((size_t)QB * 4 * 8);
  const size_t d_spi = dc.take((size_t)QB * sp_cap * 4), d_spv = dc.take((size_t)QB * sp_cap * 8), d_spc = dc.take((size_t)QB * 4);
  // results of a sub-batch, downloaded with one copy: [ids | final | per-source | count]
  const size_t out_bytes = (size_t)QB * ((size_t)pool * 48 + 4) + 8;  // (+ the sparse-row error word)
  const size_t d_out = dc.take(out_bytes);
  const size_t h_up = hc.take(n_ent * 16 + (size_t)nq * 5 * 8), h_out = hc.take(out_bytes);
  FuseArena &ar = g_fd_arena[device];
  std::lock_guard<std::mutex> lock(ar.mu);
  ANR_TRY(ar.reserve(dc.off, hc.off));
  char *D = ar.dev, *Hs = ar.host;
  ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_fd_scan<0, 0>), (int)sizeof(FsShared)));
  ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_fd_scan<0, 1>), (int)sizeof(FsShared)));
  ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_fd_scan<1, 0>), (int)sizeof(FsShared)));
  ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_fd_scan<1, 1>), (int)sizeof(FsShared)));
  ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_fd_build), (int)sizeof(FdShared)));
  if (sparse) {
    if (!sp_big) ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_fs_sort), sp_p2 * 8));
    ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_fs_stage<false>), (int)sizeof(FsStageShared)));
    ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_fs_stage<true>), (int)sizeof(FsStageShared)));
  }
  ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_fuse<true>), (int)sizeof(FuseShared)));
  hipEvent_t ev[2] = {nullptr, nullptr};
  const bool timed = stats != nullptr;
  if (timed) {
    ANR_HIP(hipEventCreate(&ev[0]));
    ANR_HIP(hipEventCreate(&ev[1]));
    *stats = anr_fuse_dense_stats{};
  }
  int rc = ANR_OK;
  hipStream_t st = nullptr;
  // one upload for the whole call
  if (n_ent) {
    std::memcpy(Hs + h_up, lids.data(), n_ent * 8);
    std::memcpy(Hs + h_up + n_ent * 8, lsc.data(), n_ent * 8);
  }
  std::memcpy(Hs + h_up + n_ent * 16, offs.data(), (size_t)nq * 5 * 8);
  {
    const hipError_t e = hipMemcpyAsync(D + d_up, Hs + h_up, n_ent * 16 + (size_t)nq * 5 * 8, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) rc = fail(ANR_EHIP, "fuse_dense upload failed: %s", hipGetErrorString(e));
  }
  const int n_cu = device_cu_count(device);
  for (int64_t q0 = 0; q0 < nq && rc == ANR_OK; q0 += QB) {
    const int64_t nb = std::min(QB, nq - q0);
    FdParams p{};
    p.method = method;
    p.r1_src = method == 1 ? r1 : -1;
    p.pool = pool;
    for (int s = 0; s < 4; ++s) {
      p.src[s].arr = src[s].array_dev;
      p.src[s].dtype = src[s].array_dtype;
      p.src[s].len = (src[s].array_dev || s == sp_src) ? src[s].array_len : 0;
      p.src[s].known_max = src[s].array_dev ? src[s].array_max_dev : nullptr;
      p.w[s] = weights[s];
    }
    p.sp.src = sp_src;
    if (sparse) {
      p.sp.cap = sp_cap;
      p.sp.in_id = src[sp_src].sparse_ids_dev;
      p.sp.in_val = src[sp_src].sparse_scores_dev;
      p.sp.in_cnt = src[sp_src].sparse_count_dev;
      p.sp.id = reinterpret_cast<unsigned *>(D + d_spi);
      p.sp.val = reinterpret_cast<double *>(D + d_spv);
      p.sp.cnt = reinterpret_cast<int *>(D + d_spc);
      p.sp.err = reinterpret_cast<unsigned *>(D + d_zero + z_ovf) + QB + 1;
    }
    p.rrf_k = rrf_k;
    p.U = U;
    p.q0 = q0;
    p.l_ids = reinterpret_cast<int64_t *>(D + d_up);
    p.l_sc = reinterpret_cast<double *>(D + d_up + n_ent * 8);
    p.l_offs = reinterpret_cast<int64_t *>(D + d_up + n_ent * 16) + q0 * 5;
    p.kprime = reinterpret_cast<int *>(D + d_kp);
    p.su_id = reinterpret_cast<unsigned *>(D + d_su);
    p.su_n = reinterpret_cast<int *>(D + d_sun);
    p.sk_hi = reinterpret_cast<unsigned long long *>(D + d_skh);
    p.sk_id = reinterpret_cast<unsigned *>(D + d_ski);
    p.sk_n = reinterpret_cast<int *>(D + d_skn);
    p.H = reinterpret_cast<unsigned *>(D + d_zero + z_H);
    p.smax_ord = reinterpret_cast<unsigned long long *>(D + d_zero + z_smax);
    p.T = reinterpret_cast<unsigned long long *>(D + d_zero + z_T);
    p.tau0 = reinterpret_cast<unsigned long long *>(D + d_zero + z_tau);
    p.c_cnt = reinterpret_cast<unsigned *>(D + d_zero + z_cnt);
    p.ovf = reinterpret_cast<unsigned *>(D + d_zero + z_ovf);
    p.ovf_any = (int)nb;
    p.c_hi = reinterpret_cast<unsigned long long *>(D + d_chi);
    p.c_id = reinterpret_cast<unsigned *>(D + d_cid);
    p.lcap = lcap;
    p.n_chunks = n_chunks;
    p.o_ids = reinterpret_cast<int64_t *>(D + d_oi);
    p.o_sc = reinterpret_cast<double *>(D + d_os);
    p.o_rank = reinterpret_cast<int *>(D + d_or);
    p.o_offs = reinterpret_cast<int64_t *>(D + d_oo);
    p.o_smax = reinterpret_cast<double *>(D + d_om);
    hipError_t e = hipMemsetAsync(D + d_zero, 0, z_bytes, st);
    if (e != hipSuccess) {
      rc = fail(ANR_EHIP, "fuse_dense setup failed: %s", hipGetErrorString(e));
      break;
    }
    if (sparse) {
      // the entries replace the stream: sort them by id (the lookup table of prep / build), then stage what the scan
      // would have left — candidates, the rank histogram, the row maximum
      if (timed) (void)hipEventRecord(ev[0], st);
      if (sp_big) hipLaunchKernelGGL(k_fs_take, dim3((unsigned)nb), dim3(kFsSortThreads), 0, st, p);
      else hipLaunchKernelGGL(k_fs_sort, dim3((unsigned)nb), dim3(kFsSortThreads), (size_t)sp_p2 * 8, st, p, sp_p2);
      hipLaunchKernelGGL(k_fd_prep, dim3((unsigned)nb), dim3(kFdThreads), 0, st, p);
      if (sp_big) hipLaunchKernelGGL(k_fs_stage<true>, dim3((unsigned)nb), dim3(kFdThreads), sizeof(FsStageShared), st, p);
      else hipLaunchKernelGGL(k_fs_stage<false>, dim3((unsigned)nb), dim3(kFdThreads), sizeof(FsStageShared), st, p);
    } else {
      hipLaunchKernelGGL(k_fd_prep, dim3((unsigned)nb), dim3(kFdThreads), 0, st, p);
      if (timed) (void)hipEventRecord(ev[0], st);
      if (method == 0) {
        bool need_pass = false;  // a max pass over the arrays only for the sources whose maxima the caller did not supply
        for (int s = 0; s < 3; ++s) need_pass = need_pass || (p.src[s].arr && !p.src[s].known_max);
        const int64_t n8 = ceil_div(U, kFdChunk);
        const int64_t gx = std::max<int64_t>(1, std::min<int64_t>(n8, ceil_div(4 * (int64_t)n_cu, nb)));
        if (need_pass) hipLaunchKernelGGL(k_fd_max, dim3((unsigned)gx, (unsigned)nb), dim3(kFdThreads), 0, st, p);
      }
      p.chunk0 = 0;
      p.prefix = 1;
      launch_scan(method, (unsigned)std::min<int64_t>(nb, 2 * n_cu), st, p, nb);
      if (n_chunks > 1) {
        p.chunk0 = 1;
        p.prefix = 0;
        const int64_t items = nb * (int64_t)(n_chunks - 1);
        // linear with one array source and finite weights: the barrier-free pass first, the regular scan for the queries
        // it flags (none on a sparse vector)
        const bool free_pass = method == 0 && n_arr == 1 && std::isfinite(weights[arr_src]) && std::isfinite(weights[3]);
        if (free_pass) {
          const dim3 g((unsigned)std::min<int64_t>(items, 4 * (int64_t)n_cu)), b(kFsThreads);
          if (src[arr_src].array_dtype == 0) hipLaunchKernelGGL(k_fd_scan_free<0>, g, b, 0, st, p, items);
          else hipLaunchKernelGGL(k_fd_scan_free<1>, g, b, 0, st, p, items);
          hipLaunchKernelGGL(k_fd_redo_prep, dim3((unsigned)nb), dim3(256), 0, st, p);
          p.only_flagged = 1;
        }
        if (method == 1) {  // rrf: the barrier-free pass keeps the rank bookkeeping and drops the staging protocol
          const dim3 g((unsigned)std::min<int64_t>(items, 4 * (int64_t)n_cu)), b(kFsThreads);
          if (src[r1].array_dtype == 0) hipLaunchKernelGGL((k_fd_scan<1, 0, true>), g, b, sizeof(FsSharedFree), st, p, items);
          else hipLaunchKernelGGL((k_fd_scan<1, 1, true>), g, b, sizeof(FsSharedFree), st, p, items);
          hipLaunchKernelGGL(k_fd_redo_prep, dim3((unsigned)nb), dim3(256), 0, st, p);
          p.only_flagged = 1;
        }
        launch_scan(method, (unsigned)std::min<int64_t>(items, 2 * n_cu), st, p, items);
        p.only_flagged = 0;
      }
    }
    if (timed) (void)hipEventRecord(ev[1], st);
    hipLaunchKernelGGL(k_fd_build, dim3((unsigned)nb), dim3(kFdThreads), sizeof(FdShared), st, p);
    // the sub-batch's results, packed: [ids nb*pool | final nb*pool | per-source nb*pool*4 | count nb]
    const size_t o_ids = 0, o_fin = (size_t)nb * pool * 8, o_src = o_fin + (size_t)nb * pool * 8,
                 o_cnt = o_src + (size_t)nb * pool * 32, o_err = o_cnt + (size_t)nb * 4, o_bytes = o_err + 4;
    FuseParams fp{};
    fp.method = method;
    fp.ids = p.o_ids;
    fp.scores = p.o_sc;
    fp.offs = p.o_offs;
    for (int s = 0; s < 4; ++s) fp.w[s] = weights[s];
    fp.rrf_k = rrf_k;
    fp.pool = pool;
    fp.out_ids = reinterpret_cast<int64_t *>(D + d_out + o_ids);
    fp.out_final = reinterpret_cast<double *>(D + d_out + o_fin);
    fp.out_src = reinterpret_cast<double *>(D + d_out + o_src);
    fp.out_count = reinterpret_cast<int *>(D + d_out + o_cnt);
    fp.smax_ovr = p.o_smax;
    fp.rank_ovr = p.o_rank;
    hipLaunchKernelGGL(k_fuse<true>, dim3((unsigned)nb), dim3(1024), sizeof(FuseShared), st, fp);
    e = hipGetLastError();
    // the sparse rows' error word rides in the same download
    if (e == hipSuccess && sparse) e = hipMemcpyAsync(D + d_out + o_err, p.sp.err, 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(Hs + h_out, D + d_out, o_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      rc = fail(ANR_EHIP, "fuse_dense failed: %s", hipGetErrorString(e));
      break;
    }
    if (sparse) {
      unsigned err = 0;
      std::memcpy(&err, Hs + h_out + o_err, 4);
      if (err) {
        rc = fail(ANR_EINVAL, "fuse_dense: malformed sparse rows in queries %lld..%lld:%s%s%s%s%s", (long long)q0, (long long)(q0 + nb - 1),
                  (err & 1u) ? " a row count < 0 (the producer's overflow mark: that query needs the N-vector form)" : "",
                  (err & 2u) ? " a row count above the row capacity" : "", (err & 4u) ? " an id >= array_len" : "",
                  (err & 8u) ? " an id listed twice" : "",
                  (err & 16u) ? " a row of a call with sparse_cap > 8192 that is not sorted by id" : "");
        break;
      }
    }
    std::memcpy(out_ids + q0 * pool, Hs + h_out + o_ids, (size_t)nb * pool * 8);
    std::memcpy(out_final + q0 * pool, Hs + h_out + o_fin, (size_t)nb * pool * 8);
    std::memcpy(out_src + q0 * pool * 4, Hs + h_out + o_src, (size_t)nb * pool * 32);
    std::memcpy(out_count + q0, Hs + h_out + o_cnt, (size_t)nb * 4);
    if (timed) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) stats->scan_ms += ms;
      int64_t bytes = 0;
      for (int s = 0; s < 4; ++s)
        if (src[s].array_dev) bytes += src[s].array_len * (src[s].array_dtype == 0 ? 8 : 4);
      if (sparse) bytes = (int64_t)sp_cap * 12;  // (an upper bound: the rows' capacity, not their fill)
      stats->scan_bytes += bytes * nb;
      stats->n_queries += nb;
      std::vector<unsigned> cc((size_t)nb * n_chunks);
      if (hipMemcpy(cc.data(), p.c_cnt, cc.size() * 4, hipMemcpyDeviceToHost) == hipSuccess)
        for (unsigned v : cc) stats->n_candidates += v;
    }
  }
  for (auto &e : ev)
    if (e) (void)hipEventDestroy(e);
  return rc;
}
