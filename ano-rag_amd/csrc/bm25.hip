// Device BM25 scoring behind anr_bm25_* (include/anorag.h): the per-query scoring loop of the reference's
// SimpleBM25.get_scores / bm25_scores (utils/bm25_search.py:43-63, :286-340) as a CSR postings scatter.
//
// The host precomputes, in float64 and with the reference's own expression, the weight of every posting
//   w(t, d) = idf_t * (tf * (k1 + 1) / (tf + k1 * (1 - b + b * (dl_d / avgdl))))
// so that score(q, d) = sum over the query tokens (with repetition, in query order) of w(token, d).
// One workgroup per query walks the query's tokens in order; inside one token every document occurs at most
// once, so its postings are added concurrently (float64 L2 atomics) and a barrier separates the tokens: every
// document receives its additions in exactly the reference's order -> bit-identical float64 scores.
// Then max-normalisation (scores / max when max > 0) as bm25_scores does.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace anr {

struct Bm25Params {
  const int64_t *indptr;   // [n_terms + 1]
  const int32_t *docs;     // [nnz], ascending inside each term's list (host-checked at anr_bm25_create)
  const double *weights;   // [nnz]
  int64_t n_docs;
  const int64_t *q_indptr; // [nq + 1]
  const int32_t *q_terms;
  double *scores;          // [nq][n_docs], zeroed
  int normalize;
  double *max_out;         // [nq] maximum of each query's OUTPUT row, or nullptr
};

__global__ __launch_bounds__(1024) void k_bm25(Bm25Params p) {
  __shared__ double s_red[16];
  __shared__ double s_max;
  const int q = blockIdx.x, tid = threadIdx.x;
  double *sc = p.scores + (int64_t)q * p.n_docs;
  int64_t touched = 0;  // postings added (an upper bound of the documents touched)
  for (int64_t t = p.q_indptr[q]; t < p.q_indptr[q + 1]; ++t) {
    const int term = p.q_terms[t];
    const int64_t lo = p.indptr[term], hi = p.indptr[term + 1];
    for (int64_t e = lo + tid; e < hi; e += 1024) atomicAdd(sc + p.docs[e], p.weights[e]);
    touched += hi - lo;
    __syncthreads();  // token order == the reference's addition order
  }
  if (!p.normalize && !p.max_out) return;
  __threadfence_block();
  // the row's maximum.  Fewer postings than documents: some document kept its 0.0, and the maximum is the larger of 0.0
  // and the largest touched score — a pass over the query's postings instead of over all n_docs scores.
  double m = 0.0;
  bool any = false;
  if (touched < p.n_docs) {
    any = true;
    for (int64_t t = p.q_indptr[q]; t < p.q_indptr[q + 1]; ++t) {
      const int term = p.q_terms[t];
      const int64_t lo = p.indptr[term], hi = p.indptr[term + 1];
      for (int64_t e = lo + tid; e < hi; e += 1024) {
        const double v = __hip_atomic_load(sc + p.docs[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        m = v > m ? v : m;
      }
    }
  } else {
    for (int64_t d = tid; d < p.n_docs; d += 1024) {
      const double v = __hip_atomic_load(sc + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      m = (!any || v > m) ? v : m;
      any = true;
    }
  }
  if (!any) m = -__builtin_inf();
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) s_red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    double mm = s_red[0];
    for (int w = 1; w < 16; ++w) mm = fmax(mm, s_red[w]);
    s_max = mm;
  }
  __syncthreads();
  const double mx = s_max;
  const bool divide = p.normalize && mx > 0.0;
  if (tid == 0 && p.max_out) p.max_out[q] = divide ? 1.0 : mx;  // (mx / mx == 1.0 exactly; division is monotone)
  if (!divide) return;
  if (touched >= p.n_docs) {  // a query whose postings cover the corpus: one pass over the row
    for (int64_t d = tid; d < p.n_docs; d += 1024) {
      const double v = __hip_atomic_load(sc + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sc[d] = v / mx;
    }
    return;
  }
  // Only the touched documents are non-zero (0 / mx = 0): divide THEM, not the whole row — at 1 M notes the row pass
  // read and wrote 16 MB per query (3.2 GB per 200 queries, most of this kernel's time) for ~10 K touched documents.
  // A document listed under several of the query's tokens must be divided once: by its FIRST occurrence, i.e. the
  // posting whose document is in no earlier token's list (each list is sorted by document: bisection).
  const int64_t t_lo = p.q_indptr[q], t_hi = p.q_indptr[q + 1];
  for (int64_t t = t_lo; t < t_hi; ++t) {
    const int term = p.q_terms[t];
    const int64_t lo = p.indptr[term], hi = p.indptr[term + 1];
    for (int64_t e = lo + tid; e < hi; e += 1024) {
      const int32_t doc = p.docs[e];
      bool first = true;
      for (int64_t u = t_lo; u < t && first; ++u) {
        const int tu = p.q_terms[u];
        int64_t a = p.indptr[tu], b = p.indptr[tu + 1];
        while (a < b) {
          const int64_t mid = (a + b) >> 1;
          if (p.docs[mid] < doc) a = mid + 1;
          else b = mid;
        }
        first = !(a < p.indptr[tu + 1] && p.docs[a] == doc);
      }
      if (first) {
        const double v = __hip_atomic_load(sc + doc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sc[doc] = v / mx;
      }
    }
  }
}

// ---- the same scoring split over DOCUMENT RANGES (round 4) -------------------------------------------------------
// k_bm25 walks a query with ONE workgroup — a barrier per token, the longest query of the batch sets the time (0.95 ms
// per 200 common-term queries at 1 M notes).  A document's additions only have to stay in token order among THEMSELVES:
// workgroup (q, r) owns the documents [r N / R, (r + 1) N / R), bisects every posting list of its query to that range once
// (all tokens' bisections run side by side, their bounds kept in LDS) and walks the tokens in order over its slice —
// the same additions to every document in the same order, so the same float64 bits.  The row maximum meets in one
// ordered 64-bit atomic per workgroup; a second launch divides the touched documents (or the whole slice of a query
// whose postings cover the corpus) exactly as k_bm25 does.
__device__ __forceinline__ unsigned long long bm_d2ord(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double bm_ord2d(unsigned long long o) {
  return __longlong_as_double((long long)((o >> 63) ? (o & 0x7fffffffffffffffull) : ~o));
}
constexpr int kBmMaxTok = 512;  // tokens of a query whose slice bounds fit the workgroup's table (longer: bisected in place)

struct Bm25RangeParams {
  Bm25Params b;
  int n_ranges;
  unsigned long long *max_ord;  // [nq], zeroed: ordered image of the largest touched score (0 = none)
};

__device__ __forceinline__ int64_t bm_lower(const int32_t *docs, int64_t a, int64_t b, int64_t doc) {
  while (a < b) {
    const int64_t mid = (a + b) >> 1;
    if (docs[mid] < doc) a = mid + 1;
    else b = mid;
  }
  return a;
}

// slice bounds of every token's posting list for the workgroup's document range: bounds[2 i], bounds[2 i + 1]
template <typename P>
__device__ __forceinline__ void bm_bounds(const P &p, int64_t t_lo, int n_tok, int64_t d0, int64_t d1, int64_t *bounds) {
  for (int i = threadIdx.x; i < 2 * n_tok; i += blockDim.x) {
    const int term = p.q_terms[t_lo + (i >> 1)];
    bounds[i] = bm_lower(p.docs, p.indptr[term], p.indptr[term + 1], (i & 1) ? d1 : d0);
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void k_bm25_range_add(Bm25RangeParams rp) {
  const Bm25Params &p = rp.b;
  __shared__ int64_t s_bounds[2 * kBmMaxTok];
  __shared__ double s_red[16];
  const int q = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  const int64_t d0 = p.n_docs * r / rp.n_ranges, d1 = p.n_docs * (r + 1) / rp.n_ranges;
  const int64_t t_lo = p.q_indptr[q], t_hi = p.q_indptr[q + 1];
  double *sc = p.scores + (int64_t)q * p.n_docs;
  int64_t touched_all = 0;  // postings of the whole query (every workgroup of the query sees the same figure)
  for (int64_t t = t_lo; t < t_hi; ++t) touched_all += p.indptr[p.q_terms[t] + 1] - p.indptr[p.q_terms[t]];
  for (int64_t tb = t_lo; tb < t_hi; tb += kBmMaxTok) {
    const int n_tok = (int)(t_hi - tb < kBmMaxTok ? t_hi - tb : kBmMaxTok);
    bm_bounds(p, tb, n_tok, d0, d1, s_bounds);
    for (int i = 0; i < n_tok; ++i) {
      const int64_t lo = s_bounds[2 * i], hi = s_bounds[2 * i + 1];
      for (int64_t e = lo + tid; e < hi; e += 1024) atomicAdd(sc + p.docs[e], p.weights[e]);
      __syncthreads();  // token order == the reference's addition order
    }
  }
  if (!p.normalize && !p.max_out) return;
  __threadfence_block();
  // this slice's share of the row maximum: its touched documents, or every document of the slice when the query's postings
  // cover the corpus (no document is then known to have kept its 0.0)
  double m = -__builtin_inf();
  if (touched_all < p.n_docs) {
    for (int64_t tb = t_lo; tb < t_hi; tb += kBmMaxTok) {
      const int n_tok = (int)(t_hi - tb < kBmMaxTok ? t_hi - tb : kBmMaxTok);
      if (t_hi - t_lo > kBmMaxTok) bm_bounds(p, tb, n_tok, d0, d1, s_bounds);  // (one block of tokens: the table is still valid)
      for (int i = 0; i < n_tok; ++i)
        for (int64_t e = s_bounds[2 * i] + tid; e < s_bounds[2 * i + 1]; e += 1024)
          m = fmax(m, __hip_atomic_load(sc + p.docs[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (t_hi - t_lo > kBmMaxTok) __syncthreads();
    }
  } else {
    for (int64_t d = d0 + tid; d < d1; d += 1024) m = fmax(m, __hip_atomic_load(sc + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) s_red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    double mm = s_red[0];
    for (int w = 1; w < 16; ++w) mm = fmax(mm, s_red[w]);
    if (mm > -__builtin_inf()) atomicMax(rp.max_ord + q, bm_d2ord(mm));
  }
}

__global__ __launch_bounds__(1024) void k_bm25_range_norm(Bm25RangeParams rp) {
  const Bm25Params &p = rp.b;
  __shared__ int64_t s_bounds[2 * kBmMaxTok];
  const int q = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  const int64_t d0 = p.n_docs * r / rp.n_ranges, d1 = p.n_docs * (r + 1) / rp.n_ranges;
  const int64_t t_lo = p.q_indptr[q], t_hi = p.q_indptr[q + 1];
  double *sc = p.scores + (int64_t)q * p.n_docs;
  int64_t touched_all = 0;
  for (int64_t t = t_lo; t < t_hi; ++t) touched_all += p.indptr[p.q_terms[t] + 1] - p.indptr[p.q_terms[t]];
  // the row's maximum: the largest touched score, and 0.0 when some document was left untouched (k_bm25's rule)
  const unsigned long long mo = rp.max_ord[q];
  double mx = mo ? bm_ord2d(mo) : -__builtin_inf();
  if (touched_all < p.n_docs) mx = fmax(mx, 0.0);
  const bool divide = p.normalize && mx > 0.0;
  if (tid == 0 && r == 0 && p.max_out) p.max_out[q] = divide ? 1.0 : mx;
  if (!divide) return;
  if (touched_all >= p.n_docs) {
    for (int64_t d = d0 + tid; d < d1; d += 1024) sc[d] = sc[d] / mx;
    return;
  }
  // divide every touched document of the slice ONCE: at its first occurrence among the query's tokens
  for (int64_t tb = t_lo; tb < t_hi; tb += kBmMaxTok) {
    const int n_tok = (int)(t_hi - tb < kBmMaxTok ? t_hi - tb : kBmMaxTok);
    bm_bounds(p, tb, n_tok, d0, d1, s_bounds);
    for (int i = 0; i < n_tok; ++i) {
      for (int64_t e = s_bounds[2 * i] + tid; e < s_bounds[2 * i + 1]; e += 1024) {
        const int32_t doc = p.docs[e];
        bool first = true;
        for (int64_t u = t_lo; u < tb + i && first; ++u) {
          const int tu = p.q_terms[u];
          const int64_t a = bm_lower(p.docs, p.indptr[tu], p.indptr[tu + 1], doc);
          first = !(a < p.indptr[tu + 1] && p.docs[a] == doc);
        }
        if (first) sc[doc] = sc[doc] / mx;
      }
    }
    __syncthreads();
  }
}

// The same scoring without the N-vector: one workgroup accumulates its query in an LDS hash table keyed by the
// document (open addressing, linear probing).  Tokens in query order with a barrier between them and every document at
// most once per token: a document's additions happen in the reference's order, one at a time -> the same float64 sums
// as k_bm25.  Output: the touched documents, unordered, for anr_fuse_dense's sparse source.
constexpr int kSpTable = 8192;    // slots (a power of two)
constexpr int kSpMaxCap = 6144;   // documents a row may hold: the table stays below 7/8 full even when every thread of
                                  // the round that crosses the limit inserts one more
constexpr unsigned kSpEmpty = 0xffffffffu;
struct Bm25SparseParams {
  const int64_t *indptr;
  const int32_t *docs;
  const double *weights;
  int64_t n_docs;
  const int64_t *q_indptr;
  const int32_t *q_terms;
  int normalize;
  int cap;
  unsigned *out_id;    // [nq][cap]
  double *out_val;     // [nq][cap]
  int *out_cnt;        // [nq]: documents written, or -1 (more than cap: the row is unusable)
  double *max_out;     // [nq] or nullptr
};

__global__ __launch_bounds__(1024) void k_bm25_sparse(Bm25SparseParams p) {
  extern __shared__ unsigned char bm_smem[];
  double *vals = reinterpret_cast<double *>(bm_smem);
  unsigned *keys = reinterpret_cast<unsigned *>(bm_smem + (size_t)kSpTable * 8);
  __shared__ double s_red[16];
  __shared__ double s_max;
  __shared__ int s_cnt, s_out;
  const int q = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < kSpTable; i += 1024) {
    keys[i] = kSpEmpty;
    vals[i] = 0.0;
  }
  if (tid == 0) {
    s_cnt = 0;
    s_out = 0;
  }
  __syncthreads();
  for (int64_t t = p.q_indptr[q]; t < p.q_indptr[q + 1]; ++t) {
    const int term = p.q_terms[t];
    const int64_t lo = p.indptr[term], hi = p.indptr[term + 1];
    for (int64_t e = lo + tid; e < hi; e += 1024) {
      if (*reinterpret_cast<volatile int *>(&s_cnt) > p.cap) break;  // overflow: the row is given up below
      const unsigned doc = (unsigned)p.docs[e];
      unsigned slot = (doc * 2654435761u) >> 19;  // 13 bits
      for (;;) {
        const unsigned old = atomicCAS(&keys[slot], kSpEmpty, doc);
        if (old == kSpEmpty) {
          atomicAdd(&s_cnt, 1);
          break;
        }
        if (old == doc) break;
        slot = (slot + 1) & (kSpTable - 1);
      }
      vals[slot] += p.weights[e];  // this token's only posting of the document
    }
    __syncthreads();
  }
  const int count = s_cnt;
  if (count > p.cap) {
    if (tid == 0) {
      p.out_cnt[q] = -1;
      if (p.max_out) p.max_out[q] = __builtin_nan("");
    }
    return;
  }
  // the row's maximum over all n_docs scores: the touched ones, and 0.0 when a document was left untouched
  double m = (int64_t)count < p.n_docs ? 0.0 : -__builtin_inf();
  for (int i = tid; i < kSpTable; i += 1024)
    if (keys[i] != kSpEmpty) m = fmax(m, vals[i]);
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) s_red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    double mm = s_red[0];
    for (int w = 1; w < 16; ++w) mm = fmax(mm, s_red[w]);
    s_max = mm;
  }
  __syncthreads();
  const double mx = s_max;
  const bool divide = p.normalize && mx > 0.0;
  for (int i = tid; i < kSpTable; i += 1024)
    if (keys[i] != kSpEmpty) {
      const int pos = atomicAdd(&s_out, 1);
      p.out_id[(int64_t)q * p.cap + pos] = keys[i];
      p.out_val[(int64_t)q * p.cap + pos] = divide ? vals[i] / mx : vals[i];
    }
  if (tid == 0) {
    p.out_cnt[q] = count;
    if (p.max_out) p.max_out[q] = divide ? 1.0 : mx;
  }
}

// ---- sparse rows beyond one LDS table (round 4): up to kSpMaxCapBig documents per row ---------------------------
// A frequent-word query touches 10 K - 60 K documents — more than one workgroup's table holds.  The row is cut into SLICES
// of the document range (query q owns slices [slice_base[q], slice_base[q + 1]): the host sizes them from the summed
// posting lengths, about kSrTarget postings each, and no wider than kSrWidth ids).  One workgroup per slice:
//   mark    one presence bit per document of the slice, from the slice of every posting list (bisected once, as above)
//   rank    prefix sum of the bit counts: a touched document's position among the slice's touched documents IS its slot —
//           no hashing, no probing, and the slots come out in ascending document order
//   add     tokens in query order, a barrier between them, `vals[slot] += weight` (a list holds a document once): the
//           additions of every document in the reference's order, one at a time -> the float64 sums of k_bm25
//   out     (id, sum) into the slice's staging run, its count, its share of the row maximum (ordered 64-bit atomicMax)
// k_bm25_slice_pack then strings a query's slices together — slices ascend in document range and each is ascending inside,
// so the packed row is SORTED BY ID, which is what anr_fuse_dense asks of rows beyond its own LDS sort — dividing by the
// row maximum exactly as k_bm25_sparse does.  A slice with more than kSrSlice documents or a row with more than `cap`
// gives the row up (count -1), like the small kernel.
constexpr int kSrSlice = 4096;     // documents per slice (float64 sums in LDS)
constexpr int kSrWidth = 131072;   // ids a slice may span
constexpr int kSrTarget = 2048;    // postings per slice the host aims at (>= documents: half the capacity as slack)
constexpr int kSrMaxSlices = 1024; // slices per query (the pack kernel's prefix table)
constexpr int kSpMaxCapBig = 65536;
constexpr int kSrLds = kSrWidth / 8 + (kSrWidth / 32) * 4 + kSrSlice * 8;

struct Bm25SliceParams {
  const int64_t *indptr;
  const int32_t *docs;
  const double *weights;
  int64_t n_docs;
  const int64_t *q_indptr;
  const int32_t *q_terms;
  const int *slice_base;  // [nq + 1]
  int nq, normalize, cap;
  unsigned *st_id;        // [slices][kSrSlice]
  double *st_val;
  int *st_cnt;            // [slices]: documents, or -1 (more than kSrSlice)
  unsigned long long *max_ord;  // [nq], zeroed
  unsigned *out_id;       // [nq][cap]
  double *out_val;
  int *out_cnt;
  double *max_out;        // [nq] or nullptr
};

__global__ __launch_bounds__(1024) void k_bm25_slice(Bm25SliceParams p) {
  extern __shared__ unsigned char bm_smem[];
  unsigned *bits = reinterpret_cast<unsigned *>(bm_smem);
  unsigned *pre = bits + kSrWidth / 32;
  double *vals = reinterpret_cast<double *>(pre + kSrWidth / 32);
  __shared__ int64_t s_bounds[2 * kBmMaxTok];
  __shared__ unsigned s_wave[16];
  __shared__ double s_red[16];
  const int sl = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int q;
  {
    int lo = 0, hi = p.nq - 1;  // the query whose slices hold sl
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (p.slice_base[mid] <= sl) lo = mid;
      else hi = mid - 1;
    }
    q = lo;
  }
  const int R = p.slice_base[q + 1] - p.slice_base[q], r = sl - p.slice_base[q];
  const int64_t d0 = p.n_docs * r / R, d1 = p.n_docs * (r + 1) / R;
  const int nW = (int)((d1 - d0 + 31) >> 5);  // <= kSrWidth / 32 (the host's R)
  const int64_t t_lo = p.q_indptr[q], t_hi = p.q_indptr[q + 1];
  const bool one_block = t_hi - t_lo <= kBmMaxTok;
  for (int i = tid; i < nW; i += 1024) bits[i] = 0u;
  for (int i = tid; i < kSrSlice; i += 1024) vals[i] = 0.0;
  __syncthreads();
  for (int64_t tb = t_lo; tb < t_hi; tb += kBmMaxTok) {
    const int n_tok = (int)(t_hi - tb < kBmMaxTok ? t_hi - tb : kBmMaxTok);
    bm_bounds(p, tb, n_tok, d0, d1, s_bounds);
    for (int i = 0; i < n_tok; ++i)
      for (int64_t e = s_bounds[2 * i] + tid; e < s_bounds[2 * i + 1]; e += 1024) {
        const unsigned rel = (unsigned)(p.docs[e] - d0);
        atomicOr(&bits[rel >> 5], 1u << (rel & 31));
      }
    __syncthreads();
  }
  // exclusive prefix of the words' bit counts: four consecutive words per thread, a wave scan, the waves' totals
  unsigned c[4], mine = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int w = tid * 4 + j;
    c[j] = w < nW ? (unsigned)__popc(bits[w]) : 0u;
    mine += c[j];
  }
  unsigned incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned base = incl - mine, total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) base += s_wave[w];
    total += s_wave[w];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int w = tid * 4 + j;
    if (w < nW) pre[w] = base;
    base += c[j];
  }
  if (total > (unsigned)kSrSlice) {  // (uniform)
    if (tid == 0) p.st_cnt[sl] = -1;
    return;
  }
  __syncthreads();
  for (int64_t tb = t_lo; tb < t_hi; tb += kBmMaxTok) {
    const int n_tok = (int)(t_hi - tb < kBmMaxTok ? t_hi - tb : kBmMaxTok);
    if (!one_block) bm_bounds(p, tb, n_tok, d0, d1, s_bounds);  // (one block: the table is still valid)
    for (int i = 0; i < n_tok; ++i) {
      for (int64_t e = s_bounds[2 * i] + tid; e < s_bounds[2 * i + 1]; e += 1024) {
        const unsigned rel = (unsigned)(p.docs[e] - d0);
        const unsigned slot = pre[rel >> 5] + (unsigned)__popc(bits[rel >> 5] & ((1u << (rel & 31)) - 1u));
        vals[slot] += p.weights[e];  // this token's only posting of the document
      }
      __syncthreads();  // token order == the reference's addition order
    }
  }
  unsigned *oi = p.st_id + (int64_t)sl * kSrSlice;
  double *ov = p.st_val + (int64_t)sl * kSrSlice;
  for (int w = tid; w < nW; w += 1024) {
    unsigned b = bits[w], pos = pre[w];
    while (b) {
      const int k = __ffs((int)b) - 1;
      oi[pos++] = (unsigned)(d0 + (int64_t)w * 32 + k);
      b &= b - 1u;
    }
  }
  double m = -__builtin_inf();
  for (int i = tid; i < (int)total; i += 1024) {
    const double v = vals[i];
    ov[i] = v;
    m = fmax(m, v);
  }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if (lane == 0) s_red[wave] = m;
  __syncthreads();
  if (tid == 0) {
    double mm = s_red[0];
    for (int w = 1; w < 16; ++w) mm = fmax(mm, s_red[w]);
    if (mm > -__builtin_inf()) atomicMax(p.max_ord + q, bm_d2ord(mm));
    p.st_cnt[sl] = (int)total;
  }
}

__global__ __launch_bounds__(1024) void k_bm25_slice_pack(Bm25SliceParams p) {
  __shared__ int s_off[kSrMaxSlices + 1];
  __shared__ int s_bad;
  const int q = blockIdx.x, tid = threadIdx.x;
  const int s0 = p.slice_base[q], R = p.slice_base[q + 1] - s0;
  if (tid == 0) s_bad = R == 0 ? 1 : 0;  // (no slices: the host gave the row up — postings beyond any capacity)
  __syncthreads();
  for (int r = tid; r < R; r += 1024) {
    const int c = p.st_cnt[s0 + r];
    s_off[r + 1] = c;
    if (c < 0) s_bad = 1;
  }
  __syncthreads();
  if (tid == 0 && !s_bad) {
    s_off[0] = 0;
    for (int r = 0; r < R; ++r) s_off[r + 1] += s_off[r];
    if (s_off[R] > p.cap) s_bad = 1;
  }
  __syncthreads();
  if (s_bad) {
    if (tid == 0) {
      p.out_cnt[q] = -1;
      if (p.max_out) p.max_out[q] = __builtin_nan("");
    }
    return;
  }
  const int count = s_off[R];
  // the row's maximum over all n_docs scores: the touched ones, and 0.0 when a document was left untouched
  const unsigned long long mo = p.max_ord[q];
  double mx = mo ? bm_ord2d(mo) : -__builtin_inf();
  if ((int64_t)count < p.n_docs) mx = fmax(mx, 0.0);
  const bool divide = p.normalize && mx > 0.0;
  for (int r = 0; r < R; ++r) {
    const int n = s_off[r + 1] - s_off[r];
    const unsigned *si = p.st_id + (int64_t)(s0 + r) * kSrSlice;
    const double *sv = p.st_val + (int64_t)(s0 + r) * kSrSlice;
    for (int i = tid; i < n; i += 1024) {
      const int64_t at = (int64_t)q * p.cap + s_off[r] + i;
      p.out_id[at] = si[i];
      p.out_val[at] = divide ? sv[i] / mx : sv[i];
    }
  }
  if (tid == 0) {
    p.out_cnt[q] = count;
    if (p.max_out) p.max_out[q] = divide ? 1.0 : mx;
  }
}

// compaction of the non-zero scores of each query: (doc, score) pairs, unordered
struct NzParams {
  const double *scores;
  int64_t n_docs;
  int cap;
  int32_t *out_docs;   // [nq][cap]
  double *out_scores;  // [nq][cap]
  int *out_count;      // [nq] (may exceed cap: truncated)
};

__global__ __launch_bounds__(1024) void k_bm25_nonzero(NzParams p) {
  __shared__ int s_cnt;
  const int q = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  const double *sc = p.scores + (int64_t)q * p.n_docs;
  for (int64_t d = tid; d < p.n_docs; d += 1024) {
    const double v = sc[d];
    if (v != 0.0) {
      const int pos = atomicAdd(&s_cnt, 1);
      if (pos < p.cap) {
        p.out_docs[(int64_t)q * p.cap + pos] = (int32_t)d;
        p.out_scores[(int64_t)q * p.cap + pos] = v;
      }
    }
  }
  __syncthreads();
  if (tid == 0) p.out_count[q] = s_cnt;
}

// FieldWeightedBM25.get_scores (utils/bm25_search.py:116-146): total = sum over the fields, in field order, of
// weight_f * field_score_f — elementwise over the documents, then the max-normalisation of
// field_weighted_bm25_scores (:222-226).  One workgroup per query.
struct CombineParams {
  const double *f[8];
  double w[8];
  int n_fields;
  int64_t n_docs;
  int normalize;
  double *out;  // [nq][n_docs]
  double *max_out;  // [nq] maximum of each query's output row, or nullptr
};

__global__ __launch_bounds__(1024) void k_bm25_combine(CombineParams p) {
  __shared__ double s_red[16];
  __shared__ double s_max;
  const int q = blockIdx.x, tid = threadIdx.x;
  double *o = p.out + (int64_t)q * p.n_docs;
  double m = -__builtin_inf();
  for (int64_t d = tid; d < p.n_docs; d += 1024) {
    double t = 0.0;
    for (int f = 0; f < p.n_fields; ++f) t += p.w[f] * p.f[f][(int64_t)q * p.n_docs + d];
    o[d] = t;
    m = fmax(m, t);
  }
  if (!p.normalize && !p.max_out) return;
  for (int s = 32; s > 0; s >>= 1) m = fmax(m, __shfl_xor(m, s));
  if ((tid & 63) == 0) s_red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    double mm = s_red[0];
    for (int w = 1; w < 16; ++w) mm = fmax(mm, s_red[w]);
    s_max = mm;
  }
  __syncthreads();
  const double mx = s_max;
  const bool divide = p.normalize && mx > 0.0;
  if (tid == 0 && p.max_out) p.max_out[q] = divide ? 1.0 : mx;
  if (divide)
    for (int64_t d = tid; d < p.n_docs; d += 1024) o[d] = o[d] / mx;
}

}  // namespace anr

using namespace anr;

struct anr_bm25 {
  int device = 0;
  int64_t n_docs = 0, n_terms = 0, nnz = 0;
  int64_t *indptr = nullptr;
  int32_t *docs = nullptr;
  double *weights = nullptr;
  hipStream_t stream = nullptr;
  std::mutex mu;
  // per-handle work space, grown on demand and kept (three hipMalloc / hipFree pairs per call were most of a
  // single-query bm25_scores): query offsets + terms, and the score rows of the host-output entry points
  char *q_buf = nullptr;
  int64_t q_cap = 0;
  double *score_buf = nullptr;
  int64_t score_cap = 0;
  unsigned long long *max_ord = nullptr;  // [max_ord_cap] row maxima of the range-split scoring (ordered images)
  int64_t max_ord_cap = 0;
  int n_cu = 256;
  // the sliced sparse rows (cap > kSpMaxCap): host copy of the posting-list offsets (the slices are sized from the summed
  // list lengths), the slices' staging runs and the slice table
  std::vector<int64_t> h_indptr;
  unsigned *st_id = nullptr;
  double *st_val = nullptr;
  int *st_cnt = nullptr;
  int64_t st_cap = 0;      // slices the staging buffers hold
  int *slice_base = nullptr;
  int64_t slice_base_cap = 0;
};

namespace {
template <typename T>
int b_alloc(T **p, int64_t n) {
  *p = nullptr;
  ANR_HIP(hipMalloc(reinterpret_cast<void **>(p), (size_t)(n > 0 ? n : 1) * sizeof(T)));
  return ANR_OK;
}
template <typename T>
void b_free(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

int validate_queries(const anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms) {
  if (!h || nq < 0 || !q_indptr) return fail(ANR_EINVAL, "bad argument");
  if (q_indptr[0] != 0) return fail(ANR_EINVAL, "q_indptr[0] must be 0");
  for (int64_t i = 0; i < nq; ++i)
    if (q_indptr[i + 1] < q_indptr[i]) return fail(ANR_EINVAL, "q_indptr must be non-decreasing");
  if (q_indptr[nq] > 0 && !q_terms) return fail(ANR_EINVAL, "null q_terms");
  for (int64_t e = 0; e < q_indptr[nq]; ++e)
    if (q_terms[e] < 0 || q_terms[e] >= h->n_terms) return fail(ANR_EINVAL, "query term id %d out of range", q_terms[e]);
  return ANR_OK;
}

// runs the scoring of one chunk of queries into the handle's score buffer (*d_scores points into it: valid until the
// next call on the handle) or, when `into` is given, into that caller-owned device buffer
// the queries' offsets (made relative) and terms into the handle's device buffer, on its stream; `rel` must outlive the copy
int upload_queries(anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms, std::vector<int64_t> &rel,
                   int64_t **dq_out, int32_t **dt_out) {
  const int64_t nt = q_indptr[nq] - q_indptr[0];
  rel.resize(nq + 1);
  for (int64_t i = 0; i <= nq; ++i) rel[i] = q_indptr[i] - q_indptr[0];
  const int64_t q_bytes = (nq + 1) * 8 + (nt > 0 ? nt : 1) * 4;
  if (q_bytes > h->q_cap) {
    b_free(h->q_buf);
    h->q_cap = 0;
    ANR_TRY(b_alloc(&h->q_buf, q_bytes + q_bytes / 2));
    h->q_cap = q_bytes + q_bytes / 2;
  }
  int64_t *dq = reinterpret_cast<int64_t *>(h->q_buf);
  int32_t *dt = reinterpret_cast<int32_t *>(h->q_buf + (nq + 1) * 8);
  hipError_t e = hipMemcpyAsync(dq, rel.data(), (size_t)(nq + 1) * 8, hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess && nt > 0)
    e = hipMemcpyAsync(dt, q_terms + q_indptr[0], (size_t)nt * 4, hipMemcpyHostToDevice, h->stream);
  if (e != hipSuccess) return fail(ANR_EHIP, "bm25 query upload failed: %s", hipGetErrorString(e));
  *dq_out = dq;
  *dt_out = dt;
  return ANR_OK;
}

int score_chunk(anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms, int normalize,
                double **d_scores, double *into = nullptr, double *max_into = nullptr) {
  std::vector<int64_t> rel;
  int64_t *dq = nullptr;
  int32_t *dt = nullptr;
  if (into) {
    *d_scores = into;
  } else {
    if (nq * h->n_docs > h->score_cap) {
      b_free(h->score_buf);
      h->score_cap = 0;
      ANR_TRY(b_alloc(&h->score_buf, nq * h->n_docs));
      h->score_cap = nq * h->n_docs;
    }
    *d_scores = h->score_buf;
  }
  ANR_TRY(upload_queries(h, nq, q_indptr, q_terms, rel, &dq, &dt));
  hipError_t e = hipMemsetAsync(*d_scores, 0, (size_t)nq * h->n_docs * 8, h->stream);
  if (e == hipSuccess) {
    Bm25Params p{h->indptr, h->docs, h->weights, h->n_docs, dq, dt, *d_scores, normalize, max_into};
    // document ranges per query: enough workgroups to fill the chip four times over (one query: 32 slices), none for a
    // corpus too small to be worth slicing
    int R = (int)std::min<int64_t>(32, std::max<int64_t>(1, ceil_div((int64_t)4 * h->n_cu, nq)));
    if (h->n_docs < (int64_t)R * 4096) R = (int)std::max<int64_t>(1, h->n_docs / 4096);
    if (R <= 1 || nq > 65535 || getenv("ANORAG_BM25_ONE_WG")) {
      hipLaunchKernelGGL(k_bm25, dim3((unsigned)nq), dim3(1024), 0, h->stream, p);
      e = hipGetLastError();
    } else {
      if (h->max_ord_cap < nq) {
        b_free(h->max_ord);
        h->max_ord_cap = 0;
        ANR_TRY(b_alloc(&h->max_ord, nq));
        h->max_ord_cap = nq;
      }
      e = hipMemsetAsync(h->max_ord, 0, (size_t)nq * 8, h->stream);
      Bm25RangeParams rp{p, R, h->max_ord};
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_bm25_range_add, dim3((unsigned)nq, (unsigned)R), dim3(1024), 0, h->stream, rp);
        if (normalize || max_into)
          hipLaunchKernelGGL(k_bm25_range_norm, dim3((unsigned)nq, (unsigned)R), dim3(1024), 0, h->stream, rp);
        e = hipGetLastError();
      }
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);  // (also: `rel` may go out of scope)
  if (e != hipSuccess) return fail(ANR_EHIP, "bm25 scoring failed: %s", hipGetErrorString(e));
  return ANR_OK;
}

// rows of up to kSpMaxCapBig documents: k_bm25_slice + k_bm25_slice_pack, a sub-batch of queries at a time so that the
// staging runs stay below ~768 MB
int sparse_sliced(anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms, int normalize, int cap,
                  uint32_t *ids_dev, double *scores_dev, int32_t *count_dev, double *max_dev, int32_t *out_count_host) {
  const int64_t width_slices = std::max<int64_t>(1, ceil_div(h->n_docs, (int64_t)kSrWidth));
  if (width_slices > kSrMaxSlices)
    return fail(ANR_EINVAL, "sparse rows beyond %d documents need a corpus of at most %lld documents", kSpMaxCap,
                (long long)kSrMaxSlices * kSrWidth);
  // slices per query from the summed posting lengths (an upper bound of the documents touched); a query whose postings
  // exceed four times the row capacity is given up here (no slices: count -1)
  std::vector<int> n_sl((size_t)nq);
  for (int64_t q = 0; q < nq; ++q) {
    int64_t sum = 0;
    for (int64_t t = q_indptr[q]; t < q_indptr[q + 1]; ++t) sum += h->h_indptr[q_terms[t] + 1] - h->h_indptr[q_terms[t]];
    n_sl[q] = sum > 4 * (int64_t)cap ? 0 : (int)std::min<int64_t>(kSrMaxSlices, std::max<int64_t>(width_slices, ceil_div(sum, (int64_t)kSrTarget)));
  }
  ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_bm25_slice), kSrLds));
  const int64_t kMaxSlices = 16384;
  std::vector<int64_t> rel;
  std::vector<int> base;
  for (int64_t q0 = 0; q0 < nq;) {
    int64_t q1 = q0, total = 0;
    while (q1 < nq && (q1 == q0 || total + n_sl[q1] <= kMaxSlices)) total += n_sl[q1++];
    const int64_t m = q1 - q0;
    base.assign((size_t)m + 1, 0);
    for (int64_t i = 0; i < m; ++i) base[i + 1] = base[i] + n_sl[q0 + i];
    if (total > h->st_cap) {
      b_free(h->st_id);
      b_free(h->st_val);
      b_free(h->st_cnt);
      h->st_cap = 0;
      const int64_t want = total + total / 4;
      ANR_TRY(b_alloc(&h->st_id, want * kSrSlice));
      ANR_TRY(b_alloc(&h->st_val, want * kSrSlice));
      ANR_TRY(b_alloc(&h->st_cnt, want));
      h->st_cap = want;
    }
    if (m + 1 > h->slice_base_cap) {
      b_free(h->slice_base);
      h->slice_base_cap = 0;
      ANR_TRY(b_alloc(&h->slice_base, 2 * (m + 1)));
      h->slice_base_cap = 2 * (m + 1);
    }
    if (h->max_ord_cap < m) {
      b_free(h->max_ord);
      h->max_ord_cap = 0;
      ANR_TRY(b_alloc(&h->max_ord, m));
      h->max_ord_cap = m;
    }
    int64_t *dq = nullptr;
    int32_t *dt = nullptr;
    ANR_TRY(upload_queries(h, m, q_indptr + q0, q_terms, rel, &dq, &dt));
    hipError_t e = hipMemcpyAsync(h->slice_base, base.data(), (size_t)(m + 1) * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->max_ord, 0, (size_t)m * 8, h->stream);
    if (e == hipSuccess) {
      Bm25SliceParams p{h->indptr, h->docs, h->weights, h->n_docs, dq, dt, h->slice_base, (int)m, normalize, cap,
                        h->st_id, h->st_val, h->st_cnt, h->max_ord, ids_dev + q0 * cap, scores_dev + q0 * cap,
                        count_dev + q0, max_dev ? max_dev + q0 : nullptr};
      if (total > 0) hipLaunchKernelGGL(k_bm25_slice, dim3((unsigned)total), dim3(1024), (size_t)kSrLds, h->stream, p);
      hipLaunchKernelGGL(k_bm25_slice_pack, dim3((unsigned)m), dim3(1024), 0, h->stream, p);
      e = hipGetLastError();
    }
    if (e == hipSuccess && out_count_host)
      e = hipMemcpyAsync(out_count_host + q0, count_dev + q0, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);  // (`rel` / `base` are re-used by the next sub-batch)
    if (e != hipSuccess) return fail(ANR_EHIP, "bm25 sparse scoring (sliced rows) failed: %s", hipGetErrorString(e));
    q0 = q1;
  }
  return ANR_OK;
}
}  // namespace

extern "C" {

int anr_bm25_create(int32_t device, int64_t n_docs, int64_t n_terms, const int64_t *indptr, const int32_t *doc_ids,
                    const double *weights, anr_bm25 **out) {
  if (!out) return fail(ANR_EINVAL, "out is null");
  *out = nullptr;
  if (n_docs < 0 || n_terms < 0 || !indptr) return fail(ANR_EINVAL, "bad argument");
  if (n_docs > 0x7fffffffll) return fail(ANR_EINVAL, "at most 2^31-1 documents");
  const int64_t nnz = indptr[n_terms];
  if (indptr[0] != 0 || nnz < 0 || (nnz > 0 && (!doc_ids || !weights))) return fail(ANR_EINVAL, "bad CSR arrays");
  for (int64_t t = 0; t < n_terms; ++t)
    if (indptr[t + 1] < indptr[t]) return fail(ANR_EINVAL, "indptr must be non-decreasing");
  for (int64_t e = 0; e < nnz; ++e)
    if (doc_ids[e] < 0 || doc_ids[e] >= n_docs) return fail(ANR_EINVAL, "posting %lld: document id out of range", (long long)e);
  for (int64_t t = 0; t < n_terms; ++t)  // one posting per (term, document), documents ascending: the scoring kernel
    for (int64_t e = indptr[t] + 1; e < indptr[t + 1]; ++e)  // adds a term's postings concurrently and bisects the lists
      if (doc_ids[e] <= doc_ids[e - 1])
        return fail(ANR_EINVAL, "term %lld: document ids must be strictly ascending inside a posting list", (long long)t);
  int ndev = anr_device_count();
  if (ndev <= 0) return fail(ANR_EHIP, "no HIP device is visible");
  if (device < 0 || device >= ndev) return fail(ANR_EINVAL, "device %d out of range", device);
  DeviceGuard g(device);
  anr_bm25 *h = new anr_bm25();
  h->device = device;
  h->n_docs = n_docs;
  h->n_terms = n_terms;
  h->nnz = nnz;
  h->h_indptr.assign(indptr, indptr + n_terms + 1);
  int rc = ANR_OK;
  h->n_cu = device_cu_count(device);
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(ANR_EHIP, "hipStreamCreate failed");
  if (rc == ANR_OK) rc = b_alloc(&h->indptr, n_terms + 1);
  if (rc == ANR_OK) rc = b_alloc(&h->docs, nnz);
  if (rc == ANR_OK) rc = b_alloc(&h->weights, nnz);
  if (rc == ANR_OK) {
    hipError_t e = hipMemcpy(h->indptr, indptr, (size_t)(n_terms + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz > 0) e = hipMemcpy(h->docs, doc_ids, (size_t)nnz * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz > 0) e = hipMemcpy(h->weights, weights, (size_t)nnz * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(ANR_EHIP, "upload failed: %s", hipGetErrorString(e));
  }
  if (rc != ANR_OK) {
    anr_bm25_destroy(h);
    return rc;
  }
  *out = h;
  return ANR_OK;
}

int anr_bm25_destroy(anr_bm25 *h) {
  if (!h) return ANR_OK;
  DeviceGuard g(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  b_free(h->indptr);
  b_free(h->docs);
  b_free(h->weights);
  b_free(h->q_buf);
  b_free(h->max_ord);
  b_free(h->st_id);
  b_free(h->st_val);
  b_free(h->st_cnt);
  b_free(h->slice_base);
  b_free(h->score_buf);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return ANR_OK;
}

int anr_bm25_scores(anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms, int32_t normalize,
                    double *out_host) {
  ANR_TRY(validate_queries(h, nq, q_indptr, q_terms));
  if (!out_host) return fail(ANR_EINVAL, "out is null");
  if (nq == 0 || h->n_docs == 0) return ANR_OK;
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(nq, (int64_t)(1ll << 30) / (h->n_docs * 8)));
  for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
    const int64_t m = std::min(chunk, nq - q0);
    double *d = nullptr;
    int rc = score_chunk(h, m, q_indptr + q0, q_terms, normalize, &d);
    if (rc == ANR_OK) {
      hipError_t e = hipMemcpy(out_host + q0 * h->n_docs, d, (size_t)m * h->n_docs * 8, hipMemcpyDeviceToHost);
      if (e != hipSuccess) rc = fail(ANR_EHIP, "download failed: %s", hipGetErrorString(e));
    }
    if (rc != ANR_OK) return rc;
  }
  return ANR_OK;
}

int anr_bm25_scores_dev(anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms, int32_t normalize,
                        double *out_dev, double *max_dev) {
  ANR_TRY(validate_queries(h, nq, q_indptr, q_terms));
  if (!out_dev) return fail(ANR_EINVAL, "out is null");
  if (nq == 0 || h->n_docs == 0) return ANR_OK;
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  double *d = nullptr;
  return score_chunk(h, nq, q_indptr, q_terms, normalize, &d, out_dev, max_dev);
}

int anr_bm25_sparse_dev(anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms, int32_t normalize,
                        int32_t cap, uint32_t *ids_dev, double *scores_dev, int32_t *count_dev, double *max_dev,
                        int32_t *out_count_host) {
  ANR_TRY(validate_queries(h, nq, q_indptr, q_terms));
  if (cap <= 0 || cap > kSpMaxCapBig) return fail(ANR_EINVAL, "cap must be in [1, %d]", kSpMaxCapBig);
  if (!ids_dev || !scores_dev || !count_dev) return fail(ANR_EINVAL, "null output arrays");
  if (nq == 0) return ANR_OK;
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  if (cap > kSpMaxCap) return sparse_sliced(h, nq, q_indptr, q_terms, normalize, cap, ids_dev, scores_dev, count_dev, max_dev, out_count_host);
  std::vector<int64_t> rel;
  int64_t *dq = nullptr;
  int32_t *dt = nullptr;
  ANR_TRY(upload_queries(h, nq, q_indptr, q_terms, rel, &dq, &dt));
  const int lds = kSpTable * 12;
  ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_bm25_sparse), lds));
  Bm25SparseParams p{h->indptr, h->docs, h->weights, h->n_docs, dq, dt, normalize, cap, ids_dev, scores_dev, count_dev, max_dev};
  hipLaunchKernelGGL(k_bm25_sparse, dim3((unsigned)nq), dim3(1024), (size_t)lds, h->stream, p);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && out_count_host)
    e = hipMemcpyAsync(out_count_host, count_dev, (size_t)nq * 4, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail(ANR_EHIP, "bm25 sparse scoring failed: %s", hipGetErrorString(e));
  return ANR_OK;
}

int anr_bm25_combine_fields(int32_t device, int32_t n_fields, const double *const *field_scores_dev, const double *weights,
                            int64_t nq, int64_t n_docs, int32_t normalize, double *out_dev, double *max_dev) {
  if (n_fields < 1 || n_fields > 8 || !field_scores_dev || !weights || nq < 0 || n_docs < 0 || !out_dev)
    return fail(ANR_EINVAL, "bad argument (1..8 fields)");
  if (nq == 0 || n_docs == 0) return ANR_OK;
  DeviceGuard g(device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", device);
  CombineParams p{};
  for (int f = 0; f < n_fields; ++f) {
    if (!field_scores_dev[f]) return fail(ANR_EINVAL, "field %d: null score array", f);
    p.f[f] = field_scores_dev[f];
    p.w[f] = weights[f];
  }
  p.n_fields = n_fields;
  p.n_docs = n_docs;
  p.normalize = normalize;
  p.out = out_dev;
  p.max_out = max_dev;
  hipLaunchKernelGGL(k_bm25_combine, dim3((unsigned)nq), dim3(1024), 0, 0, p);
  ANR_HIP(hipGetLastError());
  ANR_HIP(hipStreamSynchronize(nullptr));
  return ANR_OK;
}

int anr_bm25_nonzero(anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms, int32_t normalize,
                     int32_t cap, int32_t *out_docs, double *out_scores, int32_t *out_count) {
  ANR_TRY(validate_queries(h, nq, q_indptr, q_terms));
  if (cap <= 0 || !out_docs || !out_scores || !out_count) return fail(ANR_EINVAL, "bad output arguments");
  if (nq == 0) return ANR_OK;
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  if (h->n_docs == 0) {
    for (int64_t i = 0; i < nq; ++i) out_count[i] = 0;
    return ANR_OK;
  }
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(nq, (int64_t)(1ll << 31) / (h->n_docs * 8)));
  for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
    const int64_t m = std::min(chunk, nq - q0);
    double *d = nullptr, *dsc = nullptr;
    int32_t *ddoc = nullptr;
    int *dcnt = nullptr;
    int rc = score_chunk(h, m, q_indptr + q0, q_terms, normalize, &d);
    if (rc == ANR_OK) rc = b_alloc(&ddoc, m * cap);
    if (rc == ANR_OK) rc = b_alloc(&dsc, m * cap);
    if (rc == ANR_OK) rc = b_alloc(&dcnt, m);
    if (rc == ANR_OK) {
      NzParams np{d, h->n_docs, cap, ddoc, dsc, dcnt};
      hipLaunchKernelGGL(k_bm25_nonzero, dim3((unsigned)m), dim3(1024), 0, h->stream, np);
      hipError_t e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (e == hipSuccess) e = hipMemcpy(out_docs + q0 * cap, ddoc, (size_t)m * cap * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(out_scores + q0 * cap, dsc, (size_t)m * cap * 8, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(out_count + q0, dcnt, (size_t)m * 4, hipMemcpyDeviceToHost);
      if (e != hipSuccess) rc = fail(ANR_EHIP, "bm25 nonzero failed: %s", hipGetErrorString(e));
    }
    b_free(ddoc);  // (d points into the handle's score buffer)
    b_free(dsc);
    b_free(dcnt);
    if (rc != ANR_OK) return rc;
  }
  return ANR_OK;
}

}  // extern "C"
