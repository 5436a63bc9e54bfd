#pragma once
#include <mutex>
// Device score fusion behind anr_fuse_lists (include/anorag.h): the arithmetic of the reference's
// HybridSearcher.fuse (retrieval/hybrid_search.py:34-103) for a batch of queries, one workgroup per query,
// everything LDS-resident.  All arithmetic is float64 in the reference's own order of operations, so the
// final similarities are bit-identical to the Python implementation:
//   linear : final = w_d*(s_d/max_d) + w_b*(s_b/max_b) + w_g*(s_g/max_g) + w_p*s_p   (max == 0 -> 0.0)
//   rrf    : final = ((w_d/(k+r_d) + w_b/(k+r_b)) + w_g/(k+r_g)) + w_p*s_p, ranks from a stable descending
//            sort of each source (ties keep list order); ids that occur only in `path` are dropped.
// Result order: final descending; ties by the reference's ranks-dict insertion order (rrf) or by id (linear,
// where the reference iterates a set).
#include "common.hpp"

namespace anr {

constexpr int kFuseMax = 4096;  // total list entries per query (LDS-resident: 28 B each)

struct FuseParams {
  int method;            // 0 linear, 1 rrf
  const int64_t *ids;    // concatenated entries of all queries
  const double *scores;
  const int64_t *offs;   // [nq][5] entry offsets of the 4 sources of each query (+ end)
  double w[4];
  double rrf_k;
  int pool;
  int64_t *out_ids;      // [nq][pool]
  double *out_final;     // [nq][pool]
  double *out_src;       // [nq][pool][4] raw source scores, NaN when the id is not in that source
  int *out_count;        // [nq]
  // overrides (anr_fuse_dense: the lists are the candidates of an N-array source, not the whole source)
  const double *smax_ovr;  // optional [nq][4]: per-source maximum of the WHOLE source (linear); NaN = take the list's
  const int *rank_ovr;     // optional, per entry: 1-based rank of the entry in its WHOLE source (rrf); 0 = count in the list
};

struct FuseShared {
  unsigned long long key[kFuseMax];  // id << 2 | source
  double val[kFuseMax];              // contribution of the entry; for a segment head: the fused final
  unsigned tie[kFuseMax];            // rrf: source << 28 | rank of the first source holding the id
  unsigned idx[kFuseMax];
  unsigned keep[kFuseMax];
  unsigned long long xk[kFuseMax];  // bitonic_pairs: exchange buffer of the passes that cross waves
  double red[16];
  double smax[4];
  unsigned cnt;
};

template <typename F>
__device__ void bitonic_idx(unsigned *idx, int n_pow2, F less) {
  // sorts idx[0..n_pow2) ascending under `less`
  const int tid = threadIdx.x;
  for (int k2 = 2; k2 <= n_pow2; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n_pow2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned a = idx[i], b = idx[ixj];
          const bool up = ((i & k2) == 0);
          if (up ? less(b, a) : less(a, b)) {
            idx[i] = b;
            idx[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
}

// Orders n_pow2 (key, value) pairs ascending by key, `tie(va, vb)` deciding between equal keys (a strict order on the
// values), and leaves the values in out[0..n_pow2).  The pairs live in REGISTERS, thread t holding elements t, t + 1024,
// ...: a compare-exchange of stride j < 64 reaches its partner with shuffles inside the wave (no LDS, no barrier), a
// stride >= 1024 pairs two registers of the same thread, and only the strides 64..512 go through LDS with barriers —
// 14 of the 66 passes at 2048 elements.  (The index-array form above, one LDS gather and one barrier per pass, was 100
// of the 113 us ONE ~1100-entry fuse call spent in k_fuse.)
template <typename KeyFn, typename TieFn>
__device__ void bitonic_pairs(unsigned long long *xk, unsigned *xv, int n_pow2, KeyFn key_of, TieFn tie, unsigned *out) {
  constexpr int EMAX = kFuseMax / 1024;
  const int tid = threadIdx.x;
  const int E = n_pow2 >= 1024 ? n_pow2 / 1024 : 1;
  const bool act = tid < n_pow2;  // (n_pow2 < 1024: the upper threads only join the barriers)
  unsigned long long k[EMAX];
  unsigned v[EMAX];
#pragma unroll
  for (int e = 0; e < EMAX; ++e)
    if (e < E) {
      const int i = tid + 1024 * e;
      v[e] = act ? xv[i] : 0xffffffffu;
      k[e] = act ? key_of(v[e]) : ~0ull;
    }
  auto before = [&](unsigned long long ka, unsigned va, unsigned long long kb, unsigned vb) {
    return ka < kb || (ka == kb && tie(va, vb));
  };
  for (int k2 = 2; k2 <= n_pow2; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      if (j >= 1024) {  // partner = another register of this thread
        const int je = j >> 10;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
          if (e < E && (e & je) == 0) {
            const int f = e | je;  // (f < E: both are bits of the element number)
            const int i = tid + 1024 * e;
            const bool up = (i & k2) == 0;
            // static indexing only: f ranges over the few set-bit patterns of EMAX
#pragma unroll
            for (int ff = 0; ff < EMAX; ++ff)
              if (ff == f) {
                const bool swap = up ? before(k[ff], v[ff], k[e], v[e]) : before(k[e], v[e], k[ff], v[ff]);
                if (swap) {
                  const unsigned long long tk = k[e];
                  const unsigned tv = v[e];
                  k[e] = k[ff];
                  v[e] = v[ff];
                  k[ff] = tk;
                  v[ff] = tv;
                }
              }
          }
      } else if (j >= 64) {  // partner in another wave: through LDS
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
          if (e < E && act) {
            xk[tid + 1024 * e] = k[e];
            xv[tid + 1024 * e] = v[e];
          }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
          if (e < E && act) {
            const int i = tid + 1024 * e, pi = i ^ j;
            const unsigned long long pk = xk[pi];
            const unsigned pv = xv[pi];
            const bool up = (i & k2) == 0, lower = (i & j) == 0;
            const bool want_min = up == lower;
            const bool mine_first = before(k[e], v[e], pk, pv);
            if (want_min != mine_first) {
              k[e] = pk;
              v[e] = pv;
            }
          }
        __syncthreads();
      } else {  // partner lane in this wave
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
          if (e < E) {
            const int i = tid + 1024 * e;
            const unsigned long long pk = __shfl_xor(k[e], j);
            const unsigned pv = __shfl_xor(v[e], j);
            const bool up = (i & k2) == 0, lower = (i & j) == 0;
            const bool want_min = up == lower;
            const bool mine_first = before(k[e], v[e], pk, pv);
            if (act && want_min != mine_first) {
              k[e] = pk;
              v[e] = pv;
            }
          }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EMAX; ++e)
    if (e < E && act) out[tid + 1024 * e] = v[e];
  __syncthreads();
}

template <bool OVR>  // OVR: FuseParams::smax_ovr / rank_ovr are honoured
__global__ __launch_bounds__(1024) void k_fuse(FuseParams p) {
  extern __shared__ unsigned char fuse_smem[];
  FuseShared &sh = *reinterpret_cast<FuseShared *>(fuse_smem);
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t *off = p.offs + (int64_t)q * 5;
  const int64_t base = off[0];
  const int T = (int)(off[4] - off[0]);
  const unsigned INVALID = 0xffffffffu;

  // 1) per-source maxima (linear normalisation, hybrid_search.py:26-32)
  for (int s = 0; s < 3; ++s) {
    double m = -__builtin_inf();
    for (int64_t e = off[s] + tid; e < off[s + 1]; e += 1024) m = fmax(m, p.scores[e]);
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if (lane == 0) sh.red[wave] = m;
    __syncthreads();
    if (tid == 0) {
      double mm = sh.red[0];
      for (int w = 1; w < 16; ++w) mm = fmax(mm, sh.red[w]);
      if (OVR && p.smax_ovr) {
        const double o = p.smax_ovr[(int64_t)q * 4 + s];
        if (o == o) mm = o;
      }
      sh.smax[s] = mm;
    }
    __syncthreads();
  }
  // 2) contribution of every entry.  rrf: the rank of an entry = its position in a stable descending sort of its
  //    source's list (:66-67) — each list is ordered with bitonic_pairs (key = the score's inverted image, ties by list
  //    position), 1-based positions land in sh.keep.  (Counting, per entry, the entries that come before it was an
  //    O(n^2) loop on one CU: 90 us for a 1000-entry list.)
  constexpr int PER = (kFuseMax + 1023) / 1024;
  if (p.method == 1) {
    for (int e = tid; e < T; e += 1024) sh.val[e] = p.scores[base + e];
    __syncthreads();
    const double *raw = sh.val;
    for (int s = 0; s < 3; ++s) {
      const int o0 = (int)(off[s] - base), ns = (int)(off[s + 1] - off[s]);
      if (ns == 0) continue;
      int np2 = 1;
      while (np2 < ns) np2 <<= 1;
      for (int i = tid; i < np2; i += 1024) sh.idx[i] = i < ns ? (unsigned)(o0 + i) : INVALID;
      __syncthreads();
      bitonic_pairs(
          sh.xk, sh.idx, np2,
          [raw](unsigned a) {
            if (a == 0xffffffffu) return ~0ull;
            double v = raw[a];
            if (!(v == v)) return ~0ull - 1ull;  // NaN scores after every number (their rank is set below)
            if (v == 0.0) v = 0.0;
            const unsigned long long u = (unsigned long long)__double_as_longlong(v);
            return ~((u >> 63) ? ~u : (u | 0x8000000000000000ull));
          },
          [](unsigned a, unsigned b) { return a < b; }, sh.idx);
      for (int r = tid; r < ns; r += 1024) sh.keep[sh.idx[r]] = (unsigned)(r + 1);
      __syncthreads();
    }
  }
  double c_mine[PER];
  unsigned tie_mine[PER];
  {
    int k = 0;
    for (int e = tid; e < T; e += 1024, ++k) {
      const int64_t g = base + e;
      int s = 0;
      while (s < 3 && g >= off[s + 1]) ++s;
      const double sc = p.scores[g];
      sh.key[e] = ((unsigned long long)p.ids[g] << 2) | (unsigned long long)s;
      double c;
      unsigned tie = INVALID;
      if (s == 3) {
        c = p.w[3] * sc;
      } else if (p.method == 0) {
        const double m = sh.smax[s];
        c = p.w[s] * (m == 0.0 ? 0.0 : sc / m);
      } else {
        // rank = 1 + entries of this source ordered before e in a stable descending sort (:66-67); a NaN score is
        // beaten by nothing under the reference's comparisons
        int rank = (OVR && p.rank_ovr) ? p.rank_ovr[g] : 0;
        if (rank == 0) rank = sc == sc ? (int)sh.keep[e] : 1;
        c = p.w[s] / (p.rrf_k + (double)rank);
        tie = ((unsigned)s << 28) | (unsigned)rank;
      }
      c_mine[k] = c;
      tie_mine[k] = tie;
    }
  }
  if (p.method == 1) __syncthreads();  // every entry has read its raw score and rank
  {
    int k = 0;
    for (int e = tid; e < T; e += 1024, ++k) {
      sh.val[e] = c_mine[k];
      sh.tie[e] = tie_mine[k];
      sh.keep[e] = 0;
    }
  }
  int Tp = 1;
  while (Tp < T) Tp <<= 1;
  for (int e = tid; e < Tp; e += 1024) sh.idx[e] = e < T ? (unsigned)e : INVALID;
  if (tid == 0) sh.cnt = 0;
  __syncthreads();

  // 3) group by id: sort entry indices by (id, source)
  {
    const unsigned long long *key = sh.key;
    // (an INVALID padding value sorts last: the maximal key; entry indices break a repeated id inside one source)
    bitonic_pairs(sh.xk, sh.idx, Tp, [key](unsigned a) { return a == 0xffffffffu ? ~0ull : key[a]; },
                  [](unsigned a, unsigned b) { return a < b; }, sh.idx);
  }
  // 4) each segment head sums its sources in the reference's order dense -> bm25 -> graph, then path
  for (int i = tid; i < T; i += 1024) {
    const unsigned e = sh.idx[i];
    const unsigned long long id = sh.key[e] >> 2;
    if (i > 0 && (sh.key[sh.idx[i - 1]] >> 2) == id) continue;
    double f = 0.0, path_c = 0.0;
    unsigned tie = INVALID;
    bool nonpath = false, has_path = false;
    for (int j = i; j < T; ++j) {
      const unsigned ej = sh.idx[j];
      if ((sh.key[ej] >> 2) != id) break;
      if ((sh.key[ej] & 3ull) == 3ull) {
        path_c = sh.val[ej];
        has_path = true;
      } else {
        f += sh.val[ej];
        nonpath = true;
        tie = sh.tie[ej] < tie ? sh.tie[ej] : tie;
      }
    }
    f = f + (has_path ? path_c : p.w[3] * 0.0);
    if (p.method == 0 || nonpath) {  // rrf drops ids that occur only in `path` (:68-69)
      sh.val[e] = f;
      sh.tie[e] = tie;
      sh.keep[e] = 1;
      atomicAdd(&sh.cnt, 1u);
    }
  }
  __syncthreads();
  const int U = (int)sh.cnt;
  __syncthreads();
  // compact the surviving heads into idx (order irrelevant, sorted next)
  if (tid == 0) sh.cnt = 0;
  __syncthreads();
  unsigned mine[(kFuseMax + 1023) / 1024];
  int nm = 0;
  for (int e = tid; e < T; e += 1024)
    if (sh.keep[e]) mine[nm++] = (unsigned)e;
  __syncthreads();
  int Up = 1;
  while (Up < U) Up <<= 1;
  for (int k = 0; k < nm; ++k) sh.idx[atomicAdd(&sh.cnt, 1u)] = mine[k];
  __syncthreads();
  for (int e = U + tid; e < Up; e += 1024) sh.idx[e] = INVALID;
  __syncthreads();
  // 5) order by final descending; ties: rrf -> the reference's dict insertion order, linear -> id
  {
    const double *val = sh.val;
    const unsigned *tie = sh.tie;
    const unsigned long long *key = sh.key;
    const int method = p.method;
    // key: the final's order-preserving 64-bit image, inverted (ascending key = descending final; -0 == +0; NaN after
    // every number; padding last); equal finals: the reference's tie rules
    bitonic_pairs(
        sh.xk, sh.idx, Up,
        [val](unsigned a) {
          if (a == 0xffffffffu) return ~0ull;
          double v = val[a];
          if (!(v == v)) return ~0ull - 1ull;
          if (v == 0.0) v = 0.0;
          const unsigned long long u = (unsigned long long)__double_as_longlong(v);
          return ~((u >> 63) ? ~u : (u | 0x8000000000000000ull));
        },
        [tie, key, method](unsigned a, unsigned b) {
          if (a == 0xffffffffu || b == 0xffffffffu) return a < b;
          if (method == 1 && tie[a] != tie[b]) return tie[a] < tie[b];
          return key[a] < key[b] || (key[a] == key[b] && a < b);
        },
        sh.idx);
  }
  // 6) emit
  const int n_out = U < p.pool ? U : p.pool;
  if (tid == 0) p.out_count[q] = n_out;
  const double nan = __builtin_nan("");
  for (int i = tid; i < p.pool; i += 1024) {
    double *osr = p.out_src + ((int64_t)q * p.pool + i) * 4;
    osr[0] = osr[1] = osr[2] = osr[3] = nan;
    if (i < n_out) {
      const unsigned e = sh.idx[i];
      p.out_ids[(int64_t)q * p.pool + i] = (int64_t)(sh.key[e] >> 2);
      p.out_final[(int64_t)q * p.pool + i] = sh.val[e];
    } else {
      p.out_ids[(int64_t)q * p.pool + i] = -1;
      p.out_final[(int64_t)q * p.pool + i] = 0.0;
    }
  }
  __syncthreads();
  // raw source scores of the emitted ids: walk each emitted segment again
  for (int e = tid; e < T; e += 1024) {
    const unsigned long long id = sh.key[e] >> 2;
    const int s = (int)(sh.key[e] & 3ull);
    for (int i = 0; i < n_out; ++i) {
      const unsigned h = sh.idx[i];
      if ((sh.key[h] >> 2) == id) {
        p.out_src[((int64_t)q * p.pool + i) * 4 + s] = p.scores[base + e];
        break;
      }
    }
  }
}

// Work space of anr_fuse_dense / anr_fuse_lists: ONE device block and one pinned host block per device (each entry
// point owns an array of these), grown on demand and kept for
// the life of the process (25 hipMalloc/hipFree pairs and pageable copies were ~1.4 ms of a 2.6 ms call).  A call holds
// the device's arena lock from start to finish, so concurrent calls on one device run one after the other.
struct FuseArena {
  std::mutex mu;
  char *dev = nullptr;
  size_t dev_cap = 0;
  char *host = nullptr;
  size_t host_cap = 0;
  int reserve(size_t dev_bytes, size_t host_bytes) {
    if (dev_bytes > dev_cap) {
      if (dev) (void)hipFree(dev);
      dev = nullptr;
      dev_cap = 0;
      const size_t want = dev_bytes + dev_bytes / 4;
      if (hipMalloc(reinterpret_cast<void **>(&dev), want) != hipSuccess) return fail(ANR_EHIP, "hipMalloc(%zu) failed", want);
      dev_cap = want;
    }
    if (host_bytes > host_cap) {
      if (host) (void)hipHostFree(host);
      host = nullptr;
      host_cap = 0;
      const size_t want = host_bytes + host_bytes / 4;
      if (hipHostMalloc(reinterpret_cast<void **>(&host), want, hipHostMallocDefault) != hipSuccess)
        return fail(ANR_EHIP, "hipHostMalloc(%zu) failed", want);
      host_cap = want;
    }
    return ANR_OK;
  }
};
constexpr int kFuseMaxDevices = 64;
struct Carve {  // bump allocation inside a block (256-byte aligned pieces)
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
  }
};

}  // namespace anr
