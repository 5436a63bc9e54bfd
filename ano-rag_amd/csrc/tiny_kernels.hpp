// Single-launch exact search for tiny corpora (BASELINE.json config C1: 10 k notes x 384-d, batch 1, top-10 — the
// regime the reference actually runs per query, vector_store/retriever.py:186-216 with top_k 20).
//
// At this size the five-kernel pipeline (prepq, scan, select, rescore, finalize + one DMA) is pure launch
// latency: ~58 us per query against numpy's 44 us.  Here ONE kernel does the whole search: every workgroup reads
// the raw query straight from pinned host memory (1.5 KB over PCIe, no DMA), normalises it exactly as k_prepq
// does, computes the EXACT scores of its slice of the stored f32 rows (f64 accumulation in k_rescore's order, so
// the values are bit-identical to the streaming pipeline's), keeps its k best, and the last workgroup to finish
// (agent-scope release / ticket / acquire) merges the partial lists, writes scores and ids straight into pinned
// host memory and raises a completion word the host spins on — no event, no stream synchronisation.
#pragma once
#include "index_kernels.hpp"

namespace anr {

constexpr int kTinyThreads = 1024;
constexpr int kTinyMaxWG = 256;       // (128 up to 64 K rows: see tiny_workgroups)
constexpr int kTinyMaxK = 128;
constexpr int kTinyMaxQ = 4;
constexpr int kTinyRowsPerWG = 1024;   // rows a workgroup ranks among themselves
constexpr int kTinyMaxMerge = 2048;    // partial-list entries the last workgroup merges by sorting (k > kTinySortK)
constexpr int kTinyMaxPrune = 4096;    // ... by pruning (k <= kTinySortK): its cost barely grows with the number of lists
constexpr int kTinySortRows = 192;     // slices longer than this are ordered by the bitonic network instead of rank counting
constexpr int kTinySortK = 32;         // merges for k beyond this likewise (the second rank-counting level grows as (16 k)^2)

struct TinyParams {
  const float *x32;      // stored rows [n_rows][dim]
  const float *q_host;   // pinned host memory (device-visible address): raw queries [nq][dim]
  int nq, dim, metric, normalize, k;
  int64_t n_rows;
  int rows_per_wg, n_wg;
  unsigned long long *cand;  // [nq][n_wg][k] keys, 0 = empty
  unsigned *ticket;          // [nq], zero between calls
  unsigned long long *R;     // pinned host [nq][k][2]: (score bits | seq << 32), (local row or 0xffffffff | seq << 32)
  unsigned seq;              // this call's sequence number: a result word is valid once its upper half equals it
  unsigned long long *stamps;  // developer aid (ANORAG_TINY_STAMPS): [16] wall-clock stamps of workgroup 0 / the last one
};

#define TINY_STAMP(slot, cond) do { if (p.stamps && (cond) && threadIdx.x == 0) p.stamps[slot] = wall_clock64(); } while (0)

// buf[0..n_pow2) (n_pow2 a power of two <= 2048, padded with 0 = below every key) sorted DESCENDING in place by a
// bitonic network over keys held in registers (thread t: elements t and t + 1024): strides below 64 exchange through wave
// shuffles, 64..512 through LDS with barriers, 1024 inside the thread.  Replaces rank counting ("how many keys beat
// mine": a dependent-LDS loop of n iterations per thread — 30 us for a workgroup's 1000 rows, 75 us for the last
// workgroup's 1600-entry second merge level at k = 100).  Keys are distinct (the row id is part of them), so the
// result is the same order.  All kTinyThreads threads call.
__device__ __forceinline__ void tiny_sort_desc(unsigned long long *buf, int n_pow2) {
  const int tid = threadIdx.x;
  const bool two = n_pow2 > kTinyThreads;
  const bool act = tid < n_pow2;
  unsigned long long k0 = act ? buf[tid] : 0ull, k1 = two ? buf[tid + kTinyThreads] : 0ull;
  for (int k2 = 2; k2 <= n_pow2; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      if (j >= kTinyThreads) {  // (only j == 1024 at n_pow2 == 2048: the thread's own two keys, whole range one run)
        if (k0 < k1) {
          const unsigned long long t = k0;
          k0 = k1;
          k1 = t;
        }
      } else if (j >= 64) {
        if (act) buf[tid] = k0;
        if (two) buf[tid + kTinyThreads] = k1;
        __syncthreads();
        {
          const int i = tid, pi = i ^ j;
          const unsigned long long pk = act ? buf[pi] : 0ull;
          const bool first_wanted = ((i & k2) == 0) == ((i & j) == 0);  // this slot takes the one that sorts first (larger)
          const bool mine_first = k0 > pk;
          if (act && first_wanted != mine_first) k0 = pk;
        }
        if (two) {
          const int i = tid + kTinyThreads, pi = i ^ j;
          const unsigned long long pk = buf[pi];
          const bool first_wanted = ((i & k2) == 0) == ((i & j) == 0);
          const bool mine_first = k1 > pk;
          if (first_wanted != mine_first) k1 = pk;
        }
        __syncthreads();
      } else {
        {
          const int i = tid;
          const unsigned long long pk = __shfl_xor(k0, j);
          const bool first_wanted = ((i & k2) == 0) == ((i & j) == 0);
          const bool mine_first = k0 > pk;
          if (act && first_wanted != mine_first) k0 = pk;
        }
        if (two) {
          const int i = tid + kTinyThreads;
          const unsigned long long pk = __shfl_xor(k1, j);
          const bool first_wanted = ((i & k2) == 0) == ((i & j) == 0);
          const bool mine_first = k1 > pk;
          if (first_wanted != mine_first) k1 = pk;
        }
      }
    }
  }
  if (act) buf[tid] = k0;
  if (two) buf[tid + kTinyThreads] = k1;
  __syncthreads();
}

// KC: 16-byte chunks per lane of one row (dim / 4 <= 64 KC), 0 = generic path (dim not a multiple of 4, or > 1024)
template <int KC>
__global__ __launch_bounds__(kTinyThreads) void k_tiny_search(TinyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tiny_smem[];
  unsigned long long *s_keys = reinterpret_cast<unsigned long long *>(tiny_smem);             // [kTinyMaxPrune]
  float *s_q = reinterpret_cast<float *>(tiny_smem + (size_t)kTinyMaxPrune * 8 + (size_t)16 * kTinyMaxK * 8);  // [dim rounded to 4]
  __shared__ double s_red[4];
  __shared__ float s_scale;
  __shared__ int s_last;
  __shared__ unsigned long long s_rank[kTinyMaxK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.y, wg = blockIdx.x;
  const float *qin = p.q_host + (int64_t)q * p.dim;
  const bool w0 = wg == 0 && q == 0;
  TINY_STAMP(0, w0);
  const int64_t row0 = (int64_t)wg * p.rows_per_wg;
  const int rows = (int)(row0 + p.rows_per_wg <= p.n_rows ? p.rows_per_wg : (p.n_rows > row0 ? p.n_rows - row0 : 0));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  // rows in flight per wave: a 1024-thread workgroup leaves 128 VGPRs per lane (RB x KC x 4 of them hold row chunks)
  constexpr int NW = kTinyThreads / 64, RB = KC <= 1 ? 8 : (KC == 2 ? 6 : (KC == 3 ? 4 : 3)), KCA = KC > 0 ? KC : 1;
  const int n4 = p.dim >> 2;
  f32x4 a[RB][KCA];
  auto load_rows = [&](int r0) {  // all RB x KC row chunks of this wave's next rows in flight at once
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int rr = r0 + r < rows ? r0 + r : rows - 1;  // clamped: the duplicate is discarded below
      const f32x4 *x4 = reinterpret_cast<const f32x4 *>(p.x32 + (row0 + rr) * p.dim);
#pragma unroll
      for (int c = 0; c < KCA; ++c) {
        const int k = lane + 64 * c;
        a[r][c] = x4[k < n4 ? k : n4 - 1];  // out-of-range chunks re-read the last one and are skipped below
      }
    }
  };
  // the first rows are requested BEFORE the query is fetched from host memory and normalised: the two latencies overlap
  if (KC > 0 && wave * RB < rows) load_rows(wave * RB);
  // --- the query: raw copy from host memory, norm in k_prepq's order (256 threads, f64), divide -------------
  for (int k = tid; k < p.dim; k += kTinyThreads) s_q[k] = qin[k];
  __syncthreads();
  TINY_STAMP(1, w0);
  if (tid < 256) {
    double acc = 0.0;
    for (int k = tid; k < p.dim; k += 256) {
      const double v = (double)s_q[k];
      acc += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) s_red[wave] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    const float nrm = (float)sqrt(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    s_scale = (p.normalize && nrm != 0.0f) ? nrm : 1.0f;
  }
  __syncthreads();
  const float scale = s_scale;
  for (int k = tid; k < p.dim; k += kTinyThreads) s_q[k] = s_q[k] / scale;
  __syncthreads();
  TINY_STAMP(2, w0);
  // --- exact scores of this workgroup's rows: one wave per row, RB rows in flight per wave, every row load issued
  //     before the first use (the loop over a row's chunks is unrolled by the template: a data-dependent trip count
  //     would serialise the loads behind each other's latency) -----------------------------------------------------
  unsigned long long *s_rowkey = s_keys;  // [rows] while ranking (rows <= kTinyRowsPerWG <= kTinyMaxMerge)
  for (int r0 = wave * RB; r0 < rows; r0 += NW * RB) {
    double acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.0;
    if (KC > 0) {
      const f32x4 *q4 = reinterpret_cast<const f32x4 *>(s_q);
      if (r0 != wave * RB) load_rows(r0);
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const int k = lane + 64 * c;
        if (k < n4) {
          const f32x4 b = q4[k];
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            if (p.metric == 0) {
              acc[r] += (double)a[r][c].x * (double)b.x;
              acc[r] += (double)a[r][c].y * (double)b.y;
              acc[r] += (double)a[r][c].z * (double)b.z;
              acc[r] += (double)a[r][c].w * (double)b.w;
            } else {
              const double d0 = (double)b.x - (double)a[r][c].x, d1 = (double)b.y - (double)a[r][c].y;
              const double d2 = (double)b.z - (double)a[r][c].z, d3 = (double)b.w - (double)a[r][c].w;
              acc[r] += d0 * d0;
              acc[r] += d1 * d1;
              acc[r] += d2 * d2;
              acc[r] += d3 * d3;
            }
          }
        }
      }
    } else {
      for (int k = lane; k < p.dim; k += 64) {
        const double b = (double)s_q[k];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int rr = r0 + r < rows ? r0 + r : rows - 1;
          const double a = (double)p.x32[(row0 + rr) * p.dim + k];
          if (p.metric == 0) acc[r] += a * b;
          else acc[r] += (b - a) * (b - a);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      for (int off = 32; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off);
      if (lane == 0 && r0 + r < rows) {
        const float e = (float)acc[r];
        s_rowkey[r0 + r] = make_key(p.metric == 0 ? e : -e, (unsigned)(row0 + r0 + r));
      }
    }
  }
  __syncthreads();
  TINY_STAMP(3, w0);
  // --- the workgroup's k best (rank counting among its rows), published as its partial list --------------------
  unsigned long long *mine = p.cand + ((int64_t)q * p.n_wg + wg) * p.k;
  if (rows > kTinySortRows) {  // a long slice: order it (see tiny_sort_desc), the first k are the list
    int rp = 2;
    while (rp < rows) rp <<= 1;  // rows <= kTinyRowsPerWG = 1024
    for (int i = rows + tid; i < rp; i += kTinyThreads) s_rowkey[i] = 0ull;
    __syncthreads();
    tiny_sort_desc(s_rowkey, rp);
    for (int i = tid; i < p.k; i += kTinyThreads)
      __hip_atomic_store(mine + i, i < rows ? s_rowkey[i] : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {  // a short slice: every row counts the rows that beat it (a loop of `rows` LDS reads)
    unsigned long long mykey = 0ull;
    int myrank = p.k;
    if (tid < rows) {
      mykey = s_rowkey[tid];
      int rank = 0;
      for (int j = 0; j < rows; ++j) rank += (s_rowkey[j] > mykey) ? 1 : 0;
      myrank = rank;
    }
    if (myrank < p.k) __hip_atomic_store(mine + myrank, mykey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = rows + tid; i < p.k; i += kTinyThreads)
      __hip_atomic_store(mine + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the ticket
  __syncthreads();
  TINY_STAMP(4, w0);
  if (tid == 0) {
    // the partial lists were stored write-through (agent-scope atomic stores, drained above) and are read back with
    // agent-scope atomic loads only, so neither a release nor an acquire fence is needed around the ticket
    // (cdna_hip_programming.md, Guideline 16, R1)
    const unsigned t = __hip_atomic_fetch_add(p.ticket + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == (unsigned)(p.n_wg - 1)) ? 1 : 0;
  }
  __syncthreads();
  TINY_STAMP(5, w0);
  if (!s_last) return;
  TINY_STAMP(8, q == 0);
  // --- the last workgroup merges n_wg * k <= kTinyMaxMerge entries, ranks them, writes host memory ----------------
  const int M = p.n_wg * p.k;
  const unsigned long long *all = p.cand + (int64_t)q * p.n_wg * p.k;
  for (int i = tid; i < M; i += kTinyThreads)
    s_keys[i] = __hip_atomic_load(all + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  TINY_STAMP(9, q == 0);
  const int64_t found = p.n_rows < p.k ? p.n_rows : p.k;
  if (p.k > kTinySortK) {
    // the n_wg * k <= 2048 keys of all partial lists, ordered by the register bitonic network (binary searches across the
    // sorted lists were tried first: ~1000 dependent LDS round trips per entry, 100 us; then two levels of rank counting:
    // 11 us at k = 10 but 75 us at k = 100)
    {
      int mp = 2;
      while (mp < M) mp <<= 1;
      for (int i = M + tid; i < mp; i += kTinyThreads) s_keys[i] = 0ull;
      __syncthreads();
      tiny_sort_desc(s_keys, mp);
      for (int i = tid; i < p.k; i += kTinyThreads) s_rank[i] = i < M ? s_keys[i] : 0ull;
      __syncthreads();
    }
  } else {
    // Prune, then rank: every partial list is sorted, so a key below the k-th largest list HEAD has k larger keys
    // (those heads) and cannot be among the k best.  (1) each list's head counts the heads above it — n_wg independent
    // LDS reads — and the k-th largest becomes the threshold; (2) the keys at or above it (at most k lists qualify: <= k^2
    // keys, ~2 k in practice) are compacted; (3) the survivors rank themselves by counting.
    // (Round 2 ranked all n_wg * k keys by counting in two levels: 11.7 us of the kernel's 22 at k = 10, its largest
    // phase; a register tournament of wave-wide maxima — 2 k dependent shuffle butterflies — took just as long;
    // binary searches across the sorted lists: 100 us.)
    unsigned long long *s_l2 = s_keys + kTinyMaxPrune;  // survivors: <= kTinySortK^2 = 1024 <= 16 * kTinyMaxK entries
    __shared__ unsigned long long s_thr;
    __shared__ unsigned s_nsurv;
    if (tid == 0) {
      s_thr = 0ull;
      s_nsurv = 0u;
    }
    __syncthreads();
    if (p.n_wg > p.k && tid < p.n_wg) {
      const unsigned long long h = s_keys[tid * p.k];
      int r = 0;
      for (int j = 0; j < p.n_wg; ++j) r += (s_keys[j * p.k] > h) ? 1 : 0;
      if (r == p.k - 1 && h != 0ull) s_thr = h;  // (keys are unique: one head has exactly k - 1 heads above it)
    }
    __syncthreads();
    const unsigned long long thr = s_thr;
    for (int i = tid; i < M; i += kTinyThreads) {
      const unsigned long long key = s_keys[i];
      if (key != 0ull && key >= thr) s_l2[atomicAdd(&s_nsurv, 1u)] = key;
    }
    __syncthreads();
    TINY_STAMP(12, q == 0);
    const int ns = (int)s_nsurv;
    for (int i = tid; i < ns; i += kTinyThreads) {
      const unsigned long long key = s_l2[i];
      int rank = 0;
      for (int j = 0; j < ns; ++j) rank += (s_l2[j] > key) ? 1 : 0;
      if (rank < p.k) s_rank[rank] = key;  // ranks are distinct: one writer per slot
    }
    __syncthreads();
    TINY_STAMP(13, q == 0);
  }
  // One wave writes the k results into pinned host memory as SELF-VALIDATING 8-byte words — the call's sequence number
  // rides in the upper half of each — with plain stores and NO fence and NO completion flag: the host spins until every
  // word of the query carries the sequence number.  (Round 2 wrote D / I plainly, fenced them once at system scope and
  // then raised a flag: the fence alone was ~5 of the kernel's 22 us; a fence per writing thread or system-scope atomic
  // stores cost 5-10 us apiece.)  An aligned 8-byte store reaches host memory as one write, so a word is either the
  // previous call's or complete; the id offset of a shard is added on the host.
  if (wave == 0) {
    unsigned long long *R = p.R + (int64_t)q * p.k * 2;
    const unsigned long long tag = (unsigned long long)p.seq << 32;
    for (int i = lane; i < p.k; i += 64) {
      unsigned sbits, row;
      if (i < (int)found) {
        const unsigned long long key = s_rank[i];
        const float v = ord2f((unsigned)(key >> 32));
        sbits = __float_as_uint(p.metric == 0 ? v : -v);
        row = 0xffffffffu - (unsigned)(key & 0xffffffffu);
      } else {  // fewer rows than k: faiss padding
        sbits = __float_as_uint(p.metric == 0 ? -3.402823466e+38f : 3.402823466e+38f);
        row = 0xffffffffu;
      }
      reinterpret_cast<volatile unsigned long long *>(R)[2 * i] = tag | sbits;
      reinterpret_cast<volatile unsigned long long *>(R)[2 * i + 1] = tag | row;
    }
  }
  TINY_STAMP(10, q == 0);
  if (tid == 0) __hip_atomic_store(p.ticket + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
  TINY_STAMP(11, q == 0);
}

}  // namespace anr
