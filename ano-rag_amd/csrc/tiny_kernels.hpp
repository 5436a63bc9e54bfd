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
constexpr int kTinyMaxMerge = 2048;    // partial-list entries the last workgroup merges
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
  float *D;                  // pinned host [nq][k]
  int64_t *I;                // pinned host [nq][k]
  unsigned *flag;            // pinned host [nq]: set to seq when query q is complete
  unsigned seq;
  int64_t id_offset;
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
  unsigned long long *s_keys = reinterpret_cast<unsigned long long *>(tiny_smem);             // [kTinyMaxMerge]
  float *s_q = reinterpret_cast<float *>(tiny_smem + (size_t)kTinyMaxMerge * 8 + (size_t)16 * kTinyMaxK * 8);  // [dim rounded to 4]
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
  float *D = p.D + (int64_t)q * p.k;
  int64_t *I = p.I + (int64_t)q * p.k;
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
    // two-level merge with INDEPENDENT LDS reads (binary searches across the sorted lists were tried: ~1000 dependent LDS
    // round trips per entry, 100 us): level 1 — each of the 16 waves ranks the entries of its share of the lists among
    // themselves and keeps the k best; level 2 — the 16 k survivors are ranked among themselves
    unsigned long long *s_l2 = s_keys + kTinyMaxMerge;  // [16][k]
    const int G = (p.n_wg + 15) / 16;                    // lists per level-1 group
    const int GE = G * p.k;                              // entries per group
    for (int i = tid; i < 16 * p.k; i += kTinyThreads) s_l2[i] = 0ull;
    __syncthreads();
    for (int i = tid; i < M; i += kTinyThreads) {
      const unsigned long long key = s_keys[i];
      if (key == 0ull) continue;
      const int g = i / GE;
      const int e0 = g * GE, e1 = e0 + GE < M ? e0 + GE : M;
      int rank = 0;
      for (int j = e0; j < e1; ++j) rank += (s_keys[j] > key) ? 1 : 0;
      if (rank < p.k) s_l2[g * p.k + rank] = key;
    }
    __syncthreads();
    const int M2 = 16 * p.k;
    for (int i = tid; i < M2; i += kTinyThreads) {
      const unsigned long long key = s_l2[i];
      if (key == 0ull) continue;
      int rank = 0;
      for (int j = 0; j < M2; ++j) rank += (s_l2[j] > key) ? 1 : 0;
      if (rank < p.k) s_rank[rank] = key;  // ranks are distinct: one writer per slot
    }
    __syncthreads();
  }
  // one wave writes the k results into pinned host memory and fences them ONCE at system scope (a fence per writing
  // thread, or system-scope atomic stores, cost 5-10 us apiece here)
  if (wave == 0) {
    for (int i = lane; i < p.k; i += 64) {
      if (i < (int)found) {
        const unsigned long long key = s_rank[i];
        const float v = ord2f((unsigned)(key >> 32));
        D[i] = p.metric == 0 ? v : -v;
        I[i] = (int64_t)(0xffffffffu - (unsigned)(key & 0xffffffffu)) + p.id_offset;
      } else {  // fewer rows than k: faiss padding
        D[i] = p.metric == 0 ? -3.402823466e+38f : 3.402823466e+38f;
        I[i] = -1;
      }
    }
    __threadfence_system();
  }
  TINY_STAMP(10, q == 0);
  __syncthreads();
  TINY_STAMP(11, q == 0);
  if (tid == 0) {
    __hip_atomic_store(p.ticket + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
    __hip_atomic_store(p.flag + q, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace anr
