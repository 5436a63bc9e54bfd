// Device-wide STABLE radix sort of (key, 32-bit value) pairs — the ordering primitive of the two corner paths that sort a
// whole corpus-sized array: top-k beyond the select window (largek.hip) and rrf with several full-corpus lists
// (fusion_rrf_long.hip).  (Rounds 1-2 called hipcub::DeviceRadixSort there.)
//
// Least-significant-digit passes of 8 bits over keys brought into an ascending unsigned form first (a float becomes its
// order-preserving 32-bit image, a descending sort complements the key), 4 passes for 32-bit keys, 8 for 64-bit:
//   k_rs_hist     every workgroup counts the digits of its 4096-element tile (LDS histogram) -> hist[digit][tile]
//   k_rs_scan     exclusive prefix sum of hist in digit-major order (one workgroup, a carry across 1024-entry chunks)
//   k_rs_scatter  every workgroup walks its tile in index order, 256 elements at a time; inside a wave an element's rank
//                 among the equal digits before it comes from eight ballots, the four waves take their turns in order
//                 -> the scatter is stable: equal keys keep their input order (what the tie rules of both callers need)
// Off the hot path: clarity over speed (a pass moves 12-24 bytes per element and synchronises 64 times per tile).
#include <utility>

#include "common.hpp"

namespace anr {

constexpr int kRsThreads = 256, kRsRounds = 16, kRsTile = kRsThreads * kRsRounds;

__device__ __forceinline__ unsigned rs_f2ord(float v) {
  unsigned u = __float_as_uint(v);
  if ((u << 1) == 0u) u = 0u;  // -0.0 and +0.0 compare equal: one key, so that their ids decide
  return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float rs_ord2f(unsigned o) { return __uint_as_float((o >> 31) ? (o & 0x7fffffffu) : ~o); }

// keys into the ascending unsigned form (and the values beside them)
template <typename K, typename KIN>
__global__ void k_rs_pre(const KIN *k_in, const unsigned *v_in, int64_t n, int descending, K *k, unsigned *v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  K x;
  if constexpr (sizeof(KIN) == 4 && sizeof(K) == 4) x = (K)rs_f2ord(*reinterpret_cast<const float *>(k_in + i));
  else x = (K)k_in[i];
  k[i] = descending ? (K)~x : x;
  v[i] = v_in[i];
}
template <typename K, typename KOUT>
__global__ void k_rs_post(const K *k, const unsigned *v, int64_t n, int descending, KOUT *k_out, unsigned *v_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  K x = k[i];
  if (descending) x = (K)~x;
  if constexpr (sizeof(KOUT) == 4 && sizeof(K) == 4) {
    const float f = rs_ord2f((unsigned)x);
    *reinterpret_cast<float *>(k_out + i) = f;
  } else {
    k_out[i] = (KOUT)x;
  }
  v_out[i] = v[i];
}

template <typename K>
__global__ __launch_bounds__(kRsThreads) void k_rs_hist(const K *k, int64_t n, int shift, unsigned *hist, int64_t nblk) {
  __shared__ unsigned h[256];
  const int tid = threadIdx.x;
  h[tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRsTile;
  for (int r = 0; r < kRsRounds; ++r) {
    const int64_t i = base + (int64_t)r * kRsThreads + tid;
    if (i < n) atomicAdd(&h[(unsigned)(k[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(int64_t)tid * nblk + blockIdx.x] = h[tid];
}

__global__ __launch_bounds__(1024) void k_rs_scan(unsigned *hist, int64_t total) {
  __shared__ unsigned wsum[16];
  __shared__ unsigned carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < total; c0 += 1024) {
    const int64_t i = c0 + tid;
    const unsigned x = i < total ? hist[i] : 0u;
    unsigned incl = x;
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned off = carry_s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (i < total) hist[i] = off + incl - x;
    __syncthreads();
    if (tid == 1023) carry_s = off + incl;
    __syncthreads();
  }
}

template <typename K>
__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(const K *k_in, const unsigned *v_in, int64_t n, int shift,
                                                           const unsigned *offs, int64_t nblk, K *k_out, unsigned *v_out) {
  __shared__ unsigned base[256];  // where this tile's next element of each digit goes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  base[tid] = offs[(int64_t)tid * nblk + blockIdx.x];
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * kRsTile;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int r = 0; r < kRsRounds; ++r) {
    const int64_t i = t0 + (int64_t)r * kRsThreads + tid;
    const bool valid = i < n;
    K key = 0;
    unsigned val = 0, d = 0;
    if (valid) {
      key = k_in[i];
      val = v_in[i];
      d = (unsigned)(key >> shift) & 255u;
    }
    // lanes of this wave holding the same digit
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long bal = __ballot(valid && ((d >> b) & 1u));
      same &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const unsigned before = (unsigned)__popcll(same & lt), count = (unsigned)__popcll(same);
    // the four waves in order: all lanes read the digit's position, then the first lane of each digit group advances it
    for (int w = 0; w < kRsThreads / 64; ++w) {
      if (wave == w && valid) {
        const unsigned pos = base[d] + before;
        k_out[pos] = key;
        v_out[pos] = val;
      }
      __syncthreads();
      if (wave == w && valid && before == 0) base[d] += count;
      __syncthreads();
    }
  }
}

size_t radix_sort_temp_bytes(int64_t n) {
  const int64_t nblk = ceil_div(n > 0 ? n : 1, kRsTile);
  // two key buffers (8 bytes each), two value buffers, the digit histogram
  return (size_t)round_up(n * 8, 256) * 2 + (size_t)round_up(n * 4, 256) * 2 + (size_t)round_up(256 * nblk * 4, 256);
}

template <typename K, typename KIN>
static int radix_sort_impl(void *temp, const KIN *k_in, KIN *k_out, const unsigned *v_in, unsigned *v_out, int64_t n,
                           bool descending, hipStream_t st) {
  if (n <= 0) return ANR_OK;
  if (n > 0xffffffffll) return fail(ANR_EINVAL, "radix sort: at most 2^32-1 elements");
  const int64_t nblk = ceil_div(n, kRsTile);
  char *T = reinterpret_cast<char *>(temp);
  K *K1 = reinterpret_cast<K *>(T), *K2 = reinterpret_cast<K *>(T + round_up(n * 8, 256));
  unsigned *V1 = reinterpret_cast<unsigned *>(T + 2 * round_up(n * 8, 256));
  unsigned *V2 = reinterpret_cast<unsigned *>(T + 2 * round_up(n * 8, 256) + round_up(n * 4, 256));
  unsigned *hist = reinterpret_cast<unsigned *>(T + 2 * round_up(n * 8, 256) + 2 * round_up(n * 4, 256));
  const unsigned g256 = (unsigned)ceil_div(n, 256);
  hipLaunchKernelGGL((k_rs_pre<K, KIN>), dim3(g256), dim3(256), 0, st, k_in, v_in, n, descending ? 1 : 0, K1, V1);
  K *ka = K1, *kb = K2;
  unsigned *va = V1, *vb = V2;
  for (int shift = 0; shift < (int)sizeof(K) * 8; shift += 8) {
    hipLaunchKernelGGL((k_rs_hist<K>), dim3((unsigned)nblk), dim3(kRsThreads), 0, st, ka, n, shift, hist, nblk);
    hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, st, hist, 256 * nblk);
    hipLaunchKernelGGL((k_rs_scatter<K>), dim3((unsigned)nblk), dim3(kRsThreads), 0, st, ka, va, n, shift, hist, nblk, kb, vb);
    std::swap(ka, kb);
    std::swap(va, vb);
  }
  hipLaunchKernelGGL((k_rs_post<K, KIN>), dim3(g256), dim3(256), 0, st, ka, va, n, descending ? 1 : 0, k_out, v_out);
  ANR_HIP(hipGetLastError());
  return ANR_OK;
}

int radix_sort_pairs_u64(void *temp, const unsigned long long *k_in, unsigned long long *k_out, const unsigned *v_in,
                         unsigned *v_out, int64_t n, bool descending, hipStream_t st) {
  return radix_sort_impl<unsigned long long, unsigned long long>(temp, k_in, k_out, v_in, v_out, n, descending, st);
}

int radix_sort_pairs_f32(void *temp, const float *k_in, float *k_out, const unsigned *v_in, unsigned *v_out, int64_t n,
                         bool descending, hipStream_t st) {
  return radix_sort_impl<unsigned, unsigned>(temp, reinterpret_cast<const unsigned *>(k_in), reinterpret_cast<unsigned *>(k_out),
                                             v_in, v_out, n, descending, st);
}

}  // namespace anr
