// anr_similarity_matrix: the all-pairs similarity of two sets of embeddings, EmbeddingManager.compute_similarity of the
// reference (vector_store/embedding_manager.py:586-629) as ONE tiled kernel:
//   cosine     rows / (||row|| + 1e-8) on both sides (float32, :602-606), then the dot product (:609)
//   dot        e1 . e2^T (:620)
//   euclidean  1 / (1 + ||a - b||) (:613-616; scipy cdist works in float64)
// float32 inputs, every product / squared difference accumulated in float64 (exact products, one rounding at the end),
// float64 output that the Python side casts to the reference's result dtype.  Off the hot path (the reference calls it
// from find_most_similar and ad-hoc diagnostics): a plain LDS-tiled 64 x 64 kernel, no MFMA — the operands are float32
// and the accumulation float64.
#include "common.hpp"

namespace anr {

constexpr int kSimTile = 64, kSimK = 16;

// one wave per row: ||row|| as float32 (float64 sum of squares -> sqrt -> float32), + 1e-8 in float32
__global__ __launch_bounds__(256) void k_sim_denoms(const float *x, int64_t rows, int d, float *den) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  double s = 0.0;
  for (int c = lane; c < d; c += 64) {
    const double v = (double)x[r * d + c];
    s += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) den[r] = (float)sqrt(s) + 1e-8f;
}

struct SimParams {
  const float *a, *b;    // [m][d], [n][d]
  const float *da, *db;  // cosine: per-row denominators
  int64_t m, n;
  int d, metric;         // 0 cosine, 1 dot, 2 euclidean
  double *out;           // [m][n]
};

__global__ __launch_bounds__(256) void k_similarity(SimParams p) {
  __shared__ float sa[kSimK][kSimTile + 1], sb[kSimK][kSimTile + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t i0 = (int64_t)blockIdx.y * kSimTile, j0 = (int64_t)blockIdx.x * kSimTile;
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
  for (int k0 = 0; k0 < p.d; k0 += kSimK) {
    // 64 rows x 16 columns of each operand: thread t loads rows (t >> 4) + 16 e, column t & 15 (coalesced over k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = ty + 16 * e, k = k0 + tx;
      float va = 0.f, vb = 0.f;
      if (k < p.d) {
        if (i0 + row < p.m) {
          va = p.a[(i0 + row) * p.d + k];
          if (p.metric == 0) va = va / p.da[i0 + row];
        }
        if (j0 + row < p.n) {
          vb = p.b[(j0 + row) * p.d + k];
          if (p.metric == 0) vb = vb / p.db[j0 + row];
        }
      }
      sa[tx][row] = va;
      sb[tx][row] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSimK; ++k) {
      double av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = (double)sa[k][ty + 16 * u];
        bv[u] = (double)sb[k][tx + 16 * u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (p.metric == 2) {
            const double df = av[u] - bv[v];
            acc[u][v] += df * df;
          } else {
            acc[u][v] += av[u] * bv[v];
          }
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int64_t i = i0 + ty + 16 * u, j = j0 + tx + 16 * v;
      if (i < p.m && j < p.n) p.out[i * p.n + j] = p.metric == 2 ? 1.0 / (1.0 + sqrt(acc[u][v])) : acc[u][v];
    }
}

}  // namespace anr

using namespace anr;

extern "C" int anr_similarity_matrix(int32_t device, const float *a_host, int64_t m, const float *b_host, int64_t n,
                                     int32_t d, int32_t metric, double *out_host) {
  if (!a_host || !b_host || !out_host || m <= 0 || n <= 0 || d <= 0) return fail(ANR_EINVAL, "bad argument");
  if (metric < 0 || metric > 2) return fail(ANR_EINVAL, "metric must be 0 (cosine), 1 (dot) or 2 (euclidean)");
  if (anr_device_count() <= 0) return fail(ANR_EHIP, "no HIP device is visible");
  DeviceGuard g(device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", device);
  const size_t ab = (size_t)m * d * 4, bb = (size_t)n * d * 4, ob = (size_t)m * n * 8;
  const size_t o_b = round_up((int64_t)ab, 256), o_da = o_b + round_up((int64_t)bb, 256), o_db = o_da + round_up(m * 4, 256),
               o_out = o_db + round_up(n * 4, 256);
  unsigned char *D = nullptr;
  ANR_HIP(hipMalloc(reinterpret_cast<void **>(&D), o_out + ob));
  int rc = ANR_OK;
  auto run = [&]() -> int {
    ANR_HIP(hipMemcpy(D, a_host, ab, hipMemcpyHostToDevice));
    ANR_HIP(hipMemcpy(D + o_b, b_host, bb, hipMemcpyHostToDevice));
    SimParams p{};
    p.a = reinterpret_cast<const float *>(D);
    p.b = reinterpret_cast<const float *>(D + o_b);
    p.m = m; p.n = n; p.d = d; p.metric = metric;
    p.out = reinterpret_cast<double *>(D + o_out);
    if (metric == 0) {
      float *da = reinterpret_cast<float *>(D + o_da), *db = reinterpret_cast<float *>(D + o_db);
      hipLaunchKernelGGL(k_sim_denoms, dim3((unsigned)ceil_div(m, 4)), dim3(256), 0, nullptr, p.a, m, d, da);
      hipLaunchKernelGGL(k_sim_denoms, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, nullptr, p.b, n, d, db);
      p.da = da;
      p.db = db;
    }
    const int64_t gx = ceil_div(n, kSimTile), gy = ceil_div(m, kSimTile);
    if (gy > 65535) return fail(ANR_EINVAL, "too many rows in the first set (%lld)", (long long)m);
    hipLaunchKernelGGL(k_similarity, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, nullptr, p);
    ANR_HIP(hipGetLastError());
    ANR_HIP(hipMemcpy(out_host, D + o_out, ob, hipMemcpyDeviceToHost));
    return ANR_OK;
  };
  rc = run();
  (void)hipFree(D);
  return rc;
}
