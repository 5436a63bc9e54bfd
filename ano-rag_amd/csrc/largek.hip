// Top-k beyond the select kernel's 1024-entry window (faiss accepts any k; the reference's retrieve() over-fetches
// top_k x 3, vector_store/retriever.py:339-512): the exact scores of every row are sorted with the device radix sort
// (radix_sort.hip) and the first k pairs are written out.  Stable sort of (score, row) pairs whose
// rows start in ascending order, so equal scores keep ascending ids — the same tie rule as the select path.  A rare
// path (k > 1024): ~N log N per query instead of one streaming pass per batch.
#include "common.hpp"

namespace anr {

__global__ void k_iota(unsigned *v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (unsigned)i;
}

__global__ void k_emit_sorted(const float *keys, const unsigned *rows, int64_t n, int k, int larger_is_better,
                              int64_t id_offset, float *D, int64_t *I) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  if (i < n) {
    D[i] = keys[i];
    I[i] = (int64_t)rows[i] + id_offset;
  } else {
    D[i] = larger_is_better ? -3.402823466e+38f : 3.402823466e+38f;
    I[i] = -1;
  }
  __threadfence_system();  // D / I may be pinned host memory
}

int sort_topk(const float *scores_dev, int64_t n, int k, bool larger_is_better, int64_t id_offset, float *D_row,
              int64_t *I_row, LargeKScratch *s, hipStream_t st) {
  if (n > 0x7fffffffLL) return fail(ANR_EINVAL, "top-k beyond 1024 supports at most 2^31-1 rows");
  if (s->n < n) {
    (void)hipFree(s->iota); (void)hipFree(s->keys); (void)hipFree(s->rows); (void)hipFree(s->temp);
    s->iota = s->rows = nullptr; s->keys = nullptr; s->temp = nullptr; s->n = 0; s->temp_bytes = 0;
    ANR_HIP(hipMalloc(reinterpret_cast<void **>(&s->iota), (size_t)n * sizeof(unsigned)));
    ANR_HIP(hipMalloc(reinterpret_cast<void **>(&s->rows), (size_t)n * sizeof(unsigned)));
    ANR_HIP(hipMalloc(reinterpret_cast<void **>(&s->keys), (size_t)n * sizeof(float)));
    hipLaunchKernelGGL(k_iota, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, s->iota, n);
    const size_t bytes = radix_sort_temp_bytes(n);
    ANR_HIP(hipMalloc(&s->temp, bytes));
    s->temp_bytes = bytes;
    s->n = n;
  }
  ANR_TRY(radix_sort_pairs_f32(s->temp, scores_dev, s->keys, s->iota, s->rows, n, larger_is_better, st));
  hipLaunchKernelGGL(k_emit_sorted, dim3((unsigned)ceil_div(k, 256)), dim3(256), 0, st, s->keys, s->rows, n, k,
                     larger_is_better ? 1 : 0, id_offset, D_row, I_row);
  ANR_HIP(hipGetLastError());
  return ANR_OK;
}

void free_largek(LargeKScratch *s) {
  (void)hipFree(s->iota); (void)hipFree(s->keys); (void)hipFree(s->rows); (void)hipFree(s->temp);
  *s = LargeKScratch{};
}

}  // namespace anr
