#include <hip/hip_runtime.h>
#include <cstdio>
// global -> LDS staging rate per CU, no math: 8 waves, each issues LPW 1-KiB loads per stage (global_load_lds or via
// registers), three-slot ring, one barrier per stage; source working set `span` bytes (L2-resident when small)
template <int MODE, int LPW>
__global__ __launch_bounds__(512) void k(const uint4 *src, size_t span16, int stages, float *out) {
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = ((size_t)blockIdx.x * 7919 * 64) % span16;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int s = 0; s < stages; ++s) {
    uint4 *slot = lds + (s % 3) * (8 * LPW * 64);
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < LPW; ++i) {
        const size_t off = (base + ((size_t)s * 8 * LPW + wave * LPW + i) * 64) % span16;
        __builtin_amdgcn_global_load_lds(src + off + lane, slot + (wave * LPW + i) * 64, 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
    } else {
      uint4 r[LPW];
#pragma unroll
      for (int i = 0; i < LPW; ++i) {
        const size_t off = (base + ((size_t)s * 8 * LPW + wave * LPW + i) * 64) % span16;
        r[i] = src[off + lane];
      }
#pragma unroll
      for (int i = 0; i < LPW; ++i) slot[(wave * LPW + i) * 64 + lane] = r[i];
    }
    __builtin_amdgcn_s_barrier();
    acc.x += lds[(s % 3) * (8 * LPW * 64) + lane].x;
  }
  out[blockIdx.x * 512 + threadIdx.x] = (float)acc.x;
}
template <int MODE, int LPW>
void run(const uint4 *src, size_t span_bytes, int stages) {
  float *out; hipMalloc(&out, 256 * 512 * 4);
  const int ldsb = 3 * 8 * LPW * 1024;
  hipFuncSetAttribute((const void *)k<MODE, LPW>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, LPW>), dim3(256), dim3(512), ldsb, 0, src, span_bytes / 16, 50, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, LPW>), dim3(256), dim3(512), ldsb, 0, src, span_bytes / 16, stages, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * stages * 8 * LPW * 1024.0;
  printf("%s LPW=%d span=%zu MiB: %.2f ms, %.1f GB/s per CU, %.2f TB/s chip\n", MODE ? "via-registers" : "lds-dma      ", LPW, span_bytes >> 20, ms,
         bytes / 256 / ms / 1e6, bytes / ms / 1e9);
  hipFree(out);
}
int main() {
  uint4 *src; const size_t big = (size_t)1 << 30; hipMalloc(&src, big); hipMemset(src, 1, big);
  for (size_t span : {(size_t)2 << 20, (size_t)64 << 20, big}) {
    run<0, 6>(src, span, 4000); run<1, 6>(src, span, 4000); run<0, 4>(src, span, 6000); run<1, 8>(src, span, 3000);
  }
  return 0;
}
