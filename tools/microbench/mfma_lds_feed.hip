// The GEMM inner loop (6 ds_read_b128 -> 8 MFMAs per wave per k-step, 8 waves) WITH the operand staging traffic of a
// 256 x 256 tile beside it: 16 KiB per k-step per CU copied global -> LDS (global_load_lds, three-slot ring, one
// barrier per 3-k-step stage), from a source of `span` bytes (2 MiB: L2-resident; 1 GiB: HBM).  The MFMAs read a
// separate, pre-filled LDS region, so there is no data dependence: this isolates port / issue contention.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 gload(const void *p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");  // not tracked by the compiler's waitcnt pass
  return v;
}
template <int FEED>
__global__ __launch_bounds__(512) void k(const uint4 *src, size_t span16, float *out, int stages) {
  extern __shared__ uint4 lds[];  // [0, 1024): MFMA operands (re-read every k-step); then ring 3 x 3 x 16 x 64
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = make_uint4(i, i * 3, i * 7, 0x3c003c00u);
  __syncthreads();
  uint4 *ring = lds + 1024;
  floatx16 acc[2][4];
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  const uint4 *L = lds + lane;
  const size_t base = ((size_t)blockIdx.x * 7919 * 64) % span16;
  constexpr int D = 3;  // register stages in flight (mode 8)
  u32x4 rs[D][6];
  if (FEED & 8) {
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int i = 0; i < 6; ++i) rs[d][i] = gload(src + (base + ((size_t)d * 48 + wave * 6 + i) * 64) % span16 + lane);
  }
  for (int s = 0; s < stages; ++s) {
    if (FEED & 8) {
      // oldest register stage has landed when at most (D - 1) * 6 loads are outstanding
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      uint4 *slot = ring + (s % 3) * (48 * 64);
      const int d = s % D;
#pragma unroll
      for (int dd = 0; dd < D; ++dd)
        if (dd == d) {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            asm volatile("" : "+v"(rs[dd][i]));
            reinterpret_cast<u32x4 *>(slot)[(wave * 6 + i) * 64 + lane] = rs[dd][i];
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) rs[dd][i] = gload(src + (base + ((size_t)(s + D) * 48 + wave * 6 + i) * 64) % span16 + lane);
        }
    }
    if (FEED & 1) {
      uint4 *slot = ring + (s % 3) * (48 * 64);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const size_t off = (base + ((size_t)s * 48 + wave * 6 + i) * 64) % span16;
        __builtin_amdgcn_global_load_lds(src + off + lane, slot + (wave * 6 + i) * 64, 16, 0, 0);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      half8 a[2], b[4];
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = __builtin_bit_cast(half8, L[((wave >> 1) * 2 + m) * 64]);
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = __builtin_bit_cast(half8, L[(8 + (wave & 1) * 4 + n) * 64]);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[n], a[m], acc[m][n], 0, 0, 0);
    }
    if (FEED & 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // the previous stage's copies have landed
    if (FEED & 2) __builtin_amdgcn_s_barrier();
    if (FEED & 4) {  // ds_write of 6 KiB per wave per stage instead of the DMA (same LDS write volume, no global traffic)
      uint4 *slot = ring + (s % 3) * (48 * 64);
#pragma unroll
      for (int i = 0; i < 6; ++i) slot[(wave * 6 + i) * 64 + lane] = make_uint4(s, i, lane, wave);
    }
  }
  float sm = 0.f;
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) sm += acc[m][n][r];
  out[blockIdx.x * 512 + threadIdx.x] = sm + (float)ring[lane].x;
}
template <int FEED>
void run(const uint4 *src, size_t span, int stages) {
  float *out; hipMalloc(&out, 256 * 512 * 4);
  const int ldsb = (1024 + 3 * 48 * 64) * 16;
  hipFuncSetAttribute((const void *)k<FEED>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<FEED>), dim3(256), dim3(512), ldsb, 0, src, span / 16, out, 20);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<FEED>), dim3(256), dim3(512), ldsb, 0, src, span / 16, out, stages);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 8 * 3 * 8 * (double)stages * 32768.0;
  printf("feed=%d span=%zu MiB: %.2f ms, %.0f TFLOP/s, staging %.1f GB/s per CU\n", FEED, span >> 20, ms, flops / ms / 1e9,
         (FEED & 13) ? 48.0 * 1024 * stages / ms / 1e6 : 0.0);
  hipFree(out);
}
int main() {
  uint4 *src; const size_t big = (size_t)1 << 30; hipMalloc(&src, big); hipMemset(src, 1, big);
  run<0>(src, 2 << 20, 6000);                  // inner loop alone
  run<2>(src, 2 << 20, 6000);                  // + one barrier per stage
  run<1>(src, (size_t)2 << 20, 6000);          // + DMA (L2-resident source), no barrier
  run<3>(src, (size_t)2 << 20, 6000);          // + DMA + barrier
  run<4>(src, (size_t)2 << 20, 6000);          // + ds_write of the same volume, no global traffic, no barrier
  run<6>(src, (size_t)2 << 20, 6000);          // + ds_write + barrier
  run<3>(src, big, 6000);                      // DMA from HBM + barrier
  run<10>(src, (size_t)2 << 20, 6000);         // register-staged, 3 stages deep, counted waits + barrier (L2)
  run<10>(src, (size_t)64 << 20, 6000);        // ... Infinity Cache
  run<10>(src, big, 6000);                     // ... HBM
  return 0;
}
