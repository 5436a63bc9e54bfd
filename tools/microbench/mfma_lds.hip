#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
// waves of a 512-thread block each do per "k-step": RA a-fragment reads + RB b-fragment reads (ds_read_b128) and RA*RB MFMAs
template <int RA, int RB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float *out, int iters) {
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 4 * 64; i += WAVES * 64) lds[i] = make_uint4(i, i * 3, i * 7, 0x3c003c00u);
  __syncthreads();
  floatx16 acc[RA][RB];
  for (int m = 0; m < RA; ++m) for (int n = 0; n < RB; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  const uint4 *L = lds + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8 a[RA], b[RB];
#pragma unroll
      for (int m = 0; m < RA; ++m) a[m] = __builtin_bit_cast(half8, L[(ks * 16 + (wave & 3) * 2 % 8 + m) * 64]);
#pragma unroll
      for (int n = 0; n < RB; ++n) b[n] = __builtin_bit_cast(half8, L[(ks * 16 + 8 + (wave & 1) * 4 + n) * 64]);
#pragma unroll
      for (int m = 0; m < RA; ++m)
#pragma unroll
        for (int n = 0; n < RB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[n], a[m], acc[m][n], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int m = 0; m < RA; ++m) for (int n = 0; n < RB; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int RA, int RB, int WAVES>
void run(int iters) {
  float *out; hipMalloc(&out, 256 * WAVES * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void *)k<RA, RB, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipLaunchKernelGGL((k<RA, RB, WAVES>), dim3(256), dim3(WAVES * 64), 64 * 1024, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<RA, RB, WAVES>), dim3(256), dim3(WAVES * 64), 64 * 1024, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * WAVES * 4 * RA * RB * (double)iters * 32768.0;
  printf("RA=%d RB=%d waves=%d (%.2f KiB LDS read per MFMA): %.2f ms, %.0f TFLOP/s\n", RA, RB, WAVES, (double)(RA + RB) / (RA * RB), ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<2, 4, 8>(4000); run<4, 4, 4>(4000); run<2, 2, 8>(8000); run<2, 4, 4>(8000); run<1, 4, 8>(8000); run<2, 4, 16>(2000);
  return 0;
}
