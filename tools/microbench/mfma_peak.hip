#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int waves_per_simd, int iters) {
  const int threads = 64 * 4 * waves_per_simd;  // per CU
  float *out; hipMalloc(&out, 256 * threads * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * (threads / 256)), block(256);
  hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid.x * 4 * NACC * (double)iters * 32768.0;
  printf("NACC=%d waves/SIMD=%d: %.2f ms, %.0f TFLOP/s\n", NACC, waves_per_simd, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<8>(1, 20000); run<8>(2, 20000); run<4>(2, 40000); run<8>(1, 200000);
  return 0;
}
