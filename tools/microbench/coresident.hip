// Microbenchmark: which side kernels can be placed BESIDE the scan's resident workgroups, and how long do they take
// there?  A "resident" kernel imitates k_scan's footprint: 256 workgroups x 768 threads, 152 vector registers per
// wave (3 waves per SIMD = 456 of 512), 100.5 KiB of LDS, and streams a 4-GiB buffer with non-temporal 16-byte loads
// for a fixed time (or only sleeps, mode 0).  While it runs, a side kernel of 64 workgroups is launched on another
// stream in several shapes (threads, registers, LDS, with / without a barrier and dependent global accesses) and timed
// from its launch to its completion event.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../ano-rag_amd/csrc -o coresident coresident.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "index_kernels.hpp"  // the real k_prepq, timed beside the resident kernel too

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(768) __attribute__((amdgpu_num_vgpr(152))) void k_resident(const u32x4 *buf, size_t n16, long long ticks,
                                                                                      int stream, unsigned *sink) {
  extern __shared__ unsigned lds[];
  asm volatile("v_mov_b32 v151, 0" ::: "v151");  // the register count of the kernel descriptor follows the highest register touched
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  unsigned acc = lds[(threadIdx.x * 7) % 768];
  size_t i = (size_t)blockIdx.x * 768 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 768;
  while (wall_clock64() - t0 < ticks) {
    if (stream) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const u32x4 v = __builtin_nontemporal_load(buf + (i % n16));
        acc += v.x ^ v.w;
        i += stride;
      }
    } else {
      __builtin_amdgcn_s_sleep(64);
    }
  }
  if (acc == 0x12345u) *sink = acc;
}

#define SIDE_KERNEL(NAME, NV, TOP, BARRIER)                                                                       \
  __global__ __attribute__((amdgpu_num_vgpr(NV))) void NAME(const float *in, float *out, int n, int rounds) {        \
    extern __shared__ float sl[];                                                                                     \
    asm volatile("v_mov_b32 " TOP ", 0" ::: TOP);                                                                      \
    const int tid = threadIdx.x;                                                                                      \
    float acc = 0.f;                                                                                                  \
    for (int r = 0; r < rounds; ++r) { /* dependent global round trips */                                             \
      const int j = (blockIdx.x * blockDim.x + tid + (int)acc) % n;                                                   \
      acc += in[j];                                                                                                   \
      if (BARRIER) {                                                                                                  \
        sl[tid] = acc;                                                                                                \
        __syncthreads();                                                                                              \
        acc += sl[(tid + 1) % blockDim.x];                                                                            \
        __syncthreads();                                                                                              \
      }                                                                                                               \
    }                                                                                                                 \
    out[blockIdx.x * blockDim.x + tid] = acc;                                                                         \
  }
SIDE_KERNEL(k_side_32_nb, 32, "v31", false)
SIDE_KERNEL(k_side_32, 32, "v31", true)
SIDE_KERNEL(k_side_56, 56, "v55", true)
SIDE_KERNEL(k_side_64, 64, "v63", true)
SIDE_KERNEL(k_side_24, 24, "v23", true)

int main(int argc, char **argv) {
  const size_t bytes = (size_t)4 << 30;
  u32x4 *buf;
  float *in, *out;
  unsigned *sink;
  hipMalloc(&buf, bytes);
  hipMemset(buf, 1, bytes);
  hipMalloc(&in, 1 << 20);
  hipMemset(in, 0, 1 << 20);
  hipMalloc(&out, 1 << 20);
  hipMalloc(&sink, 4);
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_resident), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  struct Side { const char *name; int threads, lds, which; };
  const Side sides[] = {{"256 thr, 32 vgpr, no lds, no barrier", 256, 0, 0},   {"256 thr, 32 vgpr, 1 KiB lds, barrier", 256, 1024, 1},
                        {"256 thr, 56 vgpr, 1 KiB lds, barrier", 256, 1024, 2}, {"256 thr, 64 vgpr, 1 KiB lds, barrier", 256, 1024, 3},
                        {"256 thr, 56 vgpr, 48 KiB lds, barrier", 256, 48 * 1024, 2}, {"256 thr, 56 vgpr, 58 KiB lds, barrier", 256, 58 * 1024, 2},
                        {"256 thr, 56 vgpr, 64 KiB lds, barrier", 256, 64 * 1024, 2}, {"512 thr, 24 vgpr, 8 KiB lds, barrier", 512, 8192, 4},
                        {"512 thr, 32 vgpr, 8 KiB lds, barrier", 512, 8192, 1}, {"64 thr, 56 vgpr, 1 KiB lds, barrier", 64, 1024, 2}};
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_side_56), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int mode = 0; mode < 3; ++mode) {  // 0: nothing resident, 1: resident sleeping, 2: resident streaming
    printf("== %s\n", mode == 0 ? "alone" : mode == 1 ? "beside resident workgroups that sleep" : "beside resident workgroups that stream HBM");
    for (const Side &s : sides) {
      float lat[8];
      for (int rep = 0; rep < 8; ++rep) {
        hipDeviceSynchronize();
        if (mode) hipLaunchKernelGGL(k_resident, dim3(256), dim3(768), 102912, sa, buf, bytes / 16, (long long)150000, mode == 2, sink);  // 1.5 ms at 100 MHz
        // give the resident kernel time to occupy every CU
        hipEventRecord(e0, sb);
        hipEventSynchronize(e0);
        for (volatile int w = 0; w < 200000; ++w) {}
        hipEventRecord(e0, sb);
        const int rounds = 4;
        switch (s.which) {
          case 0: hipLaunchKernelGGL(k_side_32_nb, dim3(64), dim3(s.threads), s.lds, sb, in, out, 1 << 18, rounds); break;
          case 1: hipLaunchKernelGGL(k_side_32, dim3(64), dim3(s.threads), s.lds, sb, in, out, 1 << 18, rounds); break;
          case 2: hipLaunchKernelGGL(k_side_56, dim3(64), dim3(s.threads), s.lds, sb, in, out, 1 << 18, rounds); break;
          case 3: hipLaunchKernelGGL(k_side_64, dim3(64), dim3(s.threads), s.lds, sb, in, out, 1 << 18, rounds); break;
          case 4: hipLaunchKernelGGL(k_side_24, dim3(64), dim3(s.threads), s.lds, sb, in, out, 1 << 18, rounds); break;
        }
        hipEventRecord(e1, sb);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&lat[rep], e0, e1);
      }
      float mn = 1e9f, mx = 0.f;
      for (int rep = 2; rep < 8; ++rep) {
        mn = lat[rep] < mn ? lat[rep] : mn;
        mx = lat[rep] > mx ? lat[rep] : mx;
      }
      printf("  %-42s  min %8.1f us  max %8.1f us\n", s.name, mn * 1e3f, mx * 1e3f);
    }
  }
  // the real query-preparation kernel: 64 workgroups x 256 threads, 27 registers
  {
    float *qin, *q32, *qstat;
    _Float16 *q16;
    hipMalloc(&qin, 64 * 768 * 4);
    hipMemset(qin, 0x3c, 64 * 768 * 4);
    hipMalloc(&q32, 64 * 768 * 4);
    hipMalloc(&q16, 64 * 768 * 2);
    hipMalloc(&qstat, 64 * 16);
    anr::PrepQParams qp{};
    qp.qin = qin; qp.nq = 64; qp.dim = 768; qp.dimp = 768; qp.kb = 48; qp.normalize = 1; qp.q32 = q32; qp.q16 = q16; qp.qstat = qstat;
    for (int mode = 0; mode < 3; ++mode) {
      float lat[8];
      for (int rep = 0; rep < 8; ++rep) {
        hipDeviceSynchronize();
        if (mode) hipLaunchKernelGGL(k_resident, dim3(256), dim3(768), 102912, sa, buf, bytes / 16, (long long)150000, mode == 2, sink);
        hipEventRecord(e0, sb);
        hipEventSynchronize(e0);
        for (volatile int w = 0; w < 200000; ++w) {}
        hipEventRecord(e0, sb);
        hipLaunchKernelGGL(anr::k_prepq, dim3(64), dim3(256), 0, sb, qp);
        hipEventRecord(e1, sb);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&lat[rep], e0, e1);
      }
      float mn = 1e9f, mx = 0.f;
      for (int rep = 2; rep < 8; ++rep) { mn = lat[rep] < mn ? lat[rep] : mn; mx = lat[rep] > mx ? lat[rep] : mx; }
      printf("k_prepq %s: min %8.1f us  max %8.1f us\n", mode == 0 ? "alone" : mode == 1 ? "beside sleeping resident" : "beside streaming resident", mn * 1e3f, mx * 1e3f);
    }
  }
  hipDeviceSynchronize();
  return 0;
}
