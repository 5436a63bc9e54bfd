// Microbenchmark: what does a grid-wide barrier between co-resident workgroups cost on this chip?
//   one counter in device memory, every workgroup adds 1 (device scope) and spins until the count reaches
//   (round + 1) * n_wg.  Reported: time per barrier for n_wg = 32 .. 256 workgroups of 256 threads.
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_barriers(unsigned *counter, int rounds, unsigned *sink) {
  const unsigned n = gridDim.x;
  unsigned acc = 0;
  for (int r = 0; r < rounds; ++r) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * n;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    acc += r;
  }
  if (acc == 0xffffffffu) *sink = acc;
}

int main() {
  unsigned *counter, *sink;
  hipMalloc(&counter, 4);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int n_wg : {8, 32, 64, 128, 256}) {
    for (int rounds : {0, 200}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemset(counter, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_barriers, dim3(n_wg), dim3(256), 0, 0, counter, rounds, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("n_wg %3d rounds %3d: %.1f us%s\n", n_wg, rounds, best * 1e3f,
             rounds ? "" : "  (empty kernel)");
      if (rounds) printf("   -> %.2f us per barrier\n", best * 1e3f / rounds);
    }
  }
  return 0;
}
