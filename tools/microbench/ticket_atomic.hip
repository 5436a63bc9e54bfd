// Microbenchmark: what does a device-scope returning fetch_add on ONE address cost — the tile ticket of a
// dynamically scheduled scan?  3072 waves (256 workgroups x 12) each take `per_wave` tickets back to back
// (one lane issues, the value is consumed before the next one), or with `sleep` cycles of s_sleep between tickets.
// Reported: tickets per microsecond for the whole device and the latency of a dependent chain on one wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o ticket_atomic ticket_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(768) void k_tickets(unsigned *counter, int per_wave, int sleep, unsigned *sink) {
  const int lane = threadIdx.x & 63;
  unsigned acc = 0;
  for (int i = 0; i < per_wave; ++i) {
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = __builtin_amdgcn_readfirstlane(t);
    acc += t;
    for (int s = 0; s < sleep; ++s) __builtin_amdgcn_s_sleep(8);
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
  struct Cfg { int wg, nt, per, sleep; };
  for (Cfg c : {Cfg{1, 64, 2000, 0}, Cfg{256, 768, 0, 0}, Cfg{256, 768, 16, 0}, Cfg{256, 768, 64, 0}, Cfg{256, 768, 16, 40},
                Cfg{256, 768, 16, 160}, Cfg{256, 64, 64, 0}, Cfg{32, 768, 64, 0}}) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(counter, 0, 4);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_tickets, dim3(c.wg), dim3(c.nt), 0, 0, counter, c.per, c.sleep, sink);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double n = (double)c.wg * (c.nt / 64) * c.per;
    printf("wg %3d threads %3d tickets/wave %4d sleep %3d: %8.1f us  -> %.1f tickets/us, %.3f us per ticket per wave\n", c.wg, c.nt,
           c.per, c.sleep, best * 1e3f, n / (best * 1e3), c.per ? best * 1e3 / c.per : 0.0);
  }
  return 0;
}
