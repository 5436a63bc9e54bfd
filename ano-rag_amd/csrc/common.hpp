// Shared host-side helpers for libanorag_hip.so (error reporting, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "anorag.h"

namespace anr {

// thread-local last-error text behind anr_last_error()
std::string &last_error_ref();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define ANR_HIP(expr)                                                                            \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return anr::fail(ANR_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                       __LINE__);                                                                \
  } while (0)

#define ANR_TRY(expr)       \
  do {                      \
    int _r = (expr);        \
    if (_r != ANR_OK) return _r; \
  } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }

// RAII device guard: every entry point switches to the handle's device and restores the caller's.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int device_cu_count(int device);

// device-wide stable radix sort of (key, u32 value) pairs (radix_sort.hip); temp: radix_sort_temp_bytes(n) bytes of device
// memory; equal keys keep their input order, also when descending
size_t radix_sort_temp_bytes(int64_t n);
int radix_sort_pairs_u64(void *temp, const unsigned long long *k_in, unsigned long long *k_out, const unsigned *v_in,
                         unsigned *v_out, int64_t n, bool descending, hipStream_t st);
int radix_sort_pairs_f32(void *temp, const float *k_in, float *k_out, const unsigned *v_in, unsigned *v_out, int64_t n,
                         bool descending, hipStream_t st);

// top-k beyond the select window (largek.hip): full device sort of one query's exact scores
struct LargeKScratch {
  unsigned *iota = nullptr, *rows = nullptr;
  float *keys = nullptr;
  void *temp = nullptr;
  size_t temp_bytes = 0;
  int64_t n = 0;
};
int sort_topk(const float *scores_dev, int64_t n, int k, bool larger_is_better, int64_t id_offset, float *D_row,
              int64_t *I_row, LargeKScratch *s, hipStream_t st);
void free_largek(LargeKScratch *s);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, current device): the attribute belongs to the
// device's copy of the kernel, so a process that drives several GPUs needs it on each; thread-safe
int ensure_dynamic_lds(const void *kernel, int bytes);

// XCD-aware workgroup -> tile mapping for GEMM-shaped grids (MB x NB tiles).  Workgroups are dealt round-robin
// over the 8 XCDs (b and b + 8 share one; observed, speed only), each XCD has its own L2, and the 32 workgroups
// an XCD runs at a time walk K in step — so the 32 are given a P x Q patch of tiles (P * Q = 32): the P + Q
// operand blocks of the patch are fetched from outside the XCD once and re-used from its L2 by the others,
// instead of every workgroup streaming both of its operands from the Infinity Cache / HBM (measured: the
// 256 x 256 self-join tile at K = 768 was pinned at ~6.5 TB/s of L2 misses = 840 TFLOP/s).
struct PatchGrid {
  int MB, NB, P, Q;
  int patches_n;      // patches along N
  int64_t n_patches;
  int contig;         // 1: XCD x takes the CONSECUTIVE patches [x * per_xcd, (x + 1) * per_xcd) — with patches numbered
                      // N-fastest an XCD then keeps the same M panels in its L2 while it walks the N panels (GEMM: every
                      // activation panel is fetched by ONE XCD only); 0: patches dealt round-robin over the XCDs
  // workgroups to launch: whole patches, a multiple of 8 of them
  __host__ __device__ int64_t per_xcd() const { return (n_patches + 7) / 8; }
  __host__ __device__ int64_t grid() const { return per_xcd() * 8 * (int64_t)(P * Q); }
};

static inline PatchGrid make_patch_grid(int64_t MB, int64_t NB) {
  // the 32-tile patch shape that wastes the fewest slots, squarer first
  static const int shapes[6][2] = {{4, 8}, {8, 4}, {2, 16}, {16, 2}, {1, 32}, {32, 1}};
  PatchGrid best{};
  int64_t best_slots = -1;
  for (const auto &sh : shapes) {
    const int64_t pm = ceil_div(MB, sh[0]), pn = ceil_div(NB, sh[1]);
    const int64_t slots = (pm * pn + 7) / 8 * 8 * 32;
    if (best_slots < 0 || slots < best_slots) {
      best_slots = slots;
      best = PatchGrid{(int)MB, (int)NB, sh[0], sh[1], (int)pn, pm * pn, 0};
    }
  }
  return best;
}

#if defined(__HIPCC__)
// false: this workgroup has no tile (grid padding)
__device__ __forceinline__ bool patch_tile(const PatchGrid &g, int64_t wg, int &bm, int &bn) {
  const int xcd = (int)(wg & 7);
  const int64_t local = wg >> 3;
  const int within = (int)(local % (g.P * g.Q));
  const int64_t patch = g.contig ? xcd * g.per_xcd() + local / (g.P * g.Q) : (local / (g.P * g.Q)) * 8 + xcd;
  if (patch >= g.n_patches) return false;
  bm = (int)(patch / g.patches_n) * g.P + within / g.Q;
  bn = (int)(patch % g.patches_n) * g.Q + within % g.Q;
  return bm < g.MB && bn < g.NB;
}
#endif

}  // namespace anr
