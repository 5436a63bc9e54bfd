// Exact flat index behind anr_index_* (include/anorag.h): host orchestration of the kernels in
// index_kernels.hpp.  Replaces faiss.IndexFlatIP / IndexFlatL2 at the call sites of the reference's
// vector_store/vector_index.py (:77-80 create, :196 add, :223 search, :415-426 reset).
//
// Search pipeline for one batch of <= 64 queries, all on one stream:
//   pre  : prepq -> scan<DENSE> on a strided tile sample -> select -> threshold ladder
//   scan : the f16 MFMA scan (threshold-gated candidate append) — the only HBM-heavy kernel
//   post : select top-K' -> rescore (f32 rows, f64 accumulate) -> finalize (sort, top-k, certificate, status)
// Batches rotate over kWorkspaces workspaces, each with its own stream, and nothing orders one batch
// against the next: the small kernels of neighbouring batches fill the ramp and the tail of the current
// scan, and the completion marker of one batch does not sit in front of the next one's first kernel
// (measured at a 1.25 M-row shard: 0.417 -> 0.376 ms per batch).  Queries whose certificate fails are
// answered by the second pass / the dense exact path when the batch is retired.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "index_kernels.hpp"
#include "tiny_kernels.hpp"

using namespace anr;


namespace {
constexpr int kBatchLogFields = ANR_BATCH_LOG_FIELDS;
constexpr int kBatchLogCap = 512;
inline int64_t host_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
constexpr int kWorkspaces = 3;
constexpr int64_t kZeroCopyResults = 64 * 1024;       // nq * k up to which results are written straight to host memory

struct Workspace {
  // device buffers of one in-flight batch
  float *q32 = nullptr;
  _Float16 *q16 = nullptr;
  float *qstat = nullptr;
  float *qstage = nullptr;  // [64][dim] staging of host queries (device memory, DMA path)
  float *qpin = nullptr;    // [64][dim] pinned host staging of host queries (source of the DMA into qstage)
  float *dense = nullptr;
  int64_t dense_ld = 0;
  float *ladder = nullptr;
  unsigned *cntb = nullptr;   // [64][n_cu]
  unsigned *ncand = nullptr;  // status block: [64] candidates | [64] overflow | [64] flags | [64] theta (float)
  int *qslots = nullptr;      // [64] query slots of a second pass
  int *lvlmax = nullptr;      // [64] highest ladder level any scan workgroup ended at
  int scan_grid = 0;          // workgroups (= candidate lists per query) of the batch's main scan
  uint2 *cand = nullptr;      // [n_cu][64][cand_cap]
  int64_t cand_alloc = 0;
  float *sel_rank = nullptr;
  unsigned *sel_row = nullptr;
  int *sel_m = nullptr;
  float *exact = nullptr;
  unsigned *cnt_host = nullptr;    // pinned [6][64] batch status, written by k_post / k_finalize (rows 4-5: the batch's time stamps)
  unsigned *cnt_dev = nullptr;     // the same memory as the device addresses it
  unsigned long long *stamps = nullptr;  // [kStamps] device time stamps of the batch's kernels (index_kernels.hpp)
  int64_t t_enq0 = 0, t_enq1 = 0;  // host clock (ns) when the batch's enqueue began / returned
  int64_t log_slot = -1;           // the batch's record in the handle's batch log (recovery time is added to it)
  // recovery of the queries whose certificate failed: launched without waiting (launch_recovery), finished at the retire
  bool rec_launched = false;
  hipEvent_t ev_rec = nullptr;
  int *qslots_pin = nullptr;       // pinned [2][64]: query slots of the two recovery chains (source of their uploads)
  float *lad_pin = nullptr;        // pinned [64][kLadder]: the fixed-threshold ladder of a second scan
  std::vector<int> rec_lists, rec_scan, rec_dense;
  int64_t seq = 0;
  bool shadow = false;             // the batch's side kernels were the shadow-sized ones
  bool f12 = false;                // the batch's streaming pass read the 12-bit image
  int f12_level = 0;               // ... at this level of the K' ladder
  hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  hipEvent_t ev_pre = nullptr, ev_scan = nullptr;  // role streams: ladder ready (pre -> main), scan done (main -> post)
  // the batch in flight
  bool in_flight = false;   // set once the batch's completion event has been recorded, cleared when it is final
  bool folded = false;      // its statistics have been added to the handle's
  bool sparse = false, timed = false, exact_all = false;
  int nq = 0, k = 0, M = 0;
  int64_t out_off = 0, sample_rows = 0, scan_bytes = 0;
  float *D = nullptr;
  int64_t *I = nullptr;
};
}  // namespace

struct anr_index {
  int dim = 0, dimp = 0, kb = 0;
  int metric = 0, normalize = 0, device = 0;
  int n_cu = 256;
  int64_t ntotal = 0, cap = 0;  // cap: rows allocated, multiple of 32
  float *x32 = nullptr;
  _Float16 *x16 = nullptr;
  float *rowbias = nullptr;
  unsigned *xstat = nullptr;  // [4]: max ||x||, max ||x16 - x||, f16-range flag, max ||x12 - x||
  unsigned *x12 = nullptr;    // the 12-bit image the streaming scan reads when scan_bits == 12 (index_kernels.hpp)
  unsigned *xstat12 = nullptr;  // [2]: max ||x||, max(||x16 - x||, ||x12 - x||) — the certificate's statistics of such batches
  int scan_bits = 0;          // ANR_OPT_SCAN_BITS: 0 auto, 12, 16
  bool f12_suspended = false; // the corpus proved too dense for the 12-bit image (adapt_overfetch): batches read x16 again
  int f12_strikes = 0;
  bool x12_no_memory = false; // the 12-bit image could not be allocated: not tried again until rows are added or the option is set
  // K' of the 12-bit batches: its own ladder of quarter steps (kF12Steps) and a step-down that backs off
  int f12_level = 0, f12_clean = 0, f12_hold = 64, f12_lowered = 0;
  hipStream_t stream = nullptr;                        // adds, exact path, copies
  hipStream_t bstream[kWorkspaces] = {nullptr, nullptr, nullptr};  // one per in-flight batch
  std::mutex mu;

  // options
  int force_exact = 0;
  int overfetch = 0;
  int sample_rows = 0;
  int64_t cand_cap = 512;   // entries per (block, query) candidate list
  int timing = 0;
  int add_raw = 0;          // adds store the rows as given (already preprocessed, e.g. a reloaded index)
  int64_t id_offset = 0;    // added to every returned id (a shard's first global row)
  int overfetch_boost = 1;  // adaptive multiplier of the automatic K' (adapt_overfetch)
  int clean_batches = 0;
  int n_streams = kWorkspaces;  // streams the batches rotate over (1 = strictly one batch after the other)

  Workspace ws[kWorkspaces];
  bool ws_ready = false;
  int next_ws = 0;
  bool f16_unusable = false;  // a stored value left the f16 range (L2 metric with large inputs)
  bool xstat_dirty = true;

  float *xdense = nullptr;      // exact dense path [4][xdense_ld]
  LargeKScratch largek;         // device sort scratch of the k > 1024 path
  int64_t xdense_ld = 0;
  float *d_out = nullptr;       // staging for host-pointer searches
  int64_t *i_out = nullptr;
  int64_t out_alloc = 0;
  unsigned char *out_pin = nullptr, *out_pin_dev = nullptr;  // pinned [nq*k f32 | nq*k i64] the kernels write in place
  int64_t out_pin_alloc = 0;                                  // results (nq * k)
  hipEvent_t ev_call[2] = {nullptr, nullptr};

  // single-launch path for tiny corpora (tiny_kernels.hpp)
  int fused_post = 1;                             // ANR_OPT_FUSED_POST
  int shadow = 0;                                 // ANR_OPT_SHADOW
  int schedule = 0;                               // ANR_OPT_SCHEDULE
  int stream_wait = 0;                            // ANR_OPT_STREAM_WAIT
  int64_t batch_seq = 0;
  std::vector<int64_t> batch_log;                 // ring of kBatchLogFields-word records, one per retired batch
  int64_t batch_log_n = 0;                        // records ever written
  int clock_khz = 100000;                         // rate of the device's constant clock (wall_clock64)
  int tiny = 1;                                   // ANR_OPT_TINY
  unsigned char *tiny_pin = nullptr, *tiny_pin_dev = nullptr;  // pinned: queries | D | I | completion words
  unsigned char *sr_buf = nullptr;                // anr_index_score_rows scratch
  int64_t sr_cap = 0;
  unsigned long long *tiny_cand = nullptr;        // [kTinyMaxQ][kTinyMaxPrune]: a query's n_wg partial lists of k keys
  unsigned *tiny_ticket = nullptr;                // [kTinyMaxQ]
  unsigned tiny_seq = 0;
  unsigned long long *tiny_stamps = nullptr;      // developer aid: ANORAG_TINY_STAMPS=1 prints the kernel's phase times

  anr_search_stats stats{};
};

namespace {

template <typename T>
int dev_alloc(T **p, int64_t count, bool zero) {
  *p = nullptr;
  if (count <= 0) count = 1;
  ANR_HIP(hipMalloc(reinterpret_cast<void **>(p), (size_t)count * sizeof(T)));
  if (zero) {
    // hipMemset on device memory is asynchronous to the host and runs on the NULL stream, which the handle's
    // non-blocking streams do not wait for: without the synchronisation a k_add enqueued right after could run
    // first and have its rows zeroed again
    ANR_HIP(hipMemset(*p, 0, (size_t)count * sizeof(T)));
    ANR_HIP(hipStreamSynchronize(nullptr));
  }
  return ANR_OK;
}

template <typename T>
void dev_free(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

int drain(anr_index *h);  // retire every in-flight batch

int grow_storage(anr_index *h, int64_t need_rows) {
  if (need_rows <= h->cap) return ANR_OK;
  ANR_TRY(drain(h));
  int64_t ncap = h->cap ? h->cap : 1024;
  while (ncap < need_rows) ncap = ncap + ncap / 2 + 1024;
  ncap = round_up(ncap, kTileRows);
  if (need_rows > (int64_t)0xfffffff0ll) return fail(ANR_EINVAL, "index holds at most 2^32-16 rows per device");
  float *nx32 = nullptr;
  _Float16 *nx16 = nullptr;
  unsigned *nx12 = nullptr;
  float *nbias = nullptr;
  int rc = dev_alloc(&nx32, ncap * h->dim, false);
  if (rc == ANR_OK) rc = dev_alloc(&nx16, ncap * h->dimp, true);
  if (rc == ANR_OK && (h->scan_bits == 12 || h->x12)) rc = dev_alloc(&nx12, ncap * h->dimp * 3 / 8, true);
  if (rc == ANR_OK && h->metric == ANR_METRIC_L2) rc = dev_alloc(&nbias, ncap, true);
  if (rc == ANR_OK && h->ntotal > 0) {
    const int64_t used = round_up(h->ntotal, kTileRows);
    hipError_t e = hipMemcpyAsync(nx32, h->x32, (size_t)h->ntotal * h->dim * sizeof(float), hipMemcpyDeviceToDevice,
                                  h->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(nx16, h->x16, (size_t)used * h->dimp * sizeof(_Float16), hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess && nx12 && h->x12)
      e = hipMemcpyAsync(nx12, h->x12, (size_t)used * h->dimp * 3 / 2, hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess && nbias)
      e = hipMemcpyAsync(nbias, h->rowbias, (size_t)h->ntotal * sizeof(float), hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) rc = fail(ANR_EHIP, "growing the index storage failed: %s", hipGetErrorString(e));
  }
  if (rc != ANR_OK) {  // the index keeps its old storage
    dev_free(nx32);
    dev_free(nx16);
    dev_free(nx12);
    dev_free(nbias);
    return rc;
  }
  dev_free(h->x32);
  dev_free(h->x16);
  dev_free(h->x12);
  dev_free(h->rowbias);
  h->x32 = nx32;
  h->x16 = nx16;
  h->x12 = nx12;
  h->rowbias = nbias;
  h->cap = ncap;
  return ANR_OK;
}

void free_workspaces(anr_index *h);

int alloc_workspaces(anr_index *h) {
  for (auto &w : h->ws) {
    ANR_TRY(dev_alloc(&w.q32, (int64_t)kQB * h->dimp, true));
    ANR_TRY(dev_alloc(&w.q16, (int64_t)kQB * h->dimp, true));
    ANR_TRY(dev_alloc(&w.qstat, kQB * 4, true));
    ANR_TRY(dev_alloc(&w.qstage, (int64_t)kQB * h->dim, true));
    ANR_HIP(hipHostMalloc(reinterpret_cast<void **>(&w.qpin), (size_t)kQB * h->dim * sizeof(float), hipHostMallocDefault));
    ANR_TRY(dev_alloc(&w.ladder, kQB * kLadder, true));
    ANR_TRY(dev_alloc(&w.cntb, (int64_t)kQB * h->n_cu, true));
    ANR_TRY(dev_alloc(&w.ncand, 4 * kQB, true));
    ANR_TRY(dev_alloc(&w.qslots, 2 * kQB, true));
    ANR_TRY(dev_alloc(&w.lvlmax, kQB, true));
    ANR_TRY(dev_alloc(&w.sel_rank, kQB * kMaxSel, true));
    ANR_TRY(dev_alloc(&w.sel_row, kQB * kMaxSel, true));
    ANR_TRY(dev_alloc(&w.sel_m, kQB, true));
    ANR_TRY(dev_alloc(&w.exact, kQB * kMaxSel, true));
    ANR_TRY(dev_alloc(&w.stamps, kStamps, true));
    ANR_HIP(hipHostMalloc(reinterpret_cast<void **>(&w.cnt_host), 7 * kQB * sizeof(unsigned), hipHostMallocDefault));
    memset(w.cnt_host, 0, 7 * kQB * sizeof(unsigned));
    ANR_HIP(hipHostMalloc(reinterpret_cast<void **>(&w.qslots_pin), 2 * kQB * sizeof(int), hipHostMallocDefault));
    ANR_HIP(hipHostMalloc(reinterpret_cast<void **>(&w.lad_pin), kQB * kLadder * sizeof(float), hipHostMallocDefault));
    ANR_HIP(hipEventCreateWithFlags(&w.ev_rec, hipEventDisableTiming));
    ANR_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&w.cnt_dev), w.cnt_host, 0));
    ANR_HIP(hipEventCreateWithFlags(&w.ev_in, hipEventDisableTiming));
    ANR_HIP(hipEventCreateWithFlags(&w.ev_pre, hipEventDisableTiming | hipEventDisableSystemFence));
    ANR_HIP(hipEventCreateWithFlags(&w.ev_scan, hipEventDisableTiming | hipEventDisableSystemFence));
    // the batch status lands in pinned memory with its own system-scope fence: no cache write-back needed here
    ANR_HIP(hipEventCreateWithFlags(&w.ev_done, hipEventDisableTiming | hipEventDisableSystemFence));
    ANR_HIP(hipEventCreateWithFlags(&w.ev_t0, hipEventDisableSystemFence));  // timing only
    ANR_HIP(hipEventCreateWithFlags(&w.ev_t1, hipEventDisableSystemFence));
  }
  for (auto &e : h->ev_call) ANR_HIP(hipEventCreate(&e));
  return ANR_OK;
}

int ensure_workspaces(anr_index *h) {
  if (h->ws_ready) return ANR_OK;
  const int rc = alloc_workspaces(h);
  if (rc != ANR_OK) {
    free_workspaces(h);  // a half-built set would leak on the next attempt
    return rc;
  }
  h->ws_ready = true;
  return ANR_OK;
}

int ensure_cand(anr_index *h, Workspace &w) {
  if (w.cand && w.cand_alloc == h->cand_cap) return ANR_OK;
  dev_free(w.cand);
  ANR_TRY(dev_alloc(&w.cand, (int64_t)h->n_cu * kQB * h->cand_cap, false));
  w.cand_alloc = h->cand_cap;
  return ANR_OK;
}

int ensure_dense(Workspace &w, int64_t ld) {
  if (w.dense && w.dense_ld >= ld) return ANR_OK;
  dev_free(w.dense);
  w.dense_ld = round_up(ld, 32);
  return dev_alloc(&w.dense, (int64_t)kQB * w.dense_ld, false);
}

int launch_select(int nblocks, const SelParams &sp, hipStream_t st) {
  ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_select), (int)sizeof(SelShared)));
  if (sp.G > kSelMaxLists) return fail(ANR_EINTERNAL, "select: too many candidate lists (%d)", sp.G);
  // small inputs (threshold sample, candidate lists) use a 256-thread block: barriers are 4x cheaper
  const int64_t n_hint = sp.dense ? sp.n : (int64_t)sp.G * 16;
  const int nt = 1024;
  (void)n_hint;
  hipLaunchKernelGGL(k_select, dim3(nblocks), dim3(nt), sizeof(SelShared), st, sp);
  ANR_HIP(hipGetLastError());
  return ANR_OK;
}

// shadow-sized side kernels (index_kernels.hpp: k_sample, k_select_shadow, k_rescore_shadow): 256 threads, <= 56 registers,
// <= 29 KiB of LDS
int launch_select_shadow(int nblocks, const SelParams &sp, hipStream_t st) {
  if (sp.G > kShadowLists || sp.M > kShadowSel) return fail(ANR_EINTERNAL, "shadow select: %d lists / %d entries", sp.G, sp.M);
  hipLaunchKernelGGL(k_select_shadow, dim3(nblocks), dim3(256), sizeof(SelSharedShadow), st, sp);
  ANR_HIP(hipGetLastError());
  return ANR_OK;
}

// k-blocks per LDS slice of the shadow sample: the largest divisor of kb that is a multiple of `ch` and <= 28 (KiB)
int sample_slice(int kb, int ch) {
  for (int s = 28 / ch * ch; s >= ch; s -= ch)
    if (kb % s == 0) return s;
  return 0;
}

int launch_sample_shadow(SampleParams p, hipStream_t st, int n_cu) {
  const int64_t per_half = std::max<int64_t>(1, std::min<int64_t>(n_cu / 2, ceil_div(p.n_tiles, 4)));
  // (kb % 8 == 0, so slices of a multiple of four blocks always exist)
  const int ch = p.rowbias ? 2 : 4;
  p.kbs = sample_slice(p.kb, ch);
  if (p.kbs <= 0) return fail(ANR_EINTERNAL, "shadow sample: no LDS slice for kb = %d", p.kb);
  const size_t lds = (size_t)p.kbs * 1024;
  if (p.rowbias) hipLaunchKernelGGL((k_sample<2, true>), dim3((unsigned)(2 * per_half)), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((k_sample<4, false>), dim3((unsigned)(2 * per_half)), dim3(256), lds, st, p);
  ANR_HIP(hipGetLastError());
  return ANR_OK;
}

int launch_post(int nblocks, const SelParams &sp, const PostParams &pp, hipStream_t st) {
  ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_post), (int)sizeof(SelShared)));
  if (sp.G > kSelMaxLists) return fail(ANR_EINTERNAL, "select: too many candidate lists (%d)", sp.G);
  hipLaunchKernelGGL(k_post, dim3(nblocks), dim3(1024), sizeof(SelShared), st, sp, pp);
  ANR_HIP(hipGetLastError());
  return ANR_OK;
}

template <bool DENSE, bool F12 = false>
int launch_scan(anr_index *h, const ScanParams &p, hipStream_t st, int max_grid, int *grid_out = nullptr) {
  if (p.n_tiles <= 0) return ANR_OK;
  const size_t lds = (size_t)2 * p.kb * 64 * 16 + kQB * kLadder * 8 + kQB * 8;
  // 12 waves per block (3 per SIMD: 168 VGPRs each, room for two 8-KiB operand buffers without spills;
  // measured faster than 16 waves at 128 VGPRs; the 12-bit loader with 12 blocks per buffer instead of 8 — 168
  // registers, half again the bytes in flight — measured no faster: 1.747-1.770 vs 1.696-1.739 ms per 10 M-row scan; 16 waves of 118 registers with 4-block buffers: 1.781) when there is enough work for every CU, else smaller
  // blocks on more CUs
  int nwaves = 12;
  if (ceil_div(p.n_tiles, nwaves) < max_grid) nwaves = 8;
  if (ceil_div(p.n_tiles, nwaves) < max_grid) nwaves = 4;
  const int nt = nwaves * 64;
  int64_t grid = ceil_div(p.n_tiles, nwaves);
  if (grid > max_grid) grid = max_grid;
  if (grid_out) *grid_out = (int)grid;
  // a pass over more than the Infinity Cache can hold streams with non-temporal loads
  const bool stream = p.tile_stride == 1 && p.n_tiles * (int64_t)p.kb * 1024 > ((int64_t)192 << 20);
#define ANR_LAUNCH_SCAN(CHV, STRV)                                                             \
  {                                                                                            \
    ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_scan<DENSE, CHV, 768, STRV, F12>), 160 * 1024)); \
    hipLaunchKernelGGL((k_scan<DENSE, CHV, 768, STRV, F12>), dim3((unsigned)grid), dim3(nt), lds, st, p); \
  }
  if (p.kb % 16 == 0) {
    if (stream) ANR_LAUNCH_SCAN(8, true) else ANR_LAUNCH_SCAN(8, false)
  } else {
    if (stream) ANR_LAUNCH_SCAN(4, true) else ANR_LAUNCH_SCAN(4, false)
  }
#undef ANR_LAUNCH_SCAN
  ANR_HIP(hipGetLastError());
  return ANR_OK;
}

hipError_t launch_join(const JoinParams &jp, int64_t n_slots, hipStream_t st, int n_cu) {
  constexpr int slots_bytes = 2 * 4 * 16 * 1024;
  if (ensure_dynamic_lds(reinterpret_cast<const void *>(&k_join_r), slots_bytes) != ANR_OK) return hipErrorInvalidValue;
  const int64_t persistent = std::min<int64_t>(n_slots, std::max(8, n_cu / 8 * 8));  // a multiple of 8: slot % 8 = XCD
  hipLaunchKernelGGL(k_join_r, dim3((unsigned)persistent), dim3(512), slots_bytes, st, jp);
  return hipGetLastError();
}

int refresh_xstat(anr_index *h);

// ANR_OPT_SCAN_BITS 0: indexes of at least this many rows scan the 12-bit image (768-d, batch 64, pipelined, 16 -> 12 bits:
// 131 072 rows 0.051 -> 0.051 ms per batch, 262 144 rows 0.089 -> 0.077, 500 000 rows 0.149 -> 0.124)
constexpr int64_t kAuto12Rows = 262144;

// the 12-bit image of what is stored (later adds keep it up to date); the caller has drained the pipeline or is about to
// enqueue on a stream that follows h->stream's work
int build_x12(anr_index *h) {
  if (h->x12 || h->cap <= 0) return ANR_OK;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&h->x12), (size_t)h->cap * h->dimp * 3 / 2);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    h->x12 = nullptr;
    return fail(ANR_EHIP, "no memory for the 12-bit image: %s", hipGetErrorString(e));
  }
  ANR_HIP(hipMemsetAsync(h->x12, 0, (size_t)h->cap * h->dimp * 3 / 2, h->stream));
  if (h->ntotal > 0) {
    Build12Params bp{h->x32, h->x16, h->x12, h->ntotal, h->dim, h->kb, h->xstat};
    hipLaunchKernelGGL(k_build12, dim3((unsigned)ceil_div(h->ntotal, kTileRows)), dim3(256), 0, h->stream, bp);
    ANR_HIP(hipGetLastError());
  }
  ANR_HIP(hipStreamSynchronize(h->stream));
  h->xstat_dirty = true;
  return refresh_xstat(h);
}

// K' of a 12-bit batch = the base (256 at k = 100) x kF12Steps[level] / 4
constexpr int kF12Steps[] = {4, 6, 8, 10, 12, 16};
constexpr int kF12Levels = 6;

int auto_overfetch(const anr_index *h, int k, bool f12 = false) {
  if (h->overfetch > 0) {
    int m = h->overfetch < k ? k : h->overfetch;
    return m > kMaxSel ? kMaxSel : m;
  }
  int extra = k / 2 > 32 ? k / 2 : 32;
  // the 12-bit image's error term is ~15x the f16 one: K' = 192 for k = 100 left 2.5 % of the certificates of a 10 M-row
  // Gaussian corpus failing, 256 none (and 320 / 384 cost more than they save: 1.737 / 1.770 / 1.776 ms per batch)
  if (f12) {
    extra = k > 64 ? k : 64;
    const int m = (int)round_up((round_up(k + extra, 64) * kF12Steps[h->f12_level]) / 4, 64);
    return m > kMaxSel ? kMaxSel : m;
  }
  int m = (int)round_up(k + extra, 64) * h->overfetch_boost;
  return m > kMaxSel ? kMaxSel : m;
}

// Adaptive overfetch: on corpora with dense neighbourhoods (hundreds of rows within the f16 error bound of the
// k-th score) K' = 192 candidates cannot certify the top-100 and every batch pays a second scan.  When more than
// a quarter of a batch's queries fail their certificate the automatic K' doubles (up to 4x, capped at 1024) —
// an exact re-score of 1024 rows per query costs ~40 us per batch, the second scan ~300 us at a 1.25 M-row
// shard — and it halves again after 16 consecutive batches without a failure.
void adapt_overfetch(anr_index *h, int n_queries, int n_failed, bool f12, int batch_level = 0) {
  if (n_queries <= 0) return;
  if (f12) {
    // (batches are retired up to three behind the front: one that ran at an older level says nothing about the current one —
    // counting it made three failing batches in flight climb three levels at once)
    if (h->overfetch == 0 && batch_level != h->f12_level) return;
    // The 12-bit image widens the certificate's error term ~15x, so the K' a corpus needs depends on how many rows lie that
    // close to the k-th score: 256 on a Gaussian corpus, ~640 on SURVEY 8d's clustered one (1024 centres, sigma 0.3: 0.294 ms
    // per 1.25 M-row batch there against 0.320 from the f16 image; the doubling ladder below bounced between 512, where
    // 39 % of the certificates fail, and 1024: 0.318).  Its own ladder in quarter steps of the base: one step up when more
    // than a quarter of a batch fails, one step down after `hold` (64, …) clean batches — and a step down that fails straight away
    // doubles `hold`, so the level settles instead of oscillating.  Failing at the top level (or at a K' the caller fixed) in
    // two batches running — near-duplicate neighbourhoods: the lists overflow into the dense exact path — sends the index
    // back to the f16 image until the option is set again or the index is reset.
    const bool failing = 4 * n_failed > n_queries;
    if (h->f12_lowered > 0) --h->f12_lowered;
    if (failing && (h->overfetch > 0 || h->f12_level >= kF12Levels - 1)) {
      if (++h->f12_strikes >= 2) {
        h->f12_suspended = true;
        h->f12_level = h->f12_clean = 0;
        h->f12_hold = 64;
        // a corpus this dense defeats the f16 image's first K' as well (tests: tight clusters need 4x): start there, the
        // f16 ladder steps down by itself after 16 clean batches
        if (h->overfetch == 0) h->overfetch_boost = 4;
        h->clean_batches = 0;
      }
      return;
    }
    if (!failing) h->f12_strikes = 0;
    if (h->overfetch > 0) return;
    if (failing) {
      // (three quarters of the batch failing: the level is far off — two steps)
      h->f12_level = std::min(kF12Levels - 1, h->f12_level + (4 * n_failed >= 3 * n_queries && h->f12_lowered == 0 ? 2 : 1));
      if (h->f12_lowered > 0 && h->f12_hold < 4096) h->f12_hold *= 2;  // the last step down was a mistake: try again later
      h->f12_lowered = 0;
      h->f12_clean = 0;
    } else if (n_failed == 0) {
      if (h->f12_level > 0 && ++h->f12_clean >= h->f12_hold) {
        --h->f12_level;
        h->f12_clean = 0;
        h->f12_lowered = 4;
      }
    } else {
      h->f12_clean = 0;
    }
    return;
  }
  if (h->overfetch > 0) return;  // the caller fixed K'
  if (4 * n_failed > n_queries) {
    if (h->overfetch_boost < 4) h->overfetch_boost *= 2;
    h->clean_batches = 0;
  } else if (n_failed == 0) {
    if (h->overfetch_boost > 1 && ++h->clean_batches >= 16) {
      h->overfetch_boost /= 2;
      h->clean_batches = 0;
    }
  } else {
    h->clean_batches = 0;
  }
}

// dense exact path for the listed batch-local query slots of workspace w (runs on h->stream, synchronous)
int run_exact(anr_index *h, Workspace &w, const std::vector<int> &slots) {
  const int64_t ld = round_up(h->ntotal, 32);
  if (!h->xdense || h->xdense_ld < ld) {
    dev_free(h->xdense);
    h->xdense_ld = ld;
    ANR_TRY(dev_alloc(&h->xdense, 4 * ld, false));
  }
  hipStream_t st = h->stream;
  for (size_t b = 0; b < slots.size(); b += 4) {
    const int nf = (int)std::min<size_t>(4, slots.size() - b);
    ExactParams ep{};
    ep.x32 = h->x32;
    ep.q32 = w.q32;
    ep.dim = h->dim;
    ep.dimp = h->dimp;
    ep.metric = h->metric;
    ep.n_rows = h->ntotal;
    ep.nf = nf;
    for (int f = 0; f < nf; ++f) ep.qidx[f] = slots[b + f];
    ep.dense = h->xdense;
    ep.ld = h->xdense_ld;
    int64_t grid = ceil_div(h->ntotal, 4);
    if (grid > (int64_t)h->n_cu * 16) grid = (int64_t)h->n_cu * 16;
    hipLaunchKernelGGL(k_exact_dense, dim3((unsigned)grid), dim3(256), 0, st, ep);
    SelParams sp{};
    sp.dense = h->xdense;
    sp.dense_ld = h->xdense_ld;
    sp.n = h->ntotal;
    sp.row0 = 0;
    sp.row_tile_stride = 1;
    sp.negate = h->metric == ANR_METRIC_L2;
    sp.M = w.k;
    sp.out_rank = w.sel_rank;
    sp.out_row = w.sel_row;
    sp.out_m = w.sel_m;
    ANR_TRY(launch_select(nf, sp, st));
    EmitParams mp{};
    mp.rank = w.sel_rank;
    mp.row = w.sel_row;
    mp.m = w.sel_m;
    mp.nf = nf;
    for (int f = 0; f < nf; ++f) {
      mp.slot[f] = f;
      mp.outq[f] = w.out_off + slots[b + f];
    }
    mp.k = w.k;
    mp.metric = h->metric;
    mp.D = w.D;
    mp.I = w.I;
    mp.id_offset = h->id_offset;
    hipLaunchKernelGGL(k_emit, dim3(nf), dim3(256), 0, st, mp);
    ANR_HIP(hipGetLastError());
  }
  ANR_HIP(hipStreamSynchronize(st));
  return ANR_OK;
}

// Recovery of the queries whose certificate failed (threshold-gated scan), in two halves so that the host never waits for
// it inside the pipeline: launch_recovery() only ENQUEUES (on the index's own stream, from pinned buffers; everything it
// decides on is in the pinned status block the batch's last kernel wrote), finish_recovery() waits at the retire.  A batch
// that is complete while later ones are still being enqueued gets its recovery launched by enqueue_batch — until round 4
// the retire did it all synchronously (two read-backs, three stream synchronisations) and a batch with ONE failed query
// cost the pipeline as much as a batch of 64.
//   * theta (below which no row of the true top-k can score) at or above the threshold EVERY workgroup emitted at: the lists
//     already hold every row that matters — their entries >= theta re-scored exactly (k_rescore_lists), a select, k_emit;
//   * otherwise a second scan with the fixed per-query threshold theta, then the same chain over its lists;
//   * no theta (fewer than k results) or a list overflow: the dense exact path at the retire.
int launch_recovery(anr_index *h, Workspace &w, const std::vector<int> &slots) {
  hipStream_t st = h->stream;
  const float inf = __builtin_inff();
  const float *theta = reinterpret_cast<const float *>(w.cnt_host + 3 * kQB);
  const float *emitted = reinterpret_cast<const float *>(w.cnt_host + 6 * kQB);
  w.rec_lists.clear();
  w.rec_scan.clear();
  w.rec_dense.clear();
  for (int q : slots) {
    if (!(theta[q] > -inf)) w.rec_dense.push_back(q);
    else if (!w.cnt_host[kQB + q] && theta[q] >= emitted[q]) w.rec_lists.push_back(q);
    else w.rec_scan.push_back(q);
  }
  auto chain = [&](const std::vector<int> &run, int which, int grid, bool filter) -> int {
    int *pin = w.qslots_pin + which * kQB;
    for (size_t i = 0; i < run.size(); ++i) pin[i] = run[i];
    ANR_HIP(hipMemcpyAsync(w.qslots + which * kQB, pin, run.size() * sizeof(int), hipMemcpyHostToDevice, st));
    const int *qs = w.qslots + which * kQB;
    RescoreListsParams rl{h->x32, w.q32, h->dim, h->dimp, h->metric, w.cand, w.cntb, grid, (unsigned)h->cand_cap, qs,
                          filter ? reinterpret_cast<const float *>(w.ncand + 3 * kQB) : nullptr};
    hipLaunchKernelGGL(k_rescore_lists, dim3((unsigned)run.size(), 4), dim3(1024), 0, st, rl);
    SelParams sp{};
    sp.cand = w.cand;
    sp.cntb = w.cntb;
    sp.G = grid;
    sp.capb = (unsigned)h->cand_cap;
    sp.M = w.k;
    sp.negate = h->metric == ANR_METRIC_L2;  // lists hold -distance: flip back on output
    sp.out_rank = w.sel_rank;
    sp.out_row = w.sel_row;
    sp.out_m = w.sel_m;
    sp.overflow = w.ncand + kQB;
    sp.qslots = qs;
    ANR_TRY(launch_select((int)run.size(), sp, st));
    EmitParams mp{};
    mp.rank = w.sel_rank;
    mp.row = w.sel_row;
    mp.m = w.sel_m;
    mp.nf = (int)run.size();
    mp.qslots = qs;
    mp.out_off = w.out_off;
    mp.k = w.k;
    mp.metric = h->metric;
    mp.D = w.D;
    mp.I = w.I;
    mp.id_offset = h->id_offset;
    mp.overflow = w.ncand + kQB;
    mp.status_host = w.cnt_dev;  // the slot's overflow flag lands in the pinned status block: no read-back
    hipLaunchKernelGGL(k_emit, dim3(mp.nf), dim3(256), 0, st, mp);
    ANR_HIP(hipGetLastError());
    return ANR_OK;
  };
  if (!w.rec_lists.empty()) {
    h->stats.n_from_lists += (int64_t)w.rec_lists.size();
    ANR_TRY(chain(w.rec_lists, 0, w.scan_grid, true));
  }
  if (!w.rec_scan.empty()) {
    for (int i = 0; i < kQB * kLadder; ++i) w.lad_pin[i] = inf;
    for (int q : w.rec_scan)
      for (int j = 0; j < kLadder; ++j) w.lad_pin[(size_t)q * kLadder + j] = theta[q];
    ANR_TRY(ensure_cand(h, w));
    ANR_HIP(hipMemcpyAsync(w.ladder, w.lad_pin, (size_t)kQB * kLadder * sizeof(float), hipMemcpyHostToDevice, st));
    ScanParams sc{};
    sc.x16 = reinterpret_cast<const uint4 *>(h->x16);
    sc.q16 = reinterpret_cast<const uint4 *>(w.q16);
    sc.kb = h->kb;
    sc.n_rows = h->ntotal;
    sc.rowbias = h->rowbias;
    sc.tile0 = 0;
    sc.tile_stride = 1;
    sc.n_tiles = ceil_div(h->ntotal, kTileRows);
    sc.ladder = w.ladder;
    sc.lvl0 = 0;
    sc.cntb = w.cntb;
    sc.cand = w.cand;
    sc.capb = (unsigned)h->cand_cap;
    sc.kprime = 0x7fffffffu;  // levels never advance: the threshold stays theta
    int grid = 0;
    ANR_TRY(launch_scan<false>(h, sc, st, h->n_cu, &grid));
    ANR_TRY(chain(w.rec_scan, 1, grid, false));
  }
  ANR_HIP(hipEventRecord(w.ev_rec, st));
  w.rec_launched = true;
  return ANR_OK;
}

// the recovery's results are in place; returns the queries that still need the dense exact path
int finish_recovery(anr_index *h, Workspace &w, std::vector<int> *dense) {
  (void)h;
  ANR_HIP(hipEventSynchronize(w.ev_rec));
  *dense = w.rec_dense;
  for (const std::vector<int> *run : {&w.rec_lists, &w.rec_scan})
    for (int q : *run)
      if (w.cnt_host[kQB + q]) dense->push_back(q);  // a list overflowed
  return ANR_OK;
}

// wait for a workspace's batch, fold its statistics, run the exact path where the certificate failed.
// The batch stays "in flight" until its recovery passes have succeeded: if one fails the error is returned and a
// later retire() of the same workspace tries again (statistics are folded once), so a batch is never reported
// final while some of its D / I rows are not.
int retire(anr_index *h, Workspace &w) {
  if (!w.in_flight) return ANR_OK;
  ANR_HIP(hipEventSynchronize(w.ev_done));
  std::vector<int> fallback;
  if (w.exact_all) {
    for (int q = 0; q < w.nq; ++q) fallback.push_back(q);
  } else {
    for (int q = 0; q < w.nq; ++q)
      if (w.cnt_host[2 * kQB + q]) fallback.push_back(q);
  }
  if (!w.folded) {
    w.folded = true;
    if (!w.exact_all && w.sparse) {
      // one record of the batch log (anr_index_batch_log): host times and the device stamps the last kernel left in the
      // status block, converted to nanoseconds of the device's constant clock
      if (h->batch_log.empty()) h->batch_log.assign((size_t)kBatchLogCap * kBatchLogFields, 0);
      int64_t *rec = h->batch_log.data() + (size_t)(h->batch_log_n % kBatchLogCap) * kBatchLogFields;
      const unsigned long long *stp = reinterpret_cast<const unsigned long long *>(w.cnt_host + 4 * kQB);
      rec[0] = w.seq;
      rec[1] = w.t_enq0;
      rec[2] = w.t_enq1;
      rec[3] = host_ns();
      for (int i = 0; i < kStamps; ++i) rec[4 + i] = (int64_t)((double)stp[i] * 1e6 / (double)h->clock_khz);
      rec[4 + kStamps] = (w.shadow ? 1 : 0) | ((int64_t)w.nq << 8);
      rec[5 + kStamps] = 0;
      w.log_slot = h->batch_log_n % kBatchLogCap;
      h->batch_log_n += 1;
    }
    if (!w.exact_all) {
      if (w.timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, w.ev_t0, w.ev_t1) == hipSuccess) h->stats.scan_ms += ms;
      }
      h->stats.scan_bytes += w.scan_bytes;
      h->stats.overfetch = w.M;
      h->stats.sample_rows = (int)w.sample_rows;
      if (w.sparse)
        for (int q = 0; q < w.nq; ++q) {
          h->stats.n_candidates += w.cnt_host[q];
          if (w.cnt_host[kQB + q]) h->stats.n_overflow += 1;
        }
      adapt_overfetch(h, w.nq, (int)fallback.size(), w.sparse && w.f12, w.f12_level);
    }
    h->stats.n_fallback += (int64_t)fallback.size();
  }
  if (!fallback.empty()) {
    const int64_t t_rec0 = host_ns();
    std::vector<int> dense;
    if (w.sparse && !w.exact_all) {
      if (!w.rec_launched) ANR_TRY(launch_recovery(h, w, fallback));
      ANR_TRY(finish_recovery(h, w, &dense));
    } else {
      dense = fallback;
    }
    if (!dense.empty()) {
      h->stats.n_dense_exact += (int64_t)dense.size();
      ANR_TRY(run_exact(h, w, dense));
    }
    if (w.log_slot >= 0 && !h->batch_log.empty())  // host time of the recovery passes (synchronous, on the index's own stream)
      h->batch_log[(size_t)w.log_slot * kBatchLogFields + 5 + kStamps] += host_ns() - t_rec0;
  }
  w.log_slot = -1;
  w.rec_launched = false;
  w.in_flight = false;
  return ANR_OK;
}

int drain(anr_index *h) {
  if (!h->ws_ready) return ANR_OK;
  // oldest first
  for (int i = 0; i < kWorkspaces; ++i) ANR_TRY(retire(h, h->ws[(h->next_ws + i) % kWorkspaces]));
  return ANR_OK;
}

// enqueue one batch of nq <= 64 device-resident queries; returns without waiting
int enqueue_batch(anr_index *h, const float *q_dev, int nq, int k, int64_t out_off, float *D_dev, int64_t *I_dev,
                  hipStream_t user, bool host_out = false, const float *h2d_src = nullptr) {
  const int ws_index = h->next_ws;
  Workspace &w = h->ws[ws_index];
  h->next_ws = (h->next_ws + 1) % kWorkspaces;
  const int64_t t_enq0 = host_ns();
  ANR_TRY(retire(h, w));  // back-pressure: the workspace's previous batch must be complete
  // another batch still in flight = a pipelined caller: this batch's side kernels take their shadow-sized forms, which
  // run BESIDE the resident workgroups of the scan in front of them instead of after it (ANR_OPT_SHADOW)
  bool others_in_flight = false;
  for (int i = 0; i < kWorkspaces; ++i) others_in_flight |= (i != ws_index && h->ws[i].in_flight);
  // a batch that has completed meanwhile and holds failed certificates: its recovery starts NOW, beside the batches being
  // enqueued, instead of when its workspace comes round again (retire() then only waits for it)
  for (int i = 0; i < kWorkspaces; ++i) {
    Workspace &o = h->ws[i];
    if (i == ws_index || !o.in_flight || o.rec_launched || !o.sparse || o.exact_all) continue;
    if (hipEventQuery(o.ev_done) != hipSuccess) continue;
    std::vector<int> fb;
    for (int q = 0; q < o.nq; ++q)
      if (o.cnt_host[2 * kQB + q]) fb.push_back(q);
    if (!fb.empty()) ANR_TRY(launch_recovery(h, o, fb));
  }
  (void)hipGetLastError();  // (hipEventQuery reports hipErrorNotReady through the sticky error too)
  w.t_enq0 = t_enq0;
  w.seq = h->batch_seq++;
  w.shadow = false;
  w.folded = false;
  w.nq = nq;
  w.k = k;
  w.out_off = out_off;
  w.D = D_dev;
  w.I = I_dev;
  w.sparse = false;
  w.timed = h->timing != 0;
  w.scan_bytes = 0;
  w.sample_rows = 0;
  w.exact_all = h->force_exact != 0 || h->f16_unusable;

  // Streams.  ANR_OPT_SCHEDULE 1 (default): ROLE streams — every batch's query preparation, threshold sample and ladder go
  // to the PRE stream, every main scan to the MAIN stream, every select / re-score / finalize to the POST stream, tied by
  // two events per batch.  The scans then run strictly one after the other in submission order, and the side kernels of the
  // neighbouring batches — shadow-sized, see k_sample — run beside them on their own hardware queues: a side kernel that
  // sat in the same queue as a scan waiting for CUs (or behind another batch's completion marker) used to wait with it, which
  // put ~45 us of side kernels between two 300-us scans at the 8-GPU shard size.  0: one stream per in-flight batch
  // (ANR_OPT_STREAMS of them), the scheme of rounds 1-3.
  const bool roles = h->schedule == 1 && h->n_streams == kWorkspaces;
  hipStream_t bs = roles ? h->bstream[0] : h->bstream[ws_index % h->n_streams];  // pre
  hipStream_t ms = roles ? h->bstream[1] : bs;                                    // main scan
  hipStream_t ps = roles ? h->bstream[2] : bs;                                    // post
  // the queries are ready once everything already enqueued on the caller's stream has run
  // (an idle caller stream has nothing to wait for: skip the cross-queue dependency, which costs the
  // command processor several microseconds per batch)
  if (h2d_src) {
    // queries staged in pinned host memory by the caller: one DMA on the batch's own stream, no cross-queue wait
    ANR_HIP(hipMemcpyAsync(const_cast<float *>(q_dev), h2d_src, (size_t)nq * h->dim * sizeof(float), hipMemcpyHostToDevice, bs));
  } else if (user != bs && hipStreamQuery(user) != hipSuccess) {
    ANR_HIP(hipEventRecord(w.ev_in, user));
    ANR_HIP(hipStreamWaitEvent(bs, w.ev_in, 0));
  }
  (void)hipGetLastError();  // hipStreamQuery reports hipErrorNotReady through the sticky error too

  PrepQParams qp{};
  qp.qin = q_dev;
  qp.nq = nq;
  qp.dim = h->dim;
  qp.dimp = h->dimp;
  qp.kb = h->kb;
  qp.normalize = h->normalize;
  qp.q32 = w.q32;
  qp.q16 = w.q16;
  qp.qstat = w.qstat;
  qp.stamps = w.stamps;
  hipLaunchKernelGGL(k_prepq, dim3(kQB), dim3(256), 0, bs, qp);

  if (w.exact_all) {
    ANR_HIP(hipEventRecord(w.ev_done, bs));
    w.in_flight = true;  // only now: a failure above leaves no half-enqueued batch to retire
    if (h->stream_wait) ANR_HIP(hipStreamWaitEvent(user, w.ev_done, 0));
    return ANR_OK;
  }

  // the image the streaming pass reads (ANR_OPT_SCAN_BITS; 0 = 12 bits from kAuto12Rows rows on: below, a batch is
  // mostly its side kernels).  The sample and every recovery scan read x16; the certificate of a 12-bit batch takes the
  // statistics of the coarser image (pp.xstat / fp.xstat below).
  bool f12 = false;
  if (!h->f12_suspended && (h->scan_bits == 12 || (h->scan_bits == 0 && h->ntotal >= kAuto12Rows))) {
    if (!h->x12 && !h->x12_no_memory && build_x12(h) != ANR_OK) h->x12_no_memory = true;  // (stays on the f16 image)
    f12 = h->x12 != nullptr;
  }
  const int64_t n_tiles = ceil_div(h->ntotal, kTileRows);
  const int64_t full_tiles = h->ntotal / kTileRows;
  // threshold sample: ~1/64 of the rows, between 4K and 16K (more rows -> tighter first threshold)
  // sample size: the sample pass costs ~S rows of streaming, the candidates it lets through cost
  // ~64 * 12 * N / S list appends (about as much per append as per streamed row); the sum is least near
  // S = 27 sqrt(N)
  const int64_t auto_sample = round_up(
      std::min<int64_t>(262144, std::max<int64_t>(4096, (int64_t)(27.0 * std::sqrt((double)h->ntotal)))), 1024);
  const int64_t sample_want = (h->sample_rows > 0 ? h->sample_rows : auto_sample) / kTileRows;
  // tile maxima: the M-th largest is backed by M distinct rows -> at least 2 M sample tiles; the threshold-gated pipeline
  // needs a corpus of eight samples.  A 12-bit batch's larger K' can push a mid-sized corpus below that: it then runs as an
  // f16 batch (the small-corpus path scores every row from x16) with the f16 K'.
  int M = auto_overfetch(h, k, f12);
  if (f12 && full_tiles < 8 * std::max<int64_t>(sample_want, 2 * (int64_t)M)) {
    f12 = false;
    M = auto_overfetch(h, k, false);
  }
  w.M = M;
  int64_t sample_tiles = sample_want;
  if (sample_tiles < 2 * M) sample_tiles = 2 * M;
  const bool sparse = full_tiles >= 8 * sample_tiles;
  w.sparse = sparse;
  // k_post runs ONE workgroup per query: a batch of a few queries leaves its select and re-score to a handful of CUs
  // (batch 1, k = 100, 20 k x 768: 92 us fused vs 84 us as three grid-wide launches); from 5 queries up it wins
  const bool shadow = sparse && M <= kShadowSel && h->n_cu <= kShadowLists &&
                      (h->shadow == 2 || (h->shadow == 1 && others_in_flight));
  w.shadow = shadow;
  const bool fused_post = !shadow && (h->fused_post == 1 ? nq > 4 : h->fused_post != 0);
  const int side_grid = h->n_cu;
  // (Leaving 8-32 CUs out of the main scan's persistent grid so that the next batch's side kernels run beside it was
  // tried at the 1.25 M-row shard size: wall 0.347 -> 0.345-0.350 ms, and -2 % at 10 M rows.  The scan's 12-wave
  // workgroups fill a CU's register file, so the side kernels of batch i + 1 run at the tail of scan i either way.)
  const int scan_grid_max = h->n_cu;

  ScanParams sc{};
  sc.x16 = reinterpret_cast<const uint4 *>(h->x16);
  sc.q16 = reinterpret_cast<const uint4 *>(w.q16);
  sc.kb = h->kb;
  sc.n_rows = h->ntotal;
  sc.rowbias = h->rowbias;

  SelParams sp{};
  sp.M = M;
  sp.out_rank = w.sel_rank;
  sp.out_row = w.sel_row;
  sp.out_m = w.sel_m;
  sp.stamps = w.stamps;

  if (!sparse) {
    // small corpus: dense scores of every row, select the candidates from them
    ANR_TRY(ensure_dense(w, n_tiles * kTileRows));
    sc.tile0 = 0;
    sc.tile_stride = 1;
    sc.n_tiles = n_tiles;
    sc.dense = w.dense;
    sc.dense_ld = w.dense_ld;
    if (roles) {
      ANR_HIP(hipEventRecord(w.ev_pre, bs));
      ANR_HIP(hipStreamWaitEvent(ms, w.ev_pre, 0));
    }
    if (w.timed) ANR_HIP(hipEventRecord(w.ev_t0, ms));
    ANR_TRY(launch_scan<true>(h, sc, ms, scan_grid_max));
    if (w.timed) ANR_HIP(hipEventRecord(w.ev_t1, ms));
    if (roles) {
      ANR_HIP(hipEventRecord(w.ev_scan, ms));
      ANR_HIP(hipStreamWaitEvent(ps, w.ev_scan, 0));
    }
    w.scan_bytes = n_tiles * kTileRows * (int64_t)h->dimp * 2;
    sp.dense = w.dense;
    sp.dense_ld = w.dense_ld;
    sp.n = h->ntotal;
    sp.row0 = 0;
    sp.row_tile_stride = 1;
    if (!fused_post) ANR_TRY(launch_select(nq, sp, ps));
  } else {
    ANR_TRY(ensure_dense(w, sample_tiles * kTileRows));
    ANR_TRY(ensure_cand(h, w));
    w.sample_rows = sample_tiles * kTileRows;
    // pre: strided sample -> ladder of valid thresholds (also clears the level counters)
    sc.tile0 = 0;
    sc.tile_stride = full_tiles / sample_tiles;
    sc.n_tiles = sample_tiles;
    sc.dense = w.dense;
    sc.dense_ld = w.dense_ld;
    sc.groupmax = 1;  // one value per 32-row tile: the K'-th largest tile maximum is a valid threshold
    if (shadow) {
      SampleParams sm{};
      sm.x16 = sc.x16;
      sm.q16 = sc.q16;
      sm.kb = sc.kb;
      sm.tile_stride = sc.tile_stride;
      sm.n_tiles = sc.n_tiles;
      sm.rowbias = sc.rowbias;
      sm.dense = sc.dense;
      sm.dense_ld = sc.dense_ld;
      sm.stamps = w.stamps;
      ANR_TRY(launch_sample_shadow(sm, bs, h->n_cu));
    } else {
      // (fewer, fuller workgroups for a pipelined caller's sample — 8 or 12 waves instead of 4, each filling 96 KiB of LDS
      // with the query operand once — measured the same per-batch time at 1.25 M rows: 0.3142 / 0.3147 / 0.3162 ms)
      ANR_TRY(launch_scan<true>(h, sc, bs, side_grid));
    }
    sc.groupmax = 0;
    // start level of the scan: the highest one whose sample rank still predicts >= 4 K' corpus rows above it
    int lvl0 = 0;
    {
      const int64_t ratio = full_tiles / sample_tiles;
      const int64_t need = std::max<int64_t>(12, ceil_div((int64_t)4 * M, ratio));
      for (int j = 1; j < kLadder - 1; ++j)
        if ((M >> j) >= need) lvl0 = j;
    }
    SelParams ss = sp;
    ss.dense = w.dense;
    ss.dense_ld = w.dense_ld;
    ss.n = sample_tiles;
    ss.row0 = 0;
    ss.row_tile_stride = 1;
    ss.ladder = w.ladder;
    ss.lvl_init = w.lvlmax;
    ss.lvl_init_value = lvl0;
    ss.live_q = nq;  // slots beyond the batch stay inert in the scan
    if (shadow) ANR_TRY(launch_select_shadow(kQB, ss, bs));
    else ANR_TRY(launch_select(kQB, ss, bs));
    // scan
    sc.tile0 = 0;
    sc.tile_stride = 1;
    sc.n_tiles = n_tiles;
    sc.dense = nullptr;
    sc.ladder = w.ladder;
    sc.lvl0 = lvl0;
    sc.lvlmax = w.lvlmax;
    sc.cntb = w.cntb;
    sc.cand = w.cand;
    sc.capb = (unsigned)h->cand_cap;
    sc.kprime = (unsigned)M;
    sc.stamps = w.stamps;
    int scan_grid = 0;
    if (roles) {
      ANR_HIP(hipEventRecord(w.ev_pre, bs));
      ANR_HIP(hipStreamWaitEvent(ms, w.ev_pre, 0));
    }
    if (w.timed) ANR_HIP(hipEventRecord(w.ev_t0, ms));
    w.f12 = f12;
    w.f12_level = h->f12_level;
    if (f12) {
      sc.x12 = h->x12;
      ANR_TRY((launch_scan<false, true>(h, sc, ms, scan_grid_max, &scan_grid)));
    } else {
      ANR_TRY(launch_scan<false>(h, sc, ms, scan_grid_max, &scan_grid));
    }
    w.scan_grid = scan_grid;
    if (w.timed) ANR_HIP(hipEventRecord(w.ev_t1, ms));
    if (roles) {
      ANR_HIP(hipEventRecord(w.ev_scan, ms));
      ANR_HIP(hipStreamWaitEvent(ps, w.ev_scan, 0));
    }
    w.scan_bytes = n_tiles * kTileRows * (int64_t)h->dimp * (f12 ? 3 : 4) / 2;
    // post
    sp.cand = w.cand;
    sp.cntb = w.cntb;
    sp.G = scan_grid;
    sp.capb = (unsigned)h->cand_cap;
    sp.overflow = w.ncand + kQB;
    sp.ncand = w.ncand;
    if (shadow) ANR_TRY(launch_select_shadow(nq, sp, ps));
    else if (!fused_post) ANR_TRY(launch_select(nq, sp, ps));
  }

  if (fused_post) {
    // select + exact re-score + finalize in one launch per batch (k_post)
    PostParams pp{};
    pp.x32 = h->x32;
    pp.q32 = w.q32;
    pp.dim = h->dim;
    pp.dimp = h->dimp;
    pp.metric = h->metric;
    pp.qstat = w.qstat;
    pp.xstat = (sparse && w.f12) ? h->xstat12 : h->xstat;
    pp.overflow = sparse ? w.ncand + kQB : nullptr;
    pp.ncand = sparse ? w.ncand : nullptr;
    pp.n_rows = h->ntotal;
    pp.Mreq = M;
    pp.k = k;
    pp.nq = nq;
    pp.out_off = out_off;
    pp.D = D_dev;
    pp.I = I_dev;
    pp.flags = reinterpret_cast<int *>(w.ncand + 2 * kQB);
    pp.theta = reinterpret_cast<float *>(w.ncand + 3 * kQB);
    pp.id_offset = h->id_offset;
    pp.status_host = w.cnt_dev;
    pp.host_out = host_out ? 1 : 0;
    pp.stamps = sparse ? w.stamps : nullptr;
    pp.ladder = sparse ? w.ladder : nullptr;
    pp.lvlmax = sparse ? w.lvlmax : nullptr;
    ANR_TRY(launch_post(nq, sp, pp, ps));
    ANR_HIP(hipEventRecord(w.ev_done, ps));
    w.in_flight = true;
    if (h->stream_wait) ANR_HIP(hipStreamWaitEvent(user, w.ev_done, 0));
    w.t_enq1 = host_ns();
    return ANR_OK;
  }

  RescoreParams rp{};
  rp.x32 = h->x32;
  rp.q32 = w.q32;
  rp.dim = h->dim;
  rp.dimp = h->dimp;
  rp.metric = h->metric;
  rp.sel_row = w.sel_row;
  rp.sel_m = w.sel_m;
  rp.exact = w.exact;
  rp.M = M;
  if (shadow) {
    // one four-wave workgroup per CU (at least one wave per query), RB rows in flight per wave
    const unsigned grid = (unsigned)std::max(h->n_cu, (nq + 3) / 4);
    const bool v4 = (h->dim & 3) == 0;
    if (h->metric == ANR_METRIC_L2) {
      if (v4) hipLaunchKernelGGL((k_rescore_shadow<4, true, true>), dim3(grid), dim3(256), 0, ps, rp, nq);
      else hipLaunchKernelGGL((k_rescore_shadow<3, true, false>), dim3(grid), dim3(256), 0, ps, rp, nq);
    } else {
      if (v4) hipLaunchKernelGGL((k_rescore_shadow<4, false, true>), dim3(grid), dim3(256), 0, ps, rp, nq);
      else hipLaunchKernelGGL((k_rescore_shadow<3, false, false>), dim3(grid), dim3(256), 0, ps, rp, nq);
    }
  } else {
    hipLaunchKernelGGL(k_rescore, dim3((unsigned)ceil_div((int64_t)nq * M, 4)), dim3(256), 0, ps, rp);
  }

  FinalParams fp{};
  fp.exact = w.exact;
  fp.approx = w.sel_rank;
  fp.sel_row = w.sel_row;
  fp.sel_m = w.sel_m;
  fp.overflow = sparse ? w.ncand + kQB : nullptr;
  fp.qstat = w.qstat;
  fp.xstat = (sparse && w.f12) ? h->xstat12 : h->xstat;
  fp.metric = h->metric;
  fp.dimp = h->dimp;
  fp.n_rows = h->ntotal;
  fp.M = M;
  fp.k = k;
  fp.nq = nq;
  fp.out_off = out_off;
  fp.D = D_dev;
  fp.I = I_dev;
  fp.flags = reinterpret_cast<int *>(w.ncand + 2 * kQB);
  fp.theta = reinterpret_cast<float *>(w.ncand + 3 * kQB);
  fp.id_offset = h->id_offset;
  fp.ncand = sparse ? w.ncand : nullptr;
  fp.status_host = w.cnt_dev;
  fp.host_out = host_out ? 1 : 0;
  fp.stamps = sparse ? w.stamps : nullptr;
  fp.ladder = sparse ? w.ladder : nullptr;
  fp.lvlmax = sparse ? w.lvlmax : nullptr;
  hipLaunchKernelGGL(k_finalize, dim3(nq), dim3(256), 0, ps, fp);
  ANR_HIP(hipGetLastError());
  ANR_HIP(hipEventRecord(w.ev_done, ps));
  w.in_flight = true;
  if (h->stream_wait) ANR_HIP(hipStreamWaitEvent(user, w.ev_done, 0));
  w.t_enq1 = host_ns();
  return ANR_OK;
}

int refresh_xstat(anr_index *h) {
  if (!h->xstat_dirty) return ANR_OK;
  unsigned host[4] = {0, 0, 0, 0};
  ANR_HIP(hipStreamSynchronize(h->stream));
  ANR_HIP(hipMemcpy(host, h->xstat, sizeof host, hipMemcpyDeviceToHost));
  h->f16_unusable = host[2] != 0;
  // (norms are floats >= 0: their bit patterns order like the values)
  const unsigned pair[2] = {host[0], host[1] > host[3] ? host[1] : host[3]};
  ANR_HIP(hipMemcpy(h->xstat12, pair, sizeof pair, hipMemcpyHostToDevice));
  h->xstat_dirty = false;
  return ANR_OK;
}

int write_empty(const anr_index *h, int64_t nq, int32_t k, float *D, int64_t *I, bool on_host) {
  // faiss returns -1 ids and the neutral score for an empty index
  std::vector<float> dd((size_t)nq * k, h->metric == 0 ? -3.402823466e+38f : 3.402823466e+38f);
  std::vector<int64_t> ii((size_t)nq * k, -1);
  if (on_host) {
    memcpy(D, dd.data(), dd.size() * sizeof(float));
    memcpy(I, ii.data(), ii.size() * sizeof(int64_t));
  } else {
    ANR_HIP(hipMemcpy(D, dd.data(), dd.size() * sizeof(float), hipMemcpyHostToDevice));
    ANR_HIP(hipMemcpy(I, ii.data(), ii.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  }
  return ANR_OK;
}

constexpr int kMaxLargeK = 1 << 20;

int check_search_args(const anr_index *h, const void *q, int64_t nq, int32_t k, const void *D, const void *I,
                      bool allow_large_k) {
  if (!h || !q || !D || !I) return fail(ANR_EINVAL, "null argument");
  if (nq < 0 || k <= 0) return fail(ANR_EINVAL, "nq must be >= 0 and k > 0");
  if (k > (allow_large_k ? kMaxLargeK : kMaxSel))
    return fail(ANR_EINVAL, "k = %d exceeds the supported maximum %d%s", k, allow_large_k ? kMaxLargeK : kMaxSel,
                allow_large_k ? "" : " of asynchronous searches (use the synchronous call)");
  return ANR_OK;
}

// k beyond the select window: exact scores of every row (k_exact_dense, 4 queries per pass), device radix sort, first k
// pairs (largek.hip).  Synchronous; D / I are device memory here.
int search_large_k(anr_index *h, const float *q, bool q_on_host, int64_t nq, int k, float *Dd, int64_t *Id) {
  hipStream_t st = h->stream;
  Workspace &w = h->ws[0];
  const int64_t ld = round_up(h->ntotal, 32);
  if (!h->xdense || h->xdense_ld < ld) {
    dev_free(h->xdense);
    h->xdense_ld = ld;
    ANR_TRY(dev_alloc(&h->xdense, 4 * ld, false));
  }
  for (int64_t q0 = 0; q0 < nq; q0 += kQB) {
    const int nb = (int)std::min<int64_t>(kQB, nq - q0);
    ANR_HIP(hipMemcpyAsync(w.qstage, q + q0 * h->dim, (size_t)nb * h->dim * sizeof(float),
                           q_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    PrepQParams qp{};
    qp.qin = w.qstage;
    qp.nq = nb;
    qp.dim = h->dim;
    qp.dimp = h->dimp;
    qp.kb = h->kb;
    qp.normalize = h->normalize;
    qp.q32 = w.q32;
    qp.q16 = w.q16;
    qp.qstat = w.qstat;
    hipLaunchKernelGGL(k_prepq, dim3(kQB), dim3(256), 0, st, qp);
    for (int b = 0; b < nb; b += 4) {
      const int nf = std::min(4, nb - b);
      ExactParams ep{};
      ep.x32 = h->x32;
      ep.q32 = w.q32;
      ep.dim = h->dim;
      ep.dimp = h->dimp;
      ep.metric = h->metric;
      ep.n_rows = h->ntotal;
      ep.nf = nf;
      for (int f = 0; f < nf; ++f) ep.qidx[f] = b + f;
      ep.dense = h->xdense;
      ep.ld = h->xdense_ld;
      int64_t grid = ceil_div(h->ntotal, 4);
      if (grid > (int64_t)h->n_cu * 16) grid = (int64_t)h->n_cu * 16;
      hipLaunchKernelGGL(k_exact_dense, dim3((unsigned)grid), dim3(256), 0, st, ep);
      for (int f = 0; f < nf; ++f)
        ANR_TRY(sort_topk(h->xdense + (int64_t)f * h->xdense_ld, h->ntotal, k, h->metric == ANR_METRIC_IP, h->id_offset,
                          Dd + (q0 + b + f) * (int64_t)k, Id + (q0 + b + f) * (int64_t)k, &h->largek, st));
    }
    ANR_HIP(hipStreamSynchronize(st));  // qstage / q32 are reused by the next block of queries
  }
  h->stats.n_dense_exact += nq;
  return ANR_OK;
}

// Tiny corpus, host buffers, a handful of queries: ONE kernel launch, completion by a word in pinned memory
// (tiny_kernels.hpp).  Exact by construction (f32 rows, f64 accumulation): no certificate, no fallback.
// workgroups of the single-launch search: enough that every CU's slice is short, few enough for the last one's merge
int64_t tiny_workgroups(const anr_index *h, int32_t k) {
  // k <= kTinySortK: the merge prunes by the k-th largest list head and hardly notices the number of lists, so every CU
  // gets a (short) slice; beyond, the merge sorts all n_wg * k keys in registers: at most kTinyMaxMerge of them
  (void)h;
  const int64_t merge_cap = k <= kTinySortK ? kTinyMaxPrune : kTinyMaxMerge;
  return std::max<int64_t>(1, std::min<int64_t>(kTinyMaxWG, merge_cap / k));
}

// Where the single launch beats the five-kernel pipeline for batches of <= 4 queries (tools/tiny_perf.py, k = 10):
// 20 k x 768 41 vs 83 us, 50 k x 768 58 vs 86, 100 k x 768 85 vs 93; from ~120 k x 768 the pipeline's f16 scan wins.
// Its cost grows with k (fewer, longer partial lists: 2048 / k workgroups, each streaming more f32 rows at one CU's
// ~25 GB/s) while the pipeline's barely does.  Measured, batch 1, bare C ABI, this path vs the pipeline: k = 10: 10 k x 384
// 33 vs 58 us, 20 k x 768 39 vs 79, 100 k x 768 88 vs 94; k = 50: 10 k x 768 63 vs 72, 20 k x 768 68 vs 83 (1.5 MB per
// workgroup), 40 k x 768 level; k = 32 at 50 k x 768 (2.4 MB) 86 vs 80; k = 20 at 100 k x 768 (3 MB) 115 vs 94;
// k = 100: 10 k x 384 63 vs 60, 20 k x 768 93 vs 85.  Hence: k <= 64 and at most 1.6 MB of rows per workgroup.
bool tiny_applies(const anr_index *h, int64_t nq, int32_t k) {
  if (!h->tiny || h->force_exact || nq < 1 || nq > kTinyMaxQ || k > kTinyMaxK || h->ntotal < 1) return false;
  const int64_t rows_per_wg = ceil_div(h->ntotal, tiny_workgroups(h, k));
  if (rows_per_wg > kTinyRowsPerWG) return false;
  if (h->tiny >= 2) return true;  // forced (tests): structural limits only
  if (k > 64) return false;
  return rows_per_wg * h->dim * 4 <= ((int64_t)1600 << 10);
}

int search_tiny(anr_index *h, const float *q, int64_t nq, int32_t k, float *D, int64_t *I) {
  const size_t q_bytes = (size_t)kTinyMaxQ * h->dim * sizeof(float);
  // pinned block: raw queries | result words [kTinyMaxQ][kTinyMaxK][2] (8 bytes each, see k_tiny_search)
  const size_t r_off = round_up((int64_t)q_bytes, 16), f_off = r_off + (size_t)kTinyMaxQ * kTinyMaxK * 16;
  if (!h->tiny_pin || !h->tiny_cand || !h->tiny_ticket) {
    // first use: the buffers are committed to the handle only as a complete set — a failed allocation leaves nothing
    // behind that a later call could mistake for an initialised path (it would launch with null list pointers)
    unsigned char *pin = nullptr, *pin_dev = nullptr;
    decltype(h->tiny_cand) cand = nullptr;
    decltype(h->tiny_ticket) ticket = nullptr;
    int rc = ANR_OK;
    if (hipHostMalloc(reinterpret_cast<void **>(&pin), f_off + 64, hipHostMallocDefault) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void **>(&pin_dev), pin, 0) != hipSuccess)
      rc = fail(ANR_EHIP, "tiny search: pinned buffer allocation failed");
    static_assert(kTinyMaxPrune >= kTinyMaxMerge, "the partial lists of a query hold up to kTinyMaxPrune keys");
    if (rc == ANR_OK) rc = dev_alloc(&cand, (int64_t)kTinyMaxQ * kTinyMaxPrune, true);
    if (rc == ANR_OK) rc = dev_alloc(&ticket, kTinyMaxQ, true);
    if (rc != ANR_OK) {
      if (pin) (void)hipHostFree(pin);
      if (cand) (void)hipFree(cand);
      if (ticket) (void)hipFree(ticket);
      return rc;
    }
    memset(pin + r_off, 0, f_off - r_off);  // sequence 0 is never used: no word is valid yet
    h->tiny_pin = pin;
    h->tiny_pin_dev = pin_dev;
    h->tiny_cand = cand;
    h->tiny_ticket = ticket;
    if (getenv("ANORAG_TINY_STAMPS")) ANR_TRY(dev_alloc(&h->tiny_stamps, 16, true));
  }
  memcpy(h->tiny_pin, q, (size_t)nq * h->dim * sizeof(float));
  TinyParams tp{};
  tp.x32 = h->x32;
  tp.q_host = reinterpret_cast<const float *>(h->tiny_pin_dev);
  tp.nq = (int)nq;
  tp.dim = h->dim;
  tp.metric = h->metric;
  tp.normalize = h->normalize;
  tp.k = k;
  tp.n_rows = h->ntotal;
  // few enough workgroups that the last one's merge holds them (tiny_workgroups), enough that every CU slice is short
  int n_wg = (int)tiny_workgroups(h, k);
  n_wg = (int)std::min<int64_t>(n_wg, ceil_div(h->ntotal, 16));
  tp.rows_per_wg = (int)ceil_div(h->ntotal, n_wg);
  tp.n_wg = (int)ceil_div(h->ntotal, tp.rows_per_wg);
  if ((int64_t)tp.n_wg * k > kTinyMaxPrune || (k > kTinySortK && (int64_t)tp.n_wg * k > kTinyMaxMerge) || tp.rows_per_wg > kTinyRowsPerWG)
    return fail(ANR_EINTERNAL, "tiny search: %d workgroups x k = %d do not fit the merge", tp.n_wg, k);
  tp.cand = h->tiny_cand;
  tp.ticket = h->tiny_ticket;
  tp.R = reinterpret_cast<unsigned long long *>(h->tiny_pin_dev + r_off);
  if (++h->tiny_seq == 0) {  // wrapped: no stale word may carry a sequence number that comes round again
    memset(h->tiny_pin + r_off, 0, f_off - r_off);
    h->tiny_seq = 1;
  }
  tp.seq = h->tiny_seq;
  tp.stamps = h->tiny_stamps;
  const size_t lds = (size_t)kTinyMaxPrune * 8 + (size_t)16 * kTinyMaxK * 8 + (size_t)round_up(h->dim, 4) * sizeof(float);
  const int kc = (h->dim % 4 == 0 && h->dim <= 1024) ? (int)ceil_div(h->dim / 4, 64) : 0;
#define ANR_TINY_LAUNCH(KC)                                                                                   \
  {                                                                                                           \
    ANR_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_tiny_search<KC>), (int)lds));                 \
    hipLaunchKernelGGL(k_tiny_search<KC>, dim3((unsigned)tp.n_wg, (unsigned)nq), dim3(kTinyThreads), lds, h->stream, tp); \
  }
  switch (kc) {
    case 1: ANR_TINY_LAUNCH(1) break;
    case 2: ANR_TINY_LAUNCH(2) break;
    case 3: ANR_TINY_LAUNCH(3) break;
    case 4: ANR_TINY_LAUNCH(4) break;
    default: ANR_TINY_LAUNCH(0) break;
  }
#undef ANR_TINY_LAUNCH
  ANR_HIP(hipGetLastError());
  // completion: spin on the result words themselves — each carries this call's sequence number in its upper half (an
  // event / stream wait costs more than the whole kernel; a separate flag needed a system-scope fence on the device);
  // after ~2 s fall back to the stream so a device fault surfaces as an error
  volatile unsigned long long *R = reinterpret_cast<volatile unsigned long long *>(h->tiny_pin + r_off);
  const int64_t n_words = nq * (int64_t)k * 2;
  {
    uint64_t spins = 0;
    for (int64_t i = n_words - 1; i >= 0; --i) {  // last words first: they are written last
      while ((unsigned)(__atomic_load_n(&R[i], __ATOMIC_ACQUIRE) >> 32) != tp.seq) {
        __builtin_ia32_pause();
        if (++spins > (1ull << 28)) {
          ANR_HIP(hipStreamSynchronize(h->stream));
          if ((unsigned)(__atomic_load_n(&R[i], __ATOMIC_ACQUIRE) >> 32) != tp.seq)
            return fail(ANR_EINTERNAL, "tiny search: the kernel finished without writing its results");
          spins = 0;
        }
      }
    }
  }
  if (h->tiny_stamps && (h->tiny_seq & 127) == 0) {
    unsigned long long st[16];
    ANR_HIP(hipStreamSynchronize(h->stream));
    ANR_HIP(hipMemcpy(st, h->tiny_stamps, sizeof st, hipMemcpyDeviceToHost));
    auto us = [&](int a, int b) { return (double)((long long)st[b] - (long long)st[a]) / 100.0; };  // 100 MHz clock
    fprintf(stderr, "[tiny] wg0: query %.2f norm %.2f score %.2f rank+publish %.2f ticket %.2f | last: start+%.2f load %.2f "
                    "merge-1 %.2f merge-2 %.2f write %.2f us\n", us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(0, 8), us(8, 9), us(9, 12), us(12, 13), us(13, 10));
  }
  for (int64_t i = 0; i < nq * (int64_t)k; ++i) {
    const unsigned sbits = (unsigned)R[2 * i], row = (unsigned)R[2 * i + 1];
    memcpy(D + i, &sbits, 4);
    I[i] = row == 0xffffffffu ? -1 : (int64_t)row + h->id_offset;
  }
  h->stats.n_dense_exact += nq;
  return ANR_OK;
}

// synchronous search (host or device buffers): enqueue every batch, then retire them all
int search_impl(anr_index *h, const float *q, bool q_on_host, int64_t nq, int32_t k, float *D, int64_t *I,
                bool out_on_host, hipStream_t st) {
  ANR_TRY(check_search_args(h, q, nq, k, D, I, true));
  DeviceGuard g(h->device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  if (!st) st = h->stream;
  ANR_TRY(drain(h));
  h->stats = anr_search_stats{};
  h->stats.n_queries = nq;
  if (nq == 0) return ANR_OK;
  if (h->ntotal == 0) return write_empty(h, nq, k, D, I, out_on_host);
  if (q_on_host && out_on_host && tiny_applies(h, nq, k)) return search_tiny(h, q, nq, k, D, I);
  ANR_TRY(ensure_workspaces(h));
  ANR_TRY(refresh_xstat(h));
  float *Dd = D;
  int64_t *Id = I;
  // small host-buffer searches: results are written by the kernels straight into pinned host memory (posted
  // PCIe writes) and the queries go through a pinned staging buffer and one DMA on the batch's own stream —
  // instead of three pageable copies, a cross-queue wait and a stream sync per call.  (Reading the queries in
  // place from host memory is NOT an option: uncoalesced 4-byte PCIe reads took ~0.8 ms per batch.)
  const bool zero_copy_out = out_on_host && nq * k <= kZeroCopyResults;
  if (zero_copy_out) {
    if (h->out_pin_alloc < nq * k) {
      if (h->out_pin) (void)hipHostFree(h->out_pin);
      h->out_pin = h->out_pin_dev = nullptr;
      h->out_pin_alloc = 0;
      const int64_t want = std::max<int64_t>(nq * k, 4096);
      // [want f32, padded to 8 B | want i64]
      ANR_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->out_pin), (size_t)round_up(want * 4, 8) + (size_t)want * 8,
                            hipHostMallocDefault));
      ANR_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->out_pin_dev), h->out_pin, 0));
      h->out_pin_alloc = want;
    }
    Dd = reinterpret_cast<float *>(h->out_pin_dev);
    Id = reinterpret_cast<int64_t *>(h->out_pin_dev + (size_t)round_up(nq * k * 4, 8));
  } else if (out_on_host) {
    if (h->out_alloc < nq * k) {
      dev_free(h->d_out);
      dev_free(h->i_out);
      ANR_TRY(dev_alloc(&h->d_out, nq * k, false));
      ANR_TRY(dev_alloc(&h->i_out, nq * k, false));
      h->out_alloc = nq * k;
    }
    Dd = h->d_out;
    Id = h->i_out;
  }
  if (h->timing) ANR_HIP(hipEventRecord(h->ev_call[0], st));
  if (k > kMaxSel) {
    ANR_TRY(search_large_k(h, q, q_on_host, nq, k, Dd, Id));
  } else {
    for (int64_t q0 = 0; q0 < nq; q0 += kQB) {
      const int nb = (int)std::min<int64_t>(kQB, nq - q0);
      const float *qd = q + q0 * h->dim;
      const float *pinned_src = nullptr;
      if (q_on_host) {
        Workspace &w = h->ws[h->next_ws];
        ANR_TRY(retire(h, w));  // its staging buffer is about to be overwritten
        memcpy(w.qpin, qd, (size_t)nb * h->dim * sizeof(float));
        pinned_src = w.qpin;
        qd = w.qstage;
      }
      ANR_TRY(enqueue_batch(h, qd, nb, k, q0, Dd, Id, st, zero_copy_out, pinned_src));
    }
    ANR_TRY(drain(h));
  }
  if (h->timing) {
    ANR_HIP(hipEventRecord(h->ev_call[1], st));
    ANR_HIP(hipEventSynchronize(h->ev_call[1]));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev_call[0], h->ev_call[1]) == hipSuccess) h->stats.total_ms = ms;
  }
  if (zero_copy_out) {
    // drain() waited for every batch's event and the kernels fenced their host writes at system scope
    memcpy(D, h->out_pin, (size_t)nq * k * sizeof(float));
    memcpy(I, h->out_pin + (size_t)round_up(nq * k * 4, 8), (size_t)nq * k * sizeof(int64_t));
  } else if (out_on_host) {
    ANR_HIP(hipMemcpyAsync(D, Dd, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
    ANR_HIP(hipMemcpyAsync(I, Id, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    ANR_HIP(hipStreamSynchronize(st));
  }
  return ANR_OK;
}

int add_impl(anr_index *h, const float *x_dev, int64_t n, hipStream_t st) {
  ANR_TRY(grow_storage(h, h->ntotal + n));
  ANR_TRY(drain(h));
  AddParams ap{};
  ap.xin = x_dev;
  ap.n = n;
  ap.row0 = h->ntotal;
  ap.dim = h->dim;
  ap.dimp = h->dimp;
  ap.kb = h->kb;
  ap.normalize = h->normalize && !h->add_raw;
  ap.x32 = h->x32;
  ap.x16 = h->x16;
  ap.rowbias = h->rowbias;
  ap.stat = h->xstat;
  ap.x12 = h->x12;  // (kept up to date whenever it exists)
  const int64_t t0 = h->ntotal / kTileRows, t1 = ceil_div(h->ntotal + n, kTileRows);
  hipLaunchKernelGGL(k_add, dim3((unsigned)(t1 - t0)), dim3(256), 0, st, ap);
  ANR_HIP(hipGetLastError());
  h->ntotal += n;
  h->xstat_dirty = true;
  h->x12_no_memory = false;
  return ANR_OK;
}

void free_workspaces(anr_index *h) {
  for (auto &w : h->ws) {
    dev_free(w.q32); dev_free(w.q16); dev_free(w.qstat); dev_free(w.qstage); dev_free(w.dense);
    dev_free(w.ladder); dev_free(w.cntb); dev_free(w.ncand); dev_free(w.cand); dev_free(w.qslots); dev_free(w.lvlmax);
    dev_free(w.sel_rank); dev_free(w.sel_row); dev_free(w.sel_m); dev_free(w.exact); dev_free(w.stamps);
    if (w.cnt_host) (void)hipHostFree(w.cnt_host);
    if (w.qpin) (void)hipHostFree(w.qpin);
    if (w.qslots_pin) (void)hipHostFree(w.qslots_pin);
    if (w.lad_pin) (void)hipHostFree(w.lad_pin);
    w.cnt_host = w.cnt_dev = nullptr;
    w.qpin = nullptr;
    w.qslots_pin = nullptr;
    w.lad_pin = nullptr;
    w.cand_alloc = 0;
    w.dense_ld = 0;
    for (hipEvent_t *e : {&w.ev_in, &w.ev_done, &w.ev_t0, &w.ev_t1, &w.ev_pre, &w.ev_scan, &w.ev_rec}) {
      if (*e) (void)hipEventDestroy(*e);
      *e = nullptr;
    }
  }
  for (auto &e : h->ev_call) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  h->ws_ready = false;
}

}  // namespace

extern "C" {

int anr_index_create(int32_t dim, int32_t metric, int32_t normalize, int32_t device, anr_index **out) {
  if (!out) return fail(ANR_EINVAL, "out is null");
  *out = nullptr;
  if (dim <= 0 || dim > 4096) return fail(ANR_EINVAL, "dim must be in 1..4096 (got %d)", dim);
  if (metric != ANR_METRIC_IP && metric != ANR_METRIC_L2) return fail(ANR_EINVAL, "unknown metric %d", metric);
  int ndev = anr_device_count();
  if (ndev <= 0) return fail(ANR_EHIP, "no HIP device is visible");
  if (device < 0 || device >= ndev) return fail(ANR_EINVAL, "device %d out of range (0..%d)", device, ndev - 1);
  DeviceGuard g(device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", device);
  anr_index *h = new anr_index();
  h->dim = dim;
  h->dimp = (int)round_up(dim, 128);
  h->kb = h->dimp / 16;
  if ((size_t)2 * h->kb * 64 * 16 + kQB * kLadder * 8 + kQB * 8 > 160 * 1024) {
    delete h;
    return fail(ANR_EINVAL, "dim %d needs more than 160 KiB of LDS for the query operand", dim);
  }
  h->metric = metric;
  h->normalize = normalize ? 1 : 0;
  h->device = device;
  h->n_cu = device_cu_count(device);
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) h->clock_khz = khz;
  }
  hipStream_t *streams[1 + kWorkspaces] = {&h->stream, &h->bstream[0], &h->bstream[1], &h->bstream[2]};
  for (auto s : streams) {
    hipError_t e = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    if (e != hipSuccess) {
      anr_index_destroy(h);
      return fail(ANR_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
  }
  int r = dev_alloc(&h->xstat, 4, true);
  if (r == ANR_OK) r = dev_alloc(&h->xstat12, 2, true);
  if (r != ANR_OK) {
    anr_index_destroy(h);
    return r;
  }
  *out = h;
  return ANR_OK;
}

int anr_index_destroy(anr_index *h) {
  if (!h) return ANR_OK;
  DeviceGuard g(h->device);
  for (hipStream_t s : {h->bstream[0], h->bstream[1], h->bstream[2], h->stream})
    if (s) (void)hipStreamSynchronize(s);
  dev_free(h->x32);
  dev_free(h->x16);
  dev_free(h->rowbias);
  dev_free(h->xstat);
  dev_free(h->xstat12);
  dev_free(h->x12);
  free_workspaces(h);
  dev_free(h->xdense);
  free_largek(&h->largek);
  dev_free(h->d_out);
  dev_free(h->i_out);
  if (h->out_pin) (void)hipHostFree(h->out_pin);
  if (h->tiny_pin) (void)hipHostFree(h->tiny_pin);
  dev_free(h->tiny_cand);
  dev_free(h->sr_buf);
  dev_free(h->tiny_ticket);
  dev_free(h->tiny_stamps);
  for (auto &e : h->ev_call)
    if (e) (void)hipEventDestroy(e);
  for (hipStream_t s : {h->bstream[0], h->bstream[1], h->bstream[2], h->stream})
    if (s) (void)hipStreamDestroy(s);
  delete h;
  return ANR_OK;
}

int anr_index_reserve(anr_index *h, int64_t n) {
  if (!h || n < 0) return fail(ANR_EINVAL, "bad argument");
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  return grow_storage(h, n);
}

int anr_index_add_dev(anr_index *h, const float *x_dev, int64_t n, void *stream) {
  if (!h || (!x_dev && n > 0) || n < 0) return fail(ANR_EINVAL, "bad argument");
  if (n == 0) return ANR_OK;
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  ANR_TRY(add_impl(h, x_dev, n, st));
  ANR_HIP(hipStreamSynchronize(st));
  return ANR_OK;
}

int anr_index_add(anr_index *h, const float *x_host, int64_t n) {
  if (!h || (!x_host && n > 0) || n < 0) return fail(ANR_EINVAL, "bad argument");
  if (n == 0) return ANR_OK;
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  ANR_TRY(grow_storage(h, h->ntotal + n));
  // stream the host rows through a bounded device staging buffer
  const int64_t chunk = std::max<int64_t>(kTileRows, std::min<int64_t>(n, (int64_t)(256ll << 20) / (h->dim * 4)));
  float *stage = nullptr;
  ANR_TRY(dev_alloc(&stage, chunk * h->dim, false));
  int rc = ANR_OK;
  for (int64_t i = 0; i < n && rc == ANR_OK; i += chunk) {
    const int64_t m = std::min(chunk, n - i);
    hipError_t e = hipMemcpyAsync(stage, x_host + i * h->dim, (size_t)m * h->dim * sizeof(float),
                                  hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) {
      rc = fail(ANR_EHIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
      break;
    }
    rc = add_impl(h, stage, m, h->stream);
    if (rc == ANR_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(ANR_EHIP, "add kernel failed");
  }
  (void)hipFree(stage);
  return rc;
}

int64_t anr_index_ntotal(const anr_index *h) { return h ? h->ntotal : 0; }
int32_t anr_index_dim(const anr_index *h) { return h ? h->dim : 0; }

int anr_index_reset(anr_index *h) {
  if (!h) return fail(ANR_EINVAL, "null handle");
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  ANR_TRY(drain(h));
  ANR_HIP(hipStreamSynchronize(h->stream));
  if (h->x16) ANR_HIP(hipMemset(h->x16, 0, (size_t)h->cap * h->dimp * sizeof(_Float16)));
  if (h->x12) ANR_HIP(hipMemset(h->x12, 0, (size_t)h->cap * h->dimp * 3 / 2));
  if (h->rowbias) ANR_HIP(hipMemset(h->rowbias, 0, (size_t)h->cap * sizeof(float)));
  ANR_HIP(hipMemset(h->xstat, 0, 4 * sizeof(unsigned)));
  ANR_HIP(hipStreamSynchronize(nullptr));  // the fills run on the NULL stream; later adds use the handle's own
  h->ntotal = 0;
  h->xstat_dirty = true;
  h->f12_suspended = false;
  h->f12_strikes = h->f12_level = h->f12_clean = h->f12_lowered = 0;
  h->f12_hold = 64;
  return ANR_OK;
}

int anr_index_reconstruct(anr_index *h, int64_t i0, int64_t n, float *out_host) {
  if (!h || !out_host || i0 < 0 || n < 0 || i0 + n > h->ntotal) return fail(ANR_EINVAL, "row range out of bounds");
  if (n == 0) return ANR_OK;
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  ANR_HIP(hipStreamSynchronize(h->stream));
  ANR_HIP(hipMemcpy(out_host, h->x32 + i0 * h->dim, (size_t)n * h->dim * sizeof(float), hipMemcpyDeviceToHost));
  return ANR_OK;
}

int anr_index_reconstruct_scan_image(anr_index *h, int32_t bits, int64_t i0, int64_t n, float *out_host) {
  if (!h || !out_host || i0 < 0 || n < 0 || i0 + n > h->ntotal) return fail(ANR_EINVAL, "row range out of bounds");
  if (bits != 12 && bits != 16) return fail(ANR_EINVAL, "bits must be 12 or 16");
  if (bits == 12 && !h->x12) return fail(ANR_ESTATE, "the index keeps no 12-bit image (ANR_OPT_SCAN_BITS)");
  if (n == 0) return ANR_OK;
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  ANR_TRY(drain(h));
  float *d = nullptr;
  ANR_TRY(dev_alloc(&d, n * h->dim, false));
  ImageRowsParams ip{h->x16, bits == 12 ? h->x12 : nullptr, i0, n, h->dim, h->kb, d};
  hipLaunchKernelGGL(k_image_rows, dim3((unsigned)ceil_div(n * h->kb * 2, 256)), dim3(256), 0, h->stream, ip);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(out_host, d, (size_t)n * h->dim * sizeof(float), hipMemcpyDeviceToHost);
  dev_free(d);
  if (e != hipSuccess) return fail(ANR_EHIP, "reading the scan image failed: %s", hipGetErrorString(e));
  return ANR_OK;
}

int anr_index_scan_image_stats(anr_index *h, int32_t *bits_in_use, float *max_norm, float *max_err16, float *max_err12) {
  if (!h) return fail(ANR_EINVAL, "null handle");
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  ANR_TRY(drain(h));
  unsigned host[4] = {0, 0, 0, 0};
  ANR_HIP(hipStreamSynchronize(h->stream));
  ANR_HIP(hipMemcpy(host, h->xstat, sizeof host, hipMemcpyDeviceToHost));
  auto f = [](unsigned u) {
    float v;
    std::memcpy(&v, &u, 4);
    return v;
  };
  const bool f12 = h->x12 && !h->f12_suspended && (h->scan_bits == 12 || (h->scan_bits == 0 && h->ntotal >= kAuto12Rows));
  if (bits_in_use) *bits_in_use = f12 ? 12 : 16;
  if (max_norm) *max_norm = f(host[0]);
  if (max_err16) *max_err16 = f(host[1]);
  if (max_err12) *max_err12 = h->x12 ? f(host[3]) : 0.0f;
  return ANR_OK;
}

int anr_index_search(anr_index *h, const float *q_host, int64_t nq, int32_t k, float *D, int64_t *I) {
  return search_impl(h, q_host, true, nq, k, D, I, true, nullptr);
}

int anr_index_search_devq(anr_index *h, const float *q_dev, int64_t nq, int32_t k, float *D, int64_t *I) {
  return search_impl(h, q_dev, false, nq, k, D, I, true, nullptr);
}

int anr_index_search_dev(anr_index *h, const float *q_dev, int64_t nq, int32_t k, float *D_dev, int64_t *I_dev,
                         void *stream) {
  return search_impl(h, q_dev, false, nq, k, D_dev, I_dev, false, reinterpret_cast<hipStream_t>(stream));
}

int anr_index_search_dev_async(anr_index *h, const float *q_dev, int64_t nq, int32_t k, float *D_dev,
                               int64_t *I_dev, void *stream) {
  ANR_TRY(check_search_args(h, q_dev, nq, k, D_dev, I_dev, false));
  DeviceGuard g(h->device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  h->stats.n_queries += nq;
  if (nq == 0) return ANR_OK;
  if (h->ntotal == 0) return write_empty(h, nq, k, D_dev, I_dev, false);
  ANR_TRY(ensure_workspaces(h));
  ANR_TRY(refresh_xstat(h));
  for (int64_t q0 = 0; q0 < nq; q0 += kQB) {
    const int nb = (int)std::min<int64_t>(kQB, nq - q0);
    ANR_TRY(enqueue_batch(h, q_dev + q0 * h->dim, nb, k, q0, D_dev, I_dev, st));
  }
  return ANR_OK;
}

int anr_index_sync(anr_index *h) {
  if (!h) return fail(ANR_EINVAL, "null handle");
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  return drain(h);
}

int anr_index_wait(anr_index *h, int32_t keep) {
  if (!h) return fail(ANR_EINVAL, "null handle");
  if (keep < 0) return fail(ANR_EINVAL, "keep must be >= 0");
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  if (!h->ws_ready) return ANR_OK;
  int flying = 0;
  for (const auto &w : h->ws) flying += w.in_flight ? 1 : 0;
  // oldest first: next_ws is the workspace the next batch will take, i.e. the one used longest ago
  for (int i = 0; i < kWorkspaces && flying > keep; ++i) {
    Workspace &w = h->ws[(h->next_ws + i) % kWorkspaces];
    if (!w.in_flight) continue;
    ANR_TRY(retire(h, w));
    --flying;
  }
  return ANR_OK;
}

int anr_index_score_rows(anr_index *h, const float *q_host, int64_t nq, const int64_t *ids_host, int32_t per_query,
                         float *out_host) {
  if (!h || !q_host || !ids_host || !out_host || nq < 0 || per_query < 0) return fail(ANR_EINVAL, "bad argument");
  const int64_t total = nq * per_query;
  if (total == 0) return ANR_OK;
  DeviceGuard g(h->device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  ANR_TRY(drain(h));
  // one scratch block per handle, grown on demand and kept (seven hipMalloc / hipFree pairs per call were most of the
  // time of a ~100-candidate call); queries go through the same preprocessing as searches, in blocks of 64
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t b_q = al((size_t)nq * h->dim * 4), b_out = al((size_t)total * 4), b_ids = al((size_t)total * 8),
               b_qn = al((size_t)round_up(nq, kQB) * h->dim * 4), b_q16 = al((size_t)kQB * h->dimp * 2),
               b_qs = al((size_t)kQB * 4 * 4), b_q32 = al((size_t)kQB * h->dimp * 4);
  const size_t need = b_q + b_out + b_ids + b_qn + b_q16 + b_qs + b_q32;
  if ((int64_t)need > h->sr_cap) {
    dev_free(h->sr_buf);
    h->sr_cap = 0;
    ANR_TRY(dev_alloc(&h->sr_buf, (int64_t)(need + need / 2), false));
    h->sr_cap = (int64_t)(need + need / 2);
  }
  unsigned char *base = h->sr_buf;
  float *dq = reinterpret_cast<float *>(base);
  float *dout = reinterpret_cast<float *>(base + b_q);
  int64_t *dids = reinterpret_cast<int64_t *>(base + b_q + b_out);
  float *dqn = reinterpret_cast<float *>(base + b_q + b_out + b_ids);
  _Float16 *dq16 = reinterpret_cast<_Float16 *>(base + b_q + b_out + b_ids + b_qn);
  float *dqs = reinterpret_cast<float *>(base + b_q + b_out + b_ids + b_qn + b_q16);
  float *dq32 = reinterpret_cast<float *>(base + b_q + b_out + b_ids + b_qn + b_q16 + b_qs);
  hipStream_t st = h->stream;
  hipError_t e = hipMemcpyAsync(dq, q_host, (size_t)nq * h->dim * sizeof(float), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(dids, ids_host, (size_t)total * sizeof(int64_t), hipMemcpyHostToDevice, st);
  for (int64_t q0 = 0; q0 < nq && e == hipSuccess; q0 += kQB) {
    const int nb = (int)std::min<int64_t>(kQB, nq - q0);
    PrepQParams qp{};
    qp.qin = dq + q0 * h->dim;
    qp.nq = nb;
    qp.dim = h->dim;
    qp.dimp = h->dimp;
    qp.kb = h->kb;
    qp.normalize = h->normalize;
    qp.q32 = dq32;
    qp.q16 = dq16;
    qp.qstat = dqs;
    hipLaunchKernelGGL(k_prepq, dim3(kQB), dim3(256), 0, st, qp);
    // compact the padded [64][dimp] block back to [nb][dim]
    e = hipMemcpy2DAsync(dqn + q0 * h->dim, (size_t)h->dim * sizeof(float), dq32, (size_t)h->dimp * sizeof(float),
                         (size_t)h->dim * sizeof(float), (size_t)nb, hipMemcpyDeviceToDevice, st);
  }
  if (e == hipSuccess) {
    ScoreRowsParams sp{h->x32, dqn, h->dim, h->metric, h->ntotal, dids, per_query, total, dout};
    hipLaunchKernelGGL(k_score_rows, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, st, sp);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out_host, dout, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return fail(ANR_EHIP, "score_rows failed: %s", hipGetErrorString(e));
  return ANR_OK;
}

int anr_index_self_join(anr_index *h, float threshold, int64_t cap, int64_t *I_host, int64_t *J_host, float *S_host,
                        int64_t *n_pairs) {
  if (!h || !n_pairs || cap < 0 || (cap > 0 && (!I_host || !J_host || !S_host))) return fail(ANR_EINVAL, "bad argument");
  if (h->metric != ANR_METRIC_IP) return fail(ANR_EINVAL, "self join is defined for the inner-product metric");
  if (!(threshold == threshold)) return fail(ANR_EINVAL, "threshold is NaN");
  DeviceGuard g(h->device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  ANR_TRY(drain(h));
  *n_pairs = 0;
  if (h->ntotal < 2) return ANR_OK;
  ANR_TRY(refresh_xstat(h));
  if (h->f16_unusable) return fail(ANR_ESTATE, "stored values exceed the f16 range: the f16 image cannot nominate pairs");
  unsigned xs[2] = {0, 0};
  ANR_HIP(hipMemcpy(xs, h->xstat, sizeof xs, hipMemcpyDeviceToHost));
  float xmax, xerr;
  memcpy(&xmax, &xs[0], 4);
  memcpy(&xerr, &xs[1], 4);
  // |x16_i . x16_j - x_i . x_j| <= |x_i||dx_j| + |dx_i||x16_j| (+ f32 accumulation of the MFMA)
  const double eps = 2.0 * xmax * xerr + (double)xerr * xerr + 2.0 * h->dimp * 5.9604645e-8 * (double)xmax * xmax;
  const int64_t n_tiles = ceil_div(h->ntotal, kTileRows);
  const int nblk = (int)ceil_div(n_tiles, 8);
  const PatchGrid pg = make_patch_grid(nblk, nblk);
  const int64_t n_blocks = pg.grid();
  if (n_blocks > 0x7fffffffLL) return fail(ANR_EINVAL, "too many rows for one self-join launch");
  // candidates = pairs within eps below the threshold as well: a little more room than the caller's cap
  const int64_t ccap = cap + cap / 4 + 65536;
  uint2 *cand = nullptr;
  unsigned long long *counts = nullptr;  // [0] candidates, [1] results
  int64_t *oi = nullptr, *oj = nullptr;
  float *os = nullptr;
  int rc = ANR_OK;
  auto cleanup = [&]() {
    dev_free(cand); dev_free(counts); dev_free(oi); dev_free(oj); dev_free(os);
  };
  if ((rc = dev_alloc(&cand, ccap, false)) || (rc = dev_alloc(&counts, 2, true)) ||
      (rc = dev_alloc(&oi, std::max<int64_t>(cap, 1), false)) || (rc = dev_alloc(&oj, std::max<int64_t>(cap, 1), false)) ||
      (rc = dev_alloc(&os, std::max<int64_t>(cap, 1), false))) {
    cleanup();
    return rc;
  }
  hipStream_t st = h->stream;
  JoinParams jp{};
  jp.x16 = reinterpret_cast<const uint4 *>(h->x16);
  jp.kb = h->kb;
  jp.n_rows = h->ntotal;
  jp.n_tiles = n_tiles;
  jp.nblk = nblk;
  jp.pg = pg;
  jp.n_slots = n_blocks;
  jp.thr_lo = (float)((double)threshold - eps - 1e-7 * (std::fabs((double)threshold) + 1.0));
  jp.cand = cand;
  jp.cap = (unsigned long long)ccap;
  jp.count = counts;
  hipError_t e = hipSuccess;
  e = launch_join(jp, n_blocks, st, h->n_cu);
  unsigned long long hc[2] = {0, 0};
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(hc, counts, sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess && hc[0] > (unsigned long long)ccap) {
    *n_pairs = -(int64_t)hc[0];
    cleanup();
    return ANR_OK;
  }
  if (e == hipSuccess && hc[0] > 0) {
    JoinRescoreParams rp{h->x32, h->dim, cand, hc[0], threshold, oi, oj, os, (unsigned long long)cap, counts + 1};
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div((int64_t)hc[0], 256), (int64_t)h->n_cu * 16);
    hipLaunchKernelGGL(k_join_rescore, dim3(grid), dim3(256), 0, st, rp);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(hc + 1, counts + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && hc[1] > (unsigned long long)cap) {
      *n_pairs = -(int64_t)hc[1];
      cleanup();
      return ANR_OK;
    }
    if (e == hipSuccess && hc[1] > 0) {
      e = hipMemcpy(I_host, oi, hc[1] * sizeof(int64_t), hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(J_host, oj, hc[1] * sizeof(int64_t), hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(S_host, os, hc[1] * sizeof(float), hipMemcpyDeviceToHost);
    }
  }
  cleanup();
  if (e != hipSuccess) return fail(ANR_EHIP, "self join failed: %s", hipGetErrorString(e));
  *n_pairs = (int64_t)hc[1];
  return ANR_OK;
}

int anr_index_batch_log(anr_index *h, int64_t *out, int32_t max_batches, int32_t *n_out, int64_t *dev_minus_host_ns) {
  if (!h || !n_out || (max_batches > 0 && !out)) return fail(ANR_EINVAL, "null argument");
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  ANR_TRY(drain(h));
  const int64_t have = std::min<int64_t>(h->batch_log_n, kBatchLogCap);
  const int64_t n = std::min<int64_t>(have, std::max(0, max_batches));
  for (int64_t i = 0; i < n; ++i) {
    const int64_t r = h->batch_log_n - n + i;
    memcpy(out + i * kBatchLogFields, h->batch_log.data() + (size_t)(r % kBatchLogCap) * kBatchLogFields,
           kBatchLogFields * sizeof(int64_t));
  }
  *n_out = (int32_t)n;
  if (dev_minus_host_ns) {
    // correlate the two clocks: a one-thread kernel writes the device clock into pinned memory, the host notes when it
    // sees it; the tightest of a few round trips (the stamp is taken 1-3 us before the host can observe it)
    ANR_TRY(ensure_workspaces(h));
    volatile unsigned long long *pin = reinterpret_cast<volatile unsigned long long *>(h->ws[0].cnt_host + 4 * kQB);
    unsigned long long *pin_dev = reinterpret_cast<unsigned long long *>(h->ws[0].cnt_dev + 4 * kQB);
    int64_t best_rt = -1, best_off = 0;
    for (int rep = 0; rep < 6; ++rep) {
      const unsigned long long saved = pin[0];
      pin[0] = 0;
      const int64_t t0 = host_ns();
      hipLaunchKernelGGL(k_clock, dim3(1), dim3(1), 0, h->stream, pin_dev);
      ANR_HIP(hipGetLastError());
      int64_t t1 = t0;
      while (pin[0] == 0) {
        t1 = host_ns();
        if (t1 - t0 > 2000000000ll) return fail(ANR_EHIP, "clock calibration kernel did not report");
      }
      t1 = host_ns();
      const int64_t dev_ns = (int64_t)((double)pin[0] * 1e6 / (double)h->clock_khz);
      ANR_HIP(hipStreamSynchronize(h->stream));
      pin[0] = saved;
      if (rep > 0 && (best_rt < 0 || t1 - t0 < best_rt)) {
        best_rt = t1 - t0;
        best_off = dev_ns - t1;
      }
    }
    *dev_minus_host_ns = best_off;
  }
  return ANR_OK;
}

int anr_index_set_option(anr_index *h, int32_t opt, int64_t value) {
  if (!h) return fail(ANR_EINVAL, "null handle");
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->mu);
  ANR_TRY(drain(h));
  switch (opt) {
    case ANR_OPT_FORCE_EXACT: h->force_exact = value != 0; break;
    case ANR_OPT_OVERFETCH:
      if (value < 0 || value > kMaxSel) return fail(ANR_EINVAL, "overfetch must be in 0..%d", kMaxSel);
      h->overfetch = (int)value;
      break;
    case ANR_OPT_SAMPLE_ROWS:
      if (value < 0 || value > (1 << 20)) return fail(ANR_EINVAL, "sample rows must be in 0..2^20");
      h->sample_rows = (int)round_up(value, kTileRows);
      break;
    case ANR_OPT_CAND_CAP:
      if (value < 64 || value > (1 << 16)) return fail(ANR_EINVAL, "candidate list capacity must be in 64..65536");
      h->cand_cap = value;
      break;
    case ANR_OPT_TIMING: h->timing = value != 0; break;
    case ANR_OPT_ADD_RAW: h->add_raw = value != 0; break;
    case ANR_OPT_ID_OFFSET:
      if (value < 0) return fail(ANR_EINVAL, "id offset must be >= 0");
      h->id_offset = value;
      break;
    case ANR_OPT_TINY: h->tiny = value < 0 ? 0 : (value > 2 ? 2 : (int)value); break;
    case ANR_OPT_FUSED_POST: h->fused_post = value < 0 ? 0 : (value > 2 ? 2 : (int)value); break;
    case ANR_OPT_SHADOW: h->shadow = value < 0 ? 0 : (value > 2 ? 2 : (int)value); break;
    case ANR_OPT_SCHEDULE: h->schedule = value != 0; break;
    case ANR_OPT_STREAM_WAIT: h->stream_wait = value != 0; break;
    case ANR_OPT_SCAN_BITS: {
      if (value != 0 && value != 12 && value != 16) return fail(ANR_EINVAL, "scan bits must be 0 (auto), 12 or 16");
      h->f12_suspended = false;
      h->x12_no_memory = false;
      h->f12_strikes = h->f12_level = h->f12_clean = h->f12_lowered = 0;
      h->f12_hold = 64;
      if ((int)value == h->scan_bits) break;
      ANR_TRY(drain(h));
      ANR_HIP(hipStreamSynchronize(h->stream));
      h->scan_bits = (int)value;
      if (value == 16) dev_free(h->x12);
      if (value == 12 && !h->x12 && h->cap > 0) ANR_TRY(build_x12(h));
      break;
    }
    case ANR_OPT_STREAMS:
      if (value < 1 || value > kWorkspaces) return fail(ANR_EINVAL, "streams must be in 1..%d", kWorkspaces);
      h->n_streams = (int)value;
      break;
    default: return fail(ANR_EINVAL, "unknown option %d", opt);
  }
  return ANR_OK;
}

int anr_index_last_stats(anr_index *h, anr_search_stats *out) {
  if (!h || !out) return fail(ANR_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(h->mu);
  *out = h->stats;
  return ANR_OK;
}

int anr_index_reset_stats(anr_index *h) {
  if (!h) return fail(ANR_EINVAL, "null handle");
  std::lock_guard<std::mutex> lk(h->mu);
  h->stats = anr_search_stats{};
  return ANR_OK;
}

int anr_normalize_rows(float *x_host, int64_t n, int32_t d, int32_t device) {
  if (!x_host || n < 0 || d <= 0) return fail(ANR_EINVAL, "bad argument");
  if (n == 0) return ANR_OK;
  anr_index *h = nullptr;
  ANR_TRY(anr_index_create(d, ANR_METRIC_IP, 1, device, &h));
  int rc = ANR_OK;
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(256ll << 20) / ((int64_t)d * 4));
  for (int64_t i = 0; i < n && rc == ANR_OK; i += chunk) {
    const int64_t m = std::min(chunk, n - i);
    rc = anr_index_reset(h);
    if (rc == ANR_OK) rc = anr_index_add(h, x_host + i * d, m);
    if (rc == ANR_OK) rc = anr_index_reconstruct(h, 0, m, x_host + i * d);
  }
  anr_index_destroy(h);
  return rc;
}

int anr_merge_topk_strided_dev(int32_t device, const float *Dp_dev, const int64_t *Ip_dev, int64_t d_stride,
                               int64_t i_stride, int32_t P, int64_t nq, int32_t k, int32_t larger_is_better,
                               float *D_dev, int64_t *I_dev, void *stream) {
  if (!Dp_dev || !Ip_dev || !D_dev || !I_dev || P <= 0 || nq < 0 || k <= 0) return fail(ANR_EINVAL, "bad argument");
  if (d_stride < nq * k || i_stride < nq * k) return fail(ANR_EINVAL, "part stride shorter than one part");
  if (nq == 0) return ANR_OK;
  DeviceGuard g(device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice(%d) failed", device);
  MergeParams mp{Dp_dev, Ip_dev, d_stride, i_stride, P, nq, k, larger_is_better, D_dev, I_dev};
  hipLaunchKernelGGL(k_merge, dim3((unsigned)nq), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), mp);
  ANR_HIP(hipGetLastError());
  return ANR_OK;
}

int anr_merge_topk_host(const float *Dp, const int64_t *Ip, int32_t P, int64_t nq, int32_t k, int32_t larger_is_better,
                        float *D, int64_t *I) {
  if (!Dp || !Ip || !D || !I || P <= 0 || nq < 0 || k <= 0) return fail(ANR_EINVAL, "bad argument");
  // P-way merge of sorted lists per query (P is the GPU count of one node: a linear scan over the heads)
  std::vector<int> head((size_t)P);
  auto before = [&](float sa, int64_t ia, float sb, int64_t ib) {
    if (ia < 0) return false;  // padding goes last
    if (ib < 0) return true;
    if (sa != sb) return larger_is_better ? (sa > sb) : (sa < sb);
    return ia < ib;
  };
  for (int64_t q = 0; q < nq; ++q) {
    std::fill(head.begin(), head.end(), 0);
    for (int j = 0; j < k; ++j) {
      int best = -1;
      for (int p = 0; p < P; ++p) {
        if (head[p] >= k) continue;
        const int64_t e = ((int64_t)p * nq + q) * k + head[p];
        if (Ip[e] < 0) continue;
        if (best < 0) {
          best = p;
          continue;
        }
        const int64_t b = ((int64_t)best * nq + q) * k + head[best];
        if (before(Dp[e], Ip[e], Dp[b], Ip[b])) best = p;
      }
      if (best < 0) {
        D[q * k + j] = larger_is_better ? -3.402823466e+38f : 3.402823466e+38f;
        I[q * k + j] = -1;
      } else {
        const int64_t b = ((int64_t)best * nq + q) * k + head[best]++;
        D[q * k + j] = Dp[b];
        I[q * k + j] = Ip[b];
      }
    }
  }
  return ANR_OK;
}

int anr_merge_topk_dev(int32_t device, const float *Dp_dev, const int64_t *Ip_dev, int32_t P, int64_t nq, int32_t k,
                       int32_t larger_is_better, float *D_dev, int64_t *I_dev, void *stream) {
  return anr_merge_topk_strided_dev(device, Dp_dev, Ip_dev, nq * k, nq * k, P, nq, k, larger_is_better, D_dev, I_dev,
                                    stream);
}

}  // extern "C"
