// Device kernels of the exact flat index (CDNA4 / gfx950 only).
//
// Data layout in HBM:
//   x32  [cap][dim]                    float32 row-major — the stored (normalised) corpus, used for the
//                                      exact re-score and for reconstruct;
//   x16  [cap/32][KB][64 lanes][8]     f16, "MFMA-blocked": one 1-KiB block is exactly the A operand of
//                                      one v_mfma_f32_32x32x16_f16 for 32 rows x 16 dims, so a wave's
//                                      16-B/lane load is one contiguous KiB and lands in operand layout;
//   x12  [cap/32][KB][64 lanes][12 B]  optional (ANR_OPT_SCAN_BITS 12): the same blocks with every f16 rounded to its top
//                                      12 bits (sign, 5 exponent, 6 mantissa bits) — what the streaming scan reads instead of
//                                      x16, 25 % fewer bytes.  Per lane three words: the high bytes of elements 0-3, of
//                                      elements 4-7, and the eight low nibbles (elements 0-3 in the low nibbles of bytes
//                                      0-3, elements 4-7 in the high nibbles) — seven vector instructions turn them back
//                                      into the 16-byte f16 operand (x12_unpack);
//   q16  [2][KB][64 lanes][8]          f16, the 64 queries of a batch in the B-operand layout.
// KB = dimp/16 with dimp = dim rounded up to 128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace anr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kTileRows = 32;    // corpus rows per MFMA tile
constexpr int kQB = 64;          // queries per scan batch
constexpr int kLadder = 8;       // threshold ladder levels
constexpr int kMaxSel = 1024;    // max entries a select can return

__device__ __forceinline__ unsigned f2ord(float f) {
  // monotone float -> uint (NaN maps below everything)
  if (f != f) return 0u;
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long make_key(float rank, unsigned row) {
  // descending key order == (rank desc, row asc)
  return ((unsigned long long)f2ord(rank) << 32) | (unsigned long long)(0xffffffffu - row);
}

// Device time stamps of a batch (anr_index_batch_log): u64 words in device memory, one set per workspace, in ticks of the
// constant 100-MHz clock.  "End" stamps are atomic maxima (the clock is monotone, so a new batch's stamps overwrite the
// previous batch's without a reset); the scan's start is a minimum over its workgroups and is reset by the kernel before.
enum { kStampPrepEnd = 0, kStampSampleEnd = 1, kStampLadderEnd = 2, kStampScanStart = 3, kStampScanPlaced = 4, kStampScanEnd = 5,
       kStampSelEnd = 6, kStampPostEnd = 7, kStamps = 8 };
__device__ __forceinline__ void stamp_max(unsigned long long *st, int slot) {
  if (st) atomicMax(st + slot, (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void stamp_min(unsigned long long *st, int slot) {
  if (st) atomicMin(st + slot, (unsigned long long)wall_clock64());
}

// clock correlation for anr_index_batch_log: the device clock, written where the host is watching
__global__ void k_clock(unsigned long long *out_pinned) {
  __hip_atomic_store(out_pinned, (unsigned long long)wall_clock64(), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the 12-bit image of an f16 operand block (ANR_OPT_SCAN_BITS 12) -------------------------------------------------
// An f16 keeps its sign, exponent and the top six mantissa bits, rounded to nearest-even at bit 4 (a carry walks into the
// exponent as IEEE intends; a value that would round up to infinity is truncated instead).  value(x12) = f16 bits (x12 << 4).
__device__ __forceinline__ unsigned f16_to_12(_Float16 q) {
  const unsigned b = (unsigned)__builtin_bit_cast(unsigned short, q);
  unsigned r = (b + 7u + ((b >> 4) & 1u)) >> 4;
  if ((r & 0x7c0u) == 0x7c0u && (b & 0x7c00u) != 0x7c00u) r = b >> 4;  // would become inf / nan: truncate
  return r & 0xfffu;
}
__device__ __forceinline__ _Float16 f12_value(unsigned r) {
  return __builtin_bit_cast(_Float16, (unsigned short)(r << 4));
}
struct X12 {
  unsigned h0, h1, nib;
};
__device__ __forceinline__ X12 x12_pack(const unsigned (&r)[8]) {
  X12 o;
  o.h0 = (r[0] >> 4) | ((r[1] >> 4) << 8) | ((r[2] >> 4) << 16) | ((r[3] >> 4) << 24);
  o.h1 = (r[4] >> 4) | ((r[5] >> 4) << 8) | ((r[6] >> 4) << 16) | ((r[7] >> 4) << 24);
  o.nib = (r[0] & 15u) | ((r[1] & 15u) << 8) | ((r[2] & 15u) << 16) | ((r[3] & 15u) << 24) | ((r[4] & 15u) << 4) |
          ((r[5] & 15u) << 12) | ((r[6] & 15u) << 20) | ((r[7] & 15u) << 28);
  return o;
}
// -> the f16 operand (element 2 j in the low half of word j).  v_perm_b32 picks bytes 0-3 from its SECOND source and 4-7 from
// its first: word 0 = [nl.b0, h0.b0, nl.b1, h0.b1], word 1 = [nl.b2, h0.b2, nl.b3, h0.b3], likewise h1 / nh.
__device__ __forceinline__ uint4 x12_unpack(unsigned h0, unsigned h1, unsigned nib) {
  const unsigned nl = (nib & 0x0f0f0f0fu) << 4, nh = nib & 0xf0f0f0f0u;
  uint4 o;
  o.x = __builtin_amdgcn_perm(h0, nl, 0x05010400u);
  o.y = __builtin_amdgcn_perm(h0, nl, 0x07030602u);
  o.z = __builtin_amdgcn_perm(h1, nh, 0x05010400u);
  o.w = __builtin_amdgcn_perm(h1, nh, 0x07030602u);
  return o;
}

// ------------------------------------------------------------------------------------------------
// add: normalise (optional), store x32, convert to the blocked f16 image, track norm / error maxima
// ------------------------------------------------------------------------------------------------
struct AddParams {
  const float *xin;   // [n][dim] rows being appended
  int64_t n;
  int64_t row0;       // global index of xin row 0 (== ntotal before the add)
  int dim, dimp, kb;
  int normalize;
  float *x32;
  _Float16 *x16;
  float *rowbias;     // [cap] -0.5*||x||^2 of the stored row (L2 metric) or nullptr
  unsigned *stat;     // [0] max ||x|| bits, [1] max ||x16 - x|| bits (floats >= 0, so uint order == float
                      // order), [2] != 0 when some |x| exceeds the f16 range (scan disabled), [3] max ||x12 - x|| bits
  unsigned *x12;      // optional 12-bit image (see the top of this file)
};

// one block (256 threads) per 32-row tile that receives rows
__global__ __launch_bounds__(256) void k_add(AddParams p) {
  __shared__ float s_scale[kTileRows];
  __shared__ float s_err[kTileRows];
  __shared__ float s_err12[kTileRows];
  __shared__ float s_n2[kTileRows];
  __shared__ float s_x[kTileRows][132];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tile = p.row0 / kTileRows + blockIdx.x;
  const int64_t trow0 = tile * kTileRows;

  // per-row norms: wave w handles rows w, w+4, ...
  for (int r = wave; r < kTileRows; r += 4) {
    const int64_t grow = trow0 + r;
    const int64_t src = grow - p.row0;
    double acc = 0.0;
    if (src >= 0 && src < p.n) {
      const float *x = p.xin + src * p.dim;
      for (int k = lane; k < p.dim; k += 64) {
        double v = (double)x[k];
        acc += v * v;
      }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
      float nrm = (float)sqrt(acc);  // correctly rounded ||x|| in f32
      float scale = 1.0f;
      if (p.normalize && nrm != 0.0f) scale = nrm;  // divide by the norm (vector_index.py:277-280)
      s_scale[r] = scale;
      s_err[r] = 0.0f;
      s_err12[r] = 0.0f;
      s_n2[r] = 0.0f;
    }
  }
  __syncthreads();

  for (int k0 = 0; k0 < p.dimp; k0 += 128) {
    // stage 32 x 128 floats (normalised) through LDS, writing x32 on the way
    for (int i = tid; i < kTileRows * 128; i += 256) {
      const int r = i >> 7, c = i & 127;
      const int64_t grow = trow0 + r;
      const int64_t src = grow - p.row0;
      const int k = k0 + c;
      float v = 0.0f;
      const bool live = (src >= 0 && src < p.n);
      if (live && k < p.dim) {
        v = p.xin[src * p.dim + k] / s_scale[r];
        p.x32[grow * p.dim + k] = v;
      }
      s_x[r][c] = v;
    }
    __syncthreads();
    // 8 k-blocks x 64 lanes x 16 B
    for (int i = tid; i < 8 * 64; i += 256) {
      const int kbl = i >> 6, l = i & 63;
      const int r = l & 31, h = l >> 5;
      const int64_t grow = trow0 + r;
      const int64_t src = grow - p.row0;
      if (src >= 0 && src < p.n) {
        half8 hv;
        unsigned r12[8];
        float e2 = 0.0f, n2 = 0.0f, e12 = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = s_x[r][kbl * 16 + h * 8 + j];
          _Float16 q = (_Float16)v;
          hv[j] = q;
          float d = (float)q - v;
          if (fabsf(v) > 65504.0f) atomicOr(&p.stat[2], 1u);
          e2 += d * d;
          n2 += v * v;
          r12[j] = f16_to_12(q);
          const float d12 = (float)f12_value(r12[j]) - v;
          e12 += d12 * d12;
        }
        const int64_t blk = tile * p.kb + (k0 >> 4) + kbl;
        *reinterpret_cast<half8 *>(p.x16 + (blk * 64 + l) * 8) = hv;
        if (p.x12) {
          const X12 pk = x12_pack(r12);
          unsigned *o = p.x12 + (blk * 64 + l) * 3;
          o[0] = pk.h0;
          o[1] = pk.h1;
          o[2] = pk.nib;
          atomicAdd(&s_err12[r], e12);
        }
        atomicAdd(&s_err[r], e2);
        atomicAdd(&s_n2[r], n2);
      }
    }
    __syncthreads();
  }
  if (tid < kTileRows) {
    const int64_t grow = trow0 + tid;
    const int64_t src = grow - p.row0;
    if (src >= 0 && src < p.n) {
      // 1.0001: slack for the f32 accumulation of the partial sums above
      atomicMax(&p.stat[0], __float_as_uint(sqrtf(s_n2[tid]) * 1.0001f));
      atomicMax(&p.stat[1], __float_as_uint(sqrtf(s_err[tid]) * 1.0001f));
      if (p.x12) atomicMax(&p.stat[3], __float_as_uint(sqrtf(s_err12[tid]) * 1.0001f));
      if (p.rowbias) p.rowbias[grow] = -0.5f * s_n2[tid];
    }
  }
}

// the 12-bit image of rows that are already stored (ANR_OPT_SCAN_BITS switched to 12 on a filled index): one block per
// 32-row tile, from x16, the error against x32
struct Build12Params {
  const float *x32;
  const _Float16 *x16;
  unsigned *x12;
  int64_t n_rows;
  int dim, kb;
  unsigned *stat;  // [3] max ||x12 - x||
};
__global__ __launch_bounds__(256) void k_build12(Build12Params p) {
  __shared__ float s_err12[kTileRows];
  const int tid = threadIdx.x;
  const int64_t tile = blockIdx.x;
  if (tid < kTileRows) s_err12[tid] = 0.0f;
  __syncthreads();
  for (int i = tid; i < p.kb * 64; i += 256) {
    const int kbl = i >> 6, l = i & 63;
    const int r = l & 31, h = l >> 5;
    const int64_t grow = tile * kTileRows + r;
    const int64_t blk = tile * p.kb + kbl;
    const half8 hv = *reinterpret_cast<const half8 *>(p.x16 + (blk * 64 + l) * 8);
    unsigned r12[8];
    float e12 = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      r12[j] = f16_to_12(hv[j]);
      const int k = kbl * 16 + h * 8 + j;
      const float v = (grow < p.n_rows && k < p.dim) ? p.x32[grow * p.dim + k] : 0.0f;
      const float d12 = (float)f12_value(r12[j]) - v;
      e12 += d12 * d12;
    }
    const X12 pk = x12_pack(r12);
    unsigned *o = p.x12 + (blk * 64 + l) * 3;
    o[0] = pk.h0;
    o[1] = pk.h1;
    o[2] = pk.nib;
    if (grow < p.n_rows) atomicAdd(&s_err12[r], e12);
  }
  __syncthreads();
  if (tid < kTileRows && tile * kTileRows + tid < p.n_rows)
    atomicMax(&p.stat[3], __float_as_uint(sqrtf(s_err12[tid]) * 1.0001f));
}

// the image the streaming scan reads, decoded back to float32 rows (anr_index_reconstruct_scan_image: how the tests pin the
// 12-bit image and its tracked error norm against the oracle's restatement).  One thread per (row, 8 values).
struct ImageRowsParams {
  const _Float16 *x16;
  const unsigned *x12;  // nullptr: decode the f16 image
  int64_t i0, n;
  int dim, kb;
  float *out;           // [n][dim]
};
__global__ __launch_bounds__(256) void k_image_rows(ImageRowsParams p) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int groups = p.kb * 2;  // 8-value groups per row
  if (t >= p.n * groups) return;
  const int64_t row = p.i0 + t / groups;
  const int gidx = (int)(t % groups), kbl = gidx >> 1, h = gidx & 1;
  const int64_t tile = row / kTileRows;
  const int l = (int)(row % kTileRows) + 32 * h;
  const int64_t blk = tile * p.kb + kbl;
  half8 hv;
  if (p.x12) {
    const unsigned *w = p.x12 + (blk * 64 + l) * 3;
    hv = __builtin_bit_cast(half8, x12_unpack(w[0], w[1], w[2]));
  } else {
    hv = *reinterpret_cast<const half8 *>(p.x16 + (blk * 64 + l) * 8);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kbl * 16 + h * 8 + j;
    if (k < p.dim) p.out[(row - p.i0) * p.dim + k] = (float)hv[j];
  }
}

// ------------------------------------------------------------------------------------------------
// prepq: one block per query slot: normalise, keep f32 copy, build the blocked f16 B operand
// ------------------------------------------------------------------------------------------------
struct PrepQParams {
  const float *qin;  // [nq][dim]
  int nq, dim, dimp, kb, normalize;
  float *q32;        // [64][dimp]
  _Float16 *q16;     // [2][kb][64][8]
  float *qstat;      // [64][4]: ||q||, ||q16||, ||q16 - q||, ||q||^2
  unsigned long long *stamps;  // optional
};

__global__ __launch_bounds__(256) void k_prepq(PrepQParams p) {
  __shared__ double s_red[4];
  __shared__ float s_scale;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool live = q < p.nq;
  double acc = 0.0;
  if (live)
    for (int k = tid; k < p.dim; k += 256) {
      double v = (double)p.qin[(int64_t)q * p.dim + k];
      acc += v * v;
    }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) s_red[wave] = acc;
  __syncthreads();
  if (tid == 0) {
    float nrm = (float)sqrt(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    s_scale = (p.normalize && nrm != 0.0f) ? nrm : 1.0f;
  }
  __syncthreads();
  const float scale = s_scale;
  double n2 = 0.0, h2 = 0.0, e2 = 0.0;
  const int qb = q >> 5, col = q & 31;
  for (int k = tid; k < p.dimp; k += 256) {
    float v = (live && k < p.dim) ? p.qin[(int64_t)q * p.dim + k] / scale : 0.0f;
    p.q32[(int64_t)q * p.dimp + k] = v;
    _Float16 hq = (_Float16)v;
    const int kbl = k >> 4, h = (k >> 3) & 1, j = k & 7;
    p.q16[(((int64_t)qb * p.kb + kbl) * 64 + col + 32 * h) * 8 + j] = hq;
    double dv = v, dh = (double)(float)hq;
    n2 += dv * dv;
    h2 += dh * dh;
    e2 += (dh - dv) * (dh - dv);
  }
  for (int off = 32; off > 0; off >>= 1) {
    n2 += __shfl_xor(n2, off);
    h2 += __shfl_xor(h2, off);
    e2 += __shfl_xor(e2, off);
  }
  __shared__ double s_r3[4][3];
  if (lane == 0) {
    s_r3[wave][0] = n2;
    s_r3[wave][1] = h2;
    s_r3[wave][2] = e2;
  }
  __syncthreads();
  if (tid == 0) {
    double a = 0, b = 0, c = 0;
    for (int w = 0; w < 4; ++w) {
      a += s_r3[w][0];
      b += s_r3[w][1];
      c += s_r3[w][2];
    }
    p.qstat[q * 4 + 0] = (float)sqrt(a) * 1.0001f;
    p.qstat[q * 4 + 1] = (float)sqrt(b) * 1.0001f;
    p.qstat[q * 4 + 2] = (float)sqrt(c) * 1.0001f;
    p.qstat[q * 4 + 3] = (float)a;
    stamp_max(p.stamps, kStampPrepEnd);
  }
}

// ------------------------------------------------------------------------------------------------
// scan: the dominant kernel.  Streams the blocked f16 corpus once; per 32-row tile a wave runs
// KB x 2 MFMAs (32 rows x 64 queries), then either writes the score tile densely (DENSE) or appends
// the scores that pass the per-query running threshold to this block's per-query candidate list.
//
// Threshold ("ladder"): the sample phase leaves, per query, kLadder ascending values
// lad[0..L-1] = the sample's tile maxima of rank K', K'/2, K'/4, ... 1.  Every block starts at level
// lvl0, chosen on the host so that the sample predicts >= 4 K' rows at or above it in the whole corpus
// (and the level is backed by >= 12 sample tiles, so the prediction is not noise).  The threshold is a
// guess, not a bound: the select kernel counts what was emitted, and a query whose lists hold fewer than
// K' rows fails its certificate and takes the second pass.  While scanning, a block that has itself seen
// K' rows at or above a higher level moves up to it (valid: those K' rows exist) — this bounds the list
// growth on clustered corpora without any global traffic.  (An earlier version kept global per-level
// counters updated with device-scope atomics; those cost ~60 us per 1.25 M-row scan and the levels lagged
// a tile round behind, so it emitted 4x more candidates than the static start does.)
// ------------------------------------------------------------------------------------------------
struct ScanParams {
  const uint4 *x16;
  const unsigned *x12;  // the 12-bit image (k_scan<.., F12 = true> reads it instead of x16)
  const uint4 *q16;
  int kb;
  int64_t n_rows;
  int64_t tile0, tile_stride, n_tiles;  // tiles visited: tile0 + i*tile_stride, i < n_tiles
  const float *rowbias;                 // L2: rank = q.x - 0.5*||x||^2 ; nullptr for IP
  // DENSE
  float *dense;      // [64][dense_ld], column = i*32 + row-in-tile
  int64_t dense_ld;
  int groupmax;      // DENSE: write only the maximum of each 32-row tile: dense[q][i]
  // sparse
  const float *ladder;   // [64][kLadder] ascending thresholds from the sample
  int lvl0;              // ladder level every block starts at
  int *lvlmax;           // optional [64]: highest level any workgroup ended at (every row at or above that level's
                         // threshold was emitted by every workgroup — what a list-only recovery needs to know)
  unsigned *cntb;        // [64][gridDim.x] list lengths, written when a block retires
  uint2 *cand;           // [gridDim.x][64][capb] (rank-score bits, row)
  unsigned capb;
  unsigned kprime;
  unsigned long long *stamps;  // optional (main scan only)
};

// STREAM: every byte is read once per batch and the corpus is far larger than the 256-MiB Infinity Cache,
// so the loads carry the non-temporal hint (measured: 6.3 -> 6.7 TB/s on the 15-GB scan).  Small corpora
// and the threshold sample keep the default policy and stay cache-resident between batches.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool STREAM>
__device__ __forceinline__ uint4 ld16(const uint4 *p) {
  if (STREAM) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  return *p;
}

// The corpus operand of the scan as it sits in registers: F12 = false, the 16-byte f16 block as loaded; F12 = true, the
// three words of the 12-bit image, unpacked (x12_unpack) right before the block's two MFMAs.  `xa` is the lane's pointer
// into the first k-block of the tile, in units of one lane's share of a block (a uint4, or three words).
template <bool F12>
struct ScanOp;
template <>
struct ScanOp<false> {
  uint4 v;
  static constexpr int kLaneBytes = 16;
  template <bool STREAM>
  __device__ __forceinline__ void load(const char *base, int j) {  // k-block j after `base` (this lane's share of it)
    v = ld16<STREAM>(reinterpret_cast<const uint4 *>(base) + (int64_t)j * 64);
  }
  __device__ __forceinline__ half8 operand() const { return __builtin_bit_cast(half8, v); }
};
template <>
struct ScanOp<true> {
  unsigned h0, h1, nib;
  static constexpr int kLaneBytes = 12;
  template <bool STREAM>
  __device__ __forceinline__ void load(const char *base, int j) {
    const unsigned *p = reinterpret_cast<const unsigned *>(base) + (int64_t)j * 64 * 3;
    if (STREAM) {
      h0 = __builtin_nontemporal_load(p);
      h1 = __builtin_nontemporal_load(p + 1);
      nib = __builtin_nontemporal_load(p + 2);
    } else {
      h0 = p[0];
      h1 = p[1];
      nib = p[2];
    }
  }
  __device__ __forceinline__ half8 operand() const { return __builtin_bit_cast(half8, x12_unpack(h0, h1, nib)); }
};

template <int CH, bool STREAM, bool F12>
__device__ __forceinline__ void scan_load(ScanOp<F12> (&a)[CH], const char *base) {
#pragma unroll
  for (int j = 0; j < CH; ++j) a[j].template load<STREAM>(base, j);
}

template <int CH, bool F12>
__device__ __forceinline__ void scan_mfma(const ScanOp<F12> (&a)[CH], const uint4 *ldsq0, const uint4 *ldsq1,
                                          int kbase, floatx16 &acc0, floatx16 &acc1) {
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const uint4 b0 = ldsq0[(kbase + j) * 64];
    const uint4 b1 = ldsq1[(kbase + j) * 64];
    const half8 av = a[j].operand();
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(half8, b0), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(half8, b1), acc1, 0, 0, 0);
  }
}

// same, and re-fills each operand register from `next` as soon as its two MFMAs have issued
template <int CH, bool STREAM, bool F12>
__device__ __forceinline__ void scan_mfma_refill(ScanOp<F12> (&a)[CH], const char *next, const uint4 *ldsq0,
                                                 const uint4 *ldsq1, int kbase, floatx16 &acc0, floatx16 &acc1) {
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const uint4 b0 = ldsq0[(kbase + j) * 64];
    const uint4 b1 = ldsq1[(kbase + j) * 64];
    const half8 av = a[j].operand();
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(half8, b0), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(half8, b1), acc1, 0, 0, 0);
    a[j].template load<STREAM>(next, j);
  }
}

__device__ __forceinline__ unsigned ld_relaxed(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// append this lane's hits of one 32x32 accumulator block (all for query q) to the block's list and
// count them against the next three ladder levels in the block's pending counters (LDS).
// Hits are rare (a handful per tile), so the lane first builds a 16-bit hit mask without branches and
// then the wave loops max-hits-per-lane times (usually once), each lane peeling its lowest hit.
__device__ __forceinline__ void scan_emit(const floatx16 &acc, float tau, int lvl, int q, int64_t row_base,
                                          unsigned valid, const float *lad, unsigned *lds_cnt,
                                          unsigned *lds_pend, const ScanParams &p) {
  unsigned m = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) m |= (acc[r] >= tau) ? (1u << r) : 0u;
  m &= valid;
  if (!__any(m != 0)) return;
  unsigned k = 0;
  if (m) k = atomicAdd(lds_cnt + q, (unsigned)__popc(m));  // LDS: slot reservation inside the block
  const int ja = lvl + 1 < kLadder ? lvl + 1 : kLadder - 1;
  const int jb = lvl + 2 < kLadder ? lvl + 2 : kLadder - 1;
  const int jc = lvl + 3 < kLadder ? lvl + 3 : kLadder - 1;
  const float la = lad[q * kLadder + ja], lb = lad[q * kLadder + jb], lc = lad[q * kLadder + jc];
  unsigned ca = 0, cb = 0, cc = 0;
  uint2 *list = p.cand + ((int64_t)blockIdx.x * kQB + q) * p.capb;
  while (__any(m != 0)) {
    const int r = __ffs(m) - 1;  // -1 on idle lanes
    float s = acc[0];
#pragma unroll
    for (int t = 1; t < 16; ++t) s = (r == t) ? acc[t] : s;
    if (m) {
      if (k < p.capb) list[k] = make_uint2(__float_as_uint(s), (unsigned)(row_base + (r & 3) + 8 * (r >> 2)));
      ++k;
      ca += (s >= la) ? 1u : 0u;
      cb += (s >= lb) ? 1u : 0u;
      cc += (s >= lc) ? 1u : 0u;
    }
    m &= m - 1;
  }
  // levels further than three above the current one are under-counted (safe, see header comment)
  if (lvl + 1 < kLadder && ca) atomicAdd(lds_pend + q * kLadder + lvl + 1, ca);
  if (lvl + 2 < kLadder && cb) atomicAdd(lds_pend + q * kLadder + lvl + 2, cb);
  if (lvl + 3 < kLadder && cc) atomicAdd(lds_pend + q * kLadder + lvl + 3, cc);
}

#if defined(ANR_SCAN_VGPR_CAP)  // experiment: cap the scan's registers so that a side kernel's wave fits beside three of its waves
#define ANR_SCAN_ATTR __attribute__((amdgpu_waves_per_eu(512 / ANR_SCAN_VGPR_CAP, 512 / ANR_SCAN_VGPR_CAP)))
#else
#define ANR_SCAN_ATTR
#endif
template <bool DENSE, int CH, int NT, bool STREAM, bool F12 = false>
__global__ __launch_bounds__(NT) ANR_SCAN_ATTR void k_scan(ScanParams p) {
  // LDS: Q operand image [2][kb][64] | ladder [64][L] | list lengths [64] | level [64] | pending [64][L]
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x, nwaves = nthreads >> 6;
  const int nq16 = 2 * p.kb * 64;
  for (int i = tid; i < nq16; i += nthreads) lds[i] = p.q16[i];
  float *lad = reinterpret_cast<float *>(lds + nq16);
  unsigned *lds_cnt = reinterpret_cast<unsigned *>(lad + kQB * kLadder);
  int *lds_lvl = reinterpret_cast<int *>(lds_cnt + kQB);
  unsigned *lds_pend = reinterpret_cast<unsigned *>(lds_lvl + kQB);
  if (!DENSE) {
    if (tid == 0 && p.stamps) {
      stamp_min(p.stamps, kStampScanStart);
      stamp_max(p.stamps, kStampScanPlaced);
    }
    for (int i = tid; i < kQB * kLadder; i += nthreads) {
      lad[i] = p.ladder[i];
      lds_pend[i] = 0;
    }
    if (tid < kQB) {
      lds_cnt[tid] = 0;
      lds_lvl[tid] = p.lvl0;
    }
  }
  __syncthreads();

  const uint4 *ldsq0 = lds + lane;
  const uint4 *ldsq1 = lds + p.kb * 64 + lane;
  const int q0 = lane & 31, half = lane >> 5;
  const int64_t wglobal = (int64_t)blockIdx.x * nwaves + wave;
  const int64_t wtotal = (int64_t)gridDim.x * nwaves;
  const int nch = p.kb / CH;  // even by construction (kb % (2*CH) == 0)

  // the first chunk of a wave's next tile is requested before the current tile's epilogue, so the wave
  // always has loads in flight (the epilogue would otherwise be a bubble in its share of the stream)
  ScanOp<F12> aA[CH], aB[CH];
  // the lane's share of one k-block is LB bytes (16: f16 image, 12: 12-bit image); a tile is kb such blocks of 64 lanes
  constexpr int64_t LB = ScanOp<F12>::kLaneBytes, BLK = 64 * LB;
  const char *xlane = (F12 ? reinterpret_cast<const char *>(p.x12) : reinterpret_cast<const char *>(p.x16)) + lane * LB;
  if (wglobal < p.n_tiles) scan_load<CH, STREAM, F12>(aA, xlane + (p.tile0 + wglobal * p.tile_stride) * p.kb * BLK);
  for (int64_t i = wglobal; i < p.n_tiles; i += wtotal) {
    const int64_t tile = p.tile0 + i * p.tile_stride;
    const char *xa = xlane + tile * p.kb * BLK;
    // no next tile: re-request this tile's last chunk instead (an L2 hit), which keeps the loop free of
    // conditional loads so the compiler can count the outstanding loads exactly
    const char *xn = (i + wtotal < p.n_tiles) ? xa + wtotal * p.tile_stride * p.kb * BLK : xa + (int64_t)(p.kb - CH) * BLK;
    floatx16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = 0.0f;
      acc1[r] = 0.0f;
    }
    for (int c = 0; c < nch; c += 2) {
      scan_load<CH, STREAM, F12>(aB, xa + (int64_t)(c + 1) * CH * BLK);
      __builtin_amdgcn_sched_barrier(0);
      scan_mfma_refill<CH, STREAM, F12>(aA, (c + 2 < nch) ? xa + (int64_t)(c + 2) * CH * BLK : xn, ldsq0, ldsq1, c * CH, acc0, acc1);
      __builtin_amdgcn_sched_barrier(0);
      scan_mfma<CH, F12>(aB, ldsq0, ldsq1, (c + 1) * CH, acc0, acc1);
    }

    const int64_t row_base = tile * kTileRows + 4 * half;
    if (p.rowbias) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4 *>(p.rowbias + row_base + 8 * g);
        acc0[4 * g + 0] += b.x; acc0[4 * g + 1] += b.y; acc0[4 * g + 2] += b.z; acc0[4 * g + 3] += b.w;
        acc1[4 * g + 0] += b.x; acc1[4 * g + 1] += b.y; acc1[4 * g + 2] += b.z; acc1[4 * g + 3] += b.w;
      }
    }

    if (DENSE && p.groupmax) {
      // threshold sample: one value per (32-row tile, query) — the tile's best score
      float m0 = acc0[0], m1 = acc1[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) {
        m0 = fmaxf(m0, acc0[r]);
        m1 = fmaxf(m1, acc1[r]);
      }
      m0 = fmaxf(m0, __shfl_xor(m0, 32));
      m1 = fmaxf(m1, __shfl_xor(m1, 32));
      if (half == 0) {
        p.dense[(int64_t)q0 * p.dense_ld + i] = m0;
        p.dense[(int64_t)(q0 + 32) * p.dense_ld + i] = m1;
      }
    } else if (DENSE) {
      const float ninf = -__builtin_inff();
      float *d0 = p.dense + (int64_t)q0 * p.dense_ld + i * kTileRows + 4 * half;
      float *d1 = d0 + 32 * p.dense_ld;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t r0 = row_base + 8 * g;
        float4 v0, v1;
        v0.x = (r0 + 0 < p.n_rows) ? acc0[4 * g + 0] : ninf;
        v0.y = (r0 + 1 < p.n_rows) ? acc0[4 * g + 1] : ninf;
        v0.z = (r0 + 2 < p.n_rows) ? acc0[4 * g + 2] : ninf;
        v0.w = (r0 + 3 < p.n_rows) ? acc0[4 * g + 3] : ninf;
        v1.x = (r0 + 0 < p.n_rows) ? acc1[4 * g + 0] : ninf;
        v1.y = (r0 + 1 < p.n_rows) ? acc1[4 * g + 1] : ninf;
        v1.z = (r0 + 2 < p.n_rows) ? acc1[4 * g + 2] : ninf;
        v1.w = (r0 + 3 < p.n_rows) ? acc1[4 * g + 3] : ninf;
        *reinterpret_cast<float4 *>(d0 + 8 * g) = v0;
        *reinterpret_cast<float4 *>(d1 + 8 * g) = v1;
      }
    } else {
      // the block's current ladder level of this lane's two queries (LDS, shared by all waves)
      const int lv0 = __hip_atomic_load(lds_lvl + q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const int lv1 = __hip_atomic_load(lds_lvl + q0 + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const float tau0 = lad[q0 * kLadder + lv0], tau1 = lad[(q0 + 32) * kLadder + lv1];
      unsigned valid = 0xffffu;
      if ((tile + 1) * kTileRows > p.n_rows) {  // the ragged last tile
        valid = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) valid |= (row_base + (r & 3) + 8 * (r >> 2) < p.n_rows) ? (1u << r) : 0u;
      }
      scan_emit(acc0, tau0, lv0, q0, row_base, valid, lad, lds_cnt, lds_pend, p);
      scan_emit(acc1, tau1, lv1, q0 + 32, row_base, valid, lad, lds_cnt, lds_pend, p);
      // a level at which this block alone has seen K' rows is a valid threshold: raise to it
      for (int e = tid; e < kQB * kLadder; e += nthreads) {
        const int q = e >> 3, j = e & (kLadder - 1);
        if (lds_pend[q * kLadder + j] >= p.kprime) atomicMax(lds_lvl + q, j);
      }
    }
  }
  if (!DENSE) {
    __syncthreads();
    if (tid < kQB) p.cntb[(int64_t)tid * gridDim.x + blockIdx.x] = lds_cnt[tid];
    if (tid < kQB && p.lvlmax && lds_lvl[tid] > p.lvl0) atomicMax(p.lvlmax + tid, lds_lvl[tid]);
    if (tid == 0) stamp_max(p.stamps, kStampScanEnd);
  }
}

// ------------------------------------------------------------------------------------------------
// sample (shadow form of k_scan<DENSE> + groupmax): the threshold sample of batch i + 1 computed BESIDE the resident
// workgroups of batch i's scan instead of after them.  A scan workgroup leaves a CU 56 vector registers per SIMD, five wave
// slots per SIMD and 59.5 KiB of LDS in up to two pieces: this kernel's workgroup is 4 waves (one per SIMD) of <= 56
// registers and <= 28 KiB of LDS.  Workgroup b serves query half b & 1 (32 queries); in a round each of its waves
// multiplies ONE sampled 32-row tile with those 32 queries, the k range cut into slices of kbs blocks whose B operand
// (kbs KiB) is staged in LDS one after the other — kb MFMAs per tile in the same k order as the scan, so the accumulators,
// hence the tile maxima and the ladder, are bit-identical to k_scan<true>'s — and writes the tile's best score per query.
// ------------------------------------------------------------------------------------------------
struct SampleParams {
  const uint4 *x16;
  const uint4 *q16;
  int kb, kbs;                   // k-blocks in all and per LDS slice (kb % kbs == 0, kbs % CH == 0, kbs KiB <= 28 KiB)
  int64_t tile_stride, n_tiles;  // sampled tile i = corpus tile i * tile_stride (all full tiles)
  const float *rowbias;
  float *dense;                  // [64][dense_ld]: dense[q][i] = best score of sampled tile i
  int64_t dense_ld;
  unsigned long long *stamps;    // optional
};

// (a plain __device__ function: the builtin written directly inside a __global__ template breaks the host-side instantiation)
__device__ __forceinline__ void sample_glds16(const uint4 *g, uint4 *l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }

template <int CH, bool BIAS>
__global__ __launch_bounds__(256) void k_sample(SampleParams p) {
  extern __shared__ uint4 lds[];  // [kbs][64] B operand slice of this workgroup's query half
  // (wave index as a scalar: tile addresses are scalar base + lane offset, which is what keeps the kernel within 56 registers)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qh = blockIdx.x & 1;
  const uint4 *qsrc = p.q16 + (int64_t)qh * p.kb * 64;
  const uint4 *ldsq = lds + lane;
  const int q0 = lane & 31, half = lane >> 5;
  const int64_t w0 = (int64_t)(blockIdx.x >> 1) * 4 + wave, wt = (int64_t)(gridDim.x >> 1) * 4;
  const int64_t rounds = (p.n_tiles - (int64_t)(blockIdx.x >> 1) * 4 + wt - 1) / wt;  // of the workgroup's FIRST wave: all waves keep its barriers
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t i = w0 + r * wt;
    const bool live = i < p.n_tiles;
    const int64_t tile = (live ? i : w0) * p.tile_stride;  // a wave without a tile in the last round re-reads its first one
    const uint4 *xa = p.x16 + tile * p.kb * 64 + lane;
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    // CH operand registers: each is re-filled with the block CH k-steps ahead as soon as its MFMA has issued
    uint4 a[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) a[j] = xa[(int64_t)j * 64];
    // ONE loop over the k-steps (a nest of slice and step loops doubled the accumulator registers); at a slice boundary
    // the workgroup swaps the LDS slice
    for (int kc = 0, c = p.kbs; kc < p.kb; kc += CH, c += CH) {
      if (c == p.kbs) {
        c = 0;
        __syncthreads();  // the previous slice (or round) has been consumed
        // global -> LDS copies that bypass the registers (a wave's 64 x 16 B land lane-linear at the uniform LDS address)
        for (int blk = wave; blk < p.kbs; blk += 4) sample_glds16(qsrc + ((int64_t)kc + blk) * 64 + lane, lds + blk * 64);
        __syncthreads();
      }
      const int kn = kc + CH < p.kb ? kc + CH : p.kb - CH;  // past the end: re-request the last blocks (an L2 hit)
      const uint4 *nx = xa + (int64_t)kn * 64;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const uint4 b = ldsq[(c + j) * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a[j]), __builtin_bit_cast(half8, b), acc, 0, 0, 0);
        a[j] = nx[j * 64];
      }
    }
    if (BIAS) {  // L2 metric: rank = q.x - 0.5 ||x||^2
      const float *rb = p.rowbias + tile * kTileRows + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bb = *reinterpret_cast<const float4 *>(rb + 8 * g);
        acc[4 * g + 0] += bb.x; acc[4 * g + 1] += bb.y; acc[4 * g + 2] += bb.z; acc[4 * g + 3] += bb.w;
        asm volatile("" ::: "memory");  // one group at a time (register budget)
      }
    }
    float m = acc[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) m = fmaxf(m, acc[e]);
    m = fmaxf(m, __shfl_xor(m, 32));
    if (live && half == 0) p.dense[(int64_t)(q0 + 32 * qh) * p.dense_ld + i] = m;
  }
  if (tid == 0) stamp_max(p.stamps, kStampSampleEnd);
}

// ------------------------------------------------------------------------------------------------
// select: one block per query; radix-select the M largest 64-bit keys (rank desc, row asc) from a
// dense score row or from the per-block candidate lists, sort them, write (rank, row) and optionally
// the threshold ladder.  Keys are staged in LDS when they fit (the usual case).
// ------------------------------------------------------------------------------------------------
constexpr int kSelLds = 16384;   // keys staged in LDS (128 KiB)
constexpr int kSelMaxLists = 1024;

struct SelParams {
  // dense input (dense != nullptr): n entries per query, entry i -> row = (i/32)*row_tile_stride*32 + row0 + i%32
  const float *dense;
  int64_t dense_ld;
  int64_t n;
  int64_t row0, row_tile_stride;
  int negate;            // rank = -value (ascending select, L2 distances)
  // list input
  const uint2 *cand;     // [G][64][capb]
  const unsigned *cntb;  // [64][G]
  int G;
  unsigned capb;
  int M;                 // entries wanted (<= kMaxSel)
  // outputs, [64][kMaxSel]
  float *out_rank;
  unsigned *out_row;
  int *out_m;            // entries written per query
  float *ladder;         // optional [64][kLadder]
  unsigned *overflow;    // optional [64]: set when a list overflowed
  unsigned *ncand;       // optional [64]: candidates seen
  const int *qslots;     // optional: block b handles query slot qslots[b]
  int *lvl_init;         // optional (ladder mode): lvl_init[q] = lvl_init_value, the scan's start level
  int lvl_init_value;
  unsigned long long *stamps;  // optional: ladder mode stamps kStampLadderEnd (and resets the scan's start), else kStampSelEnd
  int live_q;            // ladder mode, > 0: query slots >= live_q carry no query — their ladder is +inf, so the scan
                         // never emits for them (a zero query would otherwise pass its own all-zero thresholds on
                         // every row of every tile)
};

// KEYS = keys a workgroup can stage in LDS, NSEL = entries it can return, NLISTS = candidate lists it can read.  Two sizes:
// the wide kernels (1024 threads, 150 KiB: a CU to themselves) and the "shadow" kernels (256 threads, <= 56 vector
// registers, 26 KiB) that are placed on a CU BESIDE a resident scan workgroup — which leaves 56 registers per SIMD and
// 59.5 KiB of LDS, the latter possibly in TWO pieces (the scan workgroup sits wherever the side workgroups resident at
// ITS placement left room), so that only a request of <= 29 KiB is certain to fit: with 53 KiB the select waited for the
// scan to end in half of the batches (r4 notes in DESIGN.md §4).
template <int KEYS, int NSEL, int NLISTS>
struct SelSharedT {
  static constexpr int kKeys = KEYS, kSel = NSEL, kLists = NLISTS;
  unsigned long long keys[KEYS];
  unsigned long long sel[NSEL];
  unsigned long long srt[NSEL];
  unsigned long long red[16][2];
  unsigned offs[NLISTS + 1];
  unsigned hist[256];
  unsigned cnt;
  int d;
  unsigned above, h, ovf;
};
typedef SelSharedT<kSelLds, kMaxSel, kSelMaxLists> SelShared;
constexpr int kShadowKeys = 2048, kShadowSel = 512, kShadowLists = 256;
typedef SelSharedT<kShadowKeys, kShadowSel, kShadowLists> SelSharedShadow;
constexpr int kShadowLds = 29 * 1024;  // what a shadow kernel may ask for
static_assert(sizeof(SelSharedShadow) <= kShadowLds, "the shadow select must fit into either piece of the LDS a scan workgroup leaves");

template <class SH>
__device__ __forceinline__ unsigned long long sel_key(const SelParams &p, const SH &sh, int q, int64_t i,
                                                      bool staged, int koff = 0) {
  if (staged) return sh.keys[koff + i];
  if (p.dense) {
    float v = p.dense[(int64_t)q * p.dense_ld + i];
    if (p.negate) v = -v;
    const int64_t row = (i >> 5) * p.row_tile_stride * 32 + p.row0 + (i & 31);
    return make_key(v, (unsigned)row);
  }
  // list entry i: binary search the block whose range holds i
  int lo = 0, hi = p.G;  // offs[lo] <= i < offs[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (sh.offs[mid] <= (unsigned)i) lo = mid;
    else hi = mid;
  }
  const uint2 c = p.cand[((int64_t)lo * kQB + q) * p.capb + ((unsigned)i - sh.offs[lo])];
  return make_key(__uint_as_float(c.x), c.y);
}

// The selection itself: leaves the M best keys, ordered best first, in sh.srt[0, M) and returns M (the same value in
// every thread; 0 when the query has no entry).  All threads of the workgroup call it.
template <class SH>
__device__ __forceinline__ int sel_run(const SelParams &p, SH &sh, const int q) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = blockDim.x, NW = NT >> 6;
  int64_t n;
  if (p.dense) {
    n = p.n;
  } else {
    if (tid == 0) sh.ovf = 0;
    __syncthreads();
    for (int b = tid; b < p.G; b += NT) {
      unsigned c = p.cntb[(int64_t)q * p.G + b];
      if (c > p.capb) {
        sh.ovf = 1;
        c = p.capb;
      }
      sh.offs[b + 1] = c;
    }
    __syncthreads();
    if (tid == 0) sh.offs[0] = 0;
    for (int base_i = 0; base_i < p.G; base_i += NT) {
      // inclusive scan of one chunk of list lengths: wave scan + wave totals, carried across chunks
      const int i = base_i + tid;
      unsigned v = i < p.G ? sh.offs[i + 1] : 0u;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(v, off);
        if (lane >= off) v += t;
      }
      if (lane == 63) sh.hist[wave] = v;
      __syncthreads();
      unsigned carry = sh.offs[base_i];
      for (int w = 0; w < wave; ++w) carry += sh.hist[w];
      __syncthreads();
      if (i < p.G) sh.offs[i + 1] = v + carry;
      __syncthreads();
    }
    n = sh.offs[p.G];
    if (tid == 0) {
      if (p.overflow) p.overflow[q] = sh.ovf;
      if (p.ncand) p.ncand[q] = (unsigned)n;
    }
  }
  int M = (int)((int64_t)p.M < n ? (int64_t)p.M : n);
  if (M == 0) return 0;
  const bool staged = n <= SH::kKeys;
  if (tid == 0) sh.cnt = 0;
  __syncthreads();
  // stage the keys in LDS (when they fit) and find the common leading bytes of all keys on the way -> those
  // radix passes are skipped (this also avoids one-bin LDS atomic storms)
  unsigned long long kmin = ~0ull, kmax = 0ull;
  if (staged) {
    if (p.dense) {
      for (int64_t i = tid; i < n; i += NT) {
        const unsigned long long k = sel_key(p, sh, q, i, false);
        sh.keys[i] = k;
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
      }
    } else {
      // one thread per candidate over the flattened lists, so every list read is in flight at once
      for (int64_t i = tid; i < n; i += NT) {
        int lo = 0, hi = p.G;  // offs[lo] <= i < offs[hi]
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (sh.offs[mid] <= (unsigned)i) lo = mid;
          else hi = mid;
        }
        const uint2 e = p.cand[((int64_t)lo * kQB + q) * p.capb + ((unsigned)i - sh.offs[lo])];
        const unsigned long long k = make_key(__uint_as_float(e.x), e.y);
        sh.keys[i] = k;
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
      }
    }
  } else {
    for (int64_t i = tid; i < n; i += NT) {
      const unsigned long long k = sel_key(p, sh, q, i, false);
      kmin = k < kmin ? k : kmin;
      kmax = k > kmax ? k : kmax;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long a = __shfl_xor(kmin, off), b = __shfl_xor(kmax, off);
    kmin = a < kmin ? a : kmin;
    kmax = b > kmax ? b : kmax;
  }
  if (lane == 0) {
    sh.red[wave][0] = kmin;
    sh.red[wave][1] = kmax;
  }
  __syncthreads();
  int koff = 0;
  kmin = sh.red[0][0];
  kmax = sh.red[0][1];
  for (int w = 1; w < NW; ++w) {
    kmin = sh.red[w][0] < kmin ? sh.red[w][0] : kmin;
    kmax = sh.red[w][1] > kmax ? sh.red[w][1] : kmax;
  }
  if (staged && n > 256 && n <= SH::kKeys / 2) {
    // shrink the ranking problem: 256 linear score bins between the smallest and largest staged score, keep the
    // keys from the top bins that together hold >= M keys (typically M plus a few dozen)
    const float smin = ord2f((unsigned)(kmin >> 32)), smax = ord2f((unsigned)(kmax >> 32));
    const float scale = smax > smin ? 255.0f / (smax - smin) : 0.0f;
    for (int i = tid; i < 256; i += NT) sh.hist[i] = 0;
    if (tid == 0) sh.cnt = 0;
    __syncthreads();
    for (int64_t i = tid; i < n; i += NT) {
      const float sc = ord2f((unsigned)(sh.keys[i] >> 32));
      int bin = (int)((sc - smin) * scale);
      bin = bin < 0 ? 0 : (bin > 255 ? 255 : bin);
      atomicAdd(&sh.hist[bin], 1u);
    }
    __syncthreads();
    if (wave == 0) {
      const int dtop = 255 - 4 * lane;
      const unsigned h0 = sh.hist[dtop], h1 = sh.hist[dtop - 1], h2 = sh.hist[dtop - 2], h3 = sh.hist[dtop - 3];
      const unsigned own = h0 + h1 + h2 + h3;
      unsigned incl = own;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
      }
      const unsigned excl = incl - own, need = (unsigned)M;
      if (excl < need && need <= incl) {
        unsigned c = excl;
        int d = dtop;
        if (c + h0 < need) { c += h0; d = dtop - 1;
          if (c + h1 < need) { c += h1; d = dtop - 2;
            if (c + h2 < need) { d = dtop - 3; } } }
        sh.d = d;
      }
    }
    __syncthreads();
    const int cut = sh.d;
    for (int64_t i = tid; i < n; i += NT) {
      const unsigned long long k = sh.keys[i];
      const float sc = ord2f((unsigned)(k >> 32));
      int bin = (int)((sc - smin) * scale);
      bin = bin < 0 ? 0 : (bin > 255 ? 255 : bin);
      if (bin >= cut) sh.keys[SH::kKeys / 2 + atomicAdd(&sh.cnt, 1u)] = k;
    }
    __syncthreads();
    n = sh.cnt;
    koff = SH::kKeys / 2;
    __syncthreads();
  }
  if (staged && n <= SH::kSel) {
    // small input: rank every key directly (distinct keys), n*n/NT compares per thread
    for (int i = tid; i < (int)n; i += NT) {
      const unsigned long long kk = sh.keys[koff + i];
      int rank = 0;
#pragma unroll 16
      for (int j = 0; j < (int)n; ++j) rank += (sh.keys[koff + j] > kk) ? 1 : 0;
      if (rank < M) sh.sel[rank] = kk;
    }
    __syncthreads();
    for (int i = tid; i < M; i += NT) sh.srt[i] = sh.sel[i];
    __syncthreads();
  } else {
  int bits = 0;
  while (bits < 64 && (kmin >> (56 - bits)) == (kmax >> (56 - bits))) bits += 8;
  unsigned long long prefix = bits ? (kmax >> (64 - bits)) : 0ull;
  unsigned need = (unsigned)M;
  bool whole = (bits == 64);
  while (!whole && bits < 64) {
    for (int i = tid; i < 256; i += NT) sh.hist[i] = 0;
    __syncthreads();
    for (int64_t i = tid; i < n; i += NT) {
      const unsigned long long k = sel_key(p, sh, q, i, staged, koff);
      if (bits == 0 || (k >> (64 - bits)) == prefix) atomicAdd(&sh.hist[(unsigned)(k >> (56 - bits)) & 255u], 1u);
    }
    __syncthreads();
    if (wave == 0) {
      // lane l owns digits 255-4l .. 252-4l; find the digit where the count from the top reaches `need`
      const int dtop = 255 - 4 * lane;
      const unsigned h0 = sh.hist[dtop], h1 = sh.hist[dtop - 1], h2 = sh.hist[dtop - 2], h3 = sh.hist[dtop - 3];
      const unsigned own = h0 + h1 + h2 + h3;
      unsigned incl = own;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
      }
      const unsigned excl = incl - own;
      if (excl < need && need <= incl) {
        unsigned c = excl;
        int d = dtop;
        unsigned hd = h0;
        if (c + h0 < need) { c += h0; d = dtop - 1; hd = h1;
          if (c + h1 < need) { c += h1; d = dtop - 2; hd = h2;
            if (c + h2 < need) { c += h2; d = dtop - 3; hd = h3; } } }
        sh.d = d;
        sh.above = c;
        sh.h = hd;
      }
    }
    __syncthreads();
    need -= sh.above;
    prefix = (prefix << 8) | (unsigned long long)sh.d;
    bits += 8;
    if (sh.h == need) whole = true;  // every key under this prefix is wanted
    __syncthreads();
  }
  // gather keys whose leading `bits` bits are >= prefix: exactly M of them
  if (tid == 0) sh.cnt = 0;
  __syncthreads();
  for (int64_t i = tid; i < n; i += NT) {
    const unsigned long long k = sel_key(p, sh, q, i, staged, koff);
    if (bits == 0 || (k >> (64 - bits)) >= prefix) {
      const unsigned pos = atomicAdd(&sh.cnt, 1u);
      if (pos < (unsigned)SH::kSel) sh.sel[pos] = k;
    }
  }
  __syncthreads();
  // order the M gathered keys by rank counting (keys are distinct): M*M/1024 compares per thread, one barrier
  for (int i = tid; i < M; i += NT) {
    const unsigned long long k = sh.sel[i];
    int rank = 0;
#pragma unroll 16
    for (int j = 0; j < M; ++j) rank += (sh.sel[j] > k) ? 1 : 0;
    sh.srt[rank] = k;
  }
  __syncthreads();
  }
  return M;
}

// write a selection out: (rank value, row) pairs, their count, and (ladder mode) the threshold ladder
template <class SH>
__device__ __forceinline__ void sel_write(const SelParams &p, const SH &sh, const int q, const int M) {
  const int tid = threadIdx.x, NT = blockDim.x;
  if (tid == 0) p.out_m[q] = M;
  for (int i = tid; i < M; i += NT) {
    const unsigned long long k = sh.srt[i];
    float v = ord2f((unsigned)(k >> 32));
    p.out_rank[q * kMaxSel + i] = p.negate ? -v : v;
    p.out_row[q * kMaxSel + i] = 0xffffffffu - (unsigned)(k & 0xffffffffu);
  }
  if (p.lvl_init && tid == 0) p.lvl_init[q] = p.lvl_init_value;
  if (p.ladder && tid < kLadder) {
    if (M == 0) {
      p.ladder[q * kLadder + tid] = -__builtin_inff();
    } else {
      // level j = value of rank max(1, M >> j) (1-based) of the sample: ascending in j
      int rk = M >> tid;
      if (rk < 1 || tid == kLadder - 1) rk = 1;
      p.ladder[q * kLadder + tid] = ord2f((unsigned)(sh.srt[rk - 1] >> 32));
    }
  }
}

template <class SH>
__device__ __forceinline__ void select_body(const SelParams &p, SH &sh) {
  const int tid = threadIdx.x;
  const int q = p.qslots ? p.qslots[blockIdx.x] : blockIdx.x;
  if (p.ladder && p.live_q > 0 && q >= p.live_q) {
    if (tid == 0) p.out_m[q] = 0;
    if (tid < kLadder) p.ladder[q * kLadder + tid] = __builtin_inff();
    if (p.lvl_init && tid == 0) p.lvl_init[q] = p.lvl_init_value;
    return;
  }
  const int M = sel_run(p, sh, q);
  sel_write(p, sh, q, M);
  if (tid == 0 && p.stamps) {
    if (p.ladder && blockIdx.x == 0) p.stamps[kStampScanStart] = ~0ull;  // the scan that follows takes the minimum
    stamp_max(p.stamps, p.ladder ? kStampLadderEnd : kStampSelEnd);
  }
}

__global__ __launch_bounds__(1024) void k_select(SelParams p) {
  extern __shared__ unsigned char sel_smem[];
  select_body(p, *reinterpret_cast<SelShared *>(sel_smem));
}

// the same selection sized to run in the shadow of a resident scan (DESIGN.md §4 "shadow kernels"): 4 waves, <= 56
// registers, 26 KiB; M <= 512 entries out of <= 256 lists (the host checks); more than 2048 entries are ranked from global
// memory (the radix passes re-read them)
__global__ __launch_bounds__(256) void k_select_shadow(SelParams p) {
  extern __shared__ unsigned char sel_smem[];
  select_body(p, *reinterpret_cast<SelSharedShadow *>(sel_smem));
}

// ------------------------------------------------------------------------------------------------
// post: select + exact re-score + finalize of one query in ONE workgroup (the three launches k_select, k_rescore,
// k_finalize of the pipeline fused: two launch gaps and two round trips through HBM less per batch).  After the
// selection the M candidate rows are re-scored exactly by the workgroup's 16 waves (f32 rows, f64 accumulation in
// k_rescore's order: bit-identical values), ranked by (exact, id), written out as the top-k, and thread 0 decides the
// certificate exactly as k_finalize does.
// ------------------------------------------------------------------------------------------------
struct PostParams {
  const float *x32;
  const float *q32;          // [64][dimp]
  int dim, dimp, metric;
  const float *qstat;        // [64][4]
  const unsigned *xstat;
  const unsigned *overflow;  // may be null: [64] set by the selection when a candidate list overflowed
  const unsigned *ncand;     // may be null
  int64_t n_rows;
  int Mreq, k, nq;
  int64_t out_off;
  float *D;
  int64_t *I;
  int *flags;
  float *theta;
  int64_t id_offset;
  unsigned *status_host;
  int host_out;
  unsigned long long *stamps;  // optional; copied to status_host[4 * 64 ...] by query 0's workgroup
  const float *ladder;       // optional (threshold-gated scan): with lvlmax, the threshold every workgroup emitted at
  const int *lvlmax;         // -> status_host[6 * 64 + q], what the host needs to choose a recovery without a read-back
};

// the scan emitted EVERY row at or above the threshold of the highest level any workgroup ended at: +inf when unknown
__device__ __forceinline__ float emitted_threshold(const float *ladder, const int *lvlmax, int q) {
  if (!ladder || !lvlmax) return __builtin_inff();
  const int l = lvlmax[q];
  return (l >= 0 && l < kLadder) ? ladder[q * kLadder + l] : __builtin_inff();
}

// the batch's stamps travel to the host in the status block (rows 4 and 5: kStamps u64 words)
__device__ __forceinline__ void stamps_to_host(unsigned long long *stamps, unsigned *status_host) {
  if (!stamps || !status_host) return;
  stamp_max(stamps, kStampPostEnd);
  unsigned long long *dst = reinterpret_cast<unsigned long long *>(status_host + 4 * kQB);
  for (int i = 0; i < kStamps; ++i) dst[i] = __hip_atomic_load(stamps + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(1024) void k_post(SelParams p, PostParams pp) {
  extern __shared__ unsigned char sel_smem[];
  SelShared &sh = *reinterpret_cast<SelShared *>(sel_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.x;
  const int m = sel_run(p, sh, q);  // sh.srt[0, m): (approximate rank score, row) best first
  __syncthreads();
  float approx_last = 0.f;
  if (m > 0) {
    const float v = ord2f((unsigned)(sh.srt[m - 1] >> 32));
    approx_last = p.negate ? -v : v;
  }
  // ---- exact value of every candidate row: one wave per row, four rows in flight per wave --------------------
  const float *qv = pp.q32 + (int64_t)q * pp.dimp;
  constexpr int RB = 4, NW = 16;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  for (int j0 = wave; j0 < m; j0 += NW * RB) {
    double acc[RB];
    unsigned row[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      acc[r] = 0.0;
      const int j = j0 + r * NW < m ? j0 + r * NW : j0;
      row[r] = 0xffffffffu - (unsigned)(sh.srt[j] & 0xffffffffu);
    }
    if ((pp.dim & 3) == 0) {
      const f32x4 *q4 = reinterpret_cast<const f32x4 *>(qv);
      const int n4 = pp.dim >> 2;
      for (int k = lane; k < n4; k += 64) {
        const f32x4 b = q4[k];
        f32x4 a[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r)
          a[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(pp.x32 + (int64_t)row[r] * pp.dim) + k);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          if (pp.metric == 0) {
            acc[r] += (double)a[r].x * (double)b.x;
            acc[r] += (double)a[r].y * (double)b.y;
            acc[r] += (double)a[r].z * (double)b.z;
            acc[r] += (double)a[r].w * (double)b.w;
          } else {
            const double d0 = (double)b.x - (double)a[r].x, d1 = (double)b.y - (double)a[r].y;
            const double d2 = (double)b.z - (double)a[r].z, d3 = (double)b.w - (double)a[r].w;
            acc[r] += d0 * d0;
            acc[r] += d1 * d1;
            acc[r] += d2 * d2;
            acc[r] += d3 * d3;
          }
        }
      }
    } else {
      for (int k = lane; k < pp.dim; k += 64) {
        const double b = (double)qv[k];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const double a = (double)pp.x32[(int64_t)row[r] * pp.dim + k];
          if (pp.metric == 0) acc[r] += a * b;
          else acc[r] += (b - a) * (b - a);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      for (int off = 32; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off);
      if (lane == 0 && j0 + r * NW < m) {
        const float e = (float)acc[r];
        sh.sel[j0 + r * NW] = make_key(pp.metric == 0 ? e : -e, row[r]);
      }
    }
  }
  __syncthreads();
  // ---- order by (exact, id): rank counting into the (now free) key staging area --------------------------------
  unsigned long long *sorted = sh.keys;
  for (int i = tid; i < m; i += 1024) {
    const unsigned long long k = sh.sel[i];
    int rank = 0;
#pragma unroll 16
    for (int j = 0; j < m; ++j) rank += (sh.sel[j] > k) ? 1 : 0;
    sorted[rank] = k;
  }
  __syncthreads();
  const int kk = pp.k < m ? pp.k : m;
  float *D = pp.D + (pp.out_off + q) * (int64_t)pp.k;
  int64_t *I = pp.I + (pp.out_off + q) * (int64_t)pp.k;
  for (int i = tid; i < pp.k; i += 1024) {
    if (i < kk) {
      const unsigned long long k = sorted[i];
      const float v = ord2f((unsigned)(k >> 32));
      D[i] = pp.metric == 0 ? v : -v;
      I[i] = (int64_t)(0xffffffffu - (unsigned)(k & 0xffffffffu)) + pp.id_offset;
    } else {
      D[i] = pp.metric == 0 ? -3.402823466e+38f : 3.402823466e+38f;
      I[i] = -1;
    }
  }
  if (pp.host_out) __threadfence_system();
  if (tid == 0) {  // the certificate: identical to k_finalize
    int flag = 0;
    float theta = -__builtin_inff();
    const bool ovf = pp.overflow && pp.overflow[q];
    if (ovf) flag = 1;
    if (kk > 0 && (int64_t)m < pp.n_rows) {
      const float xmax = __uint_as_float(pp.xstat[0]), xerr = __uint_as_float(pp.xstat[1]);
      const float qn = pp.qstat[q * 4 + 0], qh = pp.qstat[q * 4 + 1], qe = pp.qstat[q * 4 + 2];
      double eps = (double)xerr * qh + (double)xmax * qe + 2.0 * pp.dimp * 5.9604645e-8 * (double)xmax * qn;
      if (pp.metric != 0) eps += 1e-6 * (double)xmax * (double)xmax;
      const double bound = (double)approx_last + eps;
      double kth = (double)ord2f((unsigned)(sorted[kk - 1] >> 32));
      if (pp.metric != 0) kth = 0.5 * (kth + (double)pp.qstat[q * 4 + 3]);
      if (m < pp.Mreq || !(bound < kth)) flag = 1;
      if (m >= pp.Mreq && kk < pp.k) flag = 1;
      if (kk >= pp.k && !ovf) theta = (float)(kth - eps - 1e-6 * (fabs(kth) + 1.0));
    }
    pp.flags[q] = flag;
    if (pp.theta) pp.theta[q] = theta;
    if (pp.status_host) {
      pp.status_host[q] = pp.ncand ? pp.ncand[q] : 0u;
      pp.status_host[kQB + q] = ovf ? 1u : 0u;
      pp.status_host[2 * kQB + q] = (unsigned)flag;
      pp.status_host[3 * kQB + q] = __float_as_uint(theta);
      pp.status_host[6 * kQB + q] = __float_as_uint(emitted_threshold(pp.ladder, pp.lvlmax, q));
      if (q == 0) stamps_to_host(pp.stamps, pp.status_host);
      __threadfence_system();
    }
  }
}


// ------------------------------------------------------------------------------------------------
// rescore: one wave per (query, candidate): exact value from the f32 rows, f64 accumulation
// ------------------------------------------------------------------------------------------------
struct RescoreParams {
  const float *x32;
  const float *q32;   // [64][dimp]
  int dim, dimp;
  int metric;         // ANR_METRIC_*
  const unsigned *sel_row;  // [64][kMaxSel]
  const int *sel_m;
  float *exact;       // [64][kMaxSel]
  int M;
};

__global__ __launch_bounds__(256) void k_rescore(RescoreParams p) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int q = w / p.M, j = w % p.M;
  if (q >= kQB || j >= p.sel_m[q]) return;
  const unsigned row = p.sel_row[q * kMaxSel + j];
  const float *x = p.x32 + (int64_t)row * p.dim;
  const float *qv = p.q32 + (int64_t)q * p.dimp;
  double acc = 0.0;
  if ((p.dim & 3) == 0) {
    // 16-B loads, the whole row in flight at once (a row is read once per batch: non-temporal)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
    const f32x4 *q4 = reinterpret_cast<const f32x4 *>(qv);
    const int n4 = p.dim >> 2;
#pragma unroll 4
    for (int k = lane; k < n4; k += 64) {
      const f32x4 a = __builtin_nontemporal_load(x4 + k), b = q4[k];
      if (p.metric == 0) {
        acc += (double)a.x * (double)b.x;
        acc += (double)a.y * (double)b.y;
        acc += (double)a.z * (double)b.z;
        acc += (double)a.w * (double)b.w;
      } else {
        const double d0 = (double)b.x - (double)a.x, d1 = (double)b.y - (double)a.y;
        const double d2 = (double)b.z - (double)a.z, d3 = (double)b.w - (double)a.w;
        acc += d0 * d0;
        acc += d1 * d1;
        acc += d2 * d2;
        acc += d3 * d3;
      }
    }
  } else if (p.metric == 0) {
    for (int k = lane; k < p.dim; k += 64) acc += (double)x[k] * (double)qv[k];
  } else {
    for (int k = lane; k < p.dim; k += 64) {
      const double d = (double)qv[k] - (double)x[k];
      acc += d * d;
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) p.exact[q * kMaxSel + j] = (float)acc;
}

// rescore, shadow form: the same values (one wave per (query, candidate), a row's products added in the same order), but as
// ONE workgroup of 4 waves per CU: the waves are dealt to the queries (W / nq waves each) and a wave walks its share of the
// query's candidates with RB rows in flight — 3072 four-wave workgroups placed one at a time beside a resident scan took
// 12 placement rounds (~120 us in the shadow of a 1.25 M-row scan)
template <int RB, bool L2, bool VEC4>
__global__ __launch_bounds__(256) void k_rescore_shadow(RescoreParams p, int nq) {
  const int lane = threadIdx.x & 63;
  // the wave index as a scalar: query, candidate slots and row addresses then live in scalar registers
  const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = gridDim.x * 4;
  const int per_q = W / nq;  // the host launches at least nq waves
  const int q = w % nq, slot = w / nq;
  if (slot >= per_q) return;
  const int m = p.sel_m[q];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const float *qv = p.q32 + (int64_t)q * p.dimp;
  for (int j0 = slot; j0 < m; j0 += per_q * RB) {
    unsigned row[RB];
    double acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int j = j0 + r * per_q < m ? j0 + r * per_q : j0;
      row[r] = p.sel_row[q * kMaxSel + j];
      acc[r] = 0.0;
    }
    if (VEC4) {  // dim % 4 == 0
      const int n4 = p.dim >> 2;
#pragma unroll 1
      for (int k = lane; k < n4; k += 64) {
        const f32x4 b = reinterpret_cast<const f32x4 *>(qv)[k];
        f32x4 a[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r)
          a[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p.x32 + (int64_t)row[r] * p.dim) + k);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          if (!L2) {
            acc[r] += (double)a[r].x * (double)b.x;
            acc[r] += (double)a[r].y * (double)b.y;
            acc[r] += (double)a[r].z * (double)b.z;
            acc[r] += (double)a[r].w * (double)b.w;
          } else {
            const double d0 = (double)b.x - (double)a[r].x, d1 = (double)b.y - (double)a[r].y;
            const double d2 = (double)b.z - (double)a[r].z, d3 = (double)b.w - (double)a[r].w;
            acc[r] += d0 * d0;
            acc[r] += d1 * d1;
            acc[r] += d2 * d2;
            acc[r] += d3 * d3;
          }
          asm volatile("" ::: "memory");  // one row's float64 temporaries at a time: the kernel must stay within 56 registers
        }
      }
    } else {
#pragma unroll 1
      for (int k = lane; k < p.dim; k += 64) {
        const double b = (double)qv[k];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const double a = (double)p.x32[(int64_t)row[r] * p.dim + k];
          if (!L2) acc[r] += a * b;
          else acc[r] += (b - a) * (b - a);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      for (int off = 32; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off);
      if (lane == 0 && j0 + r * per_q < m) p.exact[q * kMaxSel + j0 + r * per_q] = (float)acc[r];
    }
  }
}

// score_rows: exact value of given (query, row) pairs — the gather(note_embeddings, ids) . q the reference
// recomputes per candidate (query/query_processor.py:3492-3589); one wave per pair, f64 accumulation
struct ScoreRowsParams {
  const float *x32;
  const float *q;      // [nq][dim] (already preprocessed like index queries)
  int dim, metric;
  int64_t n_rows;
  const int64_t *ids;  // [nq][per]
  int per;
  int64_t total;       // nq * per
  float *out;          // [nq][per]; rows outside [0, n_rows) give NaN
};

__global__ __launch_bounds__(256) void k_score_rows(ScoreRowsParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= p.total) return;
  const int64_t row = p.ids[w];
  if (row < 0 || row >= p.n_rows) {
    if (lane == 0) p.out[w] = __builtin_nanf("");
    return;
  }
  const float *x = p.x32 + row * p.dim;
  const float *qv = p.q + (w / p.per) * p.dim;
  double acc = 0.0;
  if (p.metric == 0) {
    for (int k = lane; k < p.dim; k += 64) acc += (double)x[k] * (double)qv[k];
  } else {
    for (int k = lane; k < p.dim; k += 64) {
      const double d = (double)qv[k] - (double)x[k];
      acc += d * d;
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) p.out[w] = (float)acc;
}

// second pass: exact rank score of EVERY entry of a query's candidate lists, written over the scan score
// (IP: the score; L2: minus the squared distance, so that larger is still better); one workgroup per query
struct RescoreListsParams {
  const float *x32;
  const float *q32;
  int dim, dimp, metric;
  uint2 *cand;
  const unsigned *cntb;
  int G;
  unsigned capb;
  const int *qslots;   // [nf] query slots to process
  const float *theta;  // optional [64]: entries whose scan score is below theta[q] are not re-scored (set to -inf)
};

__global__ __launch_bounds__(1024) void k_rescore_lists(RescoreListsParams p) {
  __shared__ unsigned offs[kSelMaxLists + 1];
  __shared__ unsigned wsum[16];
  const int q = p.qslots[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // exclusive prefix of the (clamped) list lengths, then one wave per flattened entry
  if (tid == 0) offs[0] = 0;
  for (int base_i = 0; base_i < p.G; base_i += 1024) {
    const int i = base_i + tid;
    unsigned v = 0;
    if (i < p.G) {
      v = p.cntb[(int64_t)q * p.G + i];
      v = v < p.capb ? v : p.capb;
    }
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned t = __shfl_up(v, off);
      if (lane >= off) v += t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    unsigned carry = offs[base_i];
    for (int w = 0; w < wave; ++w) carry += wsum[w];
    __syncthreads();
    if (i < p.G) offs[i + 1] = v + carry;
    __syncthreads();
  }
  const unsigned total = offs[p.G];
  const float *qv = p.q32 + (int64_t)q * p.dimp;
  for (unsigned e = blockIdx.y * 16 + wave; e < total; e += 16 * gridDim.y) {
    int lo = 0, hi = p.G;  // offs[lo] <= e < offs[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offs[mid] <= e) lo = mid;
      else hi = mid;
    }
    uint2 *ent = p.cand + ((int64_t)lo * kQB + q) * p.capb + (e - offs[lo]);
    if (p.theta && __uint_as_float(ent->x) < p.theta[q]) {
      if (lane == 0) ent->x = __float_as_uint(-__builtin_inff());
      continue;
    }
    const unsigned row = ent->y;
    const float *x = p.x32 + (int64_t)row * p.dim;
    double acc = 0.0;
    if ((p.dim & 3) == 0) {
      // 16-byte loads, the whole row in flight at once, products added in k_rescore's order (the same bits as the scores
      // the post kernel computes for the same row); the scalar loop below took ~5 us per row and wave
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x), *q4 = reinterpret_cast<const f32x4 *>(qv);
      const int n4 = p.dim >> 2;
#pragma unroll 4
      for (int k = lane; k < n4; k += 64) {
        const f32x4 a = __builtin_nontemporal_load(x4 + k), b = q4[k];
        if (p.metric == 0) {
          acc += (double)a.x * (double)b.x;
          acc += (double)a.y * (double)b.y;
          acc += (double)a.z * (double)b.z;
          acc += (double)a.w * (double)b.w;
        } else {
          const double d0 = (double)b.x - (double)a.x, d1 = (double)b.y - (double)a.y;
          const double d2 = (double)b.z - (double)a.z, d3 = (double)b.w - (double)a.w;
          acc += d0 * d0;
          acc += d1 * d1;
          acc += d2 * d2;
          acc += d3 * d3;
        }
      }
      if (p.metric != 0) acc = -acc;
    } else if (p.metric == 0) {
      for (int k = lane; k < p.dim; k += 64) acc += (double)x[k] * (double)qv[k];
    } else {
      for (int k = lane; k < p.dim; k += 64) {
        const double d = (double)qv[k] - (double)x[k];
        acc += d * d;
      }
      acc = -acc;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) ent->x = __float_as_uint((float)acc);
  }
}

// ------------------------------------------------------------------------------------------------
// finalize: per query sort the re-scored candidates, write the top-k, decide the certificate
// ------------------------------------------------------------------------------------------------
struct FinalParams {
  const float *exact;        // [64][kMaxSel]
  const float *approx;       // [64][kMaxSel] approximate rank score, sorted desc
  const unsigned *sel_row;
  const int *sel_m;
  const unsigned *overflow;  // may be null
  const float *qstat;        // [64][4]
  const unsigned *xstat;     // [2] bits of max ||x||, max ||x16-x||
  int metric, dimp;
  int64_t n_rows;
  int M;                     // overfetch requested of the select
  int k;
  int nq;
  int64_t out_off;           // first query of this batch in D/I
  float *D;
  int64_t *I;
  int *flags;                // [64] 0 = certified, 1 = needs the exact path
  float *theta;              // [64] scan-score threshold of the second pass: rank(k-th exact) - eps
  int64_t id_offset;
  const unsigned *ncand;     // may be null: [64] candidates seen (statistics)
  unsigned *status_host;     // pinned host memory [7][64]: candidates | overflow | flag | theta bits | time stamps (2 rows) | emitted threshold
  int host_out;              // D / I are pinned host memory: fence the writes at system scope
  unsigned long long *stamps;  // optional
  const float *ladder;       // optional, see PostParams
  const int *lvlmax;
};

__global__ __launch_bounds__(256) void k_finalize(FinalParams p) {
  __shared__ unsigned long long sel[kMaxSel];
  __shared__ unsigned long long srt[kMaxSel];
  const int q = blockIdx.x, tid = threadIdx.x;
  if (q >= p.nq) return;
  const int m = p.sel_m[q];
  for (int i = tid; i < m; i += 256) {
    const float e = p.exact[q * kMaxSel + i];
    srt[i] = make_key(p.metric == 0 ? e : -e, p.sel_row[q * kMaxSel + i]);
  }
  __syncthreads();
  for (int i = tid; i < m; i += 256) {
    const unsigned long long k = srt[i];
    int rank = 0;
#pragma unroll 16
    for (int j = 0; j < m; ++j) rank += (srt[j] > k) ? 1 : 0;
    sel[rank] = k;
  }
  __syncthreads();
  const int kk = p.k < m ? p.k : m;
  float *D = p.D + (p.out_off + q) * (int64_t)p.k;
  int64_t *I = p.I + (p.out_off + q) * (int64_t)p.k;
  for (int i = tid; i < p.k; i += 256) {
    if (i < kk) {
      const unsigned long long k = sel[i];
      const float v = ord2f((unsigned)(k >> 32));
      D[i] = p.metric == 0 ? v : -v;
      I[i] = (int64_t)(0xffffffffu - (unsigned)(k & 0xffffffffu)) + p.id_offset;
    } else {
      D[i] = p.metric == 0 ? -3.402823466e+38f : 3.402823466e+38f;
      I[i] = -1;
    }
  }
  if (p.host_out) __threadfence_system();
  if (tid == 0) {
    int flag = 0;
    float theta = -__builtin_inff();
    if (p.overflow && p.overflow[q]) flag = 1;
    if (kk > 0 && (int64_t)m < p.n_rows) {
      // rows outside the candidate set have approx rank <= approx[m-1]; their exact rank is at most
      // that + eps.  Certified when even that cannot reach the k-th exact rank.
      const float xmax = __uint_as_float(p.xstat[0]), xerr = __uint_as_float(p.xstat[1]);
      const float qn = p.qstat[q * 4 + 0], qh = p.qstat[q * 4 + 1], qe = p.qstat[q * 4 + 2];
      double eps = (double)xerr * qh + (double)xmax * qe + 2.0 * p.dimp * 5.9604645e-8 * (double)xmax * qn;
      if (p.metric != 0) eps += 1e-6 * (double)xmax * (double)xmax;  // f32 rounding of the stored -0.5||x||^2
      const double bound = (double)p.approx[q * kMaxSel + m - 1] + eps;
      double kth = (double)ord2f((unsigned)(sel[kk - 1] >> 32));  // exact rank score of the k-th result
      if (p.metric != 0) kth = 0.5 * (kth + (double)p.qstat[q * 4 + 3]);  // -dist -> q.x - 0.5||x||^2
      if (m < p.M || !(bound < kth)) flag = 1;
      if (m >= p.M && kk < p.k) flag = 1;
      // every row of the true top-k has exact rank >= kth, hence scan score >= kth - eps (rounded down a little)
      if (kk >= p.k && !(p.overflow && p.overflow[q])) theta = (float)(kth - eps - 1e-6 * (fabs(kth) + 1.0));
    }
    p.flags[q] = flag;
    if (p.theta) p.theta[q] = theta;
    if (p.status_host) {
      // the host reads these after the batch's event; written straight to pinned memory so that no
      // device-to-host copy (and no cache write-back in front of it) sits between two batches
      p.status_host[q] = p.ncand ? p.ncand[q] : 0u;
      p.status_host[kQB + q] = p.overflow ? p.overflow[q] : 0u;
      p.status_host[2 * kQB + q] = (unsigned)flag;
      p.status_host[3 * kQB + q] = __float_as_uint(theta);
      p.status_host[6 * kQB + q] = __float_as_uint(emitted_threshold(p.ladder, p.lvlmax, q));
      if (q == 0) stamps_to_host(p.stamps, p.status_host);
      __threadfence_system();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// exact dense path: exact values of up to 4 queries against every row (f32 rows, f64 accumulate)
// ------------------------------------------------------------------------------------------------
struct ExactParams {
  const float *x32;
  const float *q32;    // [64][dimp]
  int dim, dimp, metric;
  int64_t n_rows;
  int nf;              // 1..4 queries
  int qidx[4];
  float *dense;        // [4][ld]
  int64_t ld;
};

__global__ __launch_bounds__(256) void k_exact_dense(ExactParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t wt = (int64_t)gridDim.x * 4;
  for (int64_t row = w0; row < p.n_rows; row += wt) {
    const float *x = p.x32 + row * p.dim;
    double acc[4] = {0, 0, 0, 0};
    for (int k = lane; k < p.dim; k += 64) {
      const double xv = (double)x[k];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        if (f < p.nf) {
          const double qv = (double)p.q32[(int64_t)p.qidx[f] * p.dimp + k];
          if (p.metric == 0) acc[f] += xv * qv;
          else acc[f] += (qv - xv) * (qv - xv);
        }
      }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
      for (int off = 32; off > 0; off >>= 1) acc[f] += __shfl_xor(acc[f], off);
    if (lane == 0)
      for (int f = 0; f < p.nf; ++f) p.dense[(int64_t)f * p.ld + row] = (float)acc[f];
  }
}

// write a select result (already exact values) as final output rows
struct EmitParams {
  const float *rank;       // [64][kMaxSel] (values as stored, i.e. distances for L2)
  const unsigned *row;
  const int *m;
  int nf;
  const int *qslots;       // device: query slot of each block, or nullptr -> slot[] below
  int slot[4];             // select slot of each query
  int64_t outq[4];         // output row of each query
  int64_t out_off;         // with qslots: output row = out_off + slot
  int sel_is_slot;         // with qslots: the select results live at index slot (else at index blockIdx)
  int k, metric;
  float *D;
  int64_t *I;
  int64_t id_offset;
  const unsigned *overflow;  // optional [64] (with qslots): the select's overflow flag of the slot ...
  unsigned *status_host;     // ... copied to the pinned status block, row 1 (the recovery's only read-back)
};

__global__ __launch_bounds__(256) void k_emit(EmitParams p) {
  const int f = blockIdx.x;
  if (f >= p.nf) return;
  const int s = p.qslots ? p.qslots[f] : p.slot[f];
  if (threadIdx.x == 0 && p.qslots && p.overflow && p.status_host) p.status_host[kQB + s] = p.overflow[s];
  const int64_t oq = p.qslots ? p.out_off + s : p.outq[f];
  const int m = p.m[s];
  for (int i = threadIdx.x; i < p.k; i += 256) {
    float *D = p.D + oq * (int64_t)p.k;
    int64_t *I = p.I + oq * (int64_t)p.k;
    if (i < m) {
      D[i] = p.rank[s * kMaxSel + i];
      I[i] = (int64_t)p.row[s * kMaxSel + i] + p.id_offset;
    } else {
      D[i] = p.metric == 0 ? -3.402823466e+38f : 3.402823466e+38f;
      I[i] = -1;
    }
  }
  __threadfence_system();  // D / I may be pinned host memory (rare path: always fenced)
}

// ------------------------------------------------------------------------------------------------
// merge of P sorted partial lists per query (row-sharded corpus)
// ------------------------------------------------------------------------------------------------
struct MergeParams {
  const float *Dp;
  const int64_t *Ip;
  int64_t d_stride, i_stride;  // elements between the lists of consecutive parts
  int P;
  int64_t nq;
  int k;
  int larger;
  float *D;
  int64_t *I;
};

__device__ __forceinline__ bool merge_before(float sa, int64_t ia, float sb, int64_t ib, int larger) {
  // true when (sa, ia) is ordered strictly before (sb, ib); padding (-1) goes last
  if (ia < 0) return false;
  if (ib < 0) return true;
  if (sa != sb) return larger ? (sa > sb) : (sa < sb);
  return ia < ib;
}

__global__ __launch_bounds__(256) void k_merge(MergeParams p) {
  const int64_t q = blockIdx.x;
  const int total = p.P * p.k;
  for (int i = threadIdx.x; i < p.k; i += 256) {
    p.D[q * p.k + i] = p.larger ? -3.402823466e+38f : 3.402823466e+38f;
    p.I[q * p.k + i] = -1;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < total; e += 256) {
    const int pe = e / p.k, je = e % p.k;
    const float s = p.Dp[pe * p.d_stride + q * p.k + je];
    const int64_t id = p.Ip[pe * p.i_stride + q * p.k + je];
    if (id < 0) continue;
    int rank = je;  // entries of its own list that precede it
    for (int po = 0; po < p.P; ++po) {
      if (po == pe) continue;
      const float *Do = p.Dp + po * p.d_stride + q * p.k;
      const int64_t *Io = p.Ip + po * p.i_stride + q * p.k;
      int lo = 0, hi = p.k;  // count entries of list po ordered before (s, id)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (merge_before(Do[mid], Io[mid], s, id, p.larger)) lo = mid + 1;
        else hi = mid;
      }
      rank += lo;
    }
    if (rank < p.k) {
      p.D[q * p.k + rank] = s;
      p.I[q * p.k + rank] = id;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// self join: all pairs (i < j) of stored rows with x_i . x_j >= threshold — the thresholded upper triangle
// of the reference's N x N similarity matrix (graph/relation_extractor.py:769-782 and :604-608) without
// ever forming the matrix.  MFMA-bound: a workgroup (8 waves, 4 x 2) multiplies a 256-row block against a
// 256-row block of the SAME blocked f16 image (an x16 tile is a valid A and a valid B operand), operands
// staged once per workgroup in LDS; hits at or above threshold - eps go to a candidate list and are
// re-scored exactly (f32 rows, f64 accumulate) afterwards.
// ------------------------------------------------------------------------------------------------
struct JoinParams {
  const uint4 *x16;
  int kb;
  int64_t n_rows, n_tiles;
  int nblk;               // 256-row blocks
  PatchGrid pg;           // XCD-aware placement of the nblk x nblk block pairs (the lower triangle exits)
  int64_t n_slots;        // patch slots to walk (k_join_r: persistent workgroups)
  float thr_lo;           // threshold - eps
  uint2 *cand;            // [cap] (i, j)
  unsigned long long cap;
  unsigned long long *count;
};

// The operand fragments travel global -> VGPR -> LDS (ds_write_b128): two LDS slots of four k-steps; the
// fragments of stage s + 1 wait in registers while stage s is multiplied and are written into the other slot in
// the middle of it.  Variants measured on this kernel and dropped, all within 5 % of each other (820-880
// TFLOP/s at K = 768, ~42 % MFMA utilisation at the 1.9 GHz the chip holds under this load): operands through
// global_load_lds and a three-slot ring; 4 waves x (128 x 128) with double-buffered fragment reads (half the
// LDS reads per MFMA); non-persistent workgroups; linear instead of XCD-patched tile order (L2 hit rate 76 % with
// the patches).  What is left is the fill and drain of the 12-stage K loop per tile and the epilogue (~25 % of
// a tile's time at K = 768; 934 TFLOP/s at 300 k rows, 891 at K = 1024).
__global__ __launch_bounds__(512) void k_join_r(JoinParams p) {
  constexpr int TM = 8, TN = 8, S = 4, F = TM + TN, LPW = S * F / 8;
  extern __shared__ uint4 lds[];  // [2][S][F][64]
  // the wave index as a scalar: everything derived from it (fragment offsets, LDS slots) then lives in SGPRs
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // persistent: one workgroup per CU walks its share of the patch slots (slot = blockIdx + i * gridDim keeps the
  // slot -> XCD relation of patch_tile); re-dispatching a 512-thread, 128-KiB-LDS workgroup per tile left each CU
  // idle for ~8 us between 22-us tiles
  for (int64_t slot = blockIdx.x; slot < p.n_slots; slot += gridDim.x) {
  int bi, bj;
  if (!patch_tile(p.pg, slot, bi, bj) || bi > bj) continue;
  const int64_t tb0 = (int64_t)bi * TM, nb0 = (int64_t)bj * TN;
  const int wm = wave >> 1, wn = wave & 1;
  const u32x4 *xl = reinterpret_cast<const u32x4 *>(p.x16) + lane;  // native vectors: HIP's uint4 struct kept the staging array in scratch
  int64_t src[LPW];  // uniform offsets (uint4 units) of this wave's fragments at k-step 0
  int dst[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int f = wave * LPW + i, ks = f / F, idx = f % F;
    int64_t tile = idx < TM ? tb0 + idx : nb0 + idx - TM;
    if (tile >= p.n_tiles) tile = p.n_tiles - 1;  // clamp: masked at emit
    src[i] = (tile * p.kb + ks) * 64;
    dst[i] = (ks * F + idx) * 64;
  }
  u32x4 *ldsl = reinterpret_cast<u32x4 *>(lds) + lane;
  floatx16 acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  const int nstages = p.kb / S;
  constexpr int BUF = S * F * 64;  // uint4 per slot
  u32x4 stg[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) stg[i] = xl[src[i]];
#pragma unroll
  for (int i = 0; i < LPW; ++i) ldsl[dst[i]] = stg[i];
  if (nstages > 1) {
#pragma unroll
    for (int i = 0; i < LPW; ++i) stg[i] = xl[src[i] + (int64_t)S * 64];
  }
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    const uint4 *L = lds + (s & 1) * BUF + lane;
#pragma unroll
    for (int ks = 0; ks < S; ++ks) {
      half8 a[2], b[4];
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = __builtin_bit_cast(half8, L[(ks * F + 2 * wm + m) * 64]);
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = __builtin_bit_cast(half8, L[(ks * F + TM + 4 * wn + n) * 64]);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[n], a[m], acc[m][n], 0, 0, 0);
      if (ks == 1 && s + 1 < nstages) {
        // stage s + 1 has had a k-step and a half to arrive: park it in the other slot (free since the barrier
        // that ended stage s - 1) and request stage s + 2
        __builtin_amdgcn_sched_barrier(0);
        u32x4 *W = ldsl + ((s + 1) & 1) * BUF;
#pragma unroll
        for (int i = 0; i < LPW; ++i) W[dst[i]] = stg[i];
        if (s + 2 < nstages) {
#pragma unroll
          for (int i = 0; i < LPW; ++i) stg[i] = xl[src[i] + (int64_t)(s + 2) * S * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // LDS writes of this wave done, then the workgroup barrier — NOT __syncthreads(), whose vmcnt(0) would also
    // wait for the stage s + 2 loads just requested
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  // C rows = rows of the j tile, C columns = rows of the i tile.  Per accumulator tile a 16-bit hit mask
  // (value test from the registers, i < j < n test as integer masks), then: count, reserve the workgroup's slots
  // with ONE atomic (a counter that every hit bumped serialised at ~12 ns per hit), write (i, j) from the masks
  __shared__ unsigned wave_hits[8];
  __shared__ unsigned long long block_base;
  unsigned hm[2][4];
  unsigned mine = 0;
  const int n_rows = (int)p.n_rows;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int gi = (int)(tb0 + 2 * wm + m) * 32 + (lane & 31);
      const int gj0 = (int)(nb0 + 4 * wn + n) * 32 + 4 * (lane >> 5);
      unsigned valid = 0, hit = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gj = gj0 + (r & 3) + 8 * (r >> 2);
        valid |= (gi < gj && gj < n_rows) ? (1u << r) : 0u;
        hit |= (acc[m][n][r] >= p.thr_lo) ? (1u << r) : 0u;
      }
      hm[m][n] = hit & valid;
      mine += (unsigned)__popc(hm[m][n]);
    }
  unsigned incl = mine;  // inclusive prefix over the lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wave_hits[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0;
    for (int w = 0; w < 8; ++w) tot += wave_hits[w];
    block_base = tot ? atomicAdd(p.count, (unsigned long long)tot) : 0ull;
  }
  __syncthreads();
  unsigned long long k = block_base + (incl - mine);
  for (int w = 0; w < wave; ++w) k += wave_hits[w];
  if (mine) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const unsigned gi = (unsigned)((tb0 + 2 * wm + m) * 32 + (lane & 31));
        const unsigned gj0 = (unsigned)((nb0 + 4 * wn + n) * 32 + 4 * (lane >> 5));
        unsigned h = hm[m][n];
        while (h) {
          const int r = __ffs(h) - 1;
          h &= h - 1;
          if (k < p.cap) p.cand[k] = make_uint2(gi, gj0 + (r & 3) + 8 * (r >> 2));
          ++k;
        }
      }
  }
  }
}

// exact value of every candidate pair; the pairs that reach the threshold are compacted into the output
struct JoinRescoreParams {
  const float *x32;
  int dim;
  const uint2 *cand;
  unsigned long long n_cand;
  float threshold;
  int64_t *out_i, *out_j;
  float *out_s;
  unsigned long long out_cap;
  unsigned long long *out_count;
};

__global__ __launch_bounds__(256) void k_join_rescore(JoinRescoreParams p) {
  const int lane = threadIdx.x & 63;
  const unsigned long long w0 = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned long long wt = (unsigned long long)gridDim.x * 4;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  // a wave takes 64 candidates at a time: lane c keeps candidate c's value, so the survivors of the batch are
  // appended with one atomic
  for (unsigned long long c0 = w0 * 64; c0 < p.n_cand; c0 += wt * 64) {
    const int nb = p.n_cand - c0 < 64 ? (int)(p.n_cand - c0) : 64;
    const uint2 my = lane < nb ? p.cand[c0 + lane] : make_uint2(0u, 0u);
    float mine = 0.f;
    for (int c = 0; c < nb; ++c) {
      const unsigned ri = __shfl(my.x, c), rj = __shfl(my.y, c);
      const float *a = p.x32 + (int64_t)ri * p.dim, *b = p.x32 + (int64_t)rj * p.dim;
      double acc = 0.0;
      if ((p.dim & 3) == 0) {
        const f32x4 *a4 = reinterpret_cast<const f32x4 *>(a), *b4 = reinterpret_cast<const f32x4 *>(b);
        const int n4 = p.dim >> 2;
#pragma unroll 4
        for (int k = lane; k < n4; k += 64) {
          const f32x4 u = a4[k], v = b4[k];
          acc += (double)u.x * (double)v.x;
          acc += (double)u.y * (double)v.y;
          acc += (double)u.z * (double)v.z;
          acc += (double)u.w * (double)v.w;
        }
      } else {
        for (int k = lane; k < p.dim; k += 64) acc += (double)a[k] * (double)b[k];
      }
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
      if (lane == c) mine = (float)acc;
    }
    const bool keep = lane < nb && mine >= p.threshold;
    const unsigned long long mask = __ballot(keep);
    if (mask) {
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(p.out_count, (unsigned long long)__popcll(mask));
      base = __shfl(base, 0);
      if (keep) {
        const unsigned long long k = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (k < p.out_cap) {
          p.out_i[k] = my.x;
          p.out_j[k] = my.y;
          p.out_s[k] = mine;
        }
      }
    }
  }
}

}  // namespace anr
