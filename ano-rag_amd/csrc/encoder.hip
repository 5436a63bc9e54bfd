// Sentence encoder behind anr_encoder_* (include/anorag.h): the forward pass the reference obtains from
// sentence_transformers.SentenceTransformer.encode (vector_store/embedding_manager.py:392-399): a
// BERT-family encoder (bert / roberta / xlm-roberta: post-LN blocks, learned absolute positions), pooling
// (masked mean or CLS) and optional L2 normalisation.  Tokenisation stays on the host.
//
// CDNA4 design: every activation lives in HBM in MFMA *operand* layout, so no kernel needs LDS or a
// transpose and every global access is a contiguous 1 KiB per wave instruction:
//   act  [T/32][KB][64 lanes][8]  f16   token (t%32) on lane&31, 8 features per lane
//        (the residual stream too: a LayerNorm adds its input activations and the f16 projection in f32, normalises
//        in f32 and writes f16; only the LAST LayerNorm's output is also kept in f32, for the pooling)
//   W    [N/32][KB][64][8]        f16   output feature (n%32) on lane&31, 8 input features per lane
// with the *same* feature permutation inside each 16-feature block kb:
//   element j of lane half h  <->  feature kb*16 + 8*(j>>2) + 4*h + (j&3)
// which is exactly where v_mfma_f32_32x32x16_f16 leaves its results (C row = (r&3)+8*(r>>2)+4*h): the
// 16 accumulator registers of a lane are two ready-made 8-element operand fragments (r>>3 picks the
// 16-block, r&7 the element), so a GEMM epilogue is two 16-byte stores and the next GEMM / the attention
// consume them as is.  GEMMs run "token on lane" (C^T = W * act^T); V is projected with the operands
// swapped so that it lands "feature on lane, keys in k" — the A operand P*V needs.
// f16 operands and activations, f32 accumulation, f32 LayerNorm arithmetic.
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "combine.hpp"
#include "common.hpp"

namespace anr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int feat_of(int kb, int h, int j) { return kb * 16 + 8 * (j >> 2) + 4 * h + (j & 3); }

// ---- weight packing ------------------------------------------------------------------------------
// W row-major [N][K] f32 -> blocked f16 operand image [N/32][K/16][64][8]
__global__ void k_pack_weight(const float *w, int N, int K, _Float16 *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte fragment element group
  const int KB = K / 16;
  const int64_t total = (int64_t)(N / 32) * KB * 64;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const int kb = (int)((i >> 6) % KB);
  const int nb = (int)((i >> 6) / KB);
  const int n = nb * 32 + (lane & 31), h = lane >> 5;
  half8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (_Float16)w[(int64_t)n * K + feat_of(kb, h, j)];
  *reinterpret_cast<half8 *>(out + i * 8) = v;
}
// bias [N] -> accumulator order [N/32][2 halves][16 regs]
__global__ void k_pack_bias_acc(const float *b, int N, float *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int nb = i / 32, rem = i % 32, h = rem / 16, r = rem % 16;
  out[i] = b[nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
}
// per-feature vector [H] (LayerNorm gamma/beta) or table rows [R][H] -> blocked order [R][KB][2][8]
__global__ void k_pack_rows(const float *src, int64_t R, int H, float *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * H) return;
  const int64_t row = i / H;
  const int c = (int)(i % H);
  const int kb = c / 16, h = (c % 16) / 8, j = c % 8;
  out[i] = src[row * H + feat_of(kb, h, j)];
}

// LayerNorm folded into the linear layer that consumes it (EPI_FOLD_GELU): W' = W diag(gamma) as a second operand image, and per
// output feature, in accumulator order like the bias: s = the row sum of the ROUNDED f16 W' (what the MFMAs multiply the mean
// component of u with, so that "- mean s" removes exactly that), c = bias + W beta.  One thread per output feature.
__global__ void k_fold_ln(const _Float16 *w, const float *g, const float *beta, const float *bias_acc, int N, int K, _Float16 *wf,
                          float *s_acc, float *c_acc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // accumulator-order index: (nb, h, r)
  if (i >= N) return;
  const int KB = K / 16;
  const int nb = i / 32, rem = i % 32, ha = rem / 16, r = rem % 16;
  const int n_in = (r & 3) + 8 * (r >> 2) + 4 * ha;  // feature within the 32-block (k_pack_bias_acc's map)
  double ssum = 0.0, csum = 0.0;
  for (int kb = 0; kb < KB; ++kb)
    for (int h = 0; h < 2; ++h) {
      const int64_t e = (((int64_t)nb * KB + kb) * 64 + n_in + 32 * h) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float wv = (float)w[e + j];
        const int gi = (kb * 2 + h) * 8 + j;
        const _Float16 f = (_Float16)(wv * g[gi]);
        wf[e + j] = f;
        ssum += (double)(float)f;
        csum += (double)wv * (double)beta[gi];
      }
    }
  s_acc[i] = (float)ssum;
  c_acc[i] = (float)((double)bias_acc[i] + csum);
}

// ---- embeddings + LayerNorm ----------------------------------------------------------------------
struct EmbedParams {
  const int *ids;      // [B][L]
  const int *types;    // [B][L] or null
  const int *lens;     // [B]
  int B, L, Lp, H, KB, pos_offset, max_pos;
  const float *word, *pos, *type;  // blocked rows
  const float *g, *b;              // blocked LN params
  float eps;
  _Float16 *act;
};

constexpr int kLnMaxKbw = 16;  // feature blocks per wave held in registers by the workgroup LN kernels (H <= 1024)

// one workgroup per 32-token block, each wave gathers a quarter of the feature blocks of word + position +
// type rows into registers; mean and variance (two exact passes over the registers) meet in LDS
__global__ __launch_bounds__(256) void k_embed_ln(EmbedParams p) {
  __shared__ float red[2][4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tb = blockIdx.x;
  const int64_t t = tb * 32 + (lane & 31);
  const int h = lane >> 5;
  const int b = (int)(t / p.Lp), pos = (int)(t % p.Lp);
  const bool real = pos < p.L;
  const int id = real ? p.ids[(int64_t)b * p.L + pos] : 0;
  const int ty = (real && p.types) ? p.types[(int64_t)b * p.L + pos] : 0;
  const int64_t prow = pos + p.pos_offset < p.max_pos ? pos + p.pos_offset : p.max_pos - 1;  // padding rows only
  const int kbw = p.KB / 4, kb0 = wave * kbw;
  float x[kLnMaxKbw][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxKbw; ++i) {
    if (i < kbw) {
      const int o = ((kb0 + i) * 2 + h) * 8;
      const float4 *w = reinterpret_cast<const float4 *>(p.word + (int64_t)id * p.H + o);
      const float4 *q = reinterpret_cast<const float4 *>(p.pos + prow * p.H + o);
      const float4 *y = reinterpret_cast<const float4 *>(p.type + (int64_t)ty * p.H + o);
      const float4 a0 = w[0], a1 = w[1], b0 = q[0], b1 = q[1], c0 = y[0], c1 = y[1];
      x[i][0] = a0.x + b0.x + c0.x; x[i][1] = a0.y + b0.y + c0.y; x[i][2] = a0.z + b0.z + c0.z; x[i][3] = a0.w + b0.w + c0.w;
      x[i][4] = a1.x + b1.x + c1.x; x[i][5] = a1.y + b1.y + c1.y; x[i][6] = a1.z + b1.z + c1.z; x[i][7] = a1.w + b1.w + c1.w;
      s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]) + (x[i][4] + x[i][5]) + (x[i][6] + x[i][7]);
    }
  }
  s += __shfl_xor(s, 32);
  if (lane < 32) red[0][wave][lane] = s;
  __syncthreads();
  const float mean = ((red[0][0][lane & 31] + red[0][1][lane & 31]) + (red[0][2][lane & 31] + red[0][3][lane & 31])) / p.H;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxKbw; ++i) {
    if (i < kbw) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = x[i][j] - mean;
        v += d * d;
      }
    }
  }
  v += __shfl_xor(v, 32);
  if (lane < 32) red[1][wave][lane] = v;
  __syncthreads();
  v = (red[1][0][lane & 31] + red[1][1][lane & 31]) + (red[1][2][lane & 31] + red[1][3][lane & 31]);
  const float rstd = rsqrtf(v / p.H + p.eps);
#pragma unroll
  for (int i = 0; i < kLnMaxKbw; ++i) {
    if (i < kbw) {
      const int kb = kb0 + i;
      const int o = (kb * 2 + h) * 8;
      float out[8];
      half8 hv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        out[j] = (x[i][j] - mean) * rstd * p.g[o + j] + p.b[o + j];
        hv[j] = (_Float16)out[j];
      }
      const int64_t e = ((tb * p.KB + kb) * 64 + lane) * 8;
      *reinterpret_cast<half8 *>(p.act + e) = hv;
    }
  }
}

// fallback (H > 1024 or H % 64 != 0): one wave per 32-token block; each lane owns (token, half) and walks the
// KB feature blocks
__global__ __launch_bounds__(256) void k_embed_ln_wave(EmbedParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t tb = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t TB = (int64_t)p.B * p.Lp / 32;
  if (tb >= TB) return;
  const int64_t t = tb * 32 + (lane & 31);
  const int h = lane >> 5;
  const int b = (int)(t / p.Lp), pos = (int)(t % p.Lp);
  const bool real = pos < p.L;
  const int id = real ? p.ids[(int64_t)b * p.L + pos] : 0;
  const int ty = (real && p.types) ? p.types[(int64_t)b * p.L + pos] : 0;
  const int64_t prow = pos + p.pos_offset < p.max_pos ? pos + p.pos_offset : p.max_pos - 1;  // padding rows only
  float s = 0.f;
  for (int kb = 0; kb < p.KB; ++kb) {
    const int o = (kb * 2 + h) * 8;
    const float4 *w = reinterpret_cast<const float4 *>(p.word + (int64_t)id * p.H + o);
    const float4 *q = reinterpret_cast<const float4 *>(p.pos + prow * p.H + o);
    const float4 *y = reinterpret_cast<const float4 *>(p.type + (int64_t)ty * p.H + o);
    const float4 a0 = w[0], a1 = w[1], b0 = q[0], b1 = q[1], c0 = y[0], c1 = y[1];
    s += (a0.x + b0.x + c0.x) + (a0.y + b0.y + c0.y) + (a0.z + b0.z + c0.z) + (a0.w + b0.w + c0.w) +
         (a1.x + b1.x + c1.x) + (a1.y + b1.y + c1.y) + (a1.z + b1.z + c1.z) + (a1.w + b1.w + c1.w);
  }
  s += __shfl_xor(s, 32);
  const float mean = s / p.H;
  float v = 0.f;
  for (int kb = 0; kb < p.KB; ++kb) {
    const int o = (kb * 2 + h) * 8;
    const float *w = p.word + (int64_t)id * p.H + o, *q = p.pos + prow * p.H + o, *y = p.type + (int64_t)ty * p.H + o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = (w[j] + q[j] + y[j]) - mean;
      v += d * d;
    }
  }
  v += __shfl_xor(v, 32);
  const float rstd = rsqrtf(v / p.H + p.eps);
  for (int kb = 0; kb < p.KB; ++kb) {
    const int o = (kb * 2 + h) * 8;
    const float *w = p.word + (int64_t)id * p.H + o, *q = p.pos + prow * p.H + o, *y = p.type + (int64_t)ty * p.H + o;
    float out[8];
    half8 hv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      out[j] = ((w[j] + q[j] + y[j]) - mean) * rstd * p.g[o + j] + p.b[o + j];
      hv[j] = (_Float16)out[j];
    }
    const int64_t e = ((tb * p.KB + kb) * 64 + lane) * 8;
    *reinterpret_cast<half8 *>(p.act + e) = hv;
  }
}

// LayerNorm of (x + delta): x = the residual stream (f16, act layout — the layer's input activations), delta = the
// projection the GEMM before it wrote (f16).  Sum, statistics and normalisation in f32; the output replaces x in place
// (act) and, for the last LayerNorm of the forward, is also kept in f32 for the pooling (res != nullptr).
// (Round 2, first half: the GEMM epilogue added an f32 residual and wrote the f32 sum, this kernel read it and wrote
// an f32 residual + the f16 activations — 225 MB per GEMM + LN pair at 16 K tokens x 768, a quarter of the layer's
// time in the ablation runs; now 100 MB.)
struct LnParams {
  const _Float16 *x;      // residual in
  const _Float16 *delta;  // projection output
  int64_t TB;
  int H, KB;
  const float *g, *b;
  float eps;
  float *res;             // optional f32 copy of the output
  _Float16 *act;          // output (may alias x)
};
__device__ __forceinline__ void ln_load8(const LnParams &p, int64_t e, float (&v)[8]) {
  const half8 a = *reinterpret_cast<const half8 *>(p.x + e);
  if (p.delta) {
    const half8 d = *reinterpret_cast<const half8 *>(p.delta + e);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)a[j] + (float)d[j];
  } else {  // the residual sum was formed by the producing GEMM's epilogue (EPI_RES_LN)
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
  }
}

// one workgroup per 32-token block; each of the 4 waves owns a quarter of the feature blocks, keeps it in
// registers (<= 16 blocks x 8 values per lane), and the per-token sums meet in LDS: the row is read once and
// 4x as many waves stream as with one wave per token block
// NW waves per 32-token block, each owning KB / NW feature blocks in registers.  <4, false> is the throughput form
// (large batches).  <16 / 8, true> is the LATENCY form for small inputs — a query at a time is the reference's own
// calling pattern, and there a forward is ~100 dependent tiny kernels: the scale / shift vectors are requested together
// with the activations (one memory round trip instead of two before the stores) and the block's work is spread over
// four times the waves.  (At one 32-token block the 4-wave kernel took 11.7 us, a third of the single-query forward.)
template <int NW, bool PRE>
__global__ __launch_bounds__(NW * 64) void k_layernorm(LnParams p) {
  constexpr int MAXI = kLnMaxKbw * 4 / NW;
  __shared__ float red[NW][32][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tb = blockIdx.x;
  const int h = lane >> 5;
  const int kbw = p.KB / NW, kb0 = wave * kbw;
  float v[MAXI][8];
  float4 gv[PRE ? MAXI : 1][2], bv[PRE ? MAXI : 1][2];
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
    if (i < kbw) {
      ln_load8(p, ((tb * p.KB + kb0 + i) * 64 + lane) * 8, v[i]);
      if (PRE) {
        const int o = ((kb0 + i) * 2 + h) * 8;
        gv[i][0] = *reinterpret_cast<const float4 *>(p.g + o);
        gv[i][1] = *reinterpret_cast<const float4 *>(p.g + o + 4);
        bv[i][0] = *reinterpret_cast<const float4 *>(p.b + o);
        bv[i][1] = *reinterpret_cast<const float4 *>(p.b + o + 4);
      }
    }
  // shifted single-pass statistics: sums of (x - c) and (x - c)^2 with c = the token's first value, so the
  // variance does not cancel
  float c0;
  {
    const int64_t e0 = (tb * p.KB * 64 + (lane & 31)) * 8;
    c0 = (float)p.x[e0] + (p.delta ? (float)p.delta[e0] : 0.f);
  }
  const float c = __shfl(c0, lane & 31);
  float s = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < kbw) {
      const float d0 = v[i][0] - c, d1 = v[i][1] - c, d2 = v[i][2] - c, d3 = v[i][3] - c, d4 = v[i][4] - c, d5 = v[i][5] - c,
                  d6 = v[i][6] - c, d7 = v[i][7] - c;
      s += (d0 + d1) + (d2 + d3) + (d4 + d5) + (d6 + d7);
      s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) + (d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7);
    }
  }
  s += __shfl_xor(s, 32);
  s2 += __shfl_xor(s2, 32);
  if (lane < 32) {
    red[wave][lane][0] = s;
    red[wave][lane][1] = s2;
  }
  __syncthreads();
  s = 0.f;
  s2 = 0.f;
  if (NW == 4) {  // (the round-1 summation order of the throughput form, kept bit for bit)
    s = (red[0][lane & 31][0] + red[1][lane & 31][0]) + (red[2][lane & 31][0] + red[3][lane & 31][0]);
    s2 = (red[0][lane & 31][1] + red[1][lane & 31][1]) + (red[2][lane & 31][1] + red[3][lane & 31][1]);
  } else {
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      s += red[w][lane & 31][0];
      s2 += red[w][lane & 31][1];
    }
  }
  const float md = s / p.H;                       // mean - c
  const float var = fmaxf(s2 / p.H - md * md, 0.f);
  const float mean = c + md;
  const float rstd = rsqrtf(var + p.eps);
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i < kbw) {
      const int kb = kb0 + i;
      const int64_t e = ((tb * p.KB + kb) * 64 + lane) * 8;
      const int o = (kb * 2 + h) * 8;
      float gg[8], bb[8];
      if (PRE) {
        gg[0] = gv[i][0].x; gg[1] = gv[i][0].y; gg[2] = gv[i][0].z; gg[3] = gv[i][0].w;
        gg[4] = gv[i][1].x; gg[5] = gv[i][1].y; gg[6] = gv[i][1].z; gg[7] = gv[i][1].w;
        bb[0] = bv[i][0].x; bb[1] = bv[i][0].y; bb[2] = bv[i][0].z; bb[3] = bv[i][0].w;
        bb[4] = bv[i][1].x; bb[5] = bv[i][1].y; bb[6] = bv[i][1].z; bb[7] = bv[i][1].w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          gg[j] = p.g[o + j];
          bb[j] = p.b[o + j];
        }
      }
      float out[8];
      half8 hv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        out[j] = (v[i][j] - mean) * rstd * gg[j] + bb[j];
        hv[j] = (_Float16)out[j];
      }
      if (p.res) {
        *reinterpret_cast<float4 *>(p.res + e) = make_float4(out[0], out[1], out[2], out[3]);
        *reinterpret_cast<float4 *>(p.res + e + 4) = make_float4(out[4], out[5], out[6], out[7]);
      }
      *reinterpret_cast<half8 *>(p.act + e) = hv;
    }
  }
}

// fallback for shapes the workgroup kernel does not cover (H > 1024 or H % 64 != 0): one wave per token block
__global__ __launch_bounds__(256) void k_layernorm_wave(LnParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t tb = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tb >= p.TB) return;
  const int h = lane >> 5;
  // one statistics pass: sums of (x - c) and (x - c)^2 with c = the token's first value (shifted, so the
  // single-pass variance does not cancel), then the normalising pass: two reads of the row instead of three
  float c0;
  {
    const int64_t e0 = (tb * p.KB * 64 + (lane & 31)) * 8;
    c0 = (float)p.x[e0] + (p.delta ? (float)p.delta[e0] : 0.f);
  }
  const float c = __shfl(c0, lane & 31);
  float s = 0.f, s2 = 0.f;
  for (int kb = 0; kb < p.KB; ++kb) {
    float v[8];
    ln_load8(p, ((tb * p.KB + kb) * 64 + lane) * 8, v);
    const float d0 = v[0] - c, d1 = v[1] - c, d2 = v[2] - c, d3 = v[3] - c, d4 = v[4] - c, d5 = v[5] - c, d6 = v[6] - c,
                d7 = v[7] - c;
    s += (d0 + d1) + (d2 + d3) + (d4 + d5) + (d6 + d7);
    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) + (d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7);
  }
  s += __shfl_xor(s, 32);
  s2 += __shfl_xor(s2, 32);
  const float md = s / p.H;                       // mean - c
  const float var = fmaxf(s2 / p.H - md * md, 0.f);
  const float mean = c + md;
  const float rstd = rsqrtf(var + p.eps);
  for (int kb = 0; kb < p.KB; ++kb) {
    const int64_t e = ((tb * p.KB + kb) * 64 + lane) * 8;
    const int o = (kb * 2 + h) * 8;
    float v[8], out[8];
    ln_load8(p, e, v);
    half8 hv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      out[j] = (v[j] - mean) * rstd * p.g[o + j] + p.b[o + j];
      hv[j] = (_Float16)out[j];
    }
    if (p.res) {
      *reinterpret_cast<float4 *>(p.res + e) = make_float4(out[0], out[1], out[2], out[3]);
      *reinterpret_cast<float4 *>(p.res + e + 4) = make_float4(out[4], out[5], out[6], out[7]);
    }
    *reinterpret_cast<half8 *>(p.act + e) = hv;
  }
}

static void launch_layernorm(const LnParams &p, hipStream_t st) {
  const bool small = p.TB <= 64;  // <= 2048 tokens: latency matters, not throughput
  if (small && p.KB % 16 == 0 && p.KB / 16 <= kLnMaxKbw / 4)
    hipLaunchKernelGGL((k_layernorm<16, true>), dim3((unsigned)p.TB), dim3(1024), 0, st, p);
  else if (small && p.KB % 8 == 0 && p.KB / 8 <= kLnMaxKbw / 2)
    hipLaunchKernelGGL((k_layernorm<8, true>), dim3((unsigned)p.TB), dim3(512), 0, st, p);
  // throughput form: 8 waves per 32-token block (round 3: 16.0 us per pass at 16 K tokens x 768 against 18.9 with 4 waves
  // holding twice the registers, 16.6 with 16 waves)
  else if (p.KB % 8 == 0 && p.KB / 8 <= kLnMaxKbw / 2)
    hipLaunchKernelGGL((k_layernorm<8, false>), dim3((unsigned)p.TB), dim3(512), 0, st, p);
  else if (p.KB % 4 == 0 && p.KB / 4 <= kLnMaxKbw)
    hipLaunchKernelGGL((k_layernorm<4, false>), dim3((unsigned)p.TB), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(k_layernorm_wave, dim3((unsigned)ceil_div(p.TB, 4)), dim3(256), 0, st, p);
}

// ---- GEMM: out[t][n] = sum_k act[t][k] * W[n][k] + bias[n] -------------------------------------------
// EPI_RES_STATS / EPI_FOLD_GELU / EPI_RES_LN (round 4, large forwards): the first LayerNorm of a layer without a pass of
// its own.  With u = x + attention projection (the un-normalised residual sum):
//   EPI_RES_STATS  (output projection)  u = f16(x + acc + bias), and per token the sum and the sum of squares of ITS
//                  slice of u (the workgroup tile's columns) into a partial-statistics buffer — one slot per column tile,
//                  summed by the readers in slot order (no atomics: the bits do not depend on timing);
//   EPI_FOLD_GELU  (FFN up) consumes u itself: LN(u) W^T + b = rstd (u W'^T - mean s) + c with W' = W diag(gamma)
//                  folded once when the weights are finalised, s = the row sums of the f16 W', c = b + W beta;
//   EPI_RES_LN     (FFN down) rebuilds the residual it needs, LN(u) = (u - mean) rstd gamma + beta, from u and the same
//                  statistics, adds its projection and writes the sum for the layer's second LayerNorm (one input).
//   EPI_RES        (either 768-wide projection) out = f16(resid + acc + bias): the residual sum formed where the projection
//                  is still in registers, so that the LayerNorm behind it reads ONE tensor instead of two.
enum { EPI_ACT = 0, EPI_GELU = 1, EPI_VT = 3, EPI_RES_STATS = 4, EPI_FOLD_GELU = 5, EPI_RES_LN = 6, EPI_RES = 7 };
constexpr int kStatSlots = 6;  // column tiles of a 768-wide output at the narrowest tile (TN = 4)

struct GemmParams {
  const uint4 *act;   // [TB][KB][64]
  const uint4 *w;     // [NB][KB][64]
  int64_t TB;
  int NB, KB;
  const float *bias_acc;  // [NB][2][16] (token-on-lane epilogues)
  const float *bias;      // [N] plain (EPI_VT)
  _Float16 *out;          // EPI_ACT / EPI_GELU: [TB][NB*2][64][8]; EPI_VT: [NB][TB*2][64][8]
  // round 4 (see the EPI list above)
  const _Float16 *resid;  // EPI_RES_STATS: x; EPI_RES_LN: u (same layout as out; may alias out)
  float *stats_out;       // EPI_RES_STATS: [TB * 32][kStatSlots][2] partial (sum, sum of squares) of u
  const float *stats_in;  // EPI_FOLD_GELU / EPI_RES_LN: the same buffer, stat_slots slots filled
  int stat_slots;         // column tiles the producer of stats_in had
  int stat_h;             // features the statistics are over (the hidden size)
  float ln_eps;
  const float *ln_g, *ln_b;    // EPI_RES_LN: gamma / beta, blocked order [KB][2][8]
  const float *fold_s;         // EPI_FOLD_GELU: row sums of the folded f16 weights, accumulator order like bias_acc
};

// per-lane LayerNorm statistics of token (tb, lane & 31) from the partial sums
__device__ __forceinline__ void gemm_token_stats(const GemmParams &p, int64_t tb, int lane, float &mean, float &rstd) {
  const float *st = p.stats_in + ((tb * 32 + (lane & 31)) * kStatSlots) * 2;
  float s = 0.f, q = 0.f;
  for (int i = 0; i < p.stat_slots; ++i) {
    s += st[2 * i];
    q += st[2 * i + 1];
  }
  mean = s / p.stat_h;
  const float var = fmaxf(q / p.stat_h - mean * mean, 0.f);
  rstd = rsqrtf(var + p.ln_eps);
}

// gelu(x) = 0.5 x (1 + erf(x / sqrt 2)); erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the f16
// the value is stored in) — the libm erff costs as much as the GEMM's MFMAs at K = 768
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// The epilogue's form: Phi(x) - 0.5 = x Q(x^2) with Q a degree-8 polynomial in t = 2 x^2 / 4.5^2 - 1 (weighted minimax
// fit on |x| <= 4.5, beyond which Phi is held at its edge value), |gelu_poly - gelu| <= 6e-5 for every float32 x in
// [-14, 14] — a sixteenth of an f16 ulp at 1.0, the format the value is stored in.  13 plain VALU operations and no
// transcendental: the erf form's reciprocal (compiled to the IEEE division sequence), exponential and select made the
// GELU epilogue cost as many cycles as the tile's MFMAs (2900 VALU instructions per lane and tile).
__device__ __forceinline__ float gelu_poly(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
  const float t = fmaf(xc * xc, 0.09876543209876543f, -1.0f);
  float q = 0.00335475942119956f;
  q = fmaf(q, t, -0.009329607710242271f);
  q = fmaf(q, t, 0.012207310646772385f);
  q = fmaf(q, t, -0.016742795705795288f);
  q = fmaf(q, t, 0.027629755437374115f);
  q = fmaf(q, t, -0.0405561588704586f);
  q = fmaf(q, t, 0.05481854826211929f);
  q = fmaf(q, t, -0.0771719291806221f);
  q = fmaf(q, t, 0.15690211951732635f);
  return x * fmaf(xc, q, 0.5f);
}

// epilogue of one 32 x 32 accumulator tile (token block tb, feature block nb)
template <int EPI>
__device__ __forceinline__ void gemm_store_tile(const GemmParams &p, const floatx16 &c, int64_t tb, int nb, int lane) {
  const int h = lane >> 5;
  if (EPI == EPI_VT) {
    const float bb = p.bias[nb * 32 + (lane & 31)];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8 hv;
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[j] = (_Float16)(c[8 * s + j] + bb);
      *reinterpret_cast<half8 *>(p.out + (((int64_t)nb * (p.TB * 2) + tb * 2 + s) * 64 + lane) * 8) = hv;
    }
  } else {
    const float *ba = p.bias_acc + ((int64_t)nb * 2 + h) * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int64_t e = ((tb * (p.NB * 2) + nb * 2 + s) * 64 + lane) * 8;
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = c[8 * s + j] + ba[8 * s + j];
      half8 hv;
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[j] = (_Float16)(EPI == EPI_GELU ? gelu_poly(o[j]) : o[j]);
      *reinterpret_cast<half8 *>(p.out + e) = hv;
    }
  }
}

// the round-4 epilogues of one 32 x 32 accumulator tile; mean / rstd: the lane's token (EPI_FOLD_GELU, EPI_RES_LN); ssum /
// qsum: the lane's running sums over the values it writes (EPI_RES_STATS)
template <int EPI>
__device__ __forceinline__ void gemm_store_tile_ln(const GemmParams &p, const floatx16 &c, int64_t tb, int nb, int lane, float mean,
                                                   float rstd, float &ssum, float &qsum) {
  const int h = lane >> 5;
  const float *ba = p.bias_acc + ((int64_t)nb * 2 + h) * 16;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int64_t e = ((tb * (p.NB * 2) + nb * 2 + s) * 64 + lane) * 8;
    float o[8];
    half8 hv;
    if (EPI == EPI_FOLD_GELU) {
      const float *fs = p.fold_s + ((int64_t)nb * 2 + h) * 16;
      const float ms = mean * rstd;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = gelu_poly(fmaf(c[8 * s + j], rstd, fmaf(-ms, fs[8 * s + j], ba[8 * s + j])));
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[j] = (_Float16)o[j];
    } else {
      const half8 r = *reinterpret_cast<const half8 *>(p.resid + e);
      if (EPI == EPI_RES_LN) {
        const int gi = ((nb * 2 + s) * 2 + h) * 8;  // blocked order of the 8 features this lane holds
        const float4 g0 = *reinterpret_cast<const float4 *>(p.ln_g + gi), g1 = *reinterpret_cast<const float4 *>(p.ln_g + gi + 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(p.ln_b + gi), b1 = *reinterpret_cast<const float4 *>(p.ln_b + gi + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (((float)r[j] - mean) * rstd * gg[j] + bb[j]) + (c[8 * s + j] + ba[8 * s + j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (float)r[j] + (c[8 * s + j] + ba[8 * s + j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[j] = (_Float16)o[j];
      if (EPI == EPI_RES_STATS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // statistics of the ROUNDED values: what the readers of u will see
          const float v = (float)hv[j];
          ssum += v;
          qsum = fmaf(v, v, qsum);
        }
      }
    }
    *reinterpret_cast<half8 *>(p.out + e) = hv;
  }
}

// each wave: MT token blocks x NT feature blocks; waves are laid out feature-group fastest so the waves of a
// workgroup share their activation fragments through L1
template <int MT, int NT, int EPI>
__global__ __launch_bounds__(256) void k_gemm(GemmParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t wg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int NG = (p.NB + NT - 1) / NT;
  const int64_t TG = (p.TB + MT - 1) / MT;
  if (wg >= TG * NG) return;
  const int64_t tb0 = (wg / NG) * MT;
  const int nb0 = (int)(wg % NG) * NT;
  floatx16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  const uint4 *ap[MT];
  const uint4 *wp[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int64_t tb = tb0 + m < p.TB ? tb0 + m : p.TB - 1;  // clamp: tail tiles recompute the last block
    ap[m] = p.act + tb * p.KB * 64 + lane;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int nb = nb0 + n < p.NB ? nb0 + n : p.NB - 1;
    wp[n] = p.w + (int64_t)nb * p.KB * 64 + lane;
  }
  uint4 a[MT], b[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a[m] = ap[m][0];
#pragma unroll
  for (int n = 0; n < NT; ++n) b[n] = wp[n][0];
  for (int kb = 0; kb < p.KB; ++kb) {
    uint4 an[MT], bn[NT];
    const int kn = kb + 1 < p.KB ? kb + 1 : kb;
#pragma unroll
    for (int m = 0; m < MT; ++m) an[m] = ap[m][(int64_t)kn * 64];
#pragma unroll
    for (int n = 0; n < NT; ++n) bn[n] = wp[n][(int64_t)kn * 64];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const half8 av = __builtin_bit_cast(half8, a[m]);
        const half8 bv = __builtin_bit_cast(half8, b[n]);
        if (EPI == EPI_VT)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[m][n], 0, 0, 0);  // rows = tokens
        else
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, av, acc[m][n], 0, 0, 0);  // rows = features
      }
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = an[m];
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = bn[n];
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
      if (tb0 + m < p.TB && nb0 + n < p.NB) gemm_store_tile<EPI>(p, acc[m][n], tb0 + m, nb0 + n, lane);
}

// Skinny GEMM for short and medium inputs (a single query up to ~5000 tokens).  The tiled kernels below would put a
// 32-token batch on 3-24 workgroups and walk K serially (33 us for the 3072-deep FFN-down GEMM, 1.7 ms per
// forward at batch 1 — and the reference encodes queries one at a time).  Here a workgroup owns
// ONE 32-feature block for <= 2 token blocks, its 4 waves split K four ways, each streaming its quarter of the
// weight block straight from global memory (deep unroll, no LDS on the operand path), and the four partial
// accumulators meet in LDS: N/32 x ceil(TB/2) workgroups, K/64 MFMA steps each.
template <int EPI>
__device__ __forceinline__ void gemm_skinny_body(const GemmParams &p, const int block, float (&red)[4][2][64][17]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nb = block % p.NB;
  const int64_t tb0 = (int64_t)(block / p.NB) * 2;
  const bool two = tb0 + 1 < p.TB;
  const int kq = p.KB / 4, k0 = wave * kq;
  const uint4 *wp = p.w + ((int64_t)nb * p.KB + k0) * 64 + lane;
  const uint4 *a0 = p.act + (tb0 * p.KB + k0) * 64 + lane;
  const uint4 *a1 = p.act + ((two ? tb0 + 1 : tb0) * p.KB + k0) * 64 + lane;
  floatx16 c0, c1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    c0[r] = 0.f;
    c1[r] = 0.f;
  }
  // Six k-steps' operands in flight per round: at 32 tokens this kernel is a chain of memory round trips (K = 768 is
  // 12 k-steps per wave: two rounds instead of six), and a query at a time is the reference's own calling pattern.
  constexpr int U = 6;
  for (int k = 0; k < kq; k += U) {
    half8 wv[U], x0[U], x1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k + u < kq ? k + u : kq - 1;  // (clamped: the tail round repeats its last k-step's loads, unused)
      wv[u] = __builtin_bit_cast(half8, wp[(int64_t)kk * 64]);
      x0[u] = __builtin_bit_cast(half8, a0[(int64_t)kk * 64]);
      x1[u] = __builtin_bit_cast(half8, a1[(int64_t)kk * 64]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k + u < kq) {
        if (EPI == EPI_VT) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0[u], wv[u], c0, 0, 0, 0);  // rows = tokens
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1[u], wv[u], c1, 0, 0, 0);
        } else {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[u], x0[u], c0, 0, 0, 0);  // rows = features
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[u], x1[u], c1, 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    red[wave][0][lane][r] = c0[r];
    red[wave][1][lane][r] = c1[r];
  }
  __syncthreads();
  // waves 0 and 1 each finish one token block
  if (wave < 2 && (wave == 0 || two)) {
    floatx16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      c[r] = (red[0][wave][lane][r] + red[1][wave][lane][r]) + (red[2][wave][lane][r] + red[3][wave][lane][r]);
    gemm_store_tile<EPI>(p, c, tb0 + wave, nb, lane);
  }
}


template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_skinny(GemmParams p) {
  __shared__ float red[4][2][64][17];  // [wave][token block][lane][16 (+1: bank spread)]
  gemm_skinny_body<EPI>(p, (int)blockIdx.x, red);
}

// (Round 3, measured and dropped: LayerNorm formed INSIDE the skinny GEMM that consumes it — every workgroup normalising
// its token blocks itself, statistics pass + per-fragment arithmetic — to save the 24 LayerNorm launches of a small
// forward: the redundant arithmetic in N / 32 workgroups cost more than the launches it saved — one 32-token query
// 0.57 -> 0.66 ms, 16 x 64 tokens 1.0 -> 1.5 ms.)
// the Q/K projection and the V projection of a layer in ONE launch (small inputs are a chain of ~100 dependent launches of
// 4-6 us: one fewer per layer); blocks [0, n_qk) run the first problem, the rest the second — same arithmetic per block
__global__ __launch_bounds__(256) void k_gemm_skinny_qkv(GemmParams pq, GemmParams pv, int n_qk) {
  __shared__ float red[4][2][64][17];
  if ((int)blockIdx.x < n_qk) gemm_skinny_body<EPI_ACT>(pq, (int)blockIdx.x, red);
  else gemm_skinny_body<EPI_VT>(pv, (int)blockIdx.x - n_qk, red);
}

// LDS-staged GEMM: a workgroup (4 waves, 2 x 2) owns 128 tokens x 256 features; per stage of two k-steps the
// 24 operand fragments (4 activation + 8 weight blocks per k-step, 1 KiB each, already in operand layout) are
// copied global -> LDS once by global_load_lds (lane-linear, so the LDS image needs no swizzle and every
// ds_read_b128 is conflict-free) and read by the waves that need them: each fragment leaves L2 once per
// workgroup instead of once per wave.  The copies of stages s+1 and s+2 are in flight while stage s is
// multiplied (three-slot ring, counted vmcnt + one raw s_barrier per stage, never a full drain in the loop).
template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_lds(GemmParams p) {
#ifndef ANR_GEMM_S
#define ANR_GEMM_S 4
#endif
  constexpr int TM = 4, TN = 8, S = ANR_GEMM_S, F = TM + TN, LPW = S * F / 4;  // fragment copies per wave per stage
  extern __shared__ uint4 g_lds[];  // ring [3][S][F][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NG = (p.NB + TN - 1) / TN;
  const int64_t tb0 = (int64_t)(blockIdx.x / NG) * TM;
  const int nb0 = (int)(blockIdx.x % NG) * TN;
  const int wm = wave >> 1, wn = wave & 1;
  const uint4 *src[LPW];
  int dst[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int f = wave * LPW + i, ks = f / F, idx = f % F;
    if (idx < TM) {
      const int64_t tb = tb0 + idx < p.TB ? tb0 + idx : p.TB - 1;
      src[i] = p.act + (tb * p.KB + ks) * 64 + lane;
    } else {
      const int nb = nb0 + idx - TM < p.NB ? nb0 + idx - TM : p.NB - 1;
      src[i] = p.w + ((int64_t)nb * p.KB + ks) * 64 + lane;
    }
    dst[i] = (ks * F + idx) * 64;
  }
  floatx16 acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  const int nstages = p.KB / S;
  constexpr int BUF = S * F * 64;  // uint4 per ring slot
  // three-slot ring, copies issued two stages ahead: one barrier per stage
#pragma unroll
  for (int i = 0; i < LPW; ++i) __builtin_amdgcn_global_load_lds(src[i], g_lds + dst[i], 16, 0, 0);
  if (nstages > 1) {
#pragma unroll
    for (int i = 0; i < LPW; ++i) __builtin_amdgcn_global_load_lds(src[i] + (int64_t)S * 64, g_lds + BUF + dst[i], 16, 0, 0);
  }
  for (int s = 0; s < nstages; ++s) {
    if (s + 1 < nstages) {  // stage s landed, the LPW copies of stage s+1 may stay in flight
      if (S == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's copies of stage s are in LDS; every wave is done with stage s-1
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < nstages) {
      uint4 *slot = g_lds + ((s + 2) % 3) * BUF;  // last read in stage s-1
#pragma unroll
      for (int i = 0; i < LPW; ++i)
        __builtin_amdgcn_global_load_lds(src[i] + (int64_t)(s + 2) * S * 64, slot + dst[i], 16, 0, 0);
    }
    const uint4 *L = g_lds + (s % 3) * BUF + lane;
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
        for (int n = 0; n < 4; ++n) {
          if (EPI == EPI_VT) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[n], acc[m][n], 0, 0, 0);
          else acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[n], a[m], acc[m][n], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int64_t tb = tb0 + 2 * wm + m;
      const int nb = nb0 + 4 * wn + n;
      if (tb < p.TB && nb < p.NB) gemm_store_tile<EPI>(p, acc[m][n], tb, nb, lane);
    }
}

__device__ __forceinline__ void gemm_glds16(const uint4 *g, uint4 *l) { __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0); }
// keep a value alive in an ablated build (plain __device__ functions: an asm with a "v" constraint written directly in
// a __global__ template breaks the host-side instantiation, like the builtin above)
__device__ __forceinline__ void gemm_keep(const half8 &x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void gemm_keep(const floatx16 &x) { asm volatile("" ::"v"(x)); }

// ---- 8-wave ping-pong GEMM ---------------------------------------------------------------------------------
// A workgroup (4 x 2 waves) owns 256 tokens x (32 TN) features, TN = 8, 6 or 4 (64 x (16 TN) per wave): each byte that
// leaves L2 feeds twice the MFMAs of the 4-wave 128 x 256 tile above, and two waves share each SIMD.  TN = 6 / 4 exist for
// the grid shape: at 16 K tokens a 768-wide output is 192 workgroups of 256 x 256 (a 256-CU chip 3/4 busy) but exactly
// 256 of 256 x 192; the launcher picks the tile with the fewest workgroup rounds x tile width.  When the 8 + TN
// fragments of a k-step do not divide over the 8 waves, the surplus copies repeat the first fragments (same bytes to the
// same LDS address).  A k-step (K = 16) is one PHASE per wave:
//     load segment : 6 ds_read_b128 (its 2 + TN/2 operand fragments of this k-step), 2 global_load_lds (its share
//                    of the k-step PF ahead), counted s_waitcnt vmcnt   -> s_barrier
//     MFMA segment : s_waitcnt lgkmcnt(0), s_setprio 1, 2 x TN/2 MFMAs (256 cycles at TN = 8), s_setprio 0 -> s_barrier
// and the two waves that share a SIMD (wave w and w + 4) run ONE BARRIER APART (waves 4-7 pass one extra s_barrier
// before the loop, waves 0-3 one after it): at every workgroup barrier one of them turns from loading to multiplying
// and the other from multiplying to loading, so the matrix pipe of every SIMD always has a wave whose operands are
// already in registers (cdna_hip_programming.md: T3+T4 counted vmcnt / raw s_barrier, T5 s_setprio).
// LDS: a ring of R = 8 k-step slots of (8 + TN) KiB; the copy of k-step ks + PF is issued in phase ks.
//   (the two copies a wave owes per k-step are issued between its MFMAs, see the loop)
//   RAW: every wave has waited (vmcnt) for its copies of k-step ks + 1 before its first barrier of phase ks, and a
//        wave starts reading k-step ks + 1 only after its second barrier of phase ks, which lies behind the first
//        barrier of phase ks of both groups;
//   WAR: slot (ks + PF) % R last held k-step ks + PF - R, whose fragment reads completed (lgkmcnt(0)) at least
//        R - PF - 1 >= 1 full phases earlier for both groups (R >= PF + 2).
// Persistent: a workgroup per CU walks the tiles in XCD-aware patches (common.hpp: the 32 workgroups of an XCD share
// the operand panels of a 4 x 8 patch through its L2), and it requests the first PF k-steps of its NEXT tile before the
// epilogue of the current one, so the epilogue's VALU work and stores overlap the next tile's operand traffic instead
// of every CU writing, then every CU reading, in lockstep (measured on the separate-launch form: the epilogues were
// 23-48 % of the four encoder GEMMs' time).
// Ring: 8 slots, copies requested 5 k-steps ahead.  Tried and dropped (tools/gemm_abl.sh, PMC SQ_VALU_MFMA_BUSY_CYCLES):
// 11 slots / 8 k-steps ahead on the 192-wide tile (slower: 103 vs 94 us on the FFN-up GEMM — latency is not what the
// loop waits for); two workgroups per CU on a 5-slot ring (the epilogue of one beside the k-loop of the other: +5 % on
// the GEMMs that ran 2 rounds, nothing overall); XCD-contiguous instead of round-robin patch assignment (no change in
// TCC misses, which are mostly the output stores).  MFMA pipes: 54 % busy at the 1.66 GHz the chip holds under this load.
template <int EPI, int TN, int ABL = 0>  // ABL: developer ablations (1 no copies in the loop, 2 no MFMAs, 3 no epilogue, 4 no k-loop barriers)
__global__ __launch_bounds__(512) void k_gemm_pp(GemmParams p, PatchGrid pg, int64_t n_slots) {
  constexpr int TM = 8, F = TM + TN, NW = TN / 2;
  constexpr int R = 8, PF = 5;
  static_assert(R >= PF + 2, "ring too small for the prefetch distance");
  static_assert(R * F <= 160, "ring exceeds the LDS");
  extern __shared__ uint4 g_lds[];  // ring [R][F][64]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, wq = wave & 3;          // the two waves of a SIMD are (wq, grp 0) and (wq, grp 1)
  const int wm = wq, wn = grp;                       // token pair wm, feature half wn
  const int fb = wave < TN ? wave : wave - TN;       // TN = 6: waves 6, 7 repeat fragments 0, 1 (same bytes, same address)
  const int dstA = wave * 64, dstB = (TM + fb) * 64;
  constexpr int SLOT = F * 64;  // uint4 per ring slot
  const int KB = p.KB;
  const uint4 *srcA = nullptr, *srcB = nullptr;
  auto find_tile = [&](int64_t &slot, int &bm, int &bn) -> bool {
    for (; slot < n_slots; slot += gridDim.x)
      if (patch_tile(pg, slot, bm, bn)) return true;
    return false;
  };
  auto request = [&](int bm, int bn) {  // this wave copies token fragment `wave` and feature fragment `fb` of every k-step
    const int64_t tb_c = (int64_t)bm * TM + wave < p.TB ? (int64_t)bm * TM + wave : p.TB - 1;
    const int nb_c = bn * TN + fb < p.NB ? bn * TN + fb : p.NB - 1;
    srcA = p.act + (tb_c * KB) * 64 + lane;
    srcB = p.w + ((int64_t)nb_c * KB) * 64 + lane;
#pragma unroll
    for (int i = 0; i < PF; ++i)
      if (i < KB) {
        gemm_glds16(srcA + (int64_t)i * 64, g_lds + i * SLOT + dstA);
        gemm_glds16(srcB + (int64_t)i * 64, g_lds + i * SLOT + dstB);
      }
  };
  int64_t slot = blockIdx.x;
  int bm = 0, bn = 0;
  bool have = find_tile(slot, bm, bn);
  if (have) request(bm, bn);
  while (have) {
    floatx16 acc[2][NW];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NW; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    if (KB >= PF) {  // 2 (PF - 1) copies may stay in flight: k-step 0 has landed
      if (PF == 8) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger
    __builtin_amdgcn_sched_barrier(0);
    int rd = 0, wr = PF % R;  // ring slots of k-step ks and of k-step ks + PF
    for (int ks = 0; ks < KB; ++ks) {
      const uint4 *L = g_lds + rd * SLOT + lane;
      half8 a[2], b[NW];
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = __builtin_bit_cast(half8, L[(2 * wm + m) * 64]);
#pragma unroll
      for (int n = 0; n < NW; ++n) b[n] = __builtin_bit_cast(half8, L[(TM + NW * wn + n) * 64]);
      // this wave's copies of k-step ks + 1 have landed; ks + 2 .. ks + PF - 1 (issued in earlier MFMA segments) stay in flight
      if (ks + PF - 1 < KB) {
        if (PF == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (ABL != 4) __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      // the two copies of k-step ks + PF are issued BETWEEN the MFMAs: an LDS-DMA piece costs the wave 60-180 cycles of
      // issue, which hides behind the 32 cycles each MFMA keeps the pipe busy instead of lengthening the load segment
      // (round 3, tried and dropped: each wave of a group issuing from its own pair of MFMA slots so that the four do not
      // queue at the CU's vector-memory path together — the uniform branches broke the MFMA cadence, FFN-up 98 -> 111 us)
      const bool more = ABL != 1 && ks + PF < KB;
      uint4 *dst = g_lds + wr * SLOT;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) {
          if (ABL == 2) {
            gemm_keep(a[m]);
            gemm_keep(b[n]);
          } else if (EPI == EPI_VT) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[n], acc[m][n], 0, 0, 0);
          else acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[n], a[m], acc[m][n], 0, 0, 0);
          if (m == 0 && n == 1 && more) gemm_glds16(srcA + (int64_t)(ks + PF) * 64, dst + dstA);
          if (m == 1 && n == 0 && more) gemm_glds16(srcB + (int64_t)(ks + PF) * 64, dst + dstB);
        }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL != 4) __builtin_amdgcn_s_barrier();
      rd = rd + 1 == R ? 0 : rd + 1;
      wr = wr + 1 == R ? 0 : wr + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // pairs with the last barrier of the staggered group: every read is done
    __builtin_amdgcn_sched_barrier(0);
    // the next tile's first k-steps are requested now and land during the epilogue
    const int cm = bm, cn = bn;
    slot += gridDim.x;
    have = find_tile(slot, bm, bn);
    if (have) request(bm, bn);
    __builtin_amdgcn_sched_barrier(0);
    if (ABL == 3) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) gemm_keep(acc[m][n]);
    } else if (EPI == EPI_RES_STATS || EPI == EPI_FOLD_GELU || EPI == EPI_RES_LN || EPI == EPI_RES) {
      __shared__ float s_part[2][4][2][32][2];  // [feature half wn][token pair wm][m][token][sum, sum of squares]
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int64_t tb = (int64_t)cm * TM + 2 * wm + m;
        float mean = 0.f, rstd = 1.f, ssum = 0.f, qsum = 0.f;
        if (EPI != EPI_RES_STATS && EPI != EPI_RES && tb < p.TB) gemm_token_stats(p, tb, lane, mean, rstd);
#pragma unroll
        for (int n = 0; n < NW; ++n) {
          const int nb = cn * TN + NW * wn + n;
          if (tb < p.TB && nb < p.NB) gemm_store_tile_ln<EPI>(p, acc[m][n], tb, nb, lane, mean, rstd, ssum, qsum);
        }
        if (EPI == EPI_RES_STATS) {
          ssum += __shfl_xor(ssum, 32);
          qsum += __shfl_xor(qsum, 32);
          if (lane < 32) {
            s_part[wn][wm][m][lane][0] = ssum;
            s_part[wn][wm][m][lane][1] = qsum;
          }
        }
      }
      if (EPI == EPI_RES_STATS) {
        // the two waves that share a token pair hold its two feature halves: meet in LDS (no vmcnt wait: the next tile's
        // operand copies stay in flight), wave (wm, 0) writes the tile's slot
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wn == 0 && lane < 32) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int64_t tb = (int64_t)cm * TM + 2 * wm + m;
            if (tb < p.TB) {
              float *st = p.stats_out + (((tb * 32 + lane) * kStatSlots) + cn) * 2;
              st[0] = s_part[0][wm][m][lane][0] + s_part[1][wm][m][lane][0];
              st[1] = s_part[0][wm][m][lane][1] + s_part[1][wm][m][lane][1];
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // s_part is free for the next tile
      }
    } else {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) {
          const int64_t tb = (int64_t)cm * TM + 2 * wm + m;
          const int nb = cn * TN + NW * wn + n;
          if (tb < p.TB && nb < p.NB) gemm_store_tile<EPI>(p, acc[m][n], tb, nb, lane);
        }
    }
  }
}

// ---- attention: one wave per (sequence, head, 32-query block), online softmax, all in registers --------
struct AttnParams {
  const uint4 *qk;    // act layout, KBqk = 2H/16: Q blocks then K blocks
  const uint4 *vt;    // [H/32][T/16][64]
  const int *lens;
  int B, Lp, H, heads, dh;
  float scale;
  _Float16 *ctx;      // act layout [TB][H/16][64][8]
  const float *relbias;  // optional (MPNet): [heads][2 rel_span - 1], added to the scaled score at index key - query + rel_span - 1
  int rel_span;
};

template <int DH>
__global__ __launch_bounds__(256) void k_attention(AttnParams p) {
  constexpr int KD = DH / 16, DF = DH / 32;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int QB = p.Lp / 32;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (int64_t)p.B * p.heads * QB) return;
  const int qb = (int)(w % QB);
  const int hd = (int)((w / QB) % p.heads);
  const int b = (int)(w / ((int64_t)QB * p.heads));
  const int len = p.lens[b];
  const int KBqk = 2 * p.H / 16, KBh = p.H / 16;
  const int64_t tb_seq = (int64_t)b * QB;
  const int64_t KT = (int64_t)p.B * p.Lp / 16;
  half8 qf[KD];
#pragma unroll
  for (int kd = 0; kd < KD; ++kd)
    qf[kd] = __builtin_bit_cast(half8, p.qk[((tb_seq + qb) * KBqk + hd * KD + kd) * 64 + lane]);
  floatx16 O[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
  float m = -__builtin_inff(), l = 0.f;
  for (int kbk = 0; kbk * 32 < len; ++kbk) {
    const int64_t tbk = tb_seq + kbk;
    floatx16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) {
      const half8 kf = __builtin_bit_cast(half8, p.qk[(tbk * KBqk + KBh + hd * KD + kd) * 64 + lane]);
      S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kd], S, 0, 0, 0);  // rows = keys, lane = query
    }
    float mx = -__builtin_inff();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float sc = S[r] * p.scale;
      if (p.relbias) sc += p.relbias[(int64_t)hd * (2 * p.rel_span - 1) + (key < len ? key : 0) - (qb * 32 + (lane & 31)) + p.rel_span - 1];
      S[r] = key < len ? sc : -__builtin_inff();
      mx = fmaxf(mx, S[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    float ps = 0.f;
    half8 pf0, pf1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(S[r] - mn);
      ps += e;
      if (r < 8) pf0[r] = (_Float16)e;
      else pf1[r - 8] = (_Float16)e;
    }
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int d = 0; d < DF; ++d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
      const int64_t fb = (int64_t)hd * DF + d;
      const half8 v0 = __builtin_bit_cast(half8, p.vt[(fb * KT + tbk * 2 + 0) * 64 + lane]);
      const half8 v1 = __builtin_bit_cast(half8, p.vt[(fb * KT + tbk * 2 + 1) * 64 + lane]);
      O[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pf0, O[d], 0, 0, 0);  // rows = head features
      O[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pf1, O[d], 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
#pragma unroll
  for (int d = 0; d < DF; ++d)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8 hv;
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[j] = (_Float16)(O[d][8 * s + j] * inv);
      const int64_t kb = (int64_t)hd * KD + d * 2 + s;
      *reinterpret_cast<half8 *>(p.ctx + (((tb_seq + qb) * KBh + kb) * 64 + lane) * 8) = hv;
    }
}

// ---- Q/K/V projection + attention in ONE kernel (round 3) ---------------------------------------------------
// For sequences of at most 128 padded tokens (queries, short notes) and 64-wide heads.  A workgroup owns 256 tokens — two
// to eight whole sequences — and ONE head: its tile is 256 tokens x 192 features (that head's 64 Q, 64 K and 64 V rows of
// the two weight images), computed by the ping-pong k-loop of k_gemm_pp with TN = 6; then the tile is turned into f16
// operand fragments in LDS (the ring is free by then: 96 KiB), every wave picks up the operands of ONE 32-query block —
// its Q, and the K and V of its sequence's key blocks — into registers, the next tile's first k-steps are requested, and
// the wave runs the attention of its query block (online softmax, exactly k_attention's arithmetic) beside that
// prefetch.  The Q/K/V activations never touch global memory: three launches and ~150 MB of writes + reads per layer
// at 16 K tokens become one launch whose only output is the 25 MB context.
// Wave (wm, wn) multiplies token blocks 2 wm, 2 wm + 1 with the feature blocks {wn, wn + 2, wn + 4} of the tile's six
// (Q0 Q1 K0 K1 V0 V1): every wave has one Q, one K and one V block, so the operand order of each MFMA — V is projected
// with the operands swapped: feature on lane, keys in k — is a compile-time property of n.
struct QkvAttnParams {
  const uint4 *act;        // [TB][KB][64]
  const uint4 *wqk, *wv;   // [2H/32][KB][64], [H/32][KB][64]
  const float *bqk_acc;    // [2H/32][2][16]
  const float *bv;         // [H]
  int64_t TB;
  int KB, H, heads, Lp;
  const int *lens;
  float scale;
  const float *relbias;    // optional (MPNet)
  int rel_span;
  _Float16 *ctx;           // act layout [TB][H/16][64][8]
};

template <int NKB>  // key blocks per sequence: Lp = 32 NKB
__global__ __launch_bounds__(512) void k_qkv_attn(QkvAttnParams p, PatchGrid pg, int64_t n_slots) {
  constexpr int TM = 8, TN = 6, F = TM + TN, NW = 3;
  constexpr int R = 8, PF = 5;
  extern __shared__ uint4 g_lds[];  // ring [R][F][64]; after the k-loop: Q [8][4] | K [8][4] | V [2][16] fragments of 64 uint4
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int wm = wq, wn = grp;
  const int fb = wave < TN ? wave : wave - TN;
  const int dstA = wave * 64, dstB = (TM + fb) * 64;
  constexpr int SLOT = F * 64;
  const int KB = p.KB, NBH = p.H / 32;  // feature blocks per projection
  const uint4 *srcA = nullptr, *srcB = nullptr;
  auto find_tile = [&](int64_t &slot, int &bm, int &bn) -> bool {
    for (; slot < n_slots; slot += gridDim.x)
      if (patch_tile(pg, slot, bm, bn)) return true;
    return false;
  };
  auto request = [&](int bm, int hd) {
    const int64_t tb_c = (int64_t)bm * TM + wave < p.TB ? (int64_t)bm * TM + wave : p.TB - 1;
    srcA = p.act + (tb_c * KB) * 64 + lane;
    // feature fragment fb of the head's tile: Q0 Q1 | K0 K1 | V0 V1
    const uint4 *w = fb < 4 ? p.wqk + ((int64_t)((fb >> 1) * NBH + hd * 2 + (fb & 1)) * KB) * 64
                            : p.wv + ((int64_t)(hd * 2 + (fb & 1)) * KB) * 64;
    srcB = w + lane;
#pragma unroll
    for (int i = 0; i < PF; ++i)
      if (i < KB) {
        gemm_glds16(srcA + (int64_t)i * 64, g_lds + i * SLOT + dstA);
        gemm_glds16(srcB + (int64_t)i * 64, g_lds + i * SLOT + dstB);
      }
  };
  int64_t slot = blockIdx.x;
  int bm = 0, hd = 0;
  bool have = find_tile(slot, bm, hd);
  if (have) request(bm, hd);
  while (have) {
    floatx16 acc[2][NW];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NW; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    if (KB >= PF) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger (see k_gemm_pp)
    __builtin_amdgcn_sched_barrier(0);
    int rd = 0, wr = PF % R;
    for (int ks = 0; ks < KB; ++ks) {
      const uint4 *L = g_lds + rd * SLOT + lane;
      half8 a[2], b[NW];
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = __builtin_bit_cast(half8, L[(2 * wm + m) * 64]);
#pragma unroll
      for (int n = 0; n < NW; ++n) b[n] = __builtin_bit_cast(half8, L[(TM + wn + 2 * n) * 64]);
      if (ks + PF - 1 < KB) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      const bool more = ks + PF < KB;
      uint4 *dst = g_lds + wr * SLOT;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) {
          if (n == 2) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[n], acc[m][n], 0, 0, 0);  // V: rows = tokens
          else acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[n], a[m], acc[m][n], 0, 0, 0);         // Q, K: rows = features
          if (m == 0 && n == 1 && more) gemm_glds16(srcA + (int64_t)(ks + PF) * 64, dst + dstA);
          if (m == 1 && n == 0 && more) gemm_glds16(srcB + (int64_t)(ks + PF) * 64, dst + dstB);
        }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      rd = rd + 1 == R ? 0 : rd + 1;
      wr = wr + 1 == R ? 0 : wr + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // every fragment read of the ring is done
    __builtin_amdgcn_sched_barrier(0);
    // ---- the tile as f16 operand fragments in LDS: Q [8][4], K [8][4] (token on lane), V [2][16] (feature on lane) ----
    uint4 *Qs = g_lds, *Ks = g_lds + 32 * 64, *Vs = g_lds + 64 * 64;
    const int h = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int tbl = 2 * wm + m;
#pragma unroll
      for (int n = 0; n < 2; ++n) {  // Q (n = 0) and K (n = 1): bias in accumulator order, as gemm_store_tile
        const float *ba = p.bqk_acc + ((int64_t)(n * NBH + hd * 2 + wn) * 2 + h) * 16;
#pragma unroll
        for (int sf = 0; sf < 2; ++sf) {
          half8 hv;
#pragma unroll
          for (int j = 0; j < 8; ++j) hv[j] = (_Float16)(acc[m][n][8 * sf + j] + ba[8 * sf + j]);
          (n == 0 ? Qs : Ks)[(tbl * 4 + 2 * wn + sf) * 64 + lane] = __builtin_bit_cast(uint4, hv);
        }
      }
      const float bb = p.bv[(hd * 2 + wn) * 32 + (lane & 31)];
#pragma unroll
      for (int sf = 0; sf < 2; ++sf) {
        half8 hv;
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = (_Float16)(acc[m][2][8 * sf + j] + bb);
        Vs[(wn * 16 + tbl * 2 + sf) * 64 + lane] = __builtin_bit_cast(uint4, hv);
      }
    }
    __syncthreads();
    // ---- this wave's query block: u = wave; its sequence's key blocks ----
    const int u = wave, kb0 = (u / NKB) * NKB;
    half8 qf[4], kf[NKB][4], vf[NKB][2][2];
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) qf[kd] = __builtin_bit_cast(half8, Qs[(u * 4 + kd) * 64 + lane]);
#pragma unroll
    for (int kk = 0; kk < NKB; ++kk) {
#pragma unroll
      for (int kd = 0; kd < 4; ++kd) kf[kk][kd] = __builtin_bit_cast(half8, Ks[((kb0 + kk) * 4 + kd) * 64 + lane]);
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int sf = 0; sf < 2; ++sf) vf[kk][d][sf] = __builtin_bit_cast(half8, Vs[(d * 16 + (kb0 + kk) * 2 + sf) * 64 + lane]);
    }
    __syncthreads();  // the fragments are in registers: the ring may be refilled
    const int cm = bm, chd = hd;
    slot += gridDim.x;
    have = find_tile(slot, bm, hd);
    if (have) request(bm, hd);
    __builtin_amdgcn_sched_barrier(0);
    // ---- attention of the query block (k_attention's arithmetic) ----
    const int64_t tb = (int64_t)cm * TM + u;
    if (tb < p.TB) {
      const int64_t tok0 = tb * 32;
      const int seq = (int)(tok0 / p.Lp), qb = (int)((tok0 % p.Lp) / 32);
      const int len = p.lens[seq];
      floatx16 O[2];
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
      float mrun = -__builtin_inff(), l = 0.f;
#pragma unroll
      for (int kk = 0; kk < NKB; ++kk) {
        if (kk * 32 < len) {
          floatx16 S;
#pragma unroll
          for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
          for (int kd = 0; kd < 4; ++kd) S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kk][kd], qf[kd], S, 0, 0, 0);  // rows = keys
          float mx = -__builtin_inff();
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float sc = S[r] * p.scale;
            if (p.relbias) sc += p.relbias[(int64_t)chd * (2 * p.rel_span - 1) + (key < len ? key : 0) - (qb * 32 + (lane & 31)) + p.rel_span - 1];
            S[r] = key < len ? sc : -__builtin_inff();
            mx = fmaxf(mx, S[r]);
          }
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          const float mn = fmaxf(mrun, mx);
          const float alpha = __expf(mrun - mn);
          float ps = 0.f;
          half8 pf0, pf1;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float ex = __expf(S[r] - mn);
            ps += ex;
            if (r < 8) pf0[r] = (_Float16)ex;
            else pf1[r - 8] = (_Float16)ex;
          }
          l = l * alpha + ps;
          mrun = mn;
#pragma unroll
          for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
            O[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kk][d][0], pf0, O[d], 0, 0, 0);  // rows = head features
            O[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kk][d][1], pf1, O[d], 0, 0, 0);
          }
        }
      }
      l += __shfl_xor(l, 32);
      const float inv = 1.0f / l;
      const int KBh = p.H / 16;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int sf = 0; sf < 2; ++sf) {
          half8 hv;
#pragma unroll
          for (int j = 0; j < 8; ++j) hv[j] = (_Float16)(O[d][8 * sf + j] * inv);
          const int64_t kb = (int64_t)chd * 4 + d * 2 + sf;
          *reinterpret_cast<half8 *>(p.ctx + ((tb * KBh + kb) * 64 + lane) * 8) = hv;
        }
    }
  }
}

// ---- pooling + normalisation: one block per sequence, thread per feature -------------------------------
struct PoolParams {
  const float *res;  // final LayerNorm output, blocked f32
  const int *lens;
  int B, Lp, H, KB, pooling, normalize;
  float *out;           // [rows][H] row-major
  const int *out_rows;  // optional [B]: sequence b is written to row out_rows[b] (else row b)
};

__global__ __launch_bounds__(256) void k_pool(PoolParams p) {
  __shared__ float s_red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int len = p.lens[b];
  float *orow = p.out + (int64_t)(p.out_rows ? p.out_rows[b] : b) * p.H;
  // the row's values stay in registers until they are final (the output may be pinned HOST memory: written once, never
  // read back); H <= 2048, checked at create
  constexpr int PER = 8;
  float val[PER];
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int f = tid + e * 256;
    val[e] = 0.f;
    if (f < p.H) {
      const int kb = f / 16, off = f % 16, g = off / 8, hh = (off % 8) / 4, jj = off % 4, j = g * 4 + jj;
      const int nt = p.pooling == 1 ? 1 : len;
      float acc = 0.f;
      for (int t = 0; t < nt; ++t) {
        const int64_t tok = (int64_t)b * p.Lp + t;
        acc += p.res[(((tok >> 5) * p.KB + kb) * 64 + (tok & 31) + 32 * hh) * 8 + j];
      }
      // sentence-transformers: sum / clamp(mask_sum, 1e-9)
      const float v = p.pooling == 1 ? acc : acc / fmaxf((float)len, 1e-9f);
      val[e] = v;
      ss += v * v;
    }
  }
  float nrm = 1.f;
  if (p.normalize) {
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((tid & 63) == 0) s_red[tid >> 6] = ss;
    __syncthreads();
    // torch.nn.functional.normalize: x / max(||x||, 1e-12)
    nrm = fmaxf(sqrtf(s_red[0] + s_red[1] + s_red[2] + s_red[3]), 1e-12f);
  }
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int f = tid + e * 256;
    if (f < p.H) orow[f] = p.normalize ? val[e] / nrm : val[e];
  }
}

}  // namespace anr

using namespace anr;

// ====================================================================================================
struct LayerW {
  _Float16 *w1f = nullptr;                   // W1 diag(ln1 gamma): the FFN-up operand of the folded path
  float *f1s = nullptr, *f1c = nullptr;      // its row sums and folded bias, accumulator order
  _Float16 *wqk = nullptr, *wv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
  float *bqk = nullptr, *bv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;  // bqk/bo/b1/b2 in acc order
  float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;          // blocked
  bool have[16] = {false};
};

struct anr_encoder {
  anr_encoder_config cfg{};
  int device = 0;
  int n_cu = 256;
  hipStream_t stream = nullptr;
  std::mutex mu;
  float *word = nullptr, *pos = nullptr, *type = nullptr, *eg = nullptr, *eb = nullptr;
  float *relbias = nullptr;  // optional relative position bias table (MPNet)
  int rel_span = 0;
  bool have_emb[5] = {false, false, false, false, false};
  std::vector<LayerW> layers;
  bool finalized = false;
  // workspace
  int64_t ws_tokens = 0;
  int *d_ids = nullptr, *d_types = nullptr, *d_lens = nullptr, *d_rows = nullptr;  // carved from d_in
  int *d_in = nullptr;      // device block [ids B*L | types B*L | lens B | rows B]
  int *pin_in = nullptr;    // the same block in pinned host memory: ONE host-to-device copy per forward
  float *pin_out = nullptr, *pin_out_dev = nullptr;  // small batches: k_pool writes the embeddings straight to pinned host memory
  int64_t ws_b = 0, ws_in = 0;
  float *res = nullptr, *out = nullptr;  // res: the last LayerNorm's output in f32 (pooling input)
  _Float16 *act = nullptr, *delta = nullptr, *qk = nullptr, *vt = nullptr, *ctx = nullptr, *ffn = nullptr;
  _Float16 *big = nullptr;  // the block qk / vt / ctx / ffn point into
  float *stats = nullptr;   // [tokens][kStatSlots][2] partial LayerNorm statistics of the folded path
  int fold_ln = 0;          // large forwards: first LayerNorm of a layer folded into the GEMMs around it (ANORAG_ENC_FOLD=1: on)
  // anr_encoder_forward_shared: the combining queue of concurrent small forwards (combine.hpp) and its second LANE — a view
  // of this handle that shares the weights and owns a stream and a workspace of its own, so that two small forwards (each a
  // chain of ~90 dependent tiny launches) run side by side.  ANORAG_ENC_LANES=1: one lane.
  static int shared_lanes() {
    const char *v = getenv("ANORAG_ENC_LANES");
    return v ? (atoi(v) >= 2 ? 2 : 1) : 2;
  }
  ForwardCombiner shared{shared_lanes()};
  bool view = false;            // a lane: the weights belong to the handle it was made from
  anr_encoder *lane1 = nullptr;
  std::mutex lane_mu;
};

namespace {

constexpr size_t kPinOutBytes = 64 << 10;  // embeddings of a small batch go straight to pinned host memory

template <typename T>
int enc_alloc(T **p, int64_t n) {
  *p = nullptr;
  ANR_HIP(hipMalloc(reinterpret_cast<void **>(p), (size_t)(n > 0 ? n : 1) * sizeof(T)));
  return ANR_OK;
}
template <typename T>
void enc_free(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

// upload a host f32 tensor to a temporary device buffer
int upload(const float *host, int64_t n, float **dev) {
  ANR_TRY(enc_alloc(dev, n));
  ANR_HIP(hipMemcpy(*dev, host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  return ANR_OK;
}

int pack_weight(anr_encoder *e, const float *host, int N, int K, _Float16 **dst, int row_off, int N_total) {
  // packs rows [row_off, row_off+N) of a (possibly concatenated) [N_total][K] operand image
  if (!*dst) ANR_TRY(enc_alloc(dst, (int64_t)N_total * K));
  float *tmp = nullptr;
  ANR_TRY(upload(host, (int64_t)N * K, &tmp));
  const int64_t total = (int64_t)(N / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, e->stream, tmp, N, K,
                     *dst + (int64_t)row_off * K);
  const hipError_t err = hipStreamSynchronize(e->stream);
  (void)hipFree(tmp);
  if (err != hipSuccess) return fail(ANR_EHIP, "packing a tensor failed: %s", hipGetErrorString(err));
  return ANR_OK;
}

int pack_bias_acc(anr_encoder *e, const float *host, int N, float **dst, int off, int N_total) {
  if (!*dst) ANR_TRY(enc_alloc(dst, N_total));
  float *tmp = nullptr;
  ANR_TRY(upload(host, N, &tmp));
  hipLaunchKernelGGL(k_pack_bias_acc, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, e->stream, tmp, N, *dst + off);
  const hipError_t err = hipStreamSynchronize(e->stream);
  (void)hipFree(tmp);
  if (err != hipSuccess) return fail(ANR_EHIP, "packing a tensor failed: %s", hipGetErrorString(err));
  return ANR_OK;
}

int pack_rows(anr_encoder *e, const float *host, int64_t R, int H, float **dst) {
  if (!*dst) ANR_TRY(enc_alloc(dst, R * H));
  float *tmp = nullptr;
  ANR_TRY(upload(host, R * H, &tmp));
  hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)ceil_div(R * H, 256)), dim3(256), 0, e->stream, tmp, R, H, *dst);
  const hipError_t err = hipStreamSynchronize(e->stream);
  (void)hipFree(tmp);
  if (err != hipSuccess) return fail(ANR_EHIP, "packing a tensor failed: %s", hipGetErrorString(err));
  return ANR_OK;
}

int plain_copy(const float *host, int64_t n, float **dst) {
  if (!*dst) ANR_TRY(enc_alloc(dst, n));
  ANR_HIP(hipMemcpy(*dst, host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  return ANR_OK;
}

template <int EPI, int TN>
void launch_gemm8(anr_encoder *e, GemmParams &g, int64_t blocks) {
  constexpr int lds_pp = 8 * (8 + TN) * 1024;
#ifdef ANR_GEMM_ABLATIONS
  static const int abl = getenv("ANORAG_GEMM_ABL") ? atoi(getenv("ANORAG_GEMM_ABL")) : 0;  // developer ablations (wrong results)
#else
  constexpr int abl = 0;
#endif
  PatchGrid pg = make_patch_grid(ceil_div(g.TB, 8), ceil_div(g.NB, TN));
  const int64_t n_slots = pg.grid();
  const int64_t grid = std::min<int64_t>(n_slots, std::max(8, e->n_cu / 8 * 8));  // a multiple of 8: slot % 8 = XCD
#define ANR_PP(A)                                                                                          \
  {                                                                                                        \
    (void)ensure_dynamic_lds(reinterpret_cast<const void *>(&k_gemm_pp<EPI, TN, A>), lds_pp);              \
    hipLaunchKernelGGL((k_gemm_pp<EPI, TN, A>), dim3((unsigned)grid), dim3(512), lds_pp, e->stream, g, pg, n_slots); \
  }
#ifdef ANR_GEMM_ABLATIONS
  if (abl == 1) ANR_PP(1) else if (abl == 2) ANR_PP(2) else if (abl == 3) ANR_PP(3) else if (abl == 4) ANR_PP(4) else
#endif
  ANR_PP(0)
  (void)abl;
  (void)blocks;
#undef ANR_PP
}

// small inputs: one workgroup per (feature block, two token blocks), K split over its four waves
bool use_skinny(const GemmParams &g) {
  static const bool simple = getenv("ANORAG_GEMM_SIMPLE") != nullptr;
  static const bool no_skinny = getenv("ANORAG_GEMM_NOSKINNY") != nullptr;  // developer switches
  // measured crossover with the tile kernels at the bge-base shape (tools/enc_perf.py): 3072 tokens 1.80 ms skinny vs
  // 2.25 tiled, 4096 tokens 2.16 vs 2.07
  static const int skinny_max = getenv("ANORAG_SKINNY_MAX") ? atoi(getenv("ANORAG_SKINNY_MAX")) : 120;
  return !simple && g.KB % ANR_GEMM_S == 0 && !no_skinny && g.TB <= skinny_max && g.KB % 4 == 0;
}

// does this GEMM run on the 8-wave ping-pong tile kernel, and with which tile width (32 TN features)?
bool pp_tile(const anr_encoder *e, const GemmParams &g, int *tn_out) {
  static const bool simple = getenv("ANORAG_GEMM_SIMPLE") != nullptr;
  static const bool wide = getenv("ANORAG_GEMM_NARROW") == nullptr;  // developer switch: the 4-wave tile everywhere
  if (simple || g.KB % ANR_GEMM_S || use_skinny(g) || !(wide && g.KB % 2 == 0 && g.TB >= 8 * 16)) return false;
  // tile width by grid shape: fewest rounds of workgroups over the CUs, weighted by the work per workgroup.  The
  // 128-wide tile exists for mid-sized inputs (8 K tokens x N = 768: 96 / 128 / 192 tiles at width 256 / 192 / 128 on
  // 256 CUs — one round each way, so the narrowest tile, which spreads the same work over the most CUs, wins).
  const int64_t rt = ceil_div(g.TB, 8);
  const int64_t b8 = rt * ceil_div(g.NB, 8), b6 = rt * ceil_div(g.NB, 6), b4 = rt * ceil_div(g.NB, 4);
  const int64_t cost8 = ceil_div(b8, e->n_cu) * 8, cost6 = ceil_div(b6, e->n_cu) * 6, cost4 = ceil_div(b4, e->n_cu) * 4;
  if (tn_out) *tn_out = (cost4 < cost6 && cost4 < cost8) ? 4 : (cost6 < cost8 ? 6 : 8);
  return true;
}

template <int EPI>
void launch_gemm(anr_encoder *e, GemmParams &g) {
  static const bool simple = getenv("ANORAG_GEMM_SIMPLE") != nullptr;  // developer switch: the LDS-free kernel
  if (simple || g.KB % ANR_GEMM_S) {
    constexpr int MT = 2, NT = 4;
    const int64_t waves = ceil_div(g.TB, MT) * ceil_div(g.NB, NT);
    hipLaunchKernelGGL((k_gemm<MT, NT, EPI>), dim3((unsigned)ceil_div(waves, 4)), dim3(256), 0, e->stream, g);
    return;
  }
  if (use_skinny(g)) {
    const int64_t blocks = (int64_t)g.NB * ceil_div(g.TB, 2);
    hipLaunchKernelGGL((k_gemm_skinny<EPI>), dim3((unsigned)blocks), dim3(256), 0, e->stream, g);
    return;
  }
  int tn = 0;
  if (pp_tile(e, g, &tn)) {
    const int64_t blocks = ceil_div(g.TB, 8) * ceil_div(g.NB, tn);
    if (tn == 4) launch_gemm8<EPI, 4>(e, g, blocks);
    else if (tn == 6) launch_gemm8<EPI, 6>(e, g, blocks);
    else launch_gemm8<EPI, 8>(e, g, blocks);
    return;
  }
  if (EPI == EPI_RES_STATS || EPI == EPI_FOLD_GELU || EPI == EPI_RES_LN || EPI == EPI_RES) return;  // (only ever launched on the 8-wave tile kernel)
  const int64_t blocks = ceil_div(g.TB, 4) * ceil_div(g.NB, 8);
  constexpr int lds_bytes = 3 * ANR_GEMM_S * 12 * 1024;
  if (lds_bytes > 64 * 1024) (void)ensure_dynamic_lds(reinterpret_cast<const void *>(&k_gemm_lds<EPI>), lds_bytes);
  hipLaunchKernelGGL((k_gemm_lds<EPI>), dim3((unsigned)blocks), dim3(256), lds_bytes, e->stream, g);
}

int ensure_ws(anr_encoder *e, int B, int L, int Lp) {
  const int64_t T = (int64_t)B * Lp;
  const auto &c = e->cfg;
  if (T > e->ws_tokens) {
    enc_free(e->res);
    enc_free(e->delta);
    enc_free(e->act);
    enc_free(e->big);
    e->qk = e->vt = e->ctx = e->ffn = nullptr;
    enc_free(e->stats);
    ANR_TRY(enc_alloc(&e->stats, T * kStatSlots * 2));
    ANR_TRY(enc_alloc(&e->res, T * c.hidden));
    ANR_TRY(enc_alloc(&e->delta, T * c.hidden));
    ANR_TRY(enc_alloc(&e->act, T * c.hidden));
    // Q/K, V, the attention context and the FFN intermediate share ONE block: the first three are dead once the
    // output projection has run, and the intermediate (as large as the three together when I = 4 H) is written
    // after that.  A layer then touches 150 MB at 16 K tokens x 768 instead of 250 MB — inside the 256 MB Infinity
    // Cache, so the activations one kernel writes are still there when the next reads them, and overwritten lines
    // never have to reach HBM.
    const int64_t attn_elems = T * c.hidden * 4, ffn_elems = T * c.intermediate;
    ANR_TRY(enc_alloc(&e->big, std::max(attn_elems, ffn_elems)));
    e->qk = e->big;
    e->vt = e->big + T * c.hidden * 2;
    e->ctx = e->big + T * c.hidden * 3;
    e->ffn = e->big;
    e->ws_tokens = T;
  }
  if (B > e->ws_b) {
    enc_free(e->out);
    ANR_TRY(enc_alloc(&e->out, (int64_t)B * c.hidden));
    e->ws_b = B;
  }
  const int64_t n_in = 2 * (int64_t)B * L + 2 * (int64_t)B;
  if (n_in > e->ws_in) {
    enc_free(e->d_in);
    if (e->pin_in) (void)hipHostFree(e->pin_in);
    e->pin_in = nullptr;
    e->ws_in = 0;
    ANR_TRY(enc_alloc(&e->d_in, n_in + n_in / 2));
    ANR_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->pin_in), (size_t)(n_in + n_in / 2) * sizeof(int), hipHostMallocDefault));
    e->ws_in = n_in + n_in / 2;
  }
  e->d_ids = e->d_in;
  e->d_types = e->d_in + (int64_t)B * L;
  e->d_lens = e->d_in + 2 * (int64_t)B * L;
  e->d_rows = e->d_lens + B;
  if (!e->pin_out) {
    ANR_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->pin_out), kPinOutBytes, hipHostMallocDefault));
    ANR_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&e->pin_out_dev), e->pin_out, 0));
  }
  return ANR_OK;
}

}  // namespace

extern "C" {

int anr_encoder_create(const anr_encoder_config *cfg, int32_t device, anr_encoder **out) {
  if (!cfg || !out) return fail(ANR_EINVAL, "null argument");
  *out = nullptr;
  const auto &c = *cfg;
  if (c.n_layers <= 0 || c.hidden <= 0 || c.n_heads <= 0 || c.intermediate <= 0 || c.vocab_size <= 0 ||
      c.max_positions <= 0 || c.type_vocab_size <= 0)
    return fail(ANR_EINVAL, "encoder config has non-positive sizes");
  if (c.hidden % 32 || c.intermediate % 32) return fail(ANR_EINVAL, "hidden and intermediate sizes must be multiples of 32");
  if (c.hidden > 2048) return fail(ANR_EINVAL, "hidden size %d not supported (at most 2048)", c.hidden);
  if (c.hidden % c.n_heads) return fail(ANR_EINVAL, "hidden not divisible by heads");
  const int dh = c.hidden / c.n_heads;
  if (dh != 32 && dh != 64 && dh != 128) return fail(ANR_EINVAL, "head size %d not supported (32, 64, 128)", dh);
  if (c.pooling != 0 && c.pooling != 1) return fail(ANR_EINVAL, "pooling must be 0 (mean) or 1 (cls)");
  if (c.act != 0) return fail(ANR_EINVAL, "only gelu (erf) activation is supported");
  int ndev = anr_device_count();
  if (ndev <= 0) return fail(ANR_EHIP, "no HIP device is visible");
  if (device < 0 || device >= ndev) return fail(ANR_EINVAL, "device %d out of range", device);
  DeviceGuard g(device);
  anr_encoder *e = new anr_encoder();
  e->cfg = c;
  e->device = device;
  e->n_cu = device_cu_count(device);
  e->layers.resize(c.n_layers);
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
    delete e;
    return fail(ANR_EHIP, "hipStreamCreate failed");
  }
  *out = e;
  return ANR_OK;
}

int anr_encoder_destroy(anr_encoder *e) {
  if (!e) return ANR_OK;
  DeviceGuard g(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->lane1) {
    (void)anr_encoder_destroy(e->lane1);
    e->lane1 = nullptr;
  }
  if (!e->view) {
  enc_free(e->word); enc_free(e->pos); enc_free(e->type); enc_free(e->eg); enc_free(e->eb); enc_free(e->relbias);
  for (auto &l : e->layers) {
    enc_free(l.wqk); enc_free(l.wv); enc_free(l.wo); enc_free(l.w1); enc_free(l.w2);
    enc_free(l.bqk); enc_free(l.bv); enc_free(l.bo); enc_free(l.b1); enc_free(l.b2);
    enc_free(l.ln1g); enc_free(l.ln1b); enc_free(l.ln2g); enc_free(l.ln2b);
    enc_free(l.w1f); enc_free(l.f1s); enc_free(l.f1c);
  }
  }
  enc_free(e->d_in);
  if (e->pin_in) (void)hipHostFree(e->pin_in);
  if (e->pin_out) (void)hipHostFree(e->pin_out);
  enc_free(e->res); enc_free(e->stats);
  enc_free(e->delta); enc_free(e->out);
  enc_free(e->act); enc_free(e->big);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return ANR_OK;
}

// tensor names: emb.word [V][H], emb.pos [P][H], emb.type [Tv][H], emb.ln.g/.b [H];
// L<i>.q.w/.k.w/.v.w/.o.w [H][H], L<i>.ffn1.w [I][H], L<i>.ffn2.w [H][I] (nn.Linear layout [out][in]),
// the matching .b vectors, L<i>.ln1.g/.b (after attention), L<i>.ln2.g/.b (after the FFN)
int anr_encoder_set_tensor(anr_encoder *e, const char *name, const float *data, int64_t n) {
  if (!e || !name || !data) return fail(ANR_EINVAL, "null argument");
  DeviceGuard g(e->device);
  {
    std::lock_guard<std::mutex> ll(e->lane_mu);  // a lane holds copies of the weight pointers: gone with the old weights
    if (e->lane1) {
      (void)anr_encoder_destroy(e->lane1);
      e->lane1 = nullptr;
    }
  }
  std::lock_guard<std::mutex> lk(e->mu);
  const auto &c = e->cfg;
  const int H = c.hidden, I = c.intermediate;
  const std::string s(name);
  auto need = [&](int64_t want) -> int {
    return n == want ? ANR_OK : fail(ANR_EINVAL, "tensor %s: expected %lld elements, got %lld", name, (long long)want, (long long)n);
  };
  e->finalized = false;
  if (s == "emb.word") { ANR_TRY(need((int64_t)c.vocab_size * H)); e->have_emb[0] = true; return pack_rows(e, data, c.vocab_size, H, &e->word); }
  if (s == "emb.pos") { ANR_TRY(need((int64_t)c.max_positions * H)); e->have_emb[1] = true; return pack_rows(e, data, c.max_positions, H, &e->pos); }
  if (s == "emb.type") { ANR_TRY(need((int64_t)c.type_vocab_size * H)); e->have_emb[2] = true; return pack_rows(e, data, c.type_vocab_size, H, &e->type); }
  if (s == "emb.ln.g") { ANR_TRY(need(H)); e->have_emb[3] = true; return pack_rows(e, data, 1, H, &e->eg); }
  if (s == "emb.ln.b") { ANR_TRY(need(H)); e->have_emb[4] = true; return pack_rows(e, data, 1, H, &e->eb); }
  if (s == "rel.bias") {
    // optional: additive attention bias by relative position, [heads][2 span - 1] with span = usable positions
    const int span = c.max_positions - c.pos_offset;
    ANR_TRY(need((int64_t)c.n_heads * (2 * span - 1)));
    e->rel_span = span;
    return plain_copy(data, n, &e->relbias);
  }
  if (s.size() < 4 || s[0] != 'L') return fail(ANR_EINVAL, "unknown tensor name %s", name);
  const size_t dot = s.find('.');
  if (dot == std::string::npos) return fail(ANR_EINVAL, "unknown tensor name %s", name);
  const int li = atoi(s.substr(1, dot - 1).c_str());
  if (li < 0 || li >= c.n_layers) return fail(ANR_EINVAL, "tensor %s: layer out of range", name);
  LayerW &l = e->layers[li];
  const std::string t = s.substr(dot + 1);
  if (t == "q.w") { ANR_TRY(need((int64_t)H * H)); l.have[0] = true; return pack_weight(e, data, H, H, &l.wqk, 0, 2 * H); }
  if (t == "k.w") { ANR_TRY(need((int64_t)H * H)); l.have[1] = true; return pack_weight(e, data, H, H, &l.wqk, H, 2 * H); }
  if (t == "v.w") { ANR_TRY(need((int64_t)H * H)); l.have[2] = true; return pack_weight(e, data, H, H, &l.wv, 0, H); }
  if (t == "o.w") { ANR_TRY(need((int64_t)H * H)); l.have[3] = true; return pack_weight(e, data, H, H, &l.wo, 0, H); }
  if (t == "ffn1.w") { ANR_TRY(need((int64_t)I * H)); l.have[4] = true; return pack_weight(e, data, I, H, &l.w1, 0, I); }
  if (t == "ffn2.w") { ANR_TRY(need((int64_t)H * I)); l.have[5] = true; return pack_weight(e, data, H, I, &l.w2, 0, H); }
  if (t == "q.b") { ANR_TRY(need(H)); l.have[6] = true; return pack_bias_acc(e, data, H, &l.bqk, 0, 2 * H); }
  if (t == "k.b") { ANR_TRY(need(H)); l.have[7] = true; return pack_bias_acc(e, data, H, &l.bqk, H, 2 * H); }
  if (t == "v.b") { ANR_TRY(need(H)); l.have[8] = true; return plain_copy(data, H, &l.bv); }
  if (t == "o.b") { ANR_TRY(need(H)); l.have[9] = true; return pack_bias_acc(e, data, H, &l.bo, 0, H); }
  if (t == "ffn1.b") { ANR_TRY(need(I)); l.have[10] = true; return pack_bias_acc(e, data, I, &l.b1, 0, I); }
  if (t == "ffn2.b") { ANR_TRY(need(H)); l.have[11] = true; return pack_bias_acc(e, data, H, &l.b2, 0, H); }
  if (t == "ln1.g") { ANR_TRY(need(H)); l.have[12] = true; return pack_rows(e, data, 1, H, &l.ln1g); }
  if (t == "ln1.b") { ANR_TRY(need(H)); l.have[13] = true; return pack_rows(e, data, 1, H, &l.ln1b); }
  if (t == "ln2.g") { ANR_TRY(need(H)); l.have[14] = true; return pack_rows(e, data, 1, H, &l.ln2g); }
  if (t == "ln2.b") { ANR_TRY(need(H)); l.have[15] = true; return pack_rows(e, data, 1, H, &l.ln2b); }
  return fail(ANR_EINVAL, "unknown tensor name %s", name);
}

int anr_encoder_finalize(anr_encoder *e) {
  if (!e) return fail(ANR_EINVAL, "null handle");
  std::lock_guard<std::mutex> lk(e->mu);
  for (int i = 0; i < 5; ++i)
    if (!e->have_emb[i]) return fail(ANR_ESTATE, "embedding tensor %d was not set", i);
  for (size_t li = 0; li < e->layers.size(); ++li)
    for (int i = 0; i < 16; ++i)
      if (!e->layers[li].have[i]) return fail(ANR_ESTATE, "layer %zu: tensor slot %d was not set", li, i);
  {  // the folded FFN-up operands (EPI_FOLD_GELU)
    DeviceGuard g(e->device);
    const int H = e->cfg.hidden, I = e->cfg.intermediate;
    for (auto &l : e->layers) {
      if (!l.w1f) ANR_TRY(enc_alloc(&l.w1f, (int64_t)I * H));
      if (!l.f1s) ANR_TRY(enc_alloc(&l.f1s, I));
      if (!l.f1c) ANR_TRY(enc_alloc(&l.f1c, I));
      hipLaunchKernelGGL(k_fold_ln, dim3((unsigned)ceil_div(I, 64)), dim3(64), 0, e->stream, l.w1, l.ln1g, l.ln1b, l.b1, I, H, l.w1f,
                         l.f1s, l.f1c);
    }
    ANR_HIP(hipGetLastError());
    ANR_HIP(hipStreamSynchronize(e->stream));
    // OFF by default: measured SLOWER than the LayerNorm pass it removes (3.72 vs 3.67 ms per 256 x 64 forward)
    e->fold_ln = getenv("ANORAG_ENC_FOLD") ? atoi(getenv("ANORAG_ENC_FOLD")) : 0;
  }
  e->finalized = true;
  return ANR_OK;
}

}  // extern "C"

namespace {
// out_host: [B][hidden] host memory; or out_dev: device memory, sequence b written to row out_rows[b] (host [B], or row b)
// every kernel of one forward, enqueued on the encoder's stream
void enqueue_forward(anr_encoder *e, int B, int L, int Lp, bool use_types, int normalize, float *out, const int *rows) {
  const auto &c = e->cfg;
  hipStream_t st = e->stream;
  const int H = c.hidden, I = c.intermediate, KB = H / 16;
  const int64_t TB = (int64_t)B * Lp / 32;

  EmbedParams ep{};
  ep.ids = e->d_ids;
  ep.types = use_types ? e->d_types : nullptr;
  ep.lens = e->d_lens;
  ep.B = B; ep.L = L; ep.Lp = Lp; ep.H = H; ep.KB = KB; ep.pos_offset = c.pos_offset; ep.max_pos = c.max_positions;
  ep.word = e->word; ep.pos = e->pos; ep.type = e->type; ep.g = e->eg; ep.b = e->eb; ep.eps = c.ln_eps;
  ep.act = e->act;
  if (KB % 4 == 0 && KB / 4 <= kLnMaxKbw) hipLaunchKernelGGL(k_embed_ln, dim3((unsigned)TB), dim3(256), 0, st, ep);
  else hipLaunchKernelGGL(k_embed_ln_wave, dim3((unsigned)ceil_div(TB, 4)), dim3(256), 0, st, ep);

  const int dh = H / c.n_heads;
  for (int li = 0; li < c.n_layers; ++li) {
    const LayerW &l = e->layers[li];
    GemmParams gq{};
    gq.act = reinterpret_cast<const uint4 *>(e->act); gq.w = reinterpret_cast<const uint4 *>(l.wqk);
    gq.TB = TB; gq.NB = 2 * H / 32; gq.KB = KB; gq.bias_acc = l.bqk; gq.out = e->qk;
    GemmParams gv{};
    gv.act = reinterpret_cast<const uint4 *>(e->act); gv.w = reinterpret_cast<const uint4 *>(l.wv);
    gv.TB = TB; gv.NB = H / 32; gv.KB = KB; gv.bias = l.bv; gv.out = e->vt;
    // 64-wide heads and sequences of <= 128 padded tokens at tile-filling sizes: projections + attention in one kernel
    const bool fused_attn = dh == 64 && (Lp == 32 || Lp == 64 || Lp == 128) && KB % 2 == 0 && TB >= 8 * 16 && !use_skinny(gq);
    if (fused_attn) {
      QkvAttnParams qa{};
      qa.act = reinterpret_cast<const uint4 *>(e->act);
      qa.wqk = reinterpret_cast<const uint4 *>(l.wqk); qa.wv = reinterpret_cast<const uint4 *>(l.wv);
      qa.bqk_acc = l.bqk; qa.bv = l.bv;
      qa.TB = TB; qa.KB = KB; qa.H = H; qa.heads = c.n_heads; qa.Lp = Lp; qa.lens = e->d_lens;
      qa.scale = 1.0f / sqrtf((float)dh); qa.relbias = e->relbias; qa.rel_span = e->rel_span; qa.ctx = e->ctx;
      PatchGrid pg = make_patch_grid(ceil_div(TB, 8), c.n_heads);
      const int64_t n_slots = pg.grid();
      const int64_t grid = std::min<int64_t>(n_slots, std::max(8, e->n_cu / 8 * 8));
      constexpr int lds_qa = 8 * 14 * 1024;
#define ANR_QA(N)                                                                                       \
  {                                                                                                     \
    (void)ensure_dynamic_lds(reinterpret_cast<const void *>(&k_qkv_attn<N>), lds_qa);                   \
    hipLaunchKernelGGL((k_qkv_attn<N>), dim3((unsigned)grid), dim3(512), lds_qa, st, qa, pg, n_slots);  \
  }
      if (Lp == 32) ANR_QA(1) else if (Lp == 64) ANR_QA(2) else ANR_QA(4)
#undef ANR_QA
    } else if (use_skinny(gq) && use_skinny(gv)) {  // small inputs: both projections in one launch
      const int n_qk = (int)((int64_t)gq.NB * ceil_div(gq.TB, 2)), n_v = (int)((int64_t)gv.NB * ceil_div(gv.TB, 2));
      hipLaunchKernelGGL(k_gemm_skinny_qkv, dim3((unsigned)(n_qk + n_v)), dim3(256), 0, st, gq, gv, n_qk);
    } else {
      launch_gemm<EPI_ACT>(e, gq);
      launch_gemm<EPI_VT>(e, gv);
    }
    if (!fused_attn) {
    AttnParams ap{};
    ap.qk = reinterpret_cast<const uint4 *>(e->qk); ap.vt = reinterpret_cast<const uint4 *>(e->vt);
    ap.lens = e->d_lens; ap.B = B; ap.Lp = Lp; ap.H = H; ap.heads = c.n_heads; ap.dh = dh;
    ap.scale = 1.0f / sqrtf((float)dh); ap.ctx = e->ctx;
    ap.relbias = e->relbias; ap.rel_span = e->rel_span;
    const int64_t aw = (int64_t)B * c.n_heads * (Lp / 32);
    if (dh == 32) hipLaunchKernelGGL(k_attention<32>, dim3((unsigned)ceil_div(aw, 4)), dim3(256), 0, st, ap);
    else if (dh == 64) hipLaunchKernelGGL(k_attention<64>, dim3((unsigned)ceil_div(aw, 4)), dim3(256), 0, st, ap);
    else hipLaunchKernelGGL(k_attention<128>, dim3((unsigned)ceil_div(aw, 4)), dim3(256), 0, st, ap);
    }
    GemmParams go{};
    go.act = reinterpret_cast<const uint4 *>(e->ctx); go.w = reinterpret_cast<const uint4 *>(l.wo);
    go.TB = TB; go.NB = H / 32; go.KB = KB; go.bias_acc = l.bo; go.out = e->delta;
    GemmParams g1{};
    g1.act = reinterpret_cast<const uint4 *>(e->act); g1.w = reinterpret_cast<const uint4 *>(l.w1);
    g1.TB = TB; g1.NB = I / 32; g1.KB = KB; g1.bias_acc = l.b1; g1.out = e->ffn;
    GemmParams g2{};
    g2.act = reinterpret_cast<const uint4 *>(e->ffn); g2.w = reinterpret_cast<const uint4 *>(l.w2);
    g2.TB = TB; g2.NB = H / 32; g2.KB = I / 16; g2.bias_acc = l.b2; g2.out = e->delta;
    // Large forwards (all three GEMMs on the 8-wave tile kernel): the layer's FIRST LayerNorm has no pass of its own —
    // the output projection writes u = x + projection and its per-token partial sums, FFN-up consumes u through the folded
    // weights, FFN-down rebuilds LN(u) for its residual and writes the sum the second LayerNorm reads alone (EPI list above).
    int tn_o = 0;
    const bool fold = e->fold_ln == 1 && pp_tile(e, go, &tn_o) && pp_tile(e, g1, nullptr) && pp_tile(e, g2, nullptr) &&
                      ceil_div(go.NB, tn_o) <= kStatSlots;
    if (fold) {
      go.resid = e->act; go.stats_out = e->stats; go.stat_h = H;
      launch_gemm<EPI_RES_STATS>(e, go);  // delta <- u
      g1.act = reinterpret_cast<const uint4 *>(e->delta); g1.w = reinterpret_cast<const uint4 *>(l.w1f);
      g1.bias_acc = l.f1c; g1.fold_s = l.f1s;
      g1.stats_in = e->stats; g1.stat_slots = (int)ceil_div(go.NB, tn_o); g1.stat_h = H; g1.ln_eps = c.ln_eps;
      launch_gemm<EPI_FOLD_GELU>(e, g1);
      g2.resid = e->delta; g2.stats_in = e->stats; g2.stat_slots = g1.stat_slots; g2.stat_h = H; g2.ln_eps = c.ln_eps;
      g2.ln_g = l.ln1g; g2.ln_b = l.ln1b;
      launch_gemm<EPI_RES_LN>(e, g2);     // delta <- LN1(u) + projection, in place
      LnParams l2{e->delta, nullptr, TB, H, KB, l.ln2g, l.ln2b, c.ln_eps, li + 1 == c.n_layers ? e->res : nullptr, e->act};
      launch_layernorm(l2, st);
      continue;
    }
    if (e->fold_ln == 2 && pp_tile(e, go, nullptr) && pp_tile(e, g2, nullptr)) {
      // residual sums formed in the projections' epilogues: each LayerNorm reads one tensor (delta) and writes act
      go.resid = e->act;
      launch_gemm<EPI_RES>(e, go);
      LnParams r1{e->delta, nullptr, TB, H, KB, l.ln1g, l.ln1b, c.ln_eps, nullptr, e->act};
      launch_layernorm(r1, st);
      launch_gemm<EPI_GELU>(e, g1);
      g2.resid = e->act;
      launch_gemm<EPI_RES>(e, g2);
      LnParams r2{e->delta, nullptr, TB, H, KB, l.ln2g, l.ln2b, c.ln_eps, li + 1 == c.n_layers ? e->res : nullptr, e->act};
      launch_layernorm(r2, st);
      continue;
    }
    launch_gemm<EPI_ACT>(e, go);
    LnParams l1{e->act, e->delta, TB, H, KB, l.ln1g, l.ln1b, c.ln_eps, nullptr, e->act};
    launch_layernorm(l1, st);
    launch_gemm<EPI_GELU>(e, g1);
    launch_gemm<EPI_ACT>(e, g2);
    // the last LayerNorm of the forward also leaves its output in f32: the pooling input
    LnParams l2{e->act, e->delta, TB, H, KB, l.ln2g, l.ln2b, c.ln_eps, li + 1 == c.n_layers ? e->res : nullptr, e->act};
    launch_layernorm(l2, st);
  }
  PoolParams pp{e->res, e->d_lens, B, Lp, H, KB, c.pooling, normalize ? 1 : 0, out, rows};
  hipLaunchKernelGGL(k_pool, dim3(B), dim3(256), 0, st, pp);
}

int check_forward_args(const anr_encoder *e, const int32_t *ids, const int32_t *lengths, const int32_t *type_ids, int32_t B,
                       int32_t L) {
  if (!e || !ids || !lengths) return fail(ANR_EINVAL, "null argument");
  if (B <= 0 || L <= 0) return fail(ANR_EINVAL, "B and L must be positive");
  const auto &c = e->cfg;
  if (L + c.pos_offset > c.max_positions) return fail(ANR_EINVAL, "sequence length %d exceeds the position table", L);
  for (int b = 0; b < B; ++b)
    if (lengths[b] <= 0 || lengths[b] > L) return fail(ANR_EINVAL, "lengths[%d] = %d out of range 1..%d", b, lengths[b], L);
  for (int64_t i = 0; i < (int64_t)B * L; ++i) {
    if (ids[i] < 0 || ids[i] >= c.vocab_size) return fail(ANR_EINVAL, "token id %d out of range", ids[i]);
    if (type_ids && (type_ids[i] < 0 || type_ids[i] >= c.type_vocab_size)) return fail(ANR_EINVAL, "type id out of range");
  }
  return ANR_OK;
}

int forward_impl(anr_encoder *e, const int32_t *ids, const int32_t *lengths, const int32_t *type_ids, int32_t B,
                 int32_t L, int32_t normalize, float *out_host, float *out_dev, const int32_t *out_rows) {
  if (!out_host && !out_dev) return fail(ANR_EINVAL, "null argument");
  ANR_TRY(check_forward_args(e, ids, lengths, type_ids, B, L));
  const auto &c = e->cfg;
  DeviceGuard g(e->device);
  if (!g.ok) return fail(ANR_EHIP, "hipSetDevice failed");
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->finalized) return fail(ANR_ESTATE, "encoder weights are not finalized");
  const int Lp = (int)round_up(L, 32);
  ANR_TRY(ensure_ws(e, B, L, Lp));
  hipStream_t st = e->stream;
  // ONE copy from pinned memory carries ids, token types, lengths and output rows (four pageable copies cost ~20 us
  // each in staging: a seventh of a single-query forward)
  {
    const int64_t BL = (int64_t)B * L;
    std::memcpy(e->pin_in, ids, (size_t)BL * sizeof(int));
    if (type_ids) std::memcpy(e->pin_in + BL, type_ids, (size_t)BL * sizeof(int));
    std::memcpy(e->pin_in + 2 * BL, lengths, (size_t)B * sizeof(int));
    if (out_dev && out_rows) std::memcpy(e->pin_in + 2 * BL + B, out_rows, (size_t)B * sizeof(int));
    ANR_HIP(hipMemcpyAsync(e->d_in, e->pin_in, (size_t)(2 * BL + 2 * B) * sizeof(int), hipMemcpyHostToDevice, st));
  }
  const bool direct_out = out_host && (size_t)B * c.hidden * sizeof(float) <= kPinOutBytes;
  float *out = out_dev ? out_dev : (direct_out ? e->pin_out_dev : e->out);
  const int *rows = (out_dev && out_rows) ? e->d_rows : nullptr;
  // (Replaying the kernel sequence from a captured hipGraph was tried for small forwards — a query at a time is ~100
  // dependent launches — and changed nothing: 0.80 vs 0.76 ms; the time is inside the tiny kernels, not between them.)
  enqueue_forward(e, B, L, Lp, type_ids != nullptr, normalize, out, rows);
  ANR_HIP(hipGetLastError());
  if (out_host && !direct_out)
    ANR_HIP(hipMemcpyAsync(out_host, e->out, (size_t)B * c.hidden * sizeof(float), hipMemcpyDeviceToHost, st));
  ANR_HIP(hipStreamSynchronize(st));
  if (direct_out) std::memcpy(out_host, e->pin_out, (size_t)B * c.hidden * sizeof(float));
  return ANR_OK;
}

// ---- concurrent small forwards share forwards (round 4): the queue is combine.hpp's, the forward is forward_impl ----------
}  // namespace

extern "C" {

// lane 0 is the handle itself; lane 1 a view of it, made on first use
static int shared_lane(anr_encoder *e, int lane, anr_encoder **out) {
  *out = e;
  if (lane == 0) return ANR_OK;
  std::lock_guard<std::mutex> ll(e->lane_mu);
  if (!e->lane1) {
    DeviceGuard g(e->device);
    anr_encoder *v = new anr_encoder();
    v->cfg = e->cfg;
    v->device = e->device;
    v->n_cu = e->n_cu;
    v->word = e->word; v->pos = e->pos; v->type = e->type; v->eg = e->eg; v->eb = e->eb;
    v->relbias = e->relbias;
    v->rel_span = e->rel_span;
    for (int i = 0; i < 5; ++i) v->have_emb[i] = e->have_emb[i];
    v->layers = e->layers;
    v->finalized = e->finalized;
    v->fold_ln = e->fold_ln;
    v->view = true;
    if (hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) != hipSuccess) {
      delete v;
      return fail(ANR_EHIP, "hipStreamCreate failed");
    }
    e->lane1 = v;
  }
  *out = e->lane1;
  return ANR_OK;
}

int anr_encoder_forward_shared(anr_encoder *e, const int32_t *ids, const int32_t *lengths, const int32_t *type_ids, int32_t B,
                               int32_t L, int32_t normalize, float *out_host) {
  if (!out_host) return fail(ANR_EINVAL, "null argument");
  ANR_TRY(check_forward_args(e, ids, lengths, type_ids, B, L));  // in the caller's thread: a bad request fails alone
  if ((int64_t)B * round_up(L, 32) > ForwardCombiner::kMaxTokens)
    return forward_impl(e, ids, lengths, type_ids, B, L, normalize, out_host, nullptr, nullptr);
  ForwardCombiner::Req req{ids, lengths, type_ids, B, L, normalize ? 1 : 0, out_host};
  auto run = [e](int lane, const int32_t *i, const int32_t *l, const int32_t *t, int b, int len, int norm, float *out,
                 std::string *err) {
    anr_encoder *le = nullptr;
    int rc = shared_lane(e, lane, &le);
    if (rc == ANR_OK) rc = forward_impl(le, i, l, t, b, len, norm, out, nullptr, nullptr);
    if (rc != ANR_OK) *err = anr_last_error();
    return rc;
  };
  const int rc = e->shared.run(req, e->cfg.hidden, run);
  return rc == ANR_OK ? ANR_OK : fail(rc, "%s", req.err.c_str());
}

int anr_encoder_shared_stats(anr_encoder *e, int64_t *forwards, int64_t *requests) {
  if (!e) return fail(ANR_EINVAL, "null handle");
  e->shared.stats(forwards, requests);
  return ANR_OK;
}

int anr_encoder_forward(anr_encoder *e, const int32_t *ids, const int32_t *lengths, const int32_t *type_ids, int32_t B,
                        int32_t L, int32_t normalize, float *out_host) {
  if (!out_host) return fail(ANR_EINVAL, "null argument");
  return forward_impl(e, ids, lengths, type_ids, B, L, normalize, out_host, nullptr, nullptr);
}

int anr_encoder_forward_dev(anr_encoder *e, const int32_t *ids, const int32_t *lengths, const int32_t *type_ids, int32_t B,
                            int32_t L, int32_t normalize, float *out_dev, const int32_t *out_rows) {
  if (!out_dev) return fail(ANR_EINVAL, "null argument");
  if (out_rows)
    for (int b = 0; b < B; ++b)
      if (out_rows[b] < 0) return fail(ANR_EINVAL, "out_rows[%d] is negative", b);
  return forward_impl(e, ids, lengths, type_ids, B, L, normalize, nullptr, out_dev, out_rows);
}

}  // extern "C"
