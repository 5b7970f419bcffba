// Greedy-decode step for SMALL batches (B <= 4) on gfx950: weight-streaming GEMV kernels with fused prologues/epilogues.
//
// replaces (per generated token): CapDecoder.decode_word's module calls -- nn.Embedding + positional add, per decoder layer the
// self-attention in-projection / SDPA / out-projection, cross-attention q-projection / SDPA / out-projection, the two FFN
// Linears, three LayerNorms, then decoder.norm and the generator (reference model/CapDecoder.py:62-79, torch
// nn/modules/transformer.py:1143-1199) -- 26 launches on the batched kernels, 14 here.
//
// At batch 1 the step is a chain of matrix-VECTOR products: 48 MB of bf16 weights are read once per token and every product
// depends on the previous one.  The MFMA tile kernels run such a product as a split-K GEMM (one output tile, partials,
// reduce): ~7.6 us per launch.  Here one kernel shape covers every stage:
//     prologue (whole workgroup, result x[B][K] fp32 in LDS -- recomputed by EVERY workgroup, the inputs are 2-8 KB):
//        token embedding + positional row | LayerNorm (one or two chained) of the previous stage's pre-norm vector |
//        self / cross attention over the KV cache (one head at a time per wave: q.k by lanes over keys, p.v by lanes over
//        the head dim) | plain vector
//     body: every wave owns a few output features; a weight row is read with 16-byte loads straight into registers (nothing
//        is shared between waves, so no LDS staging), dotted with x held in registers, reduced with wave shuffles
//     epilogue: + bias, GELU / ReLU, + residual, store fp32 (next stage's input) or the compute dtype (q | k | v into the
//        KV-cache slot, logits)
// so the LayerNorm / attention / embedding launches disappear into the consumers' prologues and the residual adds into the
// producers' epilogues; activations between stages are fp32 vectors of <= 2048 elements.  Weights are bf16 (throughput
// mode) or fp32 (parity mode: greedy ids must match the reference exactly).
#include "vct_common.h"

namespace vct {

struct DecP {
  int B, N, K, rpw;                // rpw: output features per wave
  const void* W; long ldw;
  const float* bias;
  int pro;                         // VCT_DEC_PRO_*
  const float* x_in; long ld_x;
  const float* g1; const float* b1; const float* g2; const float* b2;
  const int64_t* ids; long id_stride; const float* table; const float* pos_row;
  const void* q; long q_bs; const void* kc; const void* vc; long kv_ld, kv_bs; int H, Lk;
  int act;
  const float* res; long ld_res;
  void* out; long ld_out; int out_native;     // out_native: store in the weight dtype (KV cache / logits) instead of fp32
  float* x_out;
};

constexpr int DEC_KMAX = 2048;
constexpr int DEC_WAVES = 8;          // waves per workgroup (512 threads: 2 per SIMD, 256 VGPRs each)

template <typename TW> struct WVec;
template <> struct WVec<bf16_t> { static constexpr int VEC = 8; };
template <> struct WVec<float> { static constexpr int VEC = 4; };

template <typename TW> __device__ __forceinline__ void load_w(const TW* p, float (&o)[WVec<TW>::VEC]) {
  if constexpr (sizeof(TW) == 2) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) { o[2 * j] = __uint_as_float(w[j] << 16); o[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
  } else {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
}

// LayerNorm of the vector a wave holds in the GEMV register layout (lane owns VEC consecutive elements of every 64*VEC
// chunk): statistics by wave shuffles, no LDS, no barrier -- every wave normalises its own copy.
template <int NB, int NCH, int VEC>
__device__ __forceinline__ void wave_ln(float (&x)[NB][NCH][VEC], int B, int K, const float* __restrict__ g,
                                        const float* __restrict__ bt, int lane) {
  constexpr int KI = 64 * VEC;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    if (b < B) {
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int u = 0; u < VEC; u++) s += x[b][c][u];
      const float mean = wave_sum(s) / (float)K;
      float q = 0.0f;
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int u = 0; u < VEC; u++) { const float d = x[b][c][u] - mean; q += d * d; }
      const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + 1e-5f);
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int u = 0; u < VEC; u++)
          x[b][c][u] = (x[b][c][u] - mean) * rstd * g[c * KI + lane * VEC + u] + bt[c * KI + lane * VEC + u];
    }
  }
}

// attention of ONE new query per batch row over Lk (<= 64) cached keys; wave w handles the (batch, head) pairs w, w + DEC_WAVES, ...
//   scores : lane l owns key l: q . k_l with hd/VEC independent 16-byte loads, softmax by wave reductions
//   values : the head's [Lk, hd] slice is read as 16-byte vectors, lane = (key within a pass, column chunk): every load of
//            the slice is independent (the first version walked the keys one dependent load at a time: ~25 us per
//            stage at batch 1, 4x that at batch 4); partial sums are folded across the lanes that share a chunk.
template <typename TW, int NB>
__device__ __forceinline__ void block_attn(float (*xs)[DEC_KMAX], const DecP& p) {
  constexpr int VEC = WVec<TW>::VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hd = p.K / p.H;
  const int cpr = hd / VEC;                 // 16-byte chunks per row of the head slice (8 / 16 / 32: a power of two <= 32)
  const int kp = 64 / cpr;                  // keys per pass
  const float scale = 1.0f / sqrtf((float)hd);
  for (int bh = wave; bh < p.B * p.H; bh += DEC_WAVES) {
    const int b = bh / p.H, h = bh % p.H;
    const TW* q = reinterpret_cast<const TW*>(p.q) + (long)b * p.q_bs + h * hd;
    const TW* kc = reinterpret_cast<const TW*>(p.kc) + (long)b * p.kv_bs + h * hd;
    const TW* vc = reinterpret_cast<const TW*>(p.vc) + (long)b * p.kv_bs + h * hd;
    const TW* kr = kc + (long)min(lane, p.Lk - 1) * p.kv_ld;
    float a = 0.0f;
    for (int j = 0; j < hd; j += VEC) {
      float kv[VEC], qv[VEC];
      load_w<TW>(kr + j, kv); load_w<TW>(q + j, qv);
#pragma unroll
      for (int u = 0; u < VEC; u++) a += kv[u] * qv[u];
    }
    const float sc = lane < p.Lk ? a * scale : -INFINITY;
    const float m = wave_max(sc);
    const float e = lane < p.Lk ? expf(sc - m) : 0.0f;
    const float pr = e / wave_sum(e);
    const int chunk = lane % cpr, kin = lane / cpr;
    float o[VEC];
#pragma unroll
    for (int u = 0; u < VEC; u++) o[u] = 0.0f;
    for (int l0 = 0; l0 < p.Lk; l0 += kp) {
      const int l = l0 + kin;
      const float pl = __shfl(pr, min(l, 63));
      float vv[VEC];
      load_w<TW>(vc + (long)min(l, p.Lk - 1) * p.kv_ld + chunk * VEC, vv);
      const float w = l < p.Lk ? pl : 0.0f;
#pragma unroll
      for (int u = 0; u < VEC; u++) o[u] += w * vv[u];
    }
    for (int off = cpr; off < 64; off <<= 1) {
#pragma unroll
      for (int u = 0; u < VEC; u++) o[u] += __shfl_xor(o[u], off);
    }
    if (lane < cpr) {
#pragma unroll
      for (int u = 0; u < VEC; u++) xs[b][h * hd + chunk * VEC + u] = o[u];
    }
  }
}

// rows of W a wave keeps in flight per trip: 64 registers (16 KB per wave) of weights, 32 at batch 4 where the input
// vectors already take up to 128 registers
constexpr int dec_rows_per_trip(int nb, int nch, int vec) {
  const int budget = nb >= 4 ? 32 : 64;
  return budget / (nch * vec) > 0 ? budget / (nch * vec) : 1;
}

// NCH = K / (64 lanes * VEC) 16-byte chunks per lane and row (compile time: registers are sized by it); RT rows per trip such
// that a wave always has 64 registers = 16 KB of weights in flight, issued BEFORE the prologue (they do not depend on it).
template <typename TW, int NB, int NCH>
__global__ __launch_bounds__(64 * DEC_WAVES) void decode_gemv_kernel(const DecP p) {
  constexpr int VEC = WVec<TW>::VEC, KI = 64 * VEC;
  constexpr int RT = dec_rows_per_trip(NB, NCH, VEC);
  __shared__ __attribute__((aligned(16))) float xs[NB][DEC_KMAX];          // attention prologues only
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K;
  const TW* W = reinterpret_cast<const TW*>(p.W);
  const int row0 = (blockIdx.x * DEC_WAVES + wave) * p.rpw;

  float w[RT][NCH][VEC];
  auto load_rows = [&](const int r) {
#pragma unroll
    for (int i = 0; i < RT; i++) {
      const TW* wr = W + (long)min(row0 + r + i, p.N - 1) * p.ldw + lane * VEC;
#pragma unroll
      for (int c = 0; c < NCH; c++) load_w<TW>(wr + c * KI, w[i][c]);
    }
  };
  if (row0 < p.N) load_rows(0);

  // ---- prologue: the input vector(s), in the register layout of the dot products (lane: VEC elements of every chunk) ----
  float xr[NB][NCH][VEC];
  const bool attn = p.pro == VCT_DEC_PRO_SELF_ATTN || p.pro == VCT_DEC_PRO_CROSS_ATTN;
  if (attn) {
    block_attn<TW, NB>(xs, p);                     // heads spread over the workgroup's waves, result through LDS
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int u = 0; u < VEC; u++) xr[b][c][u] = b < p.B ? xs[b][c * KI + lane * VEC + u] : 0.0f;
  } else {
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const int bb = min(b, p.B - 1);
      const float* src;
      if (p.pro == VCT_DEC_PRO_EMBED) src = p.table + p.ids[(long)bb * p.id_stride] * (long)K;
      else src = p.x_in + (long)bb * p.ld_x;
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int u = 0; u < VEC; u += 4) {
          const float4 v = *reinterpret_cast<const float4*>(src + c * KI + lane * VEC + u);
          xr[b][c][u] = v.x; xr[b][c][u + 1] = v.y; xr[b][c][u + 2] = v.z; xr[b][c][u + 3] = v.w;
        }
    }
    if (p.pro == VCT_DEC_PRO_EMBED) {
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int u = 0; u < VEC; u++) {
          const float pv = p.pos_row[c * KI + lane * VEC + u];
#pragma unroll
          for (int b = 0; b < NB; b++) xr[b][c][u] += pv;
        }
    }
    if constexpr (NCH * VEC <= 16) {          // LayerNorm inputs are d_model wide (<= 1024): wider instantiations skip the code
      if (p.pro == VCT_DEC_PRO_LN || p.pro == VCT_DEC_PRO_LN_LN) wave_ln<NB, NCH, VEC>(xr, p.B, K, p.g1, p.b1, lane);
      if (p.pro == VCT_DEC_PRO_LN_LN) wave_ln<NB, NCH, VEC>(xr, p.B, K, p.g2, p.b2, lane);
    }
  }
  if (p.x_out != nullptr && blockIdx.x == 0 && wave == 0) {
#pragma unroll
    for (int b = 0; b < NB; b++)
      if (b < p.B) {
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int u = 0; u < VEC; u++) p.x_out[(long)b * K + c * KI + lane * VEC + u] = xr[b][c][u];
      }
  }

  // ---- body ----------------------------------------------------------------------------------------------------------------
  for (int r = 0; r < p.rpw; r += RT) {
    if (row0 + r >= p.N) break;
    float acc[RT][NB];
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
      for (int b = 0; b < NB; b++) {
        float a = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int u = 0; u < VEC; u++) a += w[i][c][u] * xr[b][c][u];
        acc[i][b] = a;
      }
    if (r + RT < p.rpw && row0 + r + RT < p.N) load_rows(r + RT);      // next trip's weights fly under the reductions
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
      for (int b = 0; b < NB; b++) acc[i][b] = wave_sum(acc[i][b]);
    // lane i finishes row i of the trip
    float mine[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
      mine[b] = 0.0f;
#pragma unroll
      for (int i = 0; i < RT; i++) mine[b] = lane == i ? acc[i][b] : mine[b];
    }
    const int n = row0 + r + lane;
    if (lane < RT && r + lane < p.rpw && n < p.N) {
      const float bias = p.bias != nullptr ? p.bias[n] : 0.0f;
#pragma unroll
      for (int b = 0; b < NB; b++) {
        if (b < p.B) {
          float v = act_f(p.act, mine[b] + bias);
          if (p.res != nullptr) v += p.res[(long)b * p.ld_res + n];
          if (p.out_native) reinterpret_cast<TW*>(p.out)[(long)b * p.ld_out + n] = from_f<TW>(v);
          else reinterpret_cast<float*>(p.out)[(long)b * p.ld_out + n] = v;
        }
      }
    }
  }
}

template <typename TW, int NB, int NCH> static int dec_launch3(const DecP& p, hipStream_t st) {
  const int waves = (p.N + p.rpw - 1) / p.rpw;
  vct::launch(decode_gemv_kernel<TW, NB, NCH>, dim3((waves + DEC_WAVES - 1) / DEC_WAVES), dim3(64 * DEC_WAVES), 0, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}
template <typename TW, int NB> static int dec_launch2(DecP& p, int rows_per_wave, hipStream_t st) {
  constexpr int VEC = WVec<TW>::VEC;
  const int nch = p.K / (64 * VEC);
  const int rt = dec_rows_per_trip(NB, nch, VEC);
  // one trip of RT rows per wave for the layer matrices (few workgroups: the prologue is recomputed by each), two for the
  // vocabulary projection
  p.rpw = rows_per_wave > 0 ? rows_per_wave : rt * (p.N >= 8192 ? 2 : 1);
  switch (nch) {
    case 1: return dec_launch3<TW, NB, 1>(p, st);
    case 2: return dec_launch3<TW, NB, 2>(p, st);
    case 4: return dec_launch3<TW, NB, 4>(p, st);
    case 8: if constexpr (sizeof(TW) == 4) return dec_launch3<TW, NB, 8>(p, st); else return VCT_E_SHAPE;
    default: return VCT_E_SHAPE;
  }
}
template <typename TW> static int dec_launch(DecP& p, int rows_per_wave, hipStream_t st) {
  if (p.B == 1) return dec_launch2<TW, 1>(p, rows_per_wave, st);
  if (p.B == 2) return dec_launch2<TW, 2>(p, rows_per_wave, st);
  return dec_launch2<TW, 4>(p, rows_per_wave, st);
}

}  // namespace vct
using namespace vct;

extern "C" int vct_decode_gemv(const vct_decode_gemv_desc* d, void* stream) {
  if (d == nullptr || d->W == nullptr || d->out == nullptr) return VCT_E_ARG;
  if (d->wdtype != VCT_F32 && d->wdtype != VCT_BF16) return VCT_E_ARG;
  if (d->B < 1 || d->B > 4 || d->N < 1 || d->K < 1 || d->K > DEC_KMAX) return VCT_E_SHAPE;
  const int ki = d->wdtype == VCT_BF16 ? 512 : 256;
  if (d->K % ki) return VCT_E_SHAPE;
  { const int nch = d->K / ki; if (nch != 1 && nch != 2 && nch != 4 && nch != 8) return VCT_E_SHAPE; }
  if ((d->ldw % (d->wdtype == VCT_BF16 ? 8 : 4)) || ((uintptr_t)d->W & 15)) return VCT_E_ALIGN;
  switch (d->pro) {
    case VCT_DEC_PRO_NONE: if (!d->x_in) return VCT_E_ARG; break;
    case VCT_DEC_PRO_LN: if (!d->x_in || !d->g1 || !d->b1) return VCT_E_ARG; if (d->K > 1024) return VCT_E_SHAPE; break;
    case VCT_DEC_PRO_LN_LN:
      if (!d->x_in || !d->g1 || !d->b1 || !d->g2 || !d->b2) return VCT_E_ARG;
      if (d->K > 1024) return VCT_E_SHAPE;
      break;
    case VCT_DEC_PRO_EMBED: if (!d->ids || !d->table || !d->pos_row) return VCT_E_ARG; break;
    case VCT_DEC_PRO_SELF_ATTN:
    case VCT_DEC_PRO_CROSS_ATTN:
      if (!d->q || !d->kc || !d->vc) return VCT_E_ARG;
      if (d->H < 1 || d->K % d->H || d->Lk < 1 || d->Lk > 64) return VCT_E_SHAPE;
      if ((d->K / d->H) % (d->wdtype == VCT_BF16 ? 8 : 4) || (d->kv_ld % (d->wdtype == VCT_BF16 ? 8 : 4)) ||
          (d->q_bs % (d->wdtype == VCT_BF16 ? 8 : 4)) || (d->kv_bs % (d->wdtype == VCT_BF16 ? 8 : 4)))
        return VCT_E_ALIGN;
      if (((uintptr_t)d->q | (uintptr_t)d->kc | (uintptr_t)d->vc) & 15) return VCT_E_ALIGN;
      break;
    default: return VCT_E_ARG;
  }
  DecP p;
  p.B = d->B; p.N = d->N; p.K = d->K;
  p.rpw = 0;
  p.W = d->W; p.ldw = d->ldw; p.bias = d->bias; p.pro = d->pro;
  p.x_in = d->x_in; p.ld_x = d->ld_x;
  p.g1 = d->g1; p.b1 = d->b1; p.g2 = d->g2; p.b2 = d->b2;
  p.ids = d->ids; p.id_stride = d->id_stride; p.table = d->table; p.pos_row = d->pos_row;
  p.q = d->q; p.q_bs = d->q_bs; p.kc = d->kc; p.vc = d->vc; p.kv_ld = d->kv_ld; p.kv_bs = d->kv_bs; p.H = d->H; p.Lk = d->Lk;
  p.act = d->act; p.res = d->res; p.ld_res = d->ld_res;
  p.out = d->out; p.ld_out = d->ld_out; p.out_native = d->out_native;
  p.x_out = d->x_out;
  hipStream_t st = (hipStream_t)stream;
  return d->wdtype == VCT_BF16 ? dec_launch<bf16_t>(p, d->rows_per_wave, st) : dec_launch<float>(p, d->rows_per_wave, st);
}
