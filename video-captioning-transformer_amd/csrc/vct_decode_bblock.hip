// Batched greedy-decode step (2 <= batch <= 256) in THREE launches per decoder layer on gfx950, bf16 weights, MFMA on 16-row tiles.
//
// replaces (per generated token, all captions of the batch at once): CapDecoder.decode_word's module calls -- per
// nn.TransformerDecoderLayer the self-attention in-projection / SDPA / out-projection, norm1, the cross-attention q-projection /
// SDPA / out-projection, norm2, linear1 / GELU / linear2, norm3 (reference model/CapDecoder.py:62-79, torch
// nn/modules/transformer.py:1143-1199): 8 launches per layer on the skinny MFMA kernels (vct_decode_linear / vct_attn_fwd), 3 here.
//
// The batch-1 block design (vct_decode_block.hip) carried to 16-row MFMA tiles: a launch per layer BLOCK instead of a launch per
// projection, because at batch 128 every projection is a 5-7 us dependent launch for < 1 us of work (19 launches = 116 us per token).
//   self block    workgroup (head h, row tile r): x = prologue(rows of r);  q_h | k_h | v_h = x W_in[rows of head h]^T (+ bias) -> cache
//                 slot;  attention of head h for the 16 captions over their cached keys;  PARTIAL out-projection
//                 a_h[rows, :] = o_h W_o[:, head h]^T
//   cross block   workgroup (h, r): q_h = x1 W_q[head h]^T; attention over the memory's K / V; partial out-projection
//   feed-forward  workgroup (c, r), c = 256 hidden units: g = act(x2 W1[rows c]^T + b1);  partial f_c = g W2[:, c]^T
// The partial outputs [rows, 512] fp32 (8 per block) are summed by the PROLOGUE of the consuming launch, in index order, together
// with the residual, the second product's bias and the LayerNorm(s); the last block's by vct_decode_bfinal (norm3 + decoder.norm).
// Weights are read from FRAGMENT-MAJOR packed copies (vct_pack_frag: [N/16][K/32][64 lanes][8 bf16], a wave instruction = 1 KiB
// contiguous; gathering fragments from the [out, in] row-major matrix runs at a third of the L2 rate, tools/ss_probe.hip) that the
// caller refreshes when the weights change; every global load a launch can issue before it has data (weights, partials) goes out
// first.
#include "vct_common.h"

namespace vct {

constexpr int BB_D = 512, BB_H = 8, BB_HD = 64, BB_NT = 512, BB_NW = 8, BB_PMAX = 8, BB_FC = 256;
constexpr int BB_XSTR = BB_D + 8;          // bf16 panel row stride (elements)
constexpr int BB_HSTR = BB_HD + 8;         // per-head [16][64] panels
constexpr int BB_GSTR = BB_FC + 8;         // feed-forward activation panel [16][256]

struct BbX {                               // how a launch builds its 16 input rows (fp32), per row tile
  const int64_t* ids; long id_stride; const float* table; const float* pos_row;      // ids != NULL: table[ids[row * id_stride]] + pos_row
  const float* res; long ld_res; const float* bias;                                  // else res[row] (+ bias) + sum_c part[c][row]
  const float* part; long part_stride; int n_part;
  const float* g1; const float* b1; const float* g2; const float* b2;               // then up to two LayerNorms
  float* x_out; long ld_xout;                                                        // workgroups with blockIdx.x == 0 publish x (the next residual)
};

struct BbP {
  int B;
  BbX x;
  const bf16_t* w_a; int a_tile[3]; const float* b_a;   // first product: packed weight, first packed 16-row tile of q / k / v (head 0), bias
  bf16_t* slot; long slot_bs;                           // self: q | k | v of the consumed token, sample stride (elements)
  const bf16_t* kc; const bf16_t* vc; long kv_ld, kv_bs; int Lk;   // cached keys / values: row stride, sample stride; keys to attend (self: incl. the fresh one)
  const bf16_t* w_b; int b_ksteps;                      // second product: packed weight [32 tiles][b_ksteps]
  float* part_out; long part_stride;
  int act;
};

struct alignas(8) BbV4 { bf16_t e[4]; };
struct alignas(16) BbV8 { bf16_t e[8]; };

__device__ __forceinline__ bf16x8 bb_frag(const bf16_t* pk, const long tile, const int ksteps, const int s, const int lane) {
  return *reinterpret_cast<const bf16x8*>(pk + ((tile * ksteps + s) << 9) + lane * 8);
}

// x rows of this tile -> bf16 panel xb[16][BB_XSTR]; thread = (row tid >> 5, 16 columns); row statistics by half-wave shuffles
__device__ __forceinline__ void bb_build_x(const BbX& s, const int B, const int m0, bf16_t* xb, const bool publish) {
  const int tid = threadIdx.x, row = tid >> 5, c0 = (tid & 31) * 16;
  const long grow = min(m0 + row, B - 1);
  float a[16];
  if (s.ids != nullptr) {
    const float* tr = s.table + s.ids[grow * s.id_stride] * (long)BB_D + c0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 t = *reinterpret_cast<const float4*>(tr + q * 4), pz = *reinterpret_cast<const float4*>(s.pos_row + c0 + q * 4);
      a[q * 4] = t.x + pz.x; a[q * 4 + 1] = t.y + pz.y; a[q * 4 + 2] = t.z + pz.z; a[q * 4 + 3] = t.w + pz.w;
    }
  } else {
    float4 r4[4], b4[4];
    const float* bp = s.bias != nullptr ? s.bias : s.res;                 // absent inputs: a valid stand-in address, masked below
    const float* pp = s.n_part > 0 ? s.part : s.res;
    const long pstr = s.n_part > 0 ? s.part_stride : 0, prow = s.n_part > 0 ? grow * BB_D : 0;
    const int np1 = max(s.n_part - 1, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      r4[q] = *reinterpret_cast<const float4*>(s.res + grow * s.ld_res + c0 + q * 4);
      b4[q] = *reinterpret_cast<const float4*>(bp + (s.bias != nullptr ? c0 + q * 4 : 0));
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      a[q * 4] = r4[q].x; a[q * 4 + 1] = r4[q].y; a[q * 4 + 2] = r4[q].z; a[q * 4 + 3] = r4[q].w;
      if (s.bias != nullptr) { a[q * 4] += b4[q].x; a[q * 4 + 1] += b4[q].y; a[q * 4 + 2] += b4[q].z; a[q * 4 + 3] += b4[q].w; }
    }
    // the partial outputs in two rounds of four (64 registers of loads in flight beside the launch's weight fragments), index order
#pragma unroll
    for (int half = 0; half < BB_PMAX / 4; half++) {
      float4 p4[4][4];
#pragma unroll
      for (int c = 0; c < 4; c++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          p4[c][q] = *reinterpret_cast<const float4*>(pp + min(half * 4 + c, np1) * pstr + prow + (s.n_part > 0 ? c0 + q * 4 : 0));
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (half * 4 + c < s.n_part) {
#pragma unroll
          for (int q = 0; q < 4; q++) { a[q * 4] += p4[c][q].x; a[q * 4 + 1] += p4[c][q].y; a[q * 4 + 2] += p4[c][q].z; a[q * 4 + 3] += p4[c][q].w; }
        }
    }
  }
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const float* g = pass == 0 ? s.g1 : s.g2;
    const float* bt = pass == 0 ? s.b1 : s.b2;
    if (g == nullptr) break;
    float4 gg[4], bb[4];                                  // issued ahead of the statistics (one memory round trip less per norm)
#pragma unroll
    for (int q = 0; q < 4; q++) { gg[q] = *reinterpret_cast<const float4*>(g + c0 + q * 4); bb[q] = *reinterpret_cast<const float4*>(bt + c0 + q * 4); }
    float sum = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; e++) sum += a[e];
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * (1.0f / (float)BB_D);
    float sq = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; e++) { const float c = a[e] - mean; sq += c * c; }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / (float)BB_D) + 1e-5f);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      a[q * 4] = (a[q * 4] - mean) * rstd * gg[q].x + bb[q].x; a[q * 4 + 1] = (a[q * 4 + 1] - mean) * rstd * gg[q].y + bb[q].y;
      a[q * 4 + 2] = (a[q * 4 + 2] - mean) * rstd * gg[q].z + bb[q].z; a[q * 4 + 3] = (a[q * 4 + 3] - mean) * rstd * gg[q].w + bb[q].w;
    }
  }
  if (publish && s.x_out != nullptr && m0 + row < B) {
#pragma unroll
    for (int q = 0; q < 4; q++) *reinterpret_cast<float4*>(s.x_out + grow * s.ld_xout + c0 + q * 4) = float4{a[q * 4], a[q * 4 + 1], a[q * 4 + 2], a[q * 4 + 3]};
  }
  BbV8 o0, o1;
#pragma unroll
  for (int e = 0; e < 8; e++) { o0.e[e] = f2bf(a[e]); o1.e[e] = f2bf(a[8 + e]); }
  *reinterpret_cast<BbV8*>(xb + row * BB_XSTR + c0) = o0;
  *reinterpret_cast<BbV8*>(xb + row * BB_XSTR + c0 + 8) = o1;
}

// ---- attention block: self (NPROJ = 3) or cross (NPROJ = 1); workgroup = (head, row tile); LK = register bound of the cached positions --
template <int NPROJ, int LK>
__global__ __launch_bounds__(BB_NT, 2) void bb_attn_kernel(const BbP p) {
  constexpr int NTA = NPROJ * 4;                       // 16-column tiles of the first product (q [| k | v] of one head)
  constexpr int TPW = NTA / 4;                         // tiles per wave: waves 0-3 take K steps 0-7, waves 4-7 K steps 8-15
  __shared__ __attribute__((aligned(16))) bf16_t xb[16 * BB_XSTR];
  __shared__ f32x4 red[NTA][64];
  __shared__ __attribute__((aligned(16))) bf16_t hq[16 * BB_HSTR], hk[16 * BB_HSTR], hv[16 * BB_HSTR], ho[16 * BB_HSTR];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int h = blockIdx.x, m0 = blockIdx.y * 16;
  const int wq = wave & 3, kh = wave >> 2;

  // ---- every weight fragment of the launch: issued before anything else -----------------------------------------------------------
  bf16x8 wa[TPW][8], wb[4][2];
#pragma unroll
  for (int i = 0; i < TPW; i++) {
    const int g = wq * TPW + i;                        // tile inside the head's q | k | v group: which = g / 4, j = g % 4
    const long tile = p.a_tile[g >> 2] + h * 4 + (g & 3);
#pragma unroll
    for (int s = 0; s < 8; s++) wa[i][s] = bb_frag(p.w_a, tile, 16, kh * 8 + s, lane);
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int s = 0; s < 2; s++) wb[i][s] = bb_frag(p.w_b, wave * 4 + i, 16, h * 2 + s, lane);

  bb_build_x(p.x, p.B, m0, xb, blockIdx.x == 0);
  __syncthreads();

  // ---- q [| k | v] of head h for the 16 rows ------------------------------------------------------------------------------------------
  f32x4 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; i++) acc[i] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 8; s++) {
    const bf16x8 af = *reinterpret_cast<const bf16x8*>(xb + li * BB_XSTR + (kh * 8 + s) * 32 + lg * 8);
#pragma unroll
    for (int i = 0; i < TPW; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[i][s], af, acc[i], 0, 0, 0);
  }
  // cached keys / values of this wave's two rows: issued now, they land while the halves are reduced
  const int Lc = NPROJ == 3 ? p.Lk - 1 : p.Lk;         // positions read from the cache (self: the fresh one comes from LDS)
  uint4 kr[2][8];
  bf16_t vr[2][LK];
#pragma unroll
  for (int rr = 0; rr < 2; rr++) {
    const long grow = min(m0 + wave * 2 + rr, p.B - 1);
    const bf16_t* kb = p.kc + grow * p.kv_bs + (long)min(lane, max(Lc - 1, 0)) * p.kv_ld + h * BB_HD;
#pragma unroll
    for (int u = 0; u < 8; u++) kr[rr][u] = *reinterpret_cast<const uint4*>(kb + u * 8);
    const bf16_t* vb = p.vc + grow * p.kv_bs + h * BB_HD + lane;
#pragma unroll
    for (int j = 0; j < LK; j++) vr[rr][j] = vb[(long)min(j, max(Lc - 1, 0)) * p.kv_ld];
  }
  if (kh == 1) {
#pragma unroll
    for (int i = 0; i < TPW; i++) red[wq * TPW + i][lane] = acc[i];
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int i = 0; i < TPW; i++) {
      const int g = wq * TPW + i, which = g >> 2, col = (g & 3) * 16 + lg * 4;
      const f32x4 o = red[g][lane];
      const float4 bb = *reinterpret_cast<const float4*>(p.b_a + (NPROJ == 3 ? which * BB_D : 0) + h * BB_HD + col);
      BbV4 v;
      v.e[0] = f2bf(acc[i][0] + o[0] + bb.x); v.e[1] = f2bf(acc[i][1] + o[1] + bb.y);
      v.e[2] = f2bf(acc[i][2] + o[2] + bb.z); v.e[3] = f2bf(acc[i][3] + o[3] + bb.w);
      bf16_t* dst = which == 0 ? hq : (which == 1 ? hk : hv);
      *reinterpret_cast<BbV4*>(dst + li * BB_HSTR + col) = v;
      if (NPROJ == 3 && m0 + li < p.B) *reinterpret_cast<BbV4*>(p.slot + (long)(m0 + li) * p.slot_bs + which * BB_D + h * BB_HD + col) = v;
    }
  }
  __syncthreads();

  // ---- attention: wave w = rows 2w, 2w + 1; lane j = cached position (scores), lane d = head dimension (values) ----------------------
#pragma unroll
  for (int rr = 0; rr < 2; rr++) {
    const int row = wave * 2 + rr;
    float sc = 0.0f;
    {
      const bool fresh = NPROJ == 3 && lane == Lc;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t kw[4] = {kr[rr][u].x, kr[rr][u].y, kr[rr][u].z, kr[rr][u].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int c = u * 8 + e * 2;
          const float k0 = fresh ? bf2f(hk[row * BB_HSTR + c]) : __uint_as_float(kw[e] << 16);
          const float k1 = fresh ? bf2f(hk[row * BB_HSTR + c + 1]) : __uint_as_float(kw[e] & 0xffff0000u);
          sc += k0 * bf2f(hq[row * BB_HSTR + c]) + k1 * bf2f(hq[row * BB_HSTR + c + 1]);
        }
      }
    }
    sc = lane < p.Lk ? sc * 0.125f : -INFINITY;
    const float mx = wave_max(sc);
    const float e = lane < p.Lk ? __expf(sc - mx) : 0.0f;
    const float pr = e / wave_sum(e);
    float o = NPROJ == 3 ? __shfl(pr, Lc) * bf2f(hv[row * BB_HSTR + lane]) : 0.0f;
#pragma unroll
    for (int j = 0; j < LK; j++)
      if (j < Lc) o += __shfl(pr, j) * bf2f(vr[rr][j]);
    ho[row * BB_HSTR + lane] = f2bf(o);
  }
  __syncthreads();

  // ---- partial out-projection: a_h[rows, n] = sum_{c in head h} o_h[rows, c] W_o[n, c], wave w = columns 64w .. 64w + 63 -------------
  f32x4 pa[4];
#pragma unroll
  for (int i = 0; i < 4; i++) pa[i] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const bf16x8 af = *reinterpret_cast<const bf16x8*>(ho + li * BB_HSTR + s * 32 + lg * 8);
#pragma unroll
    for (int i = 0; i < 4; i++) pa[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[i][s], af, pa[i], 0, 0, 0);
  }
  if (m0 + li < p.B) {
    float* dst = p.part_out + (long)h * p.part_stride + (long)(m0 + li) * BB_D + wave * 64 + lg * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<f32x4*>(dst + i * 16) = pa[i];
  }
}

// ---- feed-forward block: workgroup = (256 hidden units c, row tile) ------------------------------------------------------------------
__global__ __launch_bounds__(BB_NT, 2) void bb_ffn_kernel(const BbP p) {
  __shared__ __attribute__((aligned(16))) bf16_t xb[16 * BB_XSTR];
  __shared__ __attribute__((aligned(16))) bf16_t gb[16 * BB_GSTR];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int c = blockIdx.x, m0 = blockIdx.y * 16;

  // linear1 rows 256c + 32w .. + 31 (two tiles) over the whole K = 512: 32 fragments per wave, all in flight
  bf16x8 w1[2][16];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int s = 0; s < 16; s++) w1[i][s] = bb_frag(p.w_a, (long)c * 16 + wave * 2 + i, 16, s, lane);

  bb_build_x(p.x, p.B, m0, xb, blockIdx.x == 0);
  __syncthreads();

  f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
  for (int s = 0; s < 16; s++) {
    const bf16x8 af = *reinterpret_cast<const bf16x8*>(xb + li * BB_XSTR + s * 32 + lg * 8);
#pragma unroll
    for (int i = 0; i < 2; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[i][s], af, acc[i], 0, 0, 0);
  }
  // linear2 columns 256c .. 256c + 255 (K steps 8c .. 8c + 7), output tiles 4w .. 4w + 3: issued into the registers linear1 has freed
  bf16x8 w2[4][8];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int s = 0; s < 8; s++) w2[i][s] = bb_frag(p.w_b, wave * 4 + i, p.b_ksteps, c * 8 + s, lane);
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int col = wave * 32 + i * 16 + lg * 4;
    const float4 bb = *reinterpret_cast<const float4*>(p.b_a + c * BB_FC + col);
    BbV4 v;
    v.e[0] = f2bf(act_fast_f(p.act, acc[i][0] + bb.x)); v.e[1] = f2bf(act_fast_f(p.act, acc[i][1] + bb.y));
    v.e[2] = f2bf(act_fast_f(p.act, acc[i][2] + bb.z)); v.e[3] = f2bf(act_fast_f(p.act, acc[i][3] + bb.w));
    *reinterpret_cast<BbV4*>(gb + li * BB_GSTR + col) = v;
  }
  __syncthreads();
  f32x4 pa[4];
#pragma unroll
  for (int i = 0; i < 4; i++) pa[i] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 8; s++) {
    const bf16x8 af = *reinterpret_cast<const bf16x8*>(gb + li * BB_GSTR + s * 32 + lg * 8);
#pragma unroll
    for (int i = 0; i < 4; i++) pa[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[i][s], af, pa[i], 0, 0, 0);
  }
  if (m0 + li < p.B) {
    float* dst = p.part_out + (long)c * p.part_stride + (long)(m0 + li) * BB_D + wave * 64 + lg * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<f32x4*>(dst + i * 16) = pa[i];
  }
}

// ---- closing rows: y = LN2(LN1(res + bias + sum of partials)) as bf16 (the vocabulary projection's input); one row tile per workgroup ----
__global__ __launch_bounds__(BB_NT) void bb_final_kernel(const BbX x, const int B, bf16_t* y, const long ldy) {
  __shared__ __attribute__((aligned(16))) bf16_t xb[16 * BB_XSTR];
  const int m0 = blockIdx.x * 16;
  bb_build_x(x, B, m0, xb, false);
  __syncthreads();
  const int tid = threadIdx.x, row = tid >> 5, c0 = (tid & 31) * 16;
  if (m0 + row < B) {
    *reinterpret_cast<BbV8*>(y + (long)(m0 + row) * ldy + c0) = *reinterpret_cast<const BbV8*>(xb + row * BB_XSTR + c0);
    *reinterpret_cast<BbV8*>(y + (long)(m0 + row) * ldy + c0 + 8) = *reinterpret_cast<const BbV8*>(xb + row * BB_XSTR + c0 + 8);
  }
}

// fragment-major packing: dst[(tile * ksteps + s) * 512 + lane * 8 + j] = W[tile * 16 + (lane & 15)][s * 32 + (lane >> 4) * 8 + j]
__global__ __launch_bounds__(256) void pack_frag_kernel(const bf16_t* __restrict__ w, const long ldw, const int ntiles, const int ksteps,
                                                        bf16_t* __restrict__ dst) {
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= (long)ntiles * ksteps * 64) return;
  const int lane = (int)(v & 63);
  const long fs = v >> 6;
  const long tile = fs / ksteps;
  const int s = (int)(fs - tile * ksteps);
  *reinterpret_cast<BbV8*>(dst + v * 8) = *reinterpret_cast<const BbV8*>(w + (tile * 16 + (lane & 15)) * ldw + s * 32 + (lane >> 4) * 8);
}

}  // namespace vct
using namespace vct;

extern "C" int vct_pack_frag(const void* w, int64_t ldw, int rows, int cols, void* dst, void* stream) {
  if (!w || !dst) return VCT_E_ARG;
  if (rows < 16 || cols < 32 || (rows % 16) || (cols % 32)) return VCT_E_SHAPE;
  if ((ldw % 8) || ((uintptr_t)w & 15) || ((uintptr_t)dst & 15)) return VCT_E_ALIGN;
  const long nv = (long)(rows / 16) * (cols / 32) * 64;
  vct::launch(pack_frag_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const bf16_t*>(w),
              (long)ldw, rows / 16, cols / 32, reinterpret_cast<bf16_t*>(dst));
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_decode_bblock_supported(int dtype, int d, int H, int ff, int B, int Lk) {
  return (dtype == VCT_BF16 && d == BB_D && H == BB_H && ff >= BB_FC && (ff % BB_FC) == 0 && ff / BB_FC <= BB_PMAX && B >= 1 && B <= 256 &&
          Lk >= 1 && Lk <= 64) ? 1 : 0;
}

static int bb_fill_x(BbX& x, const vct_decode_bblock_desc* q) {
  x.ids = q->ids; x.id_stride = q->id_stride; x.table = q->table; x.pos_row = q->pos_row;
  x.res = q->res; x.ld_res = q->ld_res; x.bias = q->res_bias; x.part = q->part; x.part_stride = q->part_stride; x.n_part = q->part ? q->n_part : 0;
  x.g1 = q->g1; x.b1 = q->b1; x.g2 = q->g2; x.b2 = q->b2; x.x_out = q->x_out; x.ld_xout = q->ld_xout;
  if (x.ids != nullptr && (!x.table || !x.pos_row)) return VCT_E_ARG;
  if (x.ids == nullptr && !x.res) return VCT_E_ARG;
  if ((x.g1 == nullptr) != (x.b1 == nullptr) || (x.g2 == nullptr) != (x.b2 == nullptr) || (x.g2 && !x.g1)) return VCT_E_ARG;
  if (x.n_part < 0 || x.n_part > BB_PMAX) return VCT_E_SHAPE;
  if (x.res && (x.ld_res % 4)) return VCT_E_ALIGN;
  return VCT_OK;
}

extern "C" int vct_decode_bblock(const vct_decode_bblock_desc* q, void* stream) {
  if (q == nullptr) return VCT_E_ARG;
  if (q->kind < 0 || q->kind > 3) return VCT_E_ARG;
  if (q->B < 1 || q->B > 256) return VCT_E_SHAPE;
  BbP p;
  p.B = q->B;
  int rc = bb_fill_x(p.x, q);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int tiles = (q->B + 15) / 16;
  if (q->kind == 3) {                                           // closing rows
    if (!q->y_out || (q->ld_y % 8) || ((uintptr_t)q->y_out & 15)) return VCT_E_ARG;
    vct::launch(bb_final_kernel, dim3(tiles), dim3(BB_NT), 0, st, p.x, q->B, reinterpret_cast<bf16_t*>(q->y_out), (long)q->ld_y);
    VCT_CHECK_LAUNCH();
    return VCT_OK;
  }
  if (!q->w_a || !q->b_a || !q->w_b || !q->part_out || (((uintptr_t)q->w_a | (uintptr_t)q->w_b | (uintptr_t)q->part_out) & 15)) return VCT_E_ARG;
  p.w_a = reinterpret_cast<const bf16_t*>(q->w_a); p.b_a = q->b_a;
  p.a_tile[0] = q->a_tile[0]; p.a_tile[1] = q->a_tile[1]; p.a_tile[2] = q->a_tile[2];
  p.w_b = reinterpret_cast<const bf16_t*>(q->w_b); p.b_ksteps = q->b_ksteps;
  p.part_out = q->part_out; p.part_stride = q->part_out_stride;
  p.slot = reinterpret_cast<bf16_t*>(q->slot); p.slot_bs = q->slot_bs;
  p.kc = reinterpret_cast<const bf16_t*>(q->kc); p.vc = reinterpret_cast<const bf16_t*>(q->vc); p.kv_ld = q->kv_ld; p.kv_bs = q->kv_bs; p.Lk = q->Lk;
  p.act = q->act;
  if (q->kind == 2) {
    if (q->ff < BB_FC || (q->ff % BB_FC) || q->ff / BB_FC > BB_PMAX || q->b_ksteps != q->ff / 32) return VCT_E_SHAPE;
    vct::launch(bb_ffn_kernel, dim3(q->ff / BB_FC, tiles), dim3(BB_NT), 0, st, p);
  } else {
    if (!q->kc || !q->vc || q->Lk < 1 || q->Lk > 64 || (q->kv_ld % 8) || (((uintptr_t)q->kc | (uintptr_t)q->vc) & 15) || q->b_ksteps != 16) return VCT_E_ARG;
    if (q->kind == 0 && (!q->slot || (q->slot_bs % 4))) return VCT_E_ARG;
    const bool small = q->Lk <= 32;
    if (q->kind == 0) {
      if (small) vct::launch(bb_attn_kernel<3, 32>, dim3(BB_H, tiles), dim3(BB_NT), 0, st, p);
      else vct::launch(bb_attn_kernel<3, 64>, dim3(BB_H, tiles), dim3(BB_NT), 0, st, p);
    } else {
      if (small) vct::launch(bb_attn_kernel<1, 32>, dim3(BB_H, tiles), dim3(BB_NT), 0, st, p);
      else vct::launch(bb_attn_kernel<1, 64>, dim3(BB_H, tiles), dim3(BB_NT), 0, st, p);
    }
  }
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}
