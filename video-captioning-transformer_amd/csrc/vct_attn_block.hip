// Fused attention block forward for short sequences on gfx950 (bf16 throughput mode):
//
//     y = LayerNorm(res + dropout(out_proj(MHA_core(q, k, v))))         -- ONE launch, one workgroup per batch element
//
// replaces (per attention block of nn.TransformerEncoderLayer / DecoderLayer, torch nn/modules/transformer.py:951-957,
// 1143-1167 as built at MMEncoder.py:236-238 / CapDecoder.py:18-20): the SDPA core, the out_proj nn.Linear, dropout,
// the residual add and the LayerNorm -- three launches (vct_attn_fwd, vct_gemm, vct_add_ln_fwd) and two HBM round
// trips of a [tokens, d] activation on the unfused path.
//
// Why this shape.  At the measured batch (B = 256, 13-19 tokens per sample, d = 512) every kernel of the block is
// latency-bound: 2048 single-wave attention problems, a 2.5 GFLOP GEMM, a 5 MB LayerNorm -- 10 + 13 + 6 us for ~1 us of
// MFMA work.  A row-block-stationary fusion is only affordable where the weight a workgroup must stream is small:
// out_proj is [d, d] = 512 KB of bf16, so one workgroup per SAMPLE (= one per CU at B = 256) pulls it once from its
// XCD's L2 (all 256 workgroups read the same 512 KB: it is L2-resident) while the attention output never leaves LDS.
//   wave h (of H):  phase 1  attention of head h (vct_attn_core.h), O tile -> global (saved for dW_o) and -> LDS panel
//                   phase 2  columns h*hd .. (h+1)*hd of  panel[rows, d] x W_o^T : A fragments from LDS, B fragments
//                            straight from global memory to registers (each wave owns its W_o rows exclusively, nothing
//                            to share through LDS), double-buffered two 64-deep K chunks ahead; the first group is
//                            issued BEFORE phase 1 and lands under the attention's latency chain.  K is consumed in a
//                            permuted order (lane group g takes k = 16g .. 16g+15 of each 64-chunk, as two MFMA steps)
//                            so that every lane reads 32 contiguous bytes and a 16-lane group covers whole 128-B lines.
//                   phase 3  per-wave LDS transpose -> 16-byte rows: + bias, saved pre-dropout output a (bf16, what the
//                            LayerNorm backward re-reads), dropout, + residual, two-pass LayerNorm statistics
//                            exchanged between the H waves through LDS, y.
// The saved tensors (o, a, mean, rstd) and the dropout index streams are exactly those of the unfused kernels, so the
// unfused backward kernels (vct_attn_bwd, vct_add_ln_bwd, the dX / dW GEMMs) run unchanged behind it.
#include "vct_attn_core.h"

namespace vct {

struct AttnBlockP {
  AttnP a;
  const bf16_t* wo; long ldw;
  const float* bo;
  const bf16_t* res; long ld_res;
  const float* gamma; const float* beta;
  bf16_t* aout; long ld_a;
  bf16_t* y; long ld_y;
  float* mean; float* rstd;
  uint32_t site2;
  int M;                 // B * Lq (row clamp for loads of padded tile rows)
  int wave_bytes;        // per-wave LDS region (attention staging, later the epilogue transpose)
  int panel_off, red_off;
  int flags;             // experiments (desc.reserved): 1 skip the attention phase, 2 skip the projection, 4 no chunk stagger
};

struct alignas(16) BV8 { bf16_t e[8]; };

template <int DT, int RT, int NW>
__global__ __launch_bounds__(64 * NW, (NW >= 4 ? NW / 4 : 1)) void attn_block_fwd_kernel(const AttnBlockP p) {
  constexpr int HD = DT * 16, ROWS = RT * 16, D = NW * HD;
  constexpr int NC = D / 64;                       // 64-deep K chunks
  constexpr int G = NC >= 2 ? 2 : 1;               // chunks per prefetch group
  constexpr int NG = NC / G;
  static_assert(D % 64 == 0 && NC % G == 0, "d_model must be a multiple of 64 (128 when >= 128)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, h = wave;
  const int li = lane & 15, lg = lane >> 4;
  bf16_t* panel = reinterpret_cast<bf16_t*>(smem + p.panel_off);
  constexpr int PSTR = D + 8;

  // ---- W_o fragments: group `gi` = chunks gi*G .. gi*G+G-1 of this wave's HD output columns ------------------------
  const bf16_t* wbase = p.wo + (long)(h * HD + li) * p.ldw + lg * 16;
  // every workgroup reads the same W_o: walking K from a different chunk per workgroup (the order of a K reduction is
  // free) spreads the simultaneous requests of an XCD's CUs over its L2 channels instead of queueing them on one line
  const int c0 = (p.flags & 4) ? 0 : (int)(blockIdx.x % NC);
  bf16x8 bq[2][G][DT][2];
  auto load_group = [&](auto GI, auto SLOT) {
    constexpr int gi = decltype(GI)::value, slot = decltype(SLOT)::value;
#pragma unroll
    for (int c = 0; c < G; c++)
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const bf16_t* src = wbase + (long)(dt * 16) * p.ldw + ((gi * G + c + c0) % NC) * 64;
        bq[slot][c][dt][0] = *reinterpret_cast<const bf16x8*>(src);
        bq[slot][c][dt][1] = *reinterpret_cast<const bf16x8*>(src + 8);
      }
  };
  load_group(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});

  // ---- phase 1: attention of head h ---------------------------------------------------------------------------------
  if (!(p.flags & 1))
    attn_fwd_wave<bf16_t, DT>(p.a, b, h, b * p.a.H + h, smem + (size_t)wave * p.wave_bytes, lane, panel, (long)PSTR);

  // residual rows / bias / LayerNorm parameters of this lane's epilogue vectors: issued now, consumed after phase 2
  constexpr int VPR = HD / 8;                      // 8-column vectors per row of the wave's slice
  constexpr int NV = ROWS * VPR;
  constexpr int IT = (NV + 63) / 64;
  BV8 rres[IT];
  float bias8[IT][8];
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int v = min(it * 64 + lane, NV - 1);
    const int row = v / VPR, col = h * HD + (v % VPR) * 8;
    const int grow = min(b * p.a.Lq + min(row, p.a.Lq - 1), p.M - 1);
    rres[it] = *reinterpret_cast<const BV8*>(p.res + (long)grow * p.ld_res + col);
    const float4 b0 = *reinterpret_cast<const float4*>(p.bo + col), b1 = *reinterpret_cast<const float4*>(p.bo + col + 4);
    bias8[it][0] = b0.x; bias8[it][1] = b0.y; bias8[it][2] = b0.z; bias8[it][3] = b0.w;
    bias8[it][4] = b1.x; bias8[it][5] = b1.y; bias8[it][6] = b1.z; bias8[it][7] = b1.w;
  }
  __syncthreads();                                 // every head's O tile is in the panel

  // ---- phase 2: out_proj columns of this wave -----------------------------------------------------------------------
  f32x4 acc[RT][DT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int dt = 0; dt < DT; dt++) acc[rt][dt] = f32x4{0, 0, 0, 0};
  const bf16_t* abase = panel + li * PSTR + lg * 16;
  if (!(p.flags & 2))
  static_for<NG>([&](auto GI) {
    constexpr int gi = decltype(GI)::value;
    if constexpr (gi + 1 < NG) load_group(std::integral_constant<int, gi + 1>{}, std::integral_constant<int, (gi + 1) & 1>{});
#pragma unroll
    for (int c = 0; c < G; c++) {
      bf16x8 af[RT][2];
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
        const bf16_t* src = abase + rt * 16 * PSTR + ((gi * G + c + c0) % NC) * 64;
        af[rt][0] = *reinterpret_cast<const bf16x8*>(src);
        af[rt][1] = *reinterpret_cast<const bf16x8*>(src + 8);
      }
#pragma unroll
      for (int s = 0; s < 2; s++)
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
          for (int dt = 0; dt < DT; dt++)
            acc[rt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt][s], bq[gi & 1][c][dt][s], acc[rt][dt], 0, 0, 0);
    }
  });

  // ---- phase 3: transpose through this wave's LDS region, epilogue on 8-column row vectors --------------------------
  constexpr int SSTR = HD + 4;
  float* stg = reinterpret_cast<float*>(smem + (size_t)wave * p.wave_bytes);       // [ROWS][SSTR], aliases the Q/K/V staging
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int r = 0; r < 4; r++) stg[(rt * 16 + lg * 4 + r) * SSTR + dt * 16 + li] = acc[rt][dt][r];
  __syncthreads();
  float* red0 = reinterpret_cast<float*>(smem + p.red_off);     // [NW][ROWS] row sums
  float* red1 = red0 + NW * ROWS;                               // [NW][ROWS] squared deviations
  const Dropout dr = make_dropout(p.a.seed, p.site2, p.a.p_drop);
  float sv[IT][8];
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int v = it * 64 + lane;
    const bool active = v < NV;
    const int vc = min(v, NV - 1);
    const int row = vc / VPR, cv = vc % VPR, col = h * HD + cv * 8;
    const int grow = b * p.a.Lq + row;
    const bool valid = active && row < p.a.Lq;
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(stg + row * SSTR + cv * 8);
    const f32x4 t1 = *reinterpret_cast<const f32x4*>(stg + row * SSTR + cv * 8 + 4);
    const float t[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
    BV8 av;
    float part = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      av.e[j] = f2bf(t[j] + bias8[it][j]);        // the saved activation is the bf16 value, and the norm is built on it
      const float s = bf2f(av.e[j]) * drop_mult(dr, (uint32_t)grow * (uint32_t)D + (uint32_t)(col + j)) + bf2f(rres[it].e[j]);
      sv[it][j] = s;
      part += s;
    }
    if (valid) *reinterpret_cast<BV8*>(p.aout + (long)grow * p.ld_a + col) = av;
#pragma unroll
    for (int o = 1; o < VPR; o <<= 1) part += __shfl_xor(part, o);
    if (active && cv == 0) red0[wave * ROWS + row] = part;
  }
  __syncthreads();
  float mean_r[IT], rstd_r[IT];
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int vc = min(it * 64 + lane, NV - 1);
    const int row = vc / VPR, cv = vc % VPR;
    float m = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; w++) m += red0[w * ROWS + row];
    m *= (1.0f / (float)D);
    mean_r[it] = m;
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) { const float c = sv[it][j] - m; sq += c * c; }
#pragma unroll
    for (int o = 1; o < VPR; o <<= 1) sq += __shfl_xor(sq, o);
    if (it * 64 + lane < NV && cv == 0) red1[wave * ROWS + row] = sq;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int v = it * 64 + lane;
    const int vc = min(v, NV - 1);
    const int row = vc / VPR, cv = vc % VPR, col = h * HD + cv * 8;
    const int grow = b * p.a.Lq + row;
    const bool valid = v < NV && row < p.a.Lq;
    float var = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; w++) var += red1[w * ROWS + row];
    const float rstd = 1.0f / sqrtf(var * (1.0f / (float)D) + 1e-5f);
    const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + col), g1 = *reinterpret_cast<const float4*>(p.gamma + col + 4);
    const float4 e0 = *reinterpret_cast<const float4*>(p.beta + col), e1 = *reinterpret_cast<const float4*>(p.beta + col + 4);
    const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float bt[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    BV8 yv;
#pragma unroll
    for (int j = 0; j < 8; j++) yv.e[j] = f2bf((sv[it][j] - mean_r[it]) * rstd * gm[j] + bt[j]);
    if (valid) {
      *reinterpret_cast<BV8*>(p.y + (long)grow * p.ld_y + col) = yv;
      if (h == 0 && cv == 0) { p.mean[grow] = mean_r[it]; p.rstd[grow] = rstd; }
    }
  }
}

template <int DT, int RT, int NW>
static int attn_block_launch(AttnBlockP& p, hipStream_t st) {
  constexpr int HD = DT * 16, ROWS = RT * 16, D = NW * HD;
  const size_t att = attn_lds_bytes<bf16_t, DT>(p.a.Lq, p.a.Lk, false);
  const size_t stg = (size_t)ROWS * (HD + 4) * sizeof(float);
  size_t wave_bytes = att > stg ? att : stg;
  wave_bytes = (wave_bytes + 15) & ~(size_t)15;
  p.wave_bytes = (int)wave_bytes;
  p.panel_off = (int)(wave_bytes * NW);
  const size_t panel = (size_t)ROWS * (D + 8) * sizeof(bf16_t);
  p.red_off = (int)((p.panel_off + panel + 15) & ~(size_t)15);
  const size_t lds = p.red_off + 2 * NW * ROWS * sizeof(float);
  if (lds > 160 * 1024) return VCT_E_SHAPE;
  static int attr = 0;
  if (lds > 64 * 1024 && (int)lds > attr) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_block_fwd_kernel<DT, RT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = (int)lds;
  }
  vct::launch(attn_block_fwd_kernel<DT, RT, NW>, dim3(p.a.B), dim3(64 * NW), lds, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

}  // namespace vct
using namespace vct;

// shape support of the fused block: bf16, one wave per head (H = 4 or 8), head_dim 16 / 32 (H = 4) or 64 (H = 8),
// d = H * hd a multiple of 64, Lq <= 32 (two query tiles), Lk <= 64 while the LDS budget holds
extern "C" int vct_attn_block_supported(int dtype, int H, int hd, int Lq, int Lk) {
  if (dtype != VCT_BF16 || Lq < 1 || Lk < 1 || Lq > 32 || Lk > 64) return 0;
  const bool combo = (H == 4 && (hd == 16 || hd == 32)) || (H == 8 && hd == 64);
  if (!combo) return 0;
  const int DT = hd / 16, RT = (Lq + 15) / 16, D = H * hd;
  const int LKT = (Lk + 15) / 16, RK = ((LKT + 1) / 2) * 32, HDK = ((DT * 16 + 31) / 32) * 32;
  const size_t att = (size_t)(RT * 16 + 2 * RK) * (HDK + 8) * 2, stg = (size_t)RT * 16 * (hd + 4) * 4;
  const size_t wave_bytes = ((att > stg ? att : stg) + 15) & ~(size_t)15;
  const size_t lds = wave_bytes * H + (size_t)RT * 16 * (D + 8) * 2 + 16 + 2 * (size_t)H * RT * 16 * 4;
  return lds <= 160 * 1024 ? 1 : 0;
}

extern "C" int vct_attn_block_fwd(const vct_attn_block_desc* d, void* stream) {
  if (d == nullptr) return VCT_E_ARG;
  const vct_attn_desc& a = d->attn;
  if (!a.q || !a.k || !a.v || !a.o || !d->w_out || !d->b_out || !d->res || !d->gamma || !d->beta || !d->a_out || !d->y ||
      !d->mean || !d->rstd)
    return VCT_E_ARG;
  if (!vct_attn_block_supported(a.dtype, a.H, a.hd, a.Lq, a.Lk) || a.B <= 0) return VCT_E_SHAPE;
  if (a.q_bs || a.k_bs || a.v_bs || a.o_bs) return VCT_E_ARG;
  const int D = a.H * a.hd;
  if ((a.ldq % 8) || (a.ldk % 8) || (a.ldv % 8) || (a.ldo % 4) || (d->ldw % 8) || (d->ld_res % 8) || (d->ld_a % 8) || (d->ld_y % 8))
    return VCT_E_ALIGN;
  if (((uintptr_t)a.o & 7) || (((uintptr_t)d->w_out | (uintptr_t)d->res | (uintptr_t)d->a_out | (uintptr_t)d->y |
                                (uintptr_t)d->b_out | (uintptr_t)d->gamma | (uintptr_t)d->beta) & 15))
    return VCT_E_ALIGN;
  if (a.key_pad_shift < 0 || (a.key_pad != nullptr && a.key_pad_shift >= a.Lk)) return VCT_E_SHAPE;
  AttnBlockP p;
  p.a.B = a.B; p.a.H = a.H; p.a.Lq = a.Lq; p.a.Lk = a.Lk; p.a.hd = a.hd; p.a.causal = a.causal;
  p.a.q = a.q; p.a.ldq = a.ldq; p.a.k = a.k; p.a.ldk = a.ldk; p.a.v = a.v; p.a.ldv = a.ldv;
  p.a.o = a.o; p.a.ldo = a.ldo;
  p.a.key_pad = a.key_pad; p.a.key_pad_shift = a.key_pad_shift;
  p.a.key_ids = a.key_ids; p.a.key_ids_bs = a.key_ids_bs; p.a.pad_id = a.pad_id;
  p.a.seed = a.seed; p.a.site = a.site; p.a.p_drop = a.p_drop;
  p.a.d_o = nullptr; p.a.ld_do = 0; p.a.dq = p.a.dk = p.a.dv = nullptr; p.a.ld_dq = p.a.ld_dk = p.a.ld_dv = 0;
  p.a.q_bs = (long)a.Lq * a.ldq; p.a.k_bs = (long)a.Lk * a.ldk; p.a.v_bs = (long)a.Lk * a.ldv; p.a.o_bs = (long)a.Lq * a.ldo;
  p.wo = reinterpret_cast<const bf16_t*>(d->w_out); p.ldw = d->ldw;
  p.bo = d->b_out;
  p.res = reinterpret_cast<const bf16_t*>(d->res); p.ld_res = d->ld_res;
  p.gamma = d->gamma; p.beta = d->beta;
  p.aout = reinterpret_cast<bf16_t*>(d->a_out); p.ld_a = d->ld_a;
  p.y = reinterpret_cast<bf16_t*>(d->y); p.ld_y = d->ld_y;
  p.mean = d->mean; p.rstd = d->rstd;
  p.site2 = d->site_res;
  p.M = a.B * a.Lq;
  p.flags = d->reserved;
  (void)D;
  hipStream_t st = (hipStream_t)stream;
  const int RT = (a.Lq + 15) / 16;
  if (a.H == 8 && a.hd == 64) return RT == 1 ? attn_block_launch<4, 1, 8>(p, st) : attn_block_launch<4, 2, 8>(p, st);
  if (a.H == 4 && a.hd == 32) return RT == 1 ? attn_block_launch<2, 1, 4>(p, st) : attn_block_launch<2, 2, 4>(p, st);
  if (a.H == 4 && a.hd == 16) return RT == 1 ? attn_block_launch<1, 1, 4>(p, st) : attn_block_launch<1, 2, 4>(p, st);
  return VCT_E_SHAPE;
}
