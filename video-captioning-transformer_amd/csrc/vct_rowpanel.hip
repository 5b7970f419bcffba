// Row-panel Linear backward for short-sequence Transformer layers on gfx950 (bf16 throughput mode): the input-gradient products of the
// layers' dX chain, one launch per product, with what follows the product row by row fused behind it.
//
//     dX[M, 512 * nblk] = epilogue(dY[M, K] W)        W = the nn.Linear weight [out = K, in = 512 * nblk] (torch F.linear backward)
//
// The tiled GEMM (vct_gemm_bf16_kernel.h) runs these M = 3328 / 4864-row, N = 512-column products as 300-600 small workgroups that
// walk 8-32 dependent K stages, every stage a first-touch fetch behind the L2 invalidate of the kernel boundary: 12-36 us per product
// for 1-4 us of MFMA work, and the LayerNorm backward behind it is another 10 us launch.  Here a workgroup owns a PANEL of 32
// consecutive rows (flat rows: nothing in these products is sample-local) and the WHOLE 512-column row block: the panel of dY sits
// in LDS, the weight comes as the stream-order packed TRANSPOSED blocks of vct_ss_pack (the layout the sample-stationary kernels
// read: 1 KiB contiguous per wave instruction, two K chunks ahead, straight into MFMA fragments) at the ~50 B/clk a CU pulls from
// L2 -- 4-5 us per 512 x 512 block -- and 152 / 104 workgroups leave a third of the chip to the weight-gradient GEMMs that run beside
// the chain.  Because a workgroup holds complete rows, the row-wise tail of the product is an epilogue:
//     epi 0  dX = acc (+ addend)                                          (out_proj / cross-attention K|V input gradients)
//     epi 1  dX = acc * act'(hpre) * dropout mask, per 512-column block   (linear2: the gradient of the feed-forward pre-activation)
//     epi 2  LayerNorm backward of the norm in FRONT of the Linear (post-norm layers: the product's result is the gradient of that
//            norm's output): gy = acc + addend; ds, dropout-masked ds, (dgamma | dbeta) partial rows -- what vct_add_ln_bwd computes
// replaces: the autograd nodes of nn.Linear / F.gelu / nn.Dropout / nn.LayerNorm between the attention cores of
// nn.TransformerEncoderLayer / nn.TransformerDecoderLayer (torch nn/modules/transformer.py:951-982,1143-1199; built at
// MMEncoder.py:236-238, CapDecoder.py:18-20) in `loss.backward()` (train.py:125) -- vct_gemm (NN form) + vct_add_ln_bwd launches of the
// unfused schedule.  Dropout masks are regenerated from the forward's counter streams.
#include "vct_layer_ss_core.h"

namespace vct {

constexpr int RP_MT = 2;                 // 16-row MFMA tiles per panel
constexpr int RP_ROWS = RP_MT * 16;
constexpr int RP_RED_BYTES = 2 * SS_NW * 32 * 4;
#ifndef RP_NB
#define RP_NB 4
#endif
constexpr int RP_NBUF = RP_NB;           // K chunks of the weight stream in flight per wave (8 KiB each): 2 was latency-bound (93 GB/s per CU)

struct RpRing { bf16x8 b[RP_NBUF][SS_TPW][2]; };
__device__ __forceinline__ void rp_ring_fill(WStream& ws, RpRing& r) {
#pragma unroll
  for (int i = 0; i < RP_NBUF; i++) ws_fetch(ws, r.b[i]);
}
// acc += W-chunk fragments x A fragments over nch chunks (a multiple of RP_NBUF); on exit the ring holds the next RP_NBUF chunks of the stream
template <int MT>
__device__ __forceinline__ void rp_wave_gemm(f32x4 (&acc)[MT][SS_TPW], const bf16_t* a, const int astr, const int nch, WStream& ws, RpRing& r) {
  for (int c = 0; c < nch; c += RP_NBUF) {
#pragma unroll
    for (int i = 0; i < RP_NBUF; i++) {
      gemm_step<MT>(acc, a, astr, c + i, r.b[i]);
      ws_fetch(ws, r.b[i]);
    }
  }
}

struct RpNorm {                          // epi 2
  const float* g; const float* mean; const float* rstd; float* ws;     // ws: [panels][2][512] partial (dgamma | dbeta) rows
  const bf16_t* xs; const bf16_t* res;   // z = res + dropout(xs) was the norm's input (xs: the sublayer output, res: the residual or NULL)
  bf16_t* ds; bf16_t* dxo;               // gradient of z; its dropout-masked copy (NULL: same values, not stored)
  uint32_t site;
};

struct RpP {
  int M, K, nblk, act;
  const bf16_t* A; long lda;
  const bf16_t* wpk;
  bf16_t* out; long ldo;                 // epi 0 / 1
  const bf16_t* addend; long ld_add;     // epi 0 / 2 (NULL: none)
  const bf16_t* hpre; long ld_h; uint32_t site_ff;     // epi 1
  RpNorm nb;
  const uint32_t* seed; float p_drop;
  int out_off;                           // byte offset of the output panels in LDS (0: they alias the A panel, which is dead by then)
};

// rows [0, L) x 512 columns of an LDS panel -> global rows row0.. (16 bytes per thread and step)
__device__ __forceinline__ void rp_panel_out(const bf16_t* panel, const int L, bf16_t* g, const long ld, const long row0, const int col0, const int tid) {
  panel_to_global<SS_D>(panel, SS_PSTR, L, g, ld, row0, col0, tid);
}

template <int EPI>
__global__ __launch_bounds__(SS_NT, SS_NW / 4) void rp_linear_kernel(const RpP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long row0 = (long)blockIdx.x * RP_ROWS;
  const int L = min(RP_ROWS, p.M - (int)row0);
  const int K = p.K, astr = K + 8, nch = K >> 6;
  bf16_t* AP = reinterpret_cast<bf16_t*>(smem);
  bf16_t* O0 = reinterpret_cast<bf16_t*>(smem + p.out_off);
  bf16_t* O1 = O0 + SS_SLOT / 2;
  float* red = reinterpret_cast<float*>(smem + max(p.out_off + 2 * SS_SLOT, RP_ROWS * astr * 2));

  // the weight stream starts before the first activation byte is here
  WStream ws;
  ws.p = p.wpk + (long)wave * SS_WSTR + lane * 8;
  ws.last = ws.p + (long)(p.nblk * nch - 1) * SS_CHUNK;
  RpRing ring;
  rp_ring_fill(ws, ring);

  // dY panel -> LDS (rows >= L zero)
  {
    const int vpr = K >> 3;
    for (int v = tid; v < RP_ROWS * vpr; v += SS_NT) {
      const int r = v / vpr, c = (v - r * vpr) * 8;
      BV8s val;
      if (r < L) val = *reinterpret_cast<const BV8s*>(p.A + (row0 + r) * p.lda + c);
      else {
#pragma unroll
        for (int j = 0; j < 8; j++) val.e[j] = 0;
      }
      *reinterpret_cast<BV8s*>(AP + r * astr + c) = val;
    }
  }
  const int ecol = wave * SS_CPW + lg * 4;
  // row-wise operands of the epilogue, in the epilogue's register layout, issued ahead of the product (they land behind the prefetch)
  BV4 adv[RP_MT][SS_TPW], xsv[RP_MT][SS_TPW], rsv[RP_MT][SS_TPW];
  const bool has_add = (EPI != 1) && p.addend != nullptr;
  const bool has_res = (EPI == 2) && p.nb.res != nullptr;
#pragma unroll
  for (int m = 0; m < RP_MT; m++)
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) {
      const long grow = row0 + min(m * 16 + li, L - 1);
      const int col = ecol + t * 16;
      BV4 z; z.e[0] = z.e[1] = z.e[2] = z.e[3] = 0;
      adv[m][t] = has_add ? *reinterpret_cast<const BV4*>(p.addend + grow * p.ld_add + col) : z;
      if constexpr (EPI == 2) {
        xsv[m][t] = *reinterpret_cast<const BV4*>(p.nb.xs + grow * SS_D + col);
        rsv[m][t] = has_res ? *reinterpret_cast<const BV4*>(p.nb.res + grow * SS_D + col) : z;
      }
    }
  ss_barrier();

  const bf16_t* a_lane = AP + li * astr + lg * 8;
  f32x4 acc[RP_MT][SS_TPW];
  if constexpr (EPI == 1) {
    const Dropout drf = make_dropout(p.seed, p.site_ff, p.p_drop);
    const int N = p.nblk * SS_D;
    for (int j = 0; j < p.nblk; j++) {
      BV4 hp[RP_MT][SS_TPW];                                   // the saved pre-activation of this block: issued ahead of the product
#pragma unroll
      for (int m = 0; m < RP_MT; m++)
#pragma unroll
        for (int t = 0; t < SS_TPW; t++)
          hp[m][t] = *reinterpret_cast<const BV4*>(p.hpre + (row0 + min(m * 16 + li, L - 1)) * p.ld_h + j * SS_D + ecol + t * 16);
      acc_zero<RP_MT>(acc);
      rp_wave_gemm<RP_MT>(acc, a_lane, astr, nch, ws, ring);
      bf16_t* OP = (j & 1) ? O1 : O0;
#pragma unroll
      for (int m = 0; m < RP_MT; m++)
#pragma unroll
        for (int t = 0; t < SS_TPW; t++) {
          const int row = m * 16 + li, col = j * SS_D + ecol + t * 16;
          float dm[4];
          drop_mults<4>(drf, (uint32_t)(row0 + row) * (uint32_t)N + (uint32_t)col, dm);
          const vf2 x0 = {bf2f(hp[m][t].e[0]), bf2f(hp[m][t].e[1])}, x1 = {bf2f(hp[m][t].e[2]), bf2f(hp[m][t].e[3])};
          const vf2 d0 = dact_fast_f2(p.act, x0) * vf2{dm[0], dm[1]}, d1 = dact_fast_f2(p.act, x1) * vf2{dm[2], dm[3]};
          BV4 o;
          o.e[0] = f2bf(acc[m][t][0] * d0[0]); o.e[1] = f2bf(acc[m][t][1] * d0[1]);
          o.e[2] = f2bf(acc[m][t][2] * d1[0]); o.e[3] = f2bf(acc[m][t][3] * d1[1]);
          *reinterpret_cast<BV4*>(OP + row * SS_PSTR + ecol + t * 16) = o;
        }
      ss_barrier();                                            // the block is complete (and panel (j - 1) & 1 was copied out a barrier ago)
      rp_panel_out(OP, L, p.out, p.ldo, row0, j * SS_D, tid);
    }
    return;
  }

  acc_zero<RP_MT>(acc);
  rp_wave_gemm<RP_MT>(acc, a_lane, astr, nch, ws, ring);
#pragma unroll
  for (int m = 0; m < RP_MT; m++)
#pragma unroll
    for (int t = 0; t < SS_TPW; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[m][t][r] += bf2f(adv[m][t].e[r]);
  if (p.out_off == 0) ss_barrier();                            // every wave is done with the dY panel: the output panels take its place

  if constexpr (EPI == 0) {
    float4 zero[SS_TPW];
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) zero[t] = float4{0.0f, 0.0f, 0.0f, 0.0f};
    epi_store<RP_MT>(acc, zero, O0, SS_PSTR, wave * SS_CPW, li, lg);
    ss_barrier();
    rp_panel_out(O0, L, p.out, p.ldo, row0, 0, tid);
    return;
  }

  if constexpr (EPI == 2) {
    // LayerNorm backward on the epilogue register layout (vct_add_ln_bwd): z = res + drop(xs), h = (z - mean) rstd,
    //   ds = rstd (gy g - mean_c(gy g) - h mean_c(gy g h));  the per-panel column partials of gy h / gy go to ws
    const RpNorm& n = p.nb;
    const Dropout dr = make_dropout(p.seed, n.site, p.p_drop);
    float4 gm[SS_TPW];
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) gm[t] = *reinterpret_cast<const float4*>(n.g + ecol + t * 16);
    f32x4 hh[RP_MT][SS_TPW], pg[SS_TPW], pb[SS_TPW];
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) { pg[t] = f32x4{0, 0, 0, 0}; pb[t] = f32x4{0, 0, 0, 0}; }
    uint32_t keep = 0u;
    float rstd[RP_MT];
    float* red0 = red;
    float* red1 = red + SS_NW * 32;
#pragma unroll
    for (int m = 0; m < RP_MT; m++) {
      const int row = m * 16 + li;
      const bool valid = row < L;
      const float mean = valid ? n.mean[row0 + row] : 0.0f;
      rstd[m] = valid ? n.rstd[row0 + row] : 0.0f;
      float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
      for (int t = 0; t < SS_TPW; t++) {
        float dm[4];
        drop_mults<4>(dr, (uint32_t)(row0 + row) * (uint32_t)SS_D + (uint32_t)(ecol + t * 16), dm);
        const float gg[4] = {gm[t].x, gm[t].y, gm[t].z, gm[t].w};
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (dm[r] != 0.0f) keep |= 1u << ((m * SS_TPW + t) * 4 + r);
          const float s = bf2f(xsv[m][t].e[r]) * dm[r] + bf2f(rsv[m][t].e[r]);
          const float h = (s - mean) * rstd[m];
          const float dyv = valid ? acc[m][t][r] : 0.0f;
          const float dh = dyv * gg[r];
          acc[m][t][r] = dh;
          hh[m][t][r] = h;
          c1 += dh; c2 += dh * h;
          pg[t][r] += dyv * h; pb[t][r] += dyv;
        }
      }
      c1 = red4_sum(c1); c2 = red4_sum(c2);
      if (lg == 0) { red0[wave * 32 + row] = c1; red1[wave * 32 + row] = c2; }
    }
    ss_barrier();
    const bool masked = n.dxo != nullptr;
#pragma unroll
    for (int m = 0; m < RP_MT; m++) {
      const int row = m * 16 + li;
      float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
      for (int w = 0; w < SS_NW; w++) { c1 += red0[w * 32 + row]; c2 += red1[w * 32 + row]; }
      c1 *= 1.0f / (float)SS_D; c2 *= 1.0f / (float)SS_D;
#pragma unroll
      for (int t = 0; t < SS_TPW; t++) {
        BV4 o, om;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float v = rstd[m] * (acc[m][t][r] - c1 - hh[m][t][r] * c2);
          o.e[r] = f2bf(v);
          om.e[r] = f2bf(((keep >> ((m * SS_TPW + t) * 4 + r)) & 1u) ? v * dr.scale : 0.0f);
        }
        *reinterpret_cast<BV4*>(O0 + row * SS_PSTR + ecol + t * 16) = o;
        if (masked) *reinterpret_cast<BV4*>(O1 + row * SS_PSTR + ecol + t * 16) = om;
      }
    }
    // column partials of this panel: over the row tiles (done above), then over the 16 lanes that share a column group
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) {
#pragma unroll
      for (int r = 0; r < 4; r++) { pg[t][r] = red16_sum(pg[t][r]); pb[t][r] = red16_sum(pb[t][r]); }
      if (li == 0) {
        *reinterpret_cast<f32x4*>(n.ws + ((long)blockIdx.x * 2 + 0) * SS_D + ecol + t * 16) = pg[t];
        *reinterpret_cast<f32x4*>(n.ws + ((long)blockIdx.x * 2 + 1) * SS_D + ecol + t * 16) = pb[t];
      }
    }
    ss_barrier();
    rp_panel_out(O0, L, n.ds, SS_D, row0, 0, tid);
    if (masked) rp_panel_out(O1, L, n.dxo, SS_D, row0, 0, tid);
  }
}

}  // namespace vct
using namespace vct;

extern "C" int vct_rp_linear_supported(int dtype, int N, int K, int epi) {
  if (dtype != VCT_BF16 || epi < 0 || epi > 2) return 0;
  if (N < SS_D || (N % SS_D) || K < 64 * RP_NBUF || (K % (64 * RP_NBUF))) return 0;
  if (epi != 1 && N != SS_D) return 0;
  const long a_bytes = (long)RP_ROWS * (K + 8) * 2;
  const long lds = (N > SS_D ? a_bytes + 2 * SS_SLOT : (a_bytes > 2 * SS_SLOT ? a_bytes : 2 * SS_SLOT)) + RP_RED_BYTES;
  return lds <= 160 * 1024;
}

extern "C" int vct_rp_linear(const vct_rp_linear_desc* d, void* stream) {
  if (d == nullptr || d->A == nullptr || d->wpk == nullptr) return VCT_E_ARG;
  const int N = d->N;
  if (d->M < 1 || !vct_rp_linear_supported(d->dtype, N, d->K, d->epi)) return VCT_E_SHAPE;
  if ((d->lda % 8) || (((uintptr_t)d->A | (uintptr_t)d->wpk) & 15)) return VCT_E_ALIGN;
  RpP p;
  memset(&p, 0, sizeof(p));
  p.M = d->M; p.K = d->K; p.nblk = N / SS_D; p.act = d->act;
  p.A = reinterpret_cast<const bf16_t*>(d->A); p.lda = (long)d->lda;
  p.wpk = reinterpret_cast<const bf16_t*>(d->wpk);
  p.out = reinterpret_cast<bf16_t*>(d->out); p.ldo = (long)d->ldo;
  p.addend = reinterpret_cast<const bf16_t*>(d->addend); p.ld_add = (long)d->ld_addend;
  p.hpre = reinterpret_cast<const bf16_t*>(d->hpre); p.ld_h = (long)d->ld_hpre; p.site_ff = d->site;
  p.seed = d->seed; p.p_drop = d->p_drop;
  if (d->epi != 2) {
    if (d->out == nullptr || (d->ldo % 8) || ((uintptr_t)d->out & 15)) return d->out ? VCT_E_ALIGN : VCT_E_ARG;
  }
  if (d->epi != 1 && d->addend != nullptr && ((d->ld_addend % 4) || ((uintptr_t)d->addend & 7))) return VCT_E_ALIGN;
  if (d->epi == 1) {
    if (d->hpre == nullptr) return VCT_E_ARG;
    if ((d->ld_hpre % 4) || ((uintptr_t)d->hpre & 7)) return VCT_E_ALIGN;
  }
  if (d->epi == 2) {
    const vct_rp_norm_bwd& n = d->norm;
    if (!n.gamma || !n.mean || !n.rstd || !n.ws || !n.xs || !n.ds) return VCT_E_ARG;
    if (((uintptr_t)n.gamma | (uintptr_t)n.ws | (uintptr_t)n.xs | (uintptr_t)n.res | (uintptr_t)n.ds | (uintptr_t)n.dxo) & 15) return VCT_E_ALIGN;
    p.nb.g = n.gamma; p.nb.mean = n.mean; p.nb.rstd = n.rstd; p.nb.ws = n.ws;
    p.nb.xs = reinterpret_cast<const bf16_t*>(n.xs); p.nb.res = reinterpret_cast<const bf16_t*>(n.res);
    p.nb.ds = reinterpret_cast<bf16_t*>(n.ds); p.nb.dxo = reinterpret_cast<bf16_t*>(n.dxo);
    p.nb.site = n.site;
  }
  const long a_bytes = (long)RP_ROWS * (d->K + 8) * 2;
  p.out_off = N > SS_D ? (int)((a_bytes + 15) & ~15L) : 0;
  const long body = p.out_off + 2 * SS_SLOT > a_bytes ? p.out_off + 2 * SS_SLOT : a_bytes;
  const int lds = (int)(((body + 15) & ~15L) + RP_RED_BYTES);
  const int panels = (d->M + RP_ROWS - 1) / RP_ROWS;
  hipStream_t st = (hipStream_t)stream;
  static vct::DynLdsOptIn optin[3];
  const void* fn = d->epi == 0 ? (const void*)rp_linear_kernel<0> : d->epi == 1 ? (const void*)rp_linear_kernel<1> : (const void*)rp_linear_kernel<2>;
  if (hipError_t e = optin[d->epi].ensure(fn, 160 * 1024); e != hipSuccess) return (int)e;
  if (d->epi == 0) vct::launch(rp_linear_kernel<0>, dim3(panels), dim3(SS_NT), (size_t)lds, st, p);
  else if (d->epi == 1) vct::launch(rp_linear_kernel<1>, dim3(panels), dim3(SS_NT), (size_t)lds, st, p);
  else vct::launch(rp_linear_kernel<2>, dim3(panels), dim3(SS_NT), (size_t)lds, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}
