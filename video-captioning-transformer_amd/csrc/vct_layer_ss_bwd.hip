// Sample-stationary Transformer layer BACKWARD (the dX chain) for short sequences on gfx950 (bf16 throughput mode).
//
//     ONE launch = the activation-gradient chain of a whole nn.TransformerEncoder stack, top layer first (+ the stack-final LayerNorm)
//     ONE 512-thread workgroup = ONE sample: the gradient rows stay in registers / LDS from the stack output to the stack input
//
// replaces, per encoder layer, the 7 launches of the unfused dX chain (engine.py: _ln_bwd / _ffn_bwd / _attn_block_bwd): LayerNorm
// backward, the two feed-forward input-gradient GEMMs (GELU' and dropout in the first one's epilogue), LayerNorm backward, the
// out_proj input-gradient GEMM, the attention backward core and the in_proj input-gradient GEMM -- the autograd nodes of torch
// nn/modules/transformer.py:951-982 as built at MMEncoder.py:236-238.  The WEIGHT gradients are not sample-local (their K dimension is
// the rows of the whole batch): this kernel stores the four output-side gradients they need (d f, d hpre, d a, d qkv: the buffers
// the unfused chain writes) and the grouped weight-gradient GEMM of the layer runs behind it unchanged; LayerNorm parameter gradients
// leave as one partial row per sample in the layout vct_ln_param_finalize_batched sums.
//
// Same skeleton as the forward (vct_layer_ss_core.h): a wave owns 64 columns of every 512-column block, operands swapped so that a lane
// holds 4 consecutive columns of a row, weights from a stream-order packed shadow two K chunks ahead -- here the TRANSPOSED blocks
// (dX = dY W: the product's "weight rows" are W's columns), packed by vct_ss_pack with `transposed` segments, in the order
//   [W2^T block j | W1^T K-slice j] x ff/512 | Wo^T | Win^T (3 K blocks)        per layer, top layer first.
// Dropout masks are regenerated from the forward's counter streams; the softmax is recomputed from q, k (saved by the forward).
#include "vct_layer_ss_core.h"

namespace vct {

static_assert(SS_NW == SS_H, "one wave per attention head");

struct SsBwdNorm { const float* g; const float* mean; const float* rstd; float* ws; };   // ws: [B][2][512] partial (dgamma | dbeta) rows

struct SsBwdW {
  const bf16_t* x; const bf16_t* qkv; const bf16_t* a; const bf16_t* x1; const bf16_t* hpre; const bf16_t* f;   // saved by the forward
  SsBwdNorm n1, n3;                  // norm1 (behind the attention block), the norm behind the feed-forward block
  bf16_t* df; bf16_t* dhpre; bf16_t* da; bf16_t* dqkv;                                                          // for the weight-gradient GEMMs
  uint32_t site_sa, site_n1, site_ff, site_n3;
};

struct SsBwdP {
  int B, L, ff, act, last, causal, nl;
  const bf16_t* wpk; int nchunks;    // packed TRANSPOSED weight stream of the nl layers in processing order
  const bf16_t* dy;                  // [B*L, 512] gradient of the stack output
  bf16_t* dx;                        // [B*L, 512] gradient of the stack input
  const bf16_t* y_last; SsBwdNorm nf;   // stack-final norm (last != 0): its input rows = the top layer's output
  const uint8_t* key_pad; int key_pad_shift;
  const int64_t* key_ids; long key_ids_bs; long pad_id;
  const uint32_t* seed; float p_drop;
  SsBwdW lw[SS_MAXL];
};

// LayerNorm backward on the epilogue register layout (what vct_add_ln_bwd computes): z = res + drop(xs), h = (z - mean) rstd,
//   ds = rstd (dy g - mean_c(dy g) - h mean_c(dy g h)) -> gy (fp32, in place);  bf16(ds * dropmask) -> panel DXO;
//   per-sample column partials of dy h / dy -> n.ws.  xs / res come from LDS panels (rows >= L zero), rows >= L give ds = 0.
template <int MT, bool RES>
__device__ __forceinline__ void ss_ln_bwd(f32x4 (&gy)[MT][SS_TPW], const bf16_t* XS, const bf16_t* RS, const SsBwdNorm& n, const Dropout& dr,
                                          const long grow0, const int b, const int L, bf16_t* DXO, float* red, const int wave,
                                          const int li_in, const int lg_in) {
  int li = li_in, lg = lg_in;
  asm volatile("" : "+v"(li), "+v"(lg));
  const int colw = wave * SS_CPW + lg * 4;
  float4 gm[SS_TPW];
#pragma unroll
  for (int t = 0; t < SS_TPW; t++) gm[t] = *reinterpret_cast<const float4*>(n.g + colw + t * 16);
  f32x4 hh[MT][SS_TPW], pg[SS_TPW], pb[SS_TPW];
#pragma unroll
  for (int t = 0; t < SS_TPW; t++) { pg[t] = f32x4{0, 0, 0, 0}; pb[t] = f32x4{0, 0, 0, 0}; }
  uint32_t keep = 0u;
  float rstd[MT];
  float* red0 = red;
  float* red1 = red + SS_NW * 32;
#pragma unroll
  for (int m = 0; m < MT; m++) {
    const int row = m * 16 + li;
    const bool valid = row < L;
    const float mean = valid ? n.mean[grow0 + row] : 0.0f;
    rstd[m] = valid ? n.rstd[grow0 + row] : 0.0f;
    float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) {
      const BV4 xs = *reinterpret_cast<const BV4*>(XS + row * SS_PSTR + colw + t * 16);
      BV4 rs;
      if constexpr (RES) rs = *reinterpret_cast<const BV4*>(RS + row * SS_PSTR + colw + t * 16);
      float dm[4];
      drop_mults<4>(dr, (uint32_t)(grow0 + row) * (uint32_t)SS_D + (uint32_t)(colw + t * 16), dm);
      const float gg[4] = {gm[t].x, gm[t].y, gm[t].z, gm[t].w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (dm[r] != 0.0f) keep |= 1u << ((m * SS_TPW + t) * 4 + r);
        float s = bf2f(xs.e[r]) * dm[r];
        if constexpr (RES) s += bf2f(rs.e[r]);
        const float h = (s - mean) * rstd[m];
        const float dyv = valid ? gy[m][t][r] : 0.0f;
        const float dh = dyv * gg[r];
        gy[m][t][r] = dh;
        hh[m][t][r] = h;
        c1 += dh; c2 += dh * h;
        pg[t][r] += dyv * h; pb[t][r] += dyv;
      }
    }
    c1 = red4_sum(c1); c2 = red4_sum(c2);
    if (lg == 0) { red0[wave * 32 + row] = c1; red1[wave * 32 + row] = c2; }
  }
  ss_barrier();
#pragma unroll
  for (int m = 0; m < MT; m++) {
    const int row = m * 16 + li;
    float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int w = 0; w < SS_NW; w++) { c1 += red0[w * 32 + row]; c2 += red1[w * 32 + row]; }
    c1 *= 1.0f / (float)SS_D; c2 *= 1.0f / (float)SS_D;
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) {
      BV4 o;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float v = rstd[m] * (gy[m][t][r] - c1 - hh[m][t][r] * c2);
        gy[m][t][r] = v;
        o.e[r] = f2bf(((keep >> ((m * SS_TPW + t) * 4 + r)) & 1u) ? v * dr.scale : 0.0f);
      }
      *reinterpret_cast<BV4*>(DXO + row * SS_PSTR + colw + t * 16) = o;
    }
  }
  // column partials of this sample: over the row tiles (done above), then over the 16 lanes that share a column group
#pragma unroll
  for (int t = 0; t < SS_TPW; t++) {
#pragma unroll
    for (int r = 0; r < 4; r++) { pg[t][r] = red16_sum(pg[t][r]); pb[t][r] = red16_sum(pb[t][r]); }
    if (li == 0) {
      *reinterpret_cast<f32x4*>(n.ws + ((long)b * 2 + 0) * SS_D + colw + t * 16) = pg[t];
      *reinterpret_cast<f32x4*>(n.ws + ((long)b * 2 + 1) * SS_D + colw + t * 16) = pb[t];
    }
  }
}

// ---- attention backward of ONE head by ONE wave, operands in LDS panels (the arithmetic and dropout stream of attn_bwd_kernel) ---------
// Qp / Kp / Vp point at this head's 64 columns of the q | k | v panel (row stride str), dOp at the head's columns of the d(attention
// output) panel.  dQ / dK / dV overwrite q / k / v IN PLACE (a head's columns belong to one wave; a key tile's k / v rows are last read
// by that tile's own step, q by the last key tile).  stat: 96 floats of LDS private to this wave.
__device__ __forceinline__ void ss_attn_bwd_wave(bf16_t* Qp, bf16_t* Kp, bf16_t* Vp, const int str, const bf16_t* dOp, const int Lq, const int Lk,
                                                 const int causal, const unsigned long long padmask, const Dropout& dr, const int bh, float* stat,
                                                 const int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int LQT = (Lq + 15) >> 4, LKT = (Lk + 15) >> 4;     // <= 2 each
  const float scale = 0.125f;
  f32x4 dq[2][4];
  // phase A: per query tile (transposed scores): softmax statistics, D = rowsum(dP P), dQ
#pragma unroll
  for (int qt = 0; qt < 2; qt++) {
#pragma unroll
    for (int dt = 0; dt < 4; dt++) dq[qt][dt] = f32x4{0, 0, 0, 0};
    if (qt < LQT) {
      f32x4 st[2], dpt[2];
      const int qq = qt * 16 + i;
#pragma unroll
      for (int t = 0; t < 2; t++) {
        st[t] = f32x4{0, 0, 0, 0};
        if (t < LKT) {
#pragma unroll
          for (int ks = 0; ks < 2; ks++)
            st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ss_frag_rowk(Kp, str, t * 16, ks, lane), ss_frag_rowk(Qp, str, qt * 16, ks, lane),
                                                            st[t], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int kk = t * 16 + g * 4 + r;
          const bool masked = kk >= Lk || (causal && kk > qq) || ((padmask >> kk) & 1ull);
          st[t][r] = (t < LKT && !masked) ? st[t][r] * scale : -INFINITY;
        }
      }
      float m = -INFINITY;
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) m = fmaxf(m, st[t][r]);
      m = red4_max(m);
      if (m == -INFINITY) m = 0.0f;
      float l = 0.0f;
#pragma unroll
      for (int t = 0; t < 2; t++) {
        if (t < LKT) {
#pragma unroll
          for (int r = 0; r < 4; r++) { st[t][r] = __expf(st[t][r] - m); l += st[t][r]; }
        } else {
          st[t] = f32x4{0, 0, 0, 0};
        }
      }
      l = red4_sum(l);
      const float inv = l > 0.0f ? 1.0f / l : 0.0f;
      float dsum = 0.0f;
#pragma unroll
      for (int t = 0; t < 2; t++) {
        dpt[t] = f32x4{0, 0, 0, 0};
        if (t < LKT) {
#pragma unroll
          for (int ks = 0; ks < 2; ks++)
            dpt[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ss_frag_rowk(Vp, str, t * 16, ks, lane), ss_frag_rowk(dOp, SS_PSTR, qt * 16, ks, lane),
                                                             dpt[t], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int kk = t * 16 + g * 4 + r;
            st[t][r] *= inv;
            dpt[t][r] *= drop_mult(dr, (uint32_t)((bh * Lq + qq) * Lk + kk));
            dsum += dpt[t][r] * st[t][r];
          }
        }
      }
      dsum = red4_sum(dsum);
      if (g == 0) { stat[qq] = m; stat[32 + qq] = inv; stat[64 + qq] = dsum; }
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) st[t][r] = st[t][r] * (dpt[t][r] - dsum) * scale;
      const bf16x8 pa = pack_p(st[0], st[1]);
      const int r1 = LKT > 1 ? 16 : 0;
#pragma unroll
      for (int dt = 0; dt < 4; dt++)
        dq[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ss_frag_colk(Kp, str, 0, r1, dt * 16, lane), pa, dq[qt][dt], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the statistics (this wave's own LDS words) before phase B reads them
  // phase B: per key tile (un-transposed scores): dV = Pd^T dO, dK = dS^T Q
#pragma unroll
  for (int t = 0; t < 2; t++) {
    if (t < LKT) {
      const int kk = t * 16 + i;
      f32x4 pd[2], dsv[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        f32x4 s = f32x4{0, 0, 0, 0}, dp = f32x4{0, 0, 0, 0};
        if (u < LQT) {
#pragma unroll
          for (int ks = 0; ks < 2; ks++) {
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ss_frag_rowk(Qp, str, u * 16, ks, lane), ss_frag_rowk(Kp, str, t * 16, ks, lane), s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ss_frag_rowk(dOp, SS_PSTR, u * 16, ks, lane), ss_frag_rowk(Vp, str, t * 16, ks, lane), dp, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int qq = u * 16 + g * 4 + r;
          float pdv = 0.0f, dv = 0.0f;
          const bool masked = kk >= Lk || (causal && kk > qq) || ((padmask >> kk) & 1ull);
          if (u < LQT && qq < Lq && !masked) {
            const float mult = drop_mult(dr, (uint32_t)((bh * Lq + qq) * Lk + kk));
            const float pv = __expf(s[r] * scale - stat[qq]) * stat[32 + qq];
            pdv = pv * mult;
            dv = pv * (dp[r] * mult - stat[64 + qq]) * scale;
          }
          pd[u][r] = pdv;
          dsv[u][r] = dv;
        }
      }
      const bf16x8 pa = pack_p(pd[0], pd[1]);
      const bf16x8 da = pack_p(dsv[0], dsv[1]);
      const int r1 = LQT > 1 ? 16 : 0;
      f32x4 av[4], ak[4];
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
        av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ss_frag_colk(dOp, SS_PSTR, 0, r1, dt * 16, lane), pa, f32x4{0, 0, 0, 0}, 0, 0, 0);
        ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ss_frag_colk(Qp, str, 0, r1, dt * 16, lane), da, f32x4{0, 0, 0, 0}, 0, 0, 0);
      }
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {                       // dV^T / dK^T tiles over this key tile's v / k rows
        BV4 ov, ok;
#pragma unroll
        for (int r = 0; r < 4; r++) { ov.e[r] = f2bf(av[dt][r]); ok.e[r] = f2bf(ak[dt][r]); }
        *reinterpret_cast<BV4*>(Vp + (t * 16 + i) * str + dt * 16 + g * 4) = ov;
        *reinterpret_cast<BV4*>(Kp + (t * 16 + i) * str + dt * 16 + g * 4) = ok;
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; qt++)
    if (qt < LQT) {
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
        BV4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) o.e[r] = f2bf(dq[qt][dt][r]);
        *reinterpret_cast<BV4*>(Qp + (qt * 16 + i) * str + dt * 16 + g * 4) = o;
      }
    }
}

// global [rows][NCOLS] -> panel with row stride pstr (rows >= L zero-filled up to `alloc`)
template <int NCOLS>
__device__ __forceinline__ void global_to_panel_w(bf16_t* panel, const int pstr, const int L, const int alloc, const bf16_t* g, const long row0,
                                                  const int tid) {
  constexpr int vpr = NCOLS / 8;
  for (int v = tid; v < alloc * vpr; v += SS_NT) {
    const int r = v / vpr, c = (v - r * vpr) * 8;
    BV8s val;
    if (r < L) val = *reinterpret_cast<const BV8s*>(g + (row0 + r) * NCOLS + c);
    else {
#pragma unroll
      for (int j = 0; j < 8; j++) val.e[j] = 0;
    }
    *reinterpret_cast<BV8s*>(panel + r * pstr + c) = val;
  }
}

template <int MT>
__global__ __launch_bounds__(SS_NT, SS_NW / 4) void layer_ss_bwd_kernel(const SsBwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int b = blockIdx.x, L = p.L;
  const long grow0 = (long)b * L;
  bf16_t* S0 = reinterpret_cast<bf16_t*>(smem + SS_R0);
  bf16_t* S1 = reinterpret_cast<bf16_t*>(smem + SS_R1A);
  bf16_t* S2 = reinterpret_cast<bf16_t*>(smem + SS_R1B);
  bf16_t* S3 = reinterpret_cast<bf16_t*>(smem + SS_R1C);
  float* red = reinterpret_cast<float*>(smem + SS_RED);
  float* stat = reinterpret_cast<float*>(smem + SS_B1) + wave * 96;       // attention statistics, private to the wave

  WStream ws;
  ws.p = p.wpk + (long)wave * SS_WSTR + lane0 * 8;
  ws.last = ws.p + (long)(p.nchunks - 1) * SS_CHUNK;
  bf16x8 b0[SS_TPW][2], b1[SS_TPW][2];
  ws_fetch(ws, b0);
  ws_fetch(ws, b1);
  const unsigned long long padmask = ss_padmask(p, b, lane0);
  const int head = wave;                                     // SS_NW == SS_H: one wave per head

  // gradient of the stack output -> registers (epilogue layout), through the stack-final norm
  f32x4 gy[MT][SS_TPW];
  {
    const int li = lane0 & 15, lg = lane0 >> 4, ecol = wave * SS_CPW + lg * 4;
    global_to_panel(S0, SS_PSTR, L, MT * 16, p.dy, SS_D, grow0, tid0);
    if (p.last) global_to_panel(S1, SS_PSTR, L, MT * 16, p.y_last, SS_D, grow0, tid0);
    ss_barrier();
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int t = 0; t < SS_TPW; t++) {
        const BV4 v = *reinterpret_cast<const BV4*>(S0 + (m * 16 + li) * SS_PSTR + ecol + t * 16);
        gy[m][t] = f32x4{bf2f(v.e[0]), bf2f(v.e[1]), bf2f(v.e[2]), bf2f(v.e[3])};
      }
    if (p.last) {
      const Dropout none = make_dropout(nullptr, 0u, 0.0f);
      const SsBwdNorm nfl = p.nf;
      ss_ln_bwd<MT, false>(gy, S1, nullptr, nfl, none, grow0, b, L, S2, red, wave, li, lg);      // (the masked copy in S2 is not used)
    }
    ss_barrier();
  }

  typedef const __attribute__((address_space(4))) SsBwdP* KargP;
  const KargP kp = (KargP)__builtin_amdgcn_kernarg_segment_ptr();
  for (int l = 0; l < p.nl; l++) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int aoff = li * SS_PSTR + lg * 8;
    const int ecol = wave * SS_CPW + lg * 4;
    const __attribute__((address_space(4))) SsBwdW& w = kp->lw[l];
    auto nrm = [](const __attribute__((address_space(4))) SsBwdNorm& n) { return SsBwdNorm{n.g, n.mean, n.rstd, n.ws}; };

    // ---- norm behind the feed-forward block: d f (dropout-masked) -> S3, ds3 stays in gy -------------------------------------------------
    global_to_panel(S1, SS_PSTR, L, MT * 16, w.f, SS_D, grow0, tid);
    global_to_panel(S2, SS_PSTR, L, MT * 16, w.x1, SS_D, grow0, tid);
    ss_barrier();
    {
      const Dropout dr = make_dropout(p.seed, w.site_n3, p.p_drop);
      ss_ln_bwd<MT, true>(gy, S1, S2, nrm(w.n3), dr, grow0, b, L, S3, red, wave, li, lg);
    }
    ss_barrier();
    panel_to_global<SS_D>(S3, SS_PSTR, L, w.df, SS_D, grow0, 0, tid);

    // ---- feed-forward block: d hpre(j) = (d f W2)[:, block j] * dropmask * act'(hpre);  d x1 += d hpre(j) W1[block j, :] ------------------
    f32x4 acc[MT][SS_TPW], facc[MT][SS_TPW];
    acc_zero<MT>(facc);
    {
      const Dropout drf = make_dropout(p.seed, w.site_ff, p.p_drop);
      const int nj = p.ff >> 9;
      for (int j = 0; j < nj; j++) {
        BV4 hp[MT][SS_TPW];                                    // the saved pre-activation of this block: issued ahead of the product
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
          for (int t = 0; t < SS_TPW; t++) {
            const int row = min(m * 16 + li, L - 1);
            hp[m][t] = *reinterpret_cast<const BV4*>(w.hpre + (grow0 + row) * p.ff + j * 512 + ecol + t * 16);
          }
        acc_zero<MT>(acc);
        wave_gemm<MT>(acc, S3 + aoff, SS_PSTR, 0, 8, ws, b0, b1);
        bf16_t* HP = (j & 1) ? S2 : S1;
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
          for (int t = 0; t < SS_TPW; t++) {
            const int row = m * 16 + li, col = j * 512 + ecol + t * 16;
            float dm[4];
            drop_mults<4>(drf, (uint32_t)(grow0 + row) * (uint32_t)p.ff + (uint32_t)col, dm);
            const vf2 x0 = {bf2f(hp[m][t].e[0]), bf2f(hp[m][t].e[1])}, x1 = {bf2f(hp[m][t].e[2]), bf2f(hp[m][t].e[3])};
            const vf2 d0 = dact_fast_f2(p.act, x0) * vf2{dm[0], dm[1]}, d1 = dact_fast_f2(p.act, x1) * vf2{dm[2], dm[3]};
            BV4 o;
            o.e[0] = f2bf(acc[m][t][0] * d0[0]); o.e[1] = f2bf(acc[m][t][1] * d0[1]);
            o.e[2] = f2bf(acc[m][t][2] * d1[0]); o.e[3] = f2bf(acc[m][t][3] * d1[1]);
            *reinterpret_cast<BV4*>(HP + row * SS_PSTR + ecol + t * 16) = o;
            if (row < L) *reinterpret_cast<BV4*>(w.dhpre + (grow0 + row) * p.ff + col) = o;
          }
        ss_barrier();
        wave_gemm<MT>(facc, HP + aoff, SS_PSTR, 0, 8, ws, b0, b1);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int t = 0; t < SS_TPW; t++) gy[m][t] += facc[m][t];        // d x1 = ds3 + (d hpre W1)
    ss_barrier();                                              // the last chunk's product has read its panel everywhere (ff = 512: that is S1)

    // ---- norm1: d a (dropout-masked) -> S0 (over the staged x), ds1 stays in gy ----------------------------------------------------------
    global_to_panel(S1, SS_PSTR, L, MT * 16, w.a, SS_D, grow0, tid);  // (S1's last readers: the product of chunk nj - 2, a barrier ago)
    global_to_panel(S0, SS_PSTR, L, MT * 16, w.x, SS_D, grow0, tid);
    ss_barrier();
    {
      const Dropout dr = make_dropout(p.seed, w.site_n1, p.p_drop);
      ss_ln_bwd<MT, true>(gy, S1, S0, nrm(w.n1), dr, grow0, b, L, S0, red, wave, li, lg);     // every lane overwrites the x it read itself
    }
    ss_barrier();                                              // d a complete; nobody reads S1..S3 any more
    panel_to_global<SS_D>(S0, SS_PSTR, L, w.da, SS_D, grow0, 0, tid);
    global_to_panel_w<3 * SS_D>(S1, SS_QSTR, L, MT * 16, w.qkv, grow0, tid);   // q | k | v panel over S1..S3, under the next product
    // ---- d o = d a Wo -> S0 (behind a barrier: the product reads all of d a) ---------------------------------------------------------------
    acc_zero<MT>(acc);
    wave_gemm<MT>(acc, S0 + aoff, SS_PSTR, 0, 8, ws, b0, b1);
    ss_barrier();
    {
      float4 zero[SS_TPW];
#pragma unroll
      for (int t = 0; t < SS_TPW; t++) zero[t] = float4{0.0f, 0.0f, 0.0f, 0.0f};
      epi_store<MT>(acc, zero, S0, SS_PSTR, wave * SS_CPW, li, lg);
    }
    ss_barrier();
    // ---- attention backward: d q | d k | d v over q | k | v in place -------------------------------------------------------------------
    {
      const Dropout dr = make_dropout(p.seed, w.site_sa, p.p_drop);
      const int hd0 = head * SS_HD;
      ss_attn_bwd_wave(S1 + hd0, S1 + SS_D + hd0, S1 + 2 * SS_D + hd0, SS_QSTR, S0 + hd0, L, L, p.causal, padmask, dr, b * SS_H + head, stat, lane);
    }
    ss_barrier();
    panel_to_global<3 * SS_D>(S1, SS_QSTR, L, w.dqkv, 3 * SS_D, grow0, 0, tid);
    // ---- d x = d qkv Win + ds1: the gradient of the layer below's output ---------------------------------------------------------------
    acc_zero<MT>(acc);
    wave_gemm<MT>(acc, S1 + li * SS_QSTR + lg * 8, SS_QSTR, 0, 24, ws, b0, b1);
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int t = 0; t < SS_TPW; t++) gy[m][t] += acc[m][t];
    ss_barrier();                                              // S1..S3 are re-staged by the next layer
  }
  {   // gradient of the stack input
    const int li = lane0 & 15, lg = lane0 >> 4, ecol = wave * SS_CPW + lg * 4;
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int t = 0; t < SS_TPW; t++) {
        const int row = m * 16 + li;
        BV4 o;
        o.e[0] = f2bf(gy[m][t][0]); o.e[1] = f2bf(gy[m][t][1]); o.e[2] = f2bf(gy[m][t][2]); o.e[3] = f2bf(gy[m][t][3]);
        if (row < L) *reinterpret_cast<BV4*>(p.dx + (grow0 + row) * SS_D + ecol + t * 16) = o;
      }
  }
}

}  // namespace vct
using namespace vct;

extern "C" int64_t vct_layer_ss_bwd_stream_chunks(int ff) {
  // per layer: ff/512 x (W2^T block + W1^T K slice) + Wo^T + Win^T (3 K blocks), 8 chunks each
  return (int64_t)8 * (2 * (ff / 512) + 4);
}

extern "C" int vct_layer_ss_bwd(const vct_layer_ss_bwd_desc* layers, int n_layers, void* stream) {
  if (layers == nullptr || n_layers < 1) return VCT_E_ARG;
  const vct_layer_ss_bwd_desc* q = layers;
  if (!vct_layer_ss_supported(q->dtype, q->d, q->H, q->ff, q->L, 0) || q->B < 1) return VCT_E_SHAPE;
  if (n_layers > SS_MAXL) return VCT_E_SHAPE;                 // one launch: the gradient rows never leave the workgroup
  if (!q->dy || !layers[n_layers - 1].dx || q->key_pad_shift < 0 || (q->key_pad != nullptr && q->key_pad_shift >= q->L)) return VCT_E_ARG;
  const int64_t per_layer = vct_layer_ss_bwd_stream_chunks(q->ff);
  auto norm_ok = [](const vct_ss_bwd_norm& n) { return n.gamma && n.mean && n.rstd && n.ws; };
  auto cvt = [](const vct_ss_bwd_norm& n) { return SsBwdNorm{n.gamma, n.mean, n.rstd, n.ws}; };
  for (int l = 0; l < n_layers; l++) {
    const vct_layer_ss_bwd_desc& d = layers[l];
    if (d.dtype != q->dtype || d.B != q->B || d.L != q->L || d.d != q->d || d.H != q->H || d.ff != q->ff || d.act != q->act ||
        d.causal != q->causal || d.key_pad_shift != q->key_pad_shift || d.key_pad != q->key_pad || d.key_ids != q->key_ids ||
        d.key_ids_bs != q->key_ids_bs || d.pad_id != q->pad_id || d.seed != q->seed || d.p_drop != q->p_drop)
      return VCT_E_ARG;
    if (d.nchunks != per_layer) return VCT_E_SHAPE;
    if (!d.wpk || (const char*)d.wpk != (const char*)q->wpk + (size_t)(l * per_layer) * SS_CHUNK * 2) return VCT_E_ARG;
    if (d.last && l != 0) return VCT_E_ARG;
    if (!d.x || !d.qkv || !d.a || !d.x1 || !d.hpre || !d.f || !d.df || !d.dhpre || !d.da || !d.dqkv) return VCT_E_ARG;
    if (!norm_ok(d.n1) || !norm_ok(d.n3) || (d.last && (!norm_ok(d.nf) || !d.y_last))) return VCT_E_ARG;
    const uintptr_t al = (uintptr_t)d.wpk | (uintptr_t)d.x | (uintptr_t)d.qkv | (uintptr_t)d.a | (uintptr_t)d.x1 | (uintptr_t)d.hpre |
                         (uintptr_t)d.f | (uintptr_t)d.df | (uintptr_t)d.dhpre | (uintptr_t)d.da | (uintptr_t)d.dqkv | (uintptr_t)d.dy |
                         (uintptr_t)d.dx | (uintptr_t)d.y_last | (uintptr_t)d.n1.gamma | (uintptr_t)d.n3.gamma | (uintptr_t)d.nf.gamma |
                         (uintptr_t)d.n1.ws | (uintptr_t)d.n3.ws | (uintptr_t)d.nf.ws;
    if (al & 15) return VCT_E_ALIGN;
  }
  hipStream_t st = (hipStream_t)stream;
  const int one = q->L <= 16 ? 1 : 0;
  static vct::DynLdsOptIn optin[2];
  const void* fn = one ? (const void*)layer_ss_bwd_kernel<1> : (const void*)layer_ss_bwd_kernel<2>;
  if (hipError_t e = optin[one].ensure(fn, SS_LDS); e != hipSuccess) return (int)e;
  SsBwdP p;
  memset(&p, 0, sizeof(p));
  p.B = q->B; p.L = q->L; p.ff = q->ff; p.act = q->act; p.last = q->last; p.causal = q->causal; p.nl = n_layers;
  p.wpk = reinterpret_cast<const bf16_t*>(q->wpk); p.nchunks = (int)(per_layer * n_layers);
  p.dy = reinterpret_cast<const bf16_t*>(q->dy); p.dx = reinterpret_cast<bf16_t*>(layers[n_layers - 1].dx);
  p.y_last = reinterpret_cast<const bf16_t*>(q->y_last); p.nf = cvt(q->nf);
  p.key_pad = q->key_pad; p.key_pad_shift = q->key_pad_shift; p.key_ids = q->key_ids; p.key_ids_bs = q->key_ids_bs; p.pad_id = q->pad_id;
  p.seed = q->seed; p.p_drop = q->p_drop;
  for (int l = 0; l < n_layers; l++) {
    const vct_layer_ss_bwd_desc& d = layers[l];
    SsBwdW& w = p.lw[l];
    w.x = reinterpret_cast<const bf16_t*>(d.x); w.qkv = reinterpret_cast<const bf16_t*>(d.qkv); w.a = reinterpret_cast<const bf16_t*>(d.a);
    w.x1 = reinterpret_cast<const bf16_t*>(d.x1); w.hpre = reinterpret_cast<const bf16_t*>(d.hpre); w.f = reinterpret_cast<const bf16_t*>(d.f);
    w.n1 = cvt(d.n1); w.n3 = cvt(d.n3);
    w.df = reinterpret_cast<bf16_t*>(d.df); w.dhpre = reinterpret_cast<bf16_t*>(d.dhpre); w.da = reinterpret_cast<bf16_t*>(d.da);
    w.dqkv = reinterpret_cast<bf16_t*>(d.dqkv);
    w.site_sa = d.site_sa; w.site_n1 = d.site_n1; w.site_ff = d.site_ff; w.site_n3 = d.site_n3;
  }
  if (one) vct::launch(layer_ss_bwd_kernel<1>, dim3(p.B), dim3(SS_NT), SS_LDS, st, p);
  else vct::launch(layer_ss_bwd_kernel<2>, dim3(p.B), dim3(SS_NT), SS_LDS, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}
