// Multi-head attention core for short sequences (Lq, Lk <= 64, head_dim <= 128) on gfx950.
// One wave per (batch, head); Q/K/V (and dO) head slices are staged once into LDS with 16-byte
// loads, every product is an MFMA tile (bf16 16x16x32 or exact-fp32 16x16x4), the softmax row
// reductions are wave shuffles, masks (causal / key padding) and dropout are applied in registers
// and the score matrix never touches HBM.
//
// Forward computes the scores TRANSPOSED (S^T = K Q^T): the C fragment of a 16x16 MFMA then holds,
// per lane, one query column and four keys -- exactly the A-operand shape of the P V product, so P
// feeds the second MFMA straight from registers.  V (and K, Q, dO in the backward) are consumed as
// B operands whose reduction index runs along LDS rows: bf16 uses the gfx950 LDS transpose read.
//
// Backward (one launch, same wave): phase A (per query tile) rebuilds S^T, P, dP^T = V dO^T,
// D = rowsum(dP*P), dS and dQ = dS K; phase B (per key tile) rebuilds S, P, dP in the un-transposed
// form whose C fragment is the A operand of the reductions over queries: dV = P^T dO, dK = dS^T Q.
#include "vct_attn_core.h"

namespace vct {

// One workgroup per (batch, head): ONE wave for the short sequences of the caption configs (two 16-row tiles at most: a second wave
// idles on a single key tile and the two contend for LDS -- measured slower in round 1), blockDim.x / 64 waves for three or four
// tiles (configs[3]: 40 x 40 x 128 -- the 70 KB of staged operands leave two workgroups per CU, i.e. two WAVES per CU with one wave
// each): the waves stage together and take the query tiles (forward, backward phase A) / key tiles (phase B) in turn.
template <typename T, int DT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  attn_fwd_wave<T, DT>(p, blockIdx.x / p.H, blockIdx.x % p.H, blockIdx.x, smem, threadIdx.x, (T*)nullptr, 0, (int)(blockDim.x >> 6));
}

template <typename T, int DT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnP p) {
  using C = AttnCfg<T, DT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int LQT = (p.Lq + 15) / 16, LKT = (p.Lk + 15) / 16;
  const int RQ = ((LQT + 1) / 2) * 32, RK = ((LKT + 1) / 2) * 32;
  T* Qs = reinterpret_cast<T*>(smem);
  T* dOs = Qs + RQ * C::STR;
  T* Ks = dOs + RQ * C::STR;
  T* Vs = Ks + RK * C::STR;
  float* stat_m = reinterpret_cast<float*>(Vs + RK * C::STR);  // [64] row max
  float* stat_i = stat_m + 64;                                  // [64] 1 / row sum
  float* stat_d = stat_i + 64;                                  // [64] D = rowsum(dP * P)
  const T* qg = reinterpret_cast<const T*>(p.q) + (long)b * p.Lq * p.ldq + (long)h * p.hd;
  const T* kg = reinterpret_cast<const T*>(p.k) + (long)b * p.Lk * p.ldk + (long)h * p.hd;
  const T* vg = reinterpret_cast<const T*>(p.v) + (long)b * p.Lk * p.ldv + (long)h * p.hd;
  const T* dog = reinterpret_cast<const T*>(p.d_o) + (long)b * p.Lq * p.ld_do + (long)h * p.hd;
  T* dqg = reinterpret_cast<T*>(p.dq) + (long)b * p.Lq * p.ld_dq + (long)h * p.hd;
  T* dkg = reinterpret_cast<T*>(p.dk) + (long)b * p.Lk * p.ld_dk + (long)h * p.hd;
  T* dvg = reinterpret_cast<T*>(p.dv) + (long)b * p.Lk * p.ld_dv + (long)h * p.hd;
  {
    const StageJob<T> jobs[4] = {{Qs, qg, p.ldq, p.Lq, RQ}, {dOs, dog, p.ld_do, p.Lq, RQ}, {Ks, kg, p.ldk, p.Lk, RK}, {Vs, vg, p.ldv, p.Lk, RK}};
    stage_multi<T, DT, 4>(jobs, p.hd, (int)threadIdx.x, nwv * 64);
  }
  __syncthreads();
  const unsigned long long kp_row = load_padmask(p, b, lane);
  const Dropout dr = make_dropout(p.seed, p.site, p.p_drop);
  const float scale = 1.0f / sqrtf((float)p.hd);
  const int hd4 = (p.hd + 3) / 4;

  // ---------------- phase A: per query tile, transposed form -> stats + dQ ----------------
  for (int qt = wv; qt < LQT; qt += nwv) {
    f32x4 st[4], dpt[4];
    scores_T<T, DT>(st, Ks, Qs, qt, LKT, hd4, scale, p, kp_row, lane);
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) m = fmaxf(m, st[t][r]);
    m = red4_max(m);
    if (m == -INFINITY) m = 0.0f;
    float l = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; t++) {           // key tiles beyond Lk hold -inf: nothing to exponentiate, hash or scale there
      if (t < LKT) {
#pragma unroll
        for (int r = 0; r < 4; r++) { st[t][r] = attn_exp<T>(st[t][r] - m); l += st[t][r]; }
      } else {
        st[t] = f32x4{0, 0, 0, 0};
      }
    }
    l = red4_sum(l);
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    const int qq = qt * 16 + i;
    // dPd^T[key][q] = V[key] . dO[q]
    float dsum = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      dpt[t] = f32x4{0, 0, 0, 0};
      if (t < LKT) {
        if constexpr (C::BF) {
#pragma unroll
          for (int ks = 0; ks < C::KS; ks++)
            dpt[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rowk<C::STR>(Vs, t * 16, ks, lane),
                                                             frag_rowk<C::STR>(dOs, qt * 16, ks, lane), dpt[t], 0, 0, 0);
        } else {
          for (int k4 = 0; k4 < hd4; k4++)
            dpt[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[(t * 16 + i) * C::STR + k4 * 4 + g],
                                                          dOs[(qt * 16 + i) * C::STR + k4 * 4 + g], dpt[t], 0, 0, 0);
        }
      }
      if (t < LKT) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int kk = t * 16 + g * 4 + r;
          st[t][r] *= inv;                                                       // P
          dpt[t][r] *= drop_mult(dr, (uint32_t)((blockIdx.x * p.Lq + qq) * p.Lk + kk));  // dP = dPd * M
          dsum += dpt[t][r] * st[t][r];
        }
      }
    }
    dsum = red4_sum(dsum);
    if (g == 0) { stat_m[qq] = m; stat_i[qq] = inv; stat_d[qq] = dsum; }
    // dS^T = P * (dP - D) ; dQ tile = scale * dS K
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) st[t][r] = st[t][r] * (dpt[t][r] - dsum) * scale;
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) acc[dt] = f32x4{0, 0, 0, 0};
    if constexpr (C::BF) {
#pragma unroll
      for (int kp = 0; kp < 2; kp++) {
        if (kp * 2 < LKT) {
          const bf16x8 pa = pack_p(st[kp * 2], st[kp * 2 + 1]);
#pragma unroll
          for (int dt = 0; dt < DT; dt++)
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_colk<C::STR>(Ks, kp * 32, kp * 32 + 16, dt * 16, lane), pa,
                                                              acc[dt], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; t++) {
        if (t < LKT) {
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int dt = 0; dt < DT; dt++)
              acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(t * 16 + g * 4 + r) * C::STR + dt * 16 + i], st[t][r], acc[dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < DT; dt++) store_row4<T>(dqg, p.ld_dq, qt * 16 + i, dt * 16 + g * 4, acc[dt], p.Lq, p.hd);   // acc = dQ^T tile
  }
  __syncthreads();  // stats visible

  // ---------------- phase B: per key tile, un-transposed form -> dV, dK ----------------
  for (int t = wv; t < LKT; t += nwv) {
    f32x4 av[DT], ak[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) { av[dt] = f32x4{0, 0, 0, 0}; ak[dt] = f32x4{0, 0, 0, 0}; }
    const int kk = t * 16 + i;  // this lane's key column in S[q][key]
#pragma unroll
    for (int qp = 0; qp < 2; qp++) {
      if (qp * 2 < LQT) {
        f32x4 pd[2], dsv[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int qt = qp * 2 + u;
          f32x4 s = f32x4{0, 0, 0, 0}, dp = f32x4{0, 0, 0, 0};
          if (qt < LQT) {
            if constexpr (C::BF) {
#pragma unroll
              for (int ks = 0; ks < C::KS; ks++) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rowk<C::STR>(Qs, qt * 16, ks, lane),
                                                            frag_rowk<C::STR>(Ks, t * 16, ks, lane), s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rowk<C::STR>(dOs, qt * 16, ks, lane),
                                                             frag_rowk<C::STR>(Vs, t * 16, ks, lane), dp, 0, 0, 0);
              }
            } else {
              for (int k4 = 0; k4 < hd4; k4++) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[(qt * 16 + i) * C::STR + k4 * 4 + g],
                                                         Ks[(t * 16 + i) * C::STR + k4 * 4 + g], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(dOs[(qt * 16 + i) * C::STR + k4 * 4 + g],
                                                          Vs[(t * 16 + i) * C::STR + k4 * 4 + g], dp, 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int qq = qt * 16 + g * 4 + r;  // row of S[q][key] held in register r
            float pv = 0.0f, dv = 0.0f, pdv = 0.0f;
            if (qt < LQT && qq < p.Lq && !key_masked(p, kp_row, qq, kk)) {
              const float mult = drop_mult(dr, (uint32_t)((blockIdx.x * p.Lq + qq) * p.Lk + kk));
              pv = attn_exp<T>(s[r] * scale - stat_m[qq]) * stat_i[qq];
              pdv = pv * mult;
              dv = pv * (dp[r] * mult - stat_d[qq]) * scale;
            }
            pd[u][r] = pdv;
            dsv[u][r] = dv;
          }
        }
        if constexpr (C::BF) {
          const bf16x8 pa = pack_p(pd[0], pd[1]);
          const bf16x8 da = pack_p(dsv[0], dsv[1]);
#pragma unroll
          for (int dt = 0; dt < DT; dt++) {
            av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_colk<C::STR>(dOs, qp * 32, qp * 32 + 16, dt * 16, lane), pa, av[dt], 0, 0, 0);
            ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_colk<C::STR>(Qs, qp * 32, qp * 32 + 16, dt * 16, lane), da, ak[dt], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int u = 0; u < 2; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int qrow = (qp * 2 + u) * 16 + g * 4 + r;
#pragma unroll
              for (int dt = 0; dt < DT; dt++) {
                av[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dOs[qrow * C::STR + dt * 16 + i], pd[u][r], av[dt], 0, 0, 0);
                ak[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[qrow * C::STR + dt * 16 + i], dsv[u][r], ak[dt], 0, 0, 0);
              }
            }
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {   // av / ak = dV^T / dK^T tiles
      store_row4<T>(dvg, p.ld_dv, t * 16 + i, dt * 16 + g * 4, av[dt], p.Lk, p.hd);
      store_row4<T>(dkg, p.ld_dk, t * 16 + i, dt * 16 + g * 4, ak[dt], p.Lk, p.hd);
    }
  }
}

template <typename T, int DT> static int attn_launch(const AttnP& p, bool bwd, hipStream_t st) {
  const size_t lds = attn_lds_bytes<T, DT>(p.Lq, p.Lk, bwd);
  if (lds > 160 * 1024) return VCT_E_SHAPE;
  if (lds > 64 * 1024) {       // opt in to the full 160 KB once per device and direction (the attribute is a maximum, not a request)
    static vct::DynLdsOptIn optin[2];
    const void* fn = bwd ? (const void*)attn_bwd_kernel<T, DT> : (const void*)attn_fwd_kernel<T, DT>;
    if (hipError_t e = optin[bwd].ensure(fn, 160 * 1024); e != hipSuccess) return (int)e;
  }
  const dim3 grid(p.B * p.H);
  const int LQT = (p.Lq + 15) / 16, LKT = (p.Lk + 15) / 16;
  const int tiles = bwd ? (LQT > LKT ? LQT : LKT) : LQT;
  static const char* wenv = getenv("VCT_ATTN_WAVES");          // A/B: 0 = by shape, n = force
  // measured (round 4, same box): configs[3] (3 x 3 tiles, head_dim 128) 18.94 -> 17.36 ms per step with 3 waves; the shipped shape
  // (2 tiles, head_dim 96) 3.57 -> 3.52 ms with 2; cfg-B (2 tiles, head_dim 64: five to six workgroups fit a CU) stays at one wave
  int nwv = tiles >= 3 ? (tiles < 4 ? tiles : 4) : (tiles == 2 && p.hd > 64 ? 2 : 1);
  if (wenv != nullptr && atoi(wenv) > 0) nwv = atoi(wenv) < 4 ? atoi(wenv) : 4;
  if (bwd) vct::launch((attn_bwd_kernel<T, DT>), grid, dim3(64 * nwv), lds, st, p);
  else vct::launch((attn_fwd_kernel<T, DT>), grid, dim3(64 * nwv), lds, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

template <typename T> static int attn_dispatch(const AttnP& p, bool bwd, hipStream_t st) {
  if (p.hd <= 16) return attn_launch<T, 1>(p, bwd, st);
  if (p.hd <= 32) return attn_launch<T, 2>(p, bwd, st);
  if (p.hd <= 64) return attn_launch<T, 4>(p, bwd, st);
  if (p.hd <= 96) return attn_launch<T, 6>(p, bwd, st);
  if (p.hd <= 128) return attn_launch<T, 8>(p, bwd, st);
  return VCT_E_SHAPE;
}

}  // namespace vct
using namespace vct;

static int attn_common(const vct_attn_desc* d, bool bwd, void* stream) {
  if (!d || !d->q || !d->k || !d->v) return VCT_E_ARG;
  if (d->dtype != VCT_F32 && d->dtype != VCT_BF16) return VCT_E_ARG;
  if (!bwd && !d->o) return VCT_E_ARG;
  if (bwd && (!d->d_o || !d->dq || !d->dk || !d->dv)) return VCT_E_ARG;
  if (d->B <= 0 || d->H <= 0 || d->Lq <= 0 || d->Lk <= 0 || d->hd <= 0) return VCT_E_SHAPE;
  if (d->Lq > 64 || d->Lk > 64 || d->hd > 128) return VCT_E_SHAPE;
  const int vec = d->dtype == VCT_BF16 ? 8 : 4;
  if (d->hd % vec || d->ldq % vec || d->ldk % vec || d->ldv % vec) return VCT_E_ALIGN;
  if (bwd && (d->ld_do % vec)) return VCT_E_ALIGN;
  // outputs are written four head-dim columns at a time
  const uintptr_t omask = d->dtype == VCT_BF16 ? 7 : 15;
  if (!bwd && ((d->ldo % 4) || ((uintptr_t)d->o & omask))) return VCT_E_ALIGN;
  if (bwd && ((d->ld_dq % 4) || (d->ld_dk % 4) || (d->ld_dv % 4) || (((uintptr_t)d->dq | (uintptr_t)d->dk | (uintptr_t)d->dv) & omask)))
    return VCT_E_ALIGN;
  AttnP p;
  p.B = d->B; p.H = d->H; p.Lq = d->Lq; p.Lk = d->Lk; p.hd = d->hd; p.causal = d->causal;
  p.q = d->q; p.ldq = d->ldq; p.k = d->k; p.ldk = d->ldk; p.v = d->v; p.ldv = d->ldv;
  p.o = d->o; p.ldo = d->ldo;
  p.key_pad = d->key_pad; p.key_pad_shift = d->key_pad_shift;
  p.key_ids = d->key_ids; p.key_ids_bs = d->key_ids_bs; p.pad_id = d->pad_id;
  if (d->key_pad_shift < 0 || (d->key_pad != nullptr && d->key_pad_shift >= d->Lk)) return VCT_E_SHAPE;
  p.seed = d->seed; p.site = d->site; p.p_drop = d->p_drop;
  p.d_o = d->d_o; p.ld_do = d->ld_do;
  p.dq = d->dq; p.ld_dq = d->ld_dq; p.dk = d->dk; p.ld_dk = d->ld_dk; p.dv = d->dv; p.ld_dv = d->ld_dv;
  p.q_bs = d->q_bs ? d->q_bs : (long)d->Lq * d->ldq;
  p.k_bs = d->k_bs ? d->k_bs : (long)d->Lk * d->ldk;
  p.v_bs = d->v_bs ? d->v_bs : (long)d->Lk * d->ldv;
  p.o_bs = d->o_bs ? d->o_bs : (long)d->Lq * d->ldo;
  if (bwd && (d->q_bs || d->k_bs || d->v_bs || d->o_bs)) return VCT_E_ARG;  // strided batches: forward (decode) only
  hipStream_t st = (hipStream_t)stream;
  return d->dtype == VCT_BF16 ? attn_dispatch<bf16_t>(p, bwd, st) : attn_dispatch<float>(p, bwd, st);
}

extern "C" int vct_attn_fwd(const vct_attn_desc* d, void* stream) { return attn_common(d, false, stream); }
extern "C" int vct_attn_bwd(const vct_attn_desc* d, void* stream) { return attn_common(d, true, stream); }
