// Shared pieces of the sample-stationary layer kernels (csrc/vct_layer_ss.hip: forward; csrc/vct_layer_ss_bwd.hip: the dX chain of the
// backward): LDS map, the weight stream, the wave-level product, panel copies, epilogues, the one-wave-per-head attention on LDS panels.
#pragma once
#include <cstring>
#include "vct_attn_core.h"

namespace vct {

constexpr int SS_D = 512, SS_H = 8, SS_HD = 64;
// Waves per workgroup: 8 (2 per SIMD, <= 256 registers) or 16 (4 per SIMD, <= 128 registers, a wave owns 32 columns per block).
// Measured at cfg-B (tools/ss_layer_stamps.hip, gpurun_out/r4f vs r4g): 16 waves finish a product block faster per wave (7-8.5 k
// cycles vs 8-9.5 k) but pay it back at every barrier (wave 0 waits 6-15 k cycles for the other fifteen after multi-block
// products) and spill in the LayerNorm epilogues: decoder layer 140 us either way, encoder layer 87 vs 81 us -> 8 waves.
constexpr int SS_NW = 8, SS_NT = SS_NW * 64;
constexpr int SS_TPW = 32 / SS_NW;                 // 16-column MFMA tiles per wave and 512-column block
constexpr int SS_CPW = SS_TPW * 16;                // columns per wave and block
constexpr int SS_WSTR = SS_TPW * 2 * 512;          // bf16 elements of a chunk that belong to one wave (SS_TPW tiles x 2 k-steps x 1 KiB)
constexpr int SS_PSTR = SS_D + 8;                  // row stride (bf16 elements) of a [32][512] panel
constexpr int SS_SLOT = 32 * SS_PSTR * 2;          // 33,280 B
constexpr int SS_QSTR = 3 * SS_D + 8;              // q | k | v panel (self-attention): 32 x 1544 bf16 = 3 slots
constexpr int SS_KVSTR = 2 * SS_D + 8;             // k | v panel of the memory (cross-attention): 16 x 1032 bf16 = 1 slot
constexpr int SS_R0 = 0, SS_R1A = SS_SLOT, SS_R1B = 2 * SS_SLOT, SS_R1C = 3 * SS_SLOT;
constexpr int SS_RM = 4 * SS_SLOT;                 // memory rows of this sample: 16 x 520 bf16
constexpr int SS_RED = SS_RM + 16 * SS_PSTR * 2;   // LayerNorm partials: 2 x [8 waves][32 rows] fp32
constexpr int SS_B1 = SS_RED + 2 * SS_NW * 32 * 4;   // linear1 bias (fp32, ff <= 2048): read by the pipelined feed-forward epilogue
constexpr int SS_FF_MAX = 2048;
constexpr int SS_LDS = SS_B1 + SS_FF_MAX * 4;
constexpr long SS_CHUNK = 32768;                   // bf16 elements per K chunk of the stream (64 KiB: 8 waves x 8 fragments x 1 KiB)
static_assert(SS_LDS <= 160 * 1024, "LDS budget");
static_assert(32 * SS_QSTR * 2 <= 3 * SS_SLOT && 16 * SS_KVSTR * 2 <= SS_SLOT, "panel slots");

struct SsNorm { const float* g; const float* b; bf16_t* y; float* mean; float* rstd; };

constexpr int SS_MAXL = 4;           // layers per launch (kernel-argument budget); deeper stacks take several launches

struct SsLayerW {                    // what differs from layer to layer
  // self-attention block
  const float* b_qkv; const float* b_o;
  bf16_t* qkv; bf16_t* o; bf16_t* a;
  SsNorm n1;
  // cross-attention block
  const float* b_cq; const float* b_ckv; const float* b_co;
  bf16_t* cq; bf16_t* ckv; bf16_t* co; bf16_t* ca;
  SsNorm n2;
  // feed-forward block
  const float* b1; const float* b2;
  bf16_t* hpre; bf16_t* h; bf16_t* f;
  SsNorm n3;
  uint32_t site_sa, site_n1, site_ca, site_n2, site_ff, site_n3;
};

struct SsLayerP {
  int B, L, Lm;                      // samples, rows per sample (<= 32), memory rows per sample (<= 16; decoder layers)
  int ff, act, last, causal, nl;
  const bf16_t* wpk; int nchunks;    // packed weight stream of ALL nl layers, back to back (stream order, see vct_ss_pack)
  bf16_t* x;                         // [B*L, 512] input of the first layer (pro == 0), or where the prologue stores the rows it builds
  const bf16_t* mem;                 // [B*Lm, 512] encoder memory (decoder layers)
  // stack prologue: 0 = x is given; 1 = encoder front end (MMEncoder.py:244-273: unify Linear, mean token, temporal encoding; the unify
  // weight is the first 8 chunks of the stream); 2 = token embedding + positional rows + dropout (CapDecoder.py:48, Embedding.py:23-25)
  int pro;
  const void* feats; int feats_f32; bf16_t* x_in; const float* b_u; const float* pe;      // pro 1: feats [B*(L-1), 512], bf16 copy out, bias, PE' rows [L, 512]
  const int64_t* emb_ids; long emb_ids_bs; const float* emb_table; const float* emb_pos; uint32_t site_emb;   // pro 2
  SsNorm nf;                         // stack-final norm behind the last layer (last != 0)
  // masks of the self-attention (as vct_attn_desc)
  const uint8_t* key_pad; int key_pad_shift;
  const int64_t* key_ids; long key_ids_bs; long pad_id;
  // dropout
  const uint32_t* seed; float p_drop;
  SsLayerW lw[SS_MAXL];
#ifdef SS_STAMPS
  unsigned long long* dbg;           // development build (tools/ss_layer_stamps.hip): [B][64] cycle stamps of wave 0 (last layer of the launch)
#endif
};
#ifdef SS_STAMPS
#ifdef SS_STAMPS_ALLWAVES      /* every wave's lane 0: [B][8 waves][64] */
#define SS_STAMP(i) do { if ((tid & 63) == 0) p.dbg[((long)blockIdx.x * 8 + (tid >> 6)) * 64 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define SS_STAMP(i) do { if (tid == 0) p.dbg[(long)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter(); } while (0)
#endif
#else
#define SS_STAMP(i) do { } while (0)
#endif

// workgroup barrier that orders LDS traffic only: the weight prefetch (and the panel copies' stores) stay in flight across it
__device__ __forceinline__ void ss_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---- weight stream: this lane's view of the layer's packed weights ------------------------------------------------------------------
// chunk c of the stream = elements [c*32768, (c+1)*32768): wave w's 2*SS_TPW fragments (column tile t, k-step s) at w*SS_WSTR + (t*2+s)*512,
// lane l's 8 bf16 at + l*8.  Loads are UNCONDITIONAL (the pointer stops at the last chunk): straight-line code, exact vmcnt waits.
struct WStream {
  const bf16_t* p;        // next chunk to fetch (this lane)
  const bf16_t* last;     // last chunk of the layer (this lane)
};
__device__ __forceinline__ void ws_fetch(WStream& ws, bf16x8 (&dst)[SS_TPW][2]) {
#pragma unroll
  for (int t = 0; t < SS_TPW; t++)
#pragma unroll
    for (int s = 0; s < 2; s++) dst[t][s] = *reinterpret_cast<const bf16x8*>(ws.p + (t * 2 + s) * 512);
  ws.p = (ws.p + SS_CHUNK <= ws.last) ? ws.p + SS_CHUNK : ws.last;
}

// acc[m][t] += W_chunk-fragments x A-fragments for `nch` (even) chunks; A = LDS panel, this lane's pointer `a` already at
// (row li, k-group lg*8), row-tile stride 16*astr, chunk kc0 first.  The stream runs TWO chunks ahead with two register buffers: on
// entry b0 / b1 hold the first two chunks (in flight), each buffer is re-fetched right behind the MFMAs that read it, and on exit they
// hold the first two chunks of whatever comes next in the stream -- so an epilogue between two products has 16 KB per wave in flight.
template <int MT>
__device__ __forceinline__ void gemm_step(f32x4 (&acc)[MT][SS_TPW], const bf16_t* a, const int astr, const int kc, const bf16x8 (&b)[SS_TPW][2]) {
  bf16x8 af[MT][2];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int s = 0; s < 2; s++) af[m][s] = *reinterpret_cast<const bf16x8*>(a + m * 16 * astr + kc * 64 + s * 32);
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int t = 0; t < SS_TPW; t++) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[t][s], af[m][s], acc[m][t], 0, 0, 0);
}
template <int MT>
__device__ __forceinline__ void wave_gemm(f32x4 (&acc)[MT][SS_TPW], const bf16_t* a, const int astr, const int kc0, const int nch, WStream& ws,
                                          bf16x8 (&b0)[SS_TPW][2], bf16x8 (&b1)[SS_TPW][2]) {
  for (int c = 0; c < nch; c += 2) {
    gemm_step<MT>(acc, a, astr, kc0 + c, b0);
    ws_fetch(ws, b0);
    gemm_step<MT>(acc, a, astr, kc0 + c + 1, b1);
    ws_fetch(ws, b1);
  }
}
// the same for exactly 8 chunks, fully unrolled, with cb(k) (k = 0..7: independent vector work) issued in front of K step k
template <int MT, class F>
__device__ __forceinline__ void wave_gemm8_cb(f32x4 (&acc)[MT][SS_TPW], const bf16_t* a, const int astr, WStream& ws, bf16x8 (&b0)[SS_TPW][2],
                                              bf16x8 (&b1)[SS_TPW][2], F&& cb) {
  static_for<4>([&](auto I) {
    constexpr int i = decltype(I)::value;
    cb(std::integral_constant<int, 2 * i>{});
    gemm_step<MT>(acc, a, astr, 2 * i, b0);
    ws_fetch(ws, b0);
    cb(std::integral_constant<int, 2 * i + 1>{});
    gemm_step<MT>(acc, a, astr, 2 * i + 1, b1);
    ws_fetch(ws, b1);
  });
}

template <int MT> __device__ __forceinline__ void acc_zero(f32x4 (&acc)[MT][SS_TPW]) {
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) acc[m][t] = f32x4{0, 0, 0, 0};
}

__device__ __forceinline__ void load_bias4(float4 (&bv)[SS_TPW], const float* bias, const int col0, const int lg) {
#pragma unroll
  for (int t = 0; t < SS_TPW; t++) bv[t] = *reinterpret_cast<const float4*>(bias + col0 + t * 16 + lg * 4);
}

struct alignas(8) BV4 { bf16_t e[4]; };
struct alignas(16) BV8s { bf16_t e[8]; };
// agent-scope streaming stores (vct_common.h) for the saved tensors: +1 % per layer (tools/ss_layer_stamps.hip, -DSS_STREAM_STORES=1): off
#ifndef SS_STREAM_STORES
#define SS_STREAM_STORES 0
#endif

// rows [0, rows) x NCOLS columns of an LDS panel -> global [row0 + r][col0 ..], 16 bytes per thread and step
template <int NCOLS>
__device__ __forceinline__ void panel_to_global(const bf16_t* panel, const int pstr, const int rows, bf16_t* g, const long ld,
                                                const long row0, const int col0, const int tid) {
  constexpr int vpr = NCOLS / 8;
  const int total = rows * vpr;
  for (int v = tid; v < total; v += SS_NT) {
    const int r = v / vpr, c = (v - r * vpr) * 8;
#if SS_STREAM_STORES
    store_stream16(g + (row0 + r) * ld + col0 + c, *reinterpret_cast<const stream_u32x4*>(panel + r * pstr + c));
#else
    *reinterpret_cast<BV8s*>(g + (row0 + r) * ld + col0 + c) = *reinterpret_cast<const BV8s*>(panel + r * pstr + c);
#endif
  }
}
// global rows -> panel (rows >= L zero-filled up to `alloc`)
__device__ __forceinline__ void global_to_panel(bf16_t* panel, const int pstr, const int L, const int alloc, const bf16_t* g, const long ld,
                                                const long row0, const int tid) {
  constexpr int vpr = SS_D / 8;
  for (int v = tid; v < alloc * vpr; v += SS_NT) {
    const int r = v / vpr, c = (v - r * vpr) * 8;
    BV8s val;
    if (r < L) val = *reinterpret_cast<const BV8s*>(g + (row0 + r) * ld + c);
    else {
#pragma unroll
      for (int j = 0; j < 8; j++) val.e[j] = 0;
    }
    *reinterpret_cast<BV8s*>(panel + r * pstr + c) = val;
  }
}

__device__ __forceinline__ void load_gb(float4 (&gm)[SS_TPW], float4 (&bt)[SS_TPW], const SsNorm& n, const int ecol) {
#pragma unroll
  for (int t = 0; t < SS_TPW; t++) {
    gm[t] = *reinterpret_cast<const float4*>(n.g + ecol + t * 16);
    bt[t] = *reinterpret_cast<const float4*>(n.b + ecol + t * 16);
  }
}

// plain epilogue: panel[row][pcol0 + ...] = bf16(acc + bias)
template <int MT>
__device__ __forceinline__ void epi_store(const f32x4 (&acc)[MT][SS_TPW], const float4 (&bv)[SS_TPW], bf16_t* panel, const int pstr, const int pcol0,
                                          const int li, const int lg) {
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) {
      BV4 o;
      o.e[0] = f2bf(acc[m][t][0] + bv[t].x); o.e[1] = f2bf(acc[m][t][1] + bv[t].y);
      o.e[2] = f2bf(acc[m][t][2] + bv[t].z); o.e[3] = f2bf(acc[m][t][3] + bv[t].w);
      *reinterpret_cast<BV4*>(panel + (m * 16 + li) * pstr + pcol0 + t * 16 + lg * 4) = o;
    }
}

// residual + dropout + LayerNorm (+ second LayerNorm) epilogue of out_proj / linear2 (what vct_add_ln_fwd / vct_add_ln_ln_fwd compute):
//   a = bf16(acc + bias) -> panel AP;  s = a * dropmask + res;  y = LN(s) -> panel YP;  [y2 = LN2(bf16 y) -> panel Y2P]
// res: 4 bf16 per (m, t) in registers.  Row statistics: this wave's 64 columns -> 4-lane-group shuffle -> LDS partials of the 8 waves.
template <int MT>
__device__ __forceinline__ void epi_ln(f32x4 (&acc)[MT][SS_TPW], const float4 (&bv)[SS_TPW], const BV4 (&res)[MT][SS_TPW], const SsNorm& n, const SsNorm* n2,
                                       const Dropout& dr, const long grow0, const int L, bf16_t* AP, bf16_t* YP, bf16_t* Y2P, float* red,
                                       const int wave, const int li_in, const int lg_in) {
  // lane indices behind an opaque barrier: otherwise the address / counter arithmetic shared by the layer's three LayerNorm
  // epilogues is computed once and kept alive (spilled) from the first to the last of them
  int li = li_in, lg = lg_in;
  asm volatile("" : "+v"(li), "+v"(lg));
  const int colw = wave * SS_CPW + lg * 4;
  // gamma / beta: issued now, first used two barriers further down -- they land behind the 16 KB of weight prefetch this wave has in
  // flight (in-order return) without anybody waiting for them
  float4 gm[SS_TPW], bt[SS_TPW];
  load_gb(gm, bt, n, colw);
  float part[MT];
#pragma unroll
  for (int m = 0; m < MT; m++) {
    part[m] = 0.0f;
    const uint32_t grow = (uint32_t)(grow0 + m * 16 + li);
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) {
      const float bb[4] = {bv[t].x, bv[t].y, bv[t].z, bv[t].w};
      float dm[4];
      drop_mults<4>(dr, grow * (uint32_t)SS_D + (uint32_t)(colw + t * 16), dm);
      BV4 av;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        av.e[r] = f2bf(acc[m][t][r] + bb[r]);          // the saved activation is the bf16 value, and the norm is built on it
        const float s = bf2f(av.e[r]) * dm[r] + bf2f(res[m][t].e[r]);
        acc[m][t][r] = s;
        part[m] += s;
      }
      *reinterpret_cast<BV4*>(AP + (m * 16 + li) * SS_PSTR + colw + t * 16) = av;
    }
    part[m] = red4_sum(part[m]);
  }
  // (Measured and dropped, round 4: ONE exchange of per-wave (sum, M2 about the wave's own mean) pairs combined exactly -- one barrier
  // less per LayerNorm, 82.4 / 138.0 us per encoder / decoder layer against 82.4 / 135.7: the second barrier costs nothing once the
  // waves are aligned by the first, the extra in-register pass does.)
  float* red0 = red;                  // [8][32]
  float* red1 = red + SS_NW * 32;
  auto stats = [&](float (&mean)[MT], float (&rstd)[MT]) {
    if (lg == 0) {
#pragma unroll
      for (int m = 0; m < MT; m++) red0[wave * 32 + m * 16 + li] = part[m];
    }
    ss_barrier();
    float sq[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
      float s = 0.0f;
#pragma unroll
      for (int w = 0; w < SS_NW; w++) s += red0[w * 32 + m * 16 + li];
      mean[m] = s * (1.0f / (float)SS_D);
      sq[m] = 0.0f;
#pragma unroll
      for (int t = 0; t < SS_TPW; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) { const float c = acc[m][t][r] - mean[m]; sq[m] += c * c; }
      sq[m] = red4_sum(sq[m]);
    }
    if (lg == 0) {
#pragma unroll
      for (int m = 0; m < MT; m++) red1[wave * 32 + m * 16 + li] = sq[m];
    }
    ss_barrier();
#pragma unroll
    for (int m = 0; m < MT; m++) {
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < SS_NW; w++) v += red1[w * 32 + m * 16 + li];
      rstd[m] = 1.0f / sqrtf(v * (1.0f / (float)SS_D) + 1e-5f);
    }
  };
  float mean[MT], rstd[MT];
  stats(mean, rstd);
  if (wave == 0 && lg == 0) {
#pragma unroll
    for (int m = 0; m < MT; m++)
      if (m * 16 + li < L) { n.mean[grow0 + m * 16 + li] = mean[m]; n.rstd[grow0 + m * 16 + li] = rstd[m]; }
  }
#pragma unroll
  for (int m = 0; m < MT; m++) {
    part[m] = 0.0f;
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) {
      const float gg[4] = {gm[t].x, gm[t].y, gm[t].z, gm[t].w}, be[4] = {bt[t].x, bt[t].y, bt[t].z, bt[t].w};
      BV4 yv;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        yv.e[r] = f2bf((acc[m][t][r] - mean[m]) * rstd[m] * gg[r] + be[r]);
        acc[m][t][r] = bf2f(yv.e[r]);                  // a second norm reads the rows as STORED
        part[m] += acc[m][t][r];
      }
      *reinterpret_cast<BV4*>(YP + (m * 16 + li) * SS_PSTR + colw + t * 16) = yv;
    }
    part[m] = red4_sum(part[m]);
  }
  if (n2 != nullptr) {
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) {
      gm[t] = *reinterpret_cast<const float4*>(n2->g + colw + t * 16);
      bt[t] = *reinterpret_cast<const float4*>(n2->b + colw + t * 16);
    }
    stats(mean, rstd);
    if (wave == 0 && lg == 0) {
#pragma unroll
      for (int m = 0; m < MT; m++)
        if (m * 16 + li < L) { n2->mean[grow0 + m * 16 + li] = mean[m]; n2->rstd[grow0 + m * 16 + li] = rstd[m]; }
    }
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int t = 0; t < SS_TPW; t++) {
        const float gg[4] = {gm[t].x, gm[t].y, gm[t].z, gm[t].w}, be[4] = {bt[t].x, bt[t].y, bt[t].z, bt[t].w};
        BV4 yv;
#pragma unroll
        for (int r = 0; r < 4; r++) yv.e[r] = f2bf((acc[m][t][r] - mean[m]) * rstd[m] * gg[r] + be[r]);
        *reinterpret_cast<BV4*>(Y2P + (m * 16 + li) * SS_PSTR + colw + t * 16) = yv;
      }
  }
}

// ---- attention of ONE head by ONE wave, operands in LDS panels (the arithmetic and the dropout stream of attn_fwd_wave) ------------
// Qp / Kp / Vp point at this head's 64 columns; rows >= Lq / Lk of the panels hold finite values.  O^T tiles -> OP[row][h*64 + ...].
__device__ __forceinline__ bf16x8 ss_frag_rowk(const bf16_t* p, const int str, const int row_base, const int ks, const int lane) {
  return *reinterpret_cast<const bf16x8*>(p + (row_base + (lane & 15)) * str + ks * 32 + (lane >> 4) * 8);
}
__device__ __forceinline__ bf16x8 ss_frag_colk(const bf16_t* p, const int str, const int r0, const int r1, const int col_base, const int lane) {
  const int i = lane & 15, g = lane >> 4;
  const s16x4 lo = lds_tr16(p + (r0 + g * 4 + (i >> 2)) * str + col_base + (i & 3) * 4);
  const s16x4 hi = lds_tr16(p + (r1 + g * 4 + (i >> 2)) * str + col_base + (i & 3) * 4);
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void ss_attn_wave(const bf16_t* Qp, const int strq, const bf16_t* Kp, const bf16_t* Vp, const int strkv, const int Lq,
                                             const int Lk, const int causal, const unsigned long long padmask, const Dropout& dr, const int bh,
                                             bf16_t* OPh, const int lane, const int qt0, const int qstep) {
  const int i = lane & 15, g = lane >> 4;
  const int LQT = (Lq + 15) >> 4, LKT = (Lk + 15) >> 4;     // <= 2 each
  const float scale = 0.125f;                                // 1 / sqrt(64)
  for (int qt = qt0; qt < LQT; qt += qstep) {        // query tiles qt0, qt0 + qstep, ...: the waves that share a head split them
    f32x4 st[2];
    const int qq = qt * 16 + i;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      st[t] = f32x4{0, 0, 0, 0};
      if (t < LKT) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
          st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ss_frag_rowk(Kp, strkv, t * 16, ks, lane), ss_frag_rowk(Qp, strq, qt * 16, ks, lane),
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
#pragma unroll
    for (int t = 0; t < 2; t++)
      if (t < LKT) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int kk = t * 16 + g * 4 + r;
          st[t][r] *= inv * drop_mult(dr, (uint32_t)((bh * Lq + qq) * Lk + kk));
        }
      }
    const bf16x8 pa = pack_p(st[0], st[1]);
    const int r1 = LKT > 1 ? 16 : 0;                         // one key tile: the second half of P is zero, re-read tile 0 (finite)
#pragma unroll
    for (int dt = 0; dt < 4; dt++) {
      f32x4 ot = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ss_frag_colk(Vp, strkv, 0, r1, dt * 16, lane), pa, f32x4{0, 0, 0, 0}, 0, 0, 0);
      BV4 o;
#pragma unroll
      for (int r = 0; r < 4; r++) o.e[r] = f2bf(ot[r]);
      *reinterpret_cast<BV4*>(OPh + (qt * 16 + i) * SS_PSTR + dt * 16 + g * 4) = o;
    }
  }
}

template <class P> __device__ __forceinline__ unsigned long long ss_padmask(const P& p, const int b, const int lane) {
  if (p.key_ids != nullptr) {
    const long id = p.key_ids[(long)b * p.key_ids_bs + min(lane, p.L - 1)];
    return __ballot(lane < p.L && id == p.pad_id);
  }
  if (p.key_pad == nullptr) return 0ull;
  const int w = p.L - p.key_pad_shift;
  const int j = min(max(lane - p.key_pad_shift, 0), w - 1);
  const uint8_t v = p.key_pad[(long)b * w + j];
  return __ballot(lane >= p.key_pad_shift && lane < p.L && v != 0);
}


}  // namespace vct
