// Sample-stationary Transformer layer forward for short sequences on gfx950 (bf16 throughput mode, training or eval).
//
//     ONE launch = ONE whole nn.TransformerEncoderLayer / nn.TransformerDecoderLayer (+ optionally the stack-final LayerNorm)
//     ONE 512-thread workgroup = ONE sample: its <= 32 token rows stay in LDS from the layer input to the layer output
//
// replaces, per layer, the 9-14 launches of the unfused schedule (engine.py: _attn_ln_fwd / _ffn_fwd / _ln_fwd): in-projection GEMM(s),
// attention core, out_proj GEMM, add + LayerNorm, cross-attention likewise, linear1 (+GELU, dropout), linear2, add + LayerNorm --
// i.e. torch nn/modules/transformer.py:951-982 (encoder layer) and :1143-1199 (decoder layer) as built at MMEncoder.py:236-238 and
// CapDecoder.py:18-20, with nn.MultiheadAttention = torch nn/functional.py:6206-6640.
//
// Why this shape (tools/ss_probe.hip).  At cfg-B a layer is ~10 dependent products of 10 GFLOP: 4 us of
// MFMA work each behind 14 us of launch + cold first fetch + epilogue + drain, and no tiling removes the dependency chain.  A sample's
// rows, however, depend on no other sample: with B = 256 samples on 256 CUs every CU can run its sample through the WHOLE layer without
// ever meeting another workgroup -- no kernel boundary, no grid barrier, no flag.  What it pays instead is weight traffic: each
// workgroup streams all of the layer's bf16 weights (8.4 MB decoder / 6.3 MB encoder layer) from its XCD's L2 straight into MFMA
// operand registers, at the L2 -> CU rate of one CU.  That rate is 15 B/clk/CU when the fragments are gathered from the [out, in]
// row-major matrix (16 rows x 64 B per wave instruction) and 48 B/clk/CU when the weights are pre-packed in STREAM ORDER -- the exact
// sequence of 1-KiB wave fragments the kernel consumes, so that the eight waves of a workgroup read 64 KB contiguous per K chunk and
// the whole layer is one sequential stream (vct_ss_pack: a private second shadow of the layer weights, rewritten behind the
// optimizer).  Measured with MFMAs, epilogue stores and product barriers in the loop: 86 us per decoder layer, 58 us per encoder layer,
// 280 us for the 2 + 2 stack, against 440 us of GEMM + attention + LayerNorm launches today.  The MFMA pipe idles > 50 % (rows are padded
// 19 -> 32, and the stream is the bound): the design buys latency, not arithmetic efficiency.
//
// Work split inside the workgroup: a product out[32, N] = A[32, K] W[N, K]^T is walked in 512-column blocks; wave w owns columns
// w*64 .. w*64+63 of the block (4 MFMA column tiles x 2 row tiles), A fragments come from the LDS panel of the previous product,
// W fragments from the stream (double-buffered one 64-deep K chunk ahead, ACROSS product boundaries: the stream does not depend on
// data).  The MFMA operands are swapped (D^T = W X^T) so that a lane ends up with 4 CONSECUTIVE output columns of one row: 8-byte LDS
// stores in the epilogue.  Everything the unfused backward kernels read (qkv, o, a, LayerNorm statistics, pre-activation, dropped
// activation, ...) is first completed in an LDS panel and then copied to HBM with 16-byte row-contiguous stores by all 512 threads;
// the dropout counter streams are those of the unfused kernels, so the unfused backward runs unchanged behind this forward.
#include "vct_layer_ss_core.h"

namespace vct {

template <bool CROSS, int MT>
__global__ __launch_bounds__(SS_NT, SS_NW / 4) void layer_ss_fwd_kernel(const SsLayerP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int b = blockIdx.x, L = p.L;
  const long grow0 = (long)b * L;
  bf16_t* R0 = reinterpret_cast<bf16_t*>(smem + SS_R0);
  bf16_t* R1A = reinterpret_cast<bf16_t*>(smem + SS_R1A);
  bf16_t* R1B = reinterpret_cast<bf16_t*>(smem + SS_R1B);
  bf16_t* R1C = reinterpret_cast<bf16_t*>(smem + SS_R1C);
  bf16_t* RMp = reinterpret_cast<bf16_t*>(smem + SS_RM);
  float* red = reinterpret_cast<float*>(smem + SS_RED);
  float* b1s = reinterpret_cast<float*>(smem + SS_B1);

  // (Waves 4-7 -- the second-dispatched half -- reach every barrier 3-6 k cycles behind waves 0-3: the older SIMD partner wins the
  // issue arbitration.  s_setprio 1 for that half just swaps the roles, 264.6 vs 264.2 k cycles per decoder layer: zero-sum, as
  // MI355X_MICROARCH.md says of two waves per SIMD.  tools/ss_layer_skew.hip.)
  WStream ws;
  ws.p = p.wpk + (long)wave * SS_WSTR + lane0 * 8;
  ws.last = ws.p + (long)(p.nchunks - 1) * SS_CHUNK;
  bf16x8 b0[SS_TPW][2], b1[SS_TPW][2];
  ws_fetch(ws, b0);                                        // the stream starts before the first activation byte is here,
  ws_fetch(ws, b1);                                        // two chunks ahead: both buffers are in flight between products

  if constexpr (CROSS) global_to_panel(RMp, SS_PSTR, p.Lm, 16, p.mem, SS_D, (long)b * p.Lm, tid0);
  const unsigned long long padmask = ss_padmask(p, b, lane0);
  if (!CROSS && p.pro == 1) {
    // ---- encoder front end: u = feats W_u^T + b_u;  rows 1..T = u + PE',  row 0 = mean_t(u) + PE'[0]  (all T frames, pads included) ----
    const int T = L - 1, li = lane0 & 15, lg = lane0 >> 4;
    for (int v = tid0; v < MT * 16 * (SS_D / 8); v += SS_NT) {          // frames -> bf16 panel R1A (rows >= T zero) [+ the bf16 copy the unify dW reads]
      const int r = v / (SS_D / 8), c = (v % (SS_D / 8)) * 8;
      BV8s val;
      if (r < T) {
        const long g = ((long)b * T + r) * SS_D + c;
        if (p.feats_f32) {
          const float4 f0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.feats) + g);
          const float4 f1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.feats) + g + 4);
          val.e[0] = f2bf(f0.x); val.e[1] = f2bf(f0.y); val.e[2] = f2bf(f0.z); val.e[3] = f2bf(f0.w);
          val.e[4] = f2bf(f1.x); val.e[5] = f2bf(f1.y); val.e[6] = f2bf(f1.z); val.e[7] = f2bf(f1.w);
          if (p.x_in != nullptr) *reinterpret_cast<BV8s*>(p.x_in + g) = val;
        } else {
          val = *reinterpret_cast<const BV8s*>(reinterpret_cast<const bf16_t*>(p.feats) + g);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) val.e[j] = 0;
      }
      *reinterpret_cast<BV8s*>(R1A + r * SS_PSTR + c) = val;
    }
    float4 bu[SS_TPW];
    load_bias4(bu, p.b_u, wave * SS_CPW, lg);
    ss_barrier();
    f32x4 au[MT][SS_TPW];
    acc_zero<MT>(au);
    wave_gemm<MT>(au, R1A + li * SS_PSTR + lg * 8, SS_PSTR, 0, 8, ws, b0, b1);
    const float invT = 1.0f / (float)T;
#pragma unroll
    for (int t = 0; t < SS_TPW; t++) {
      const int col = wave * SS_CPW + t * 16 + lg * 4;
      float cs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      const float bb[4] = {bu[t].x, bu[t].y, bu[t].z, bu[t].w};
#pragma unroll
      for (int m = 0; m < MT; m++) {
        const int fr = m * 16 + li;                                     // frame index; its row in the stack input is fr + 1
        const bool valid = fr < T;
        const float4 pe4 = *reinterpret_cast<const float4*>(p.pe + (long)min(fr + 1, T) * SS_D + col);
        const float pp[4] = {pe4.x, pe4.y, pe4.z, pe4.w};
        BV4 z;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float u = bf2f(f2bf(au[m][t][r] + bb[r]));              // u as the unfused path stores it (bf16), then + PE'
          cs[r] += valid ? u : 0.0f;
          z.e[r] = valid ? f2bf(u + pp[r]) : (bf16_t)0;
        }
        if (fr + 1 < MT * 16) *reinterpret_cast<BV4*>(R0 + (fr + 1) * SS_PSTR + col) = z;
      }
#pragma unroll
      for (int r = 0; r < 4; r++) cs[r] = red16_sum(cs[r]);             // over the frames (the 16 lanes that share lg)
      if (li == 0) {
        const float4 pe0 = *reinterpret_cast<const float4*>(p.pe + col);
        BV4 z;
        z.e[0] = f2bf(cs[0] * invT + pe0.x); z.e[1] = f2bf(cs[1] * invT + pe0.y); z.e[2] = f2bf(cs[2] * invT + pe0.z); z.e[3] = f2bf(cs[3] * invT + pe0.w);
        *reinterpret_cast<BV4*>(R0 + col) = z;
      }
    }
    ss_barrier();
    panel_to_global<SS_D>(R0, SS_PSTR, L, p.x, SS_D, grow0, 0, tid0);
  } else if (CROSS && p.pro == 2) {
    // ---- token embedding: x = dropout(Emb[ids] + pos) (embed_fwd_kernel's arithmetic and counter stream) --------------------------------
    const Dropout dre = make_dropout(p.seed, p.site_emb, p.p_drop);
    for (int v = tid0; v < MT * 16 * (SS_D / 8); v += SS_NT) {
      const int r = v / (SS_D / 8), c = (v % (SS_D / 8)) * 8;
      BV8s val;
      if (r < L) {
        const long id = p.emb_ids[(long)b * p.emb_ids_bs + r];
        const float* tr = p.emb_table + id * SS_D + c;
        const float* pr = p.emb_pos + (long)r * SS_D + c;
        const float4 t0 = *reinterpret_cast<const float4*>(tr), t1 = *reinterpret_cast<const float4*>(tr + 4);
        const float4 q0 = *reinterpret_cast<const float4*>(pr), q1 = *reinterpret_cast<const float4*>(pr + 4);
        float dm[8];
        drop_mults<8>(dre, (uint32_t)(grow0 + r) * (uint32_t)SS_D + (uint32_t)c, dm);
        val.e[0] = f2bf((t0.x + q0.x) * dm[0]); val.e[1] = f2bf((t0.y + q0.y) * dm[1]); val.e[2] = f2bf((t0.z + q0.z) * dm[2]);
        val.e[3] = f2bf((t0.w + q0.w) * dm[3]); val.e[4] = f2bf((t1.x + q1.x) * dm[4]); val.e[5] = f2bf((t1.y + q1.y) * dm[5]);
        val.e[6] = f2bf((t1.z + q1.z) * dm[6]); val.e[7] = f2bf((t1.w + q1.w) * dm[7]);
        *reinterpret_cast<BV8s*>(p.x + (grow0 + r) * SS_D + c) = val;
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) val.e[j] = 0;
      }
      *reinterpret_cast<BV8s*>(R0 + r * SS_PSTR + c) = val;
    }
  } else {
    global_to_panel(R0, SS_PSTR, L, MT * 16, p.x, SS_D, grow0, tid0);
  }
  constexpr int SS_WPH = SS_NW / SS_H;                     // waves per attention head: they take the query tiles in turn
  const int head = wave / SS_WPH, qt0 = wave % SS_WPH, hd0 = head * SS_HD;
  // panels of the feed-forward phase (slot plan: DESIGN.md): input rows, the two activation buffers, f, y (= the next layer's x), y2
  bf16_t* FIN = CROSS ? R0 : R1B;
  bf16_t* HP0 = CROSS ? R1B : R0;
  bf16_t* HP1 = R1C;
  bf16_t* FP = R1A;
  bf16_t* YP = R0;
  bf16_t* Y2P = R1C;

  // the per-layer table is indexed with a loop variable: read it through the kernel-argument segment (scalar loads) -- indexing the
  // by-value argument itself makes the compiler copy all of it to scratch
  typedef const __attribute__((address_space(4))) SsLayerP* KargP;
  const KargP kp = (KargP)__builtin_amdgcn_kernarg_segment_ptr();
  for (int l = 0; l < p.nl; l++) {
    // per-lane indices, re-materialised per layer behind an opaque barrier: left visible, the compiler hoists every address
    // derived from them out of the layer loop and keeps ~80 loop-invariant registers alive through the whole body (spills)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int aoff = li * SS_PSTR + lg * 8;                // this lane's A-fragment origin inside a [32][SS_PSTR] panel
    const int ecol = wave * SS_CPW + lg * 4;                   // this lane's first epilogue column inside a 512-column block (+ t*16)
    const __attribute__((address_space(4))) SsLayerW& w = kp->lw[l];     // fields are fetched (scalar loads) where they are used
    auto nrm = [](const __attribute__((address_space(4))) SsNorm& n) { return SsNorm{n.g, n.b, n.y, n.mean, n.rstd}; };
    const bool fin = p.last && l == p.nl - 1;
    // the layer's linear1 bias -> LDS (the x panel of layers l > 0 is already in R0: the previous layer's output)
    for (int v = tid; v < (p.ff >> 2); v += SS_NT) reinterpret_cast<float4*>(b1s)[v] = reinterpret_cast<const float4*>(w.b1)[v];
    SS_STAMP(0);
    ss_barrier();
    SS_STAMP(1);

    f32x4 acc[MT][SS_TPW];
    float4 bv[SS_TPW];
    // ---- self-attention block --------------------------------------------------------------------------------------------------------
    for (int nb = 0; nb < 3; nb++) {                       // q | k | v = x W_in^T + b_in  -> panel [32][1544] in R1A..R1C
      load_bias4(bv, w.b_qkv, nb * 512 + wave * SS_CPW, lg);
      acc_zero<MT>(acc);
      wave_gemm<MT>(acc, R0 + aoff, SS_PSTR, 0, 8, ws, b0, b1);
      epi_store<MT>(acc, bv, R1A, SS_QSTR, nb * 512 + wave * SS_CPW, li, lg);
    }
    // residual rows of the out_proj epilogue: out of the x panel, which is about to become the attention output panel
    BV4 res[MT][SS_TPW];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int t = 0; t < SS_TPW; t++) res[m][t] = *reinterpret_cast<const BV4*>(R0 + (m * 16 + li) * SS_PSTR + ecol + t * 16);
    SS_STAMP(2);
    ss_barrier();
    SS_STAMP(3);
    panel_to_global<3 * SS_D>(R1A, SS_QSTR, L, w.qkv, 3 * SS_D, grow0, 0, tid);
    SS_STAMP(4);
    {
      const Dropout dr = make_dropout(p.seed, w.site_sa, p.p_drop);
      ss_attn_wave(R1A + hd0, SS_QSTR, R1A + SS_D + hd0, R1A + 2 * SS_D + hd0, SS_QSTR, L, L, p.causal, padmask, dr,
                   b * SS_H + head, R0 + hd0, lane, qt0, SS_WPH);
    }
    SS_STAMP(5);
    ss_barrier();
    SS_STAMP(6);
    panel_to_global<SS_D>(R0, SS_PSTR, L, w.o, SS_D, grow0, 0, tid);
    SS_STAMP(7);
    {                                                      // a = o W_o^T + b_o;  x1 = LN1(x + drop(a))
      load_bias4(bv, w.b_o, wave * SS_CPW, lg);
      acc_zero<MT>(acc);
      wave_gemm<MT>(acc, R0 + aoff, SS_PSTR, 0, 8, ws, b0, b1);
      SS_STAMP(8);
      const Dropout dr = make_dropout(p.seed, w.site_n1, p.p_drop);
      epi_ln<MT>(acc, bv, res, nrm(w.n1), nullptr, dr, grow0, L, R1A, R1B, nullptr, red, wave, li, lg);
      SS_STAMP(9);
    }
    ss_barrier();
    panel_to_global<SS_D>(R1A, SS_PSTR, L, w.a, SS_D, grow0, 0, tid);
    panel_to_global<SS_D>(R1B, SS_PSTR, L, w.n1.y, SS_D, grow0, 0, tid);
    SS_STAMP(10);

    if constexpr (CROSS) {
      // ---- cross-attention block -----------------------------------------------------------------------------------------------------
      const int Lm = p.Lm;
      load_bias4(bv, w.b_cq, wave * SS_CPW, lg);               // q = x1 W_q^T + b_q -> R1C
      acc_zero<MT>(acc);
      wave_gemm<MT>(acc, R1B + aoff, SS_PSTR, 0, 8, ws, b0, b1);
      epi_store<MT>(acc, bv, R1C, SS_PSTR, wave * SS_CPW, li, lg);
      SS_STAMP(11);
      for (int nb = 0; nb < 2; nb++) {                     // k | v = mem W_kv^T + b_kv -> panel [16][1032] in R0
        f32x4 acm[1][SS_TPW];
        load_bias4(bv, w.b_ckv, nb * 512 + wave * SS_CPW, lg);
        acc_zero<1>(acm);
        wave_gemm<1>(acm, RMp + aoff, SS_PSTR, 0, 8, ws, b0, b1);
        epi_store<1>(acm, bv, R0, SS_KVSTR, nb * 512 + wave * SS_CPW, li, lg);
      }
      SS_STAMP(12);
      ss_barrier();
      panel_to_global<SS_D>(R1C, SS_PSTR, L, w.cq, SS_D, grow0, 0, tid);
      panel_to_global<2 * SS_D>(R0, SS_KVSTR, Lm, w.ckv, 2 * SS_D, (long)b * Lm, 0, tid);
      SS_STAMP(13);
      {
        const Dropout dr = make_dropout(p.seed, w.site_ca, p.p_drop);
        // rows >= Lm of the 16-row k | v panel: computed from zero memory rows = the bias, finite
        ss_attn_wave(R1C + hd0, SS_PSTR, R0 + hd0, R0 + SS_D + hd0, SS_KVSTR, L, Lm, 0, 0ull, dr, b * SS_H + head,
                     R1A + hd0, lane, qt0, SS_WPH);
      }
      SS_STAMP(14);
      ss_barrier();
      panel_to_global<SS_D>(R1A, SS_PSTR, L, w.co, SS_D, grow0, 0, tid);
      SS_STAMP(15);
      {                                                    // a2 = o2 W_o^T + b_o;  x2 = LN2(x1 + drop(a2)); residual x1 still in R1B
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
          for (int t = 0; t < SS_TPW; t++) res[m][t] = *reinterpret_cast<const BV4*>(R1B + (m * 16 + li) * SS_PSTR + ecol + t * 16);
        load_bias4(bv, w.b_co, wave * SS_CPW, lg);
        acc_zero<MT>(acc);
        wave_gemm<MT>(acc, R1A + aoff, SS_PSTR, 0, 8, ws, b0, b1);
        SS_STAMP(16);
        const Dropout dr = make_dropout(p.seed, w.site_n2, p.p_drop);
        epi_ln<MT>(acc, bv, res, nrm(w.n2), nullptr, dr, grow0, L, R1C, R0, nullptr, red, wave, li, lg);
        SS_STAMP(17);
      }
      ss_barrier();
      panel_to_global<SS_D>(R1C, SS_PSTR, L, w.ca, SS_D, grow0, 0, tid);
      panel_to_global<SS_D>(R0, SS_PSTR, L, w.n2.y, SS_D, grow0, 0, tid);
      SS_STAMP(18);
    }

    // ---- feed-forward block, software-pipelined over the ff / 512 chunks ----------------------------------------------------------------
    //   linear1(0) | [linear1(j+1) GEMM with the GELU / dropout epilogue of chunk j between its K steps] | barrier | linear2(j) K slice | ...
    // The epilogue of a chunk is ~9 k cycles of vector ALU work that nothing else in the workgroup could overlap; issued between the
    // K steps of the NEXT chunk's linear1 it runs in the shadow of that product's weight stream.  The activation panel is double-
    // buffered (one barrier per chunk); the pre-activation goes to HBM straight from the registers (8-byte stores).
    f32x4 facc[MT][SS_TPW];
    BV4 hpk[MT][SS_TPW];                                          // pre-activation of the chunk whose epilogue is pending, as stored (bf16)
    acc_zero<MT>(facc);
    const Dropout drf = make_dropout(p.seed, w.site_ff, p.p_drop);
    const int nj = p.ff >> 9;
    auto ffn_pre = [&](const int j) {                        // hpre(j) = bf16(acc + b1): to HBM now, kept packed for the pipelined GELU
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int t = 0; t < SS_TPW; t++) {
          const int row = m * 16 + li, col = j * 512 + ecol + t * 16;
          const float4 bb = *reinterpret_cast<const float4*>(b1s + col);
          BV4 pv;
          pv.e[0] = f2bf(acc[m][t][0] + bb.x); pv.e[1] = f2bf(acc[m][t][1] + bb.y);
          pv.e[2] = f2bf(acc[m][t][2] + bb.z); pv.e[3] = f2bf(acc[m][t][3] + bb.w);
          hpk[m][t] = pv;
#if SS_STREAM_STORES
          if (row < L) store_stream8(w.hpre + (grow0 + row) * p.ff + col, __builtin_bit_cast(stream_u32x2, pv));
#else
          if (row < L) *reinterpret_cast<BV4*>(w.hpre + (grow0 + row) * p.ff + col) = pv;
#endif
        }
    };
    auto ffn_tile = [&](const int j, auto K) {               // GELU + dropout of ONE 16 x 16 tile of chunk j -> activation panel
      constexpr int k = decltype(K)::value;
      if constexpr (k < MT * SS_TPW) {
        constexpr int m = k / SS_TPW, t = k % SS_TPW;
        const int row = m * 16 + li;
        const int col = j * 512 + ecol + t * 16;
        float dm[4];
        drop_mults<4>(drf, (uint32_t)(grow0 + row) * (uint32_t)p.ff + (uint32_t)col, dm);
        // the activation is built on the pre-activation AS STORED (bf16): what the backward's GELU' reads
        const vf2 x0 = {bf2f(hpk[m][t].e[0]), bf2f(hpk[m][t].e[1])}, x1 = {bf2f(hpk[m][t].e[2]), bf2f(hpk[m][t].e[3])};
        const vf2 a0 = act_fast_f2(p.act, x0) * vf2{dm[0], dm[1]}, a1 = act_fast_f2(p.act, x1) * vf2{dm[2], dm[3]};
        BV4 hv;
        hv.e[0] = f2bf(a0[0]); hv.e[1] = f2bf(a0[1]); hv.e[2] = f2bf(a1[0]); hv.e[3] = f2bf(a1[1]);
        *reinterpret_cast<BV4*>(((j & 1) ? HP1 : HP0) + row * SS_PSTR + ecol + t * 16) = hv;
      }
    };
    load_bias4(bv, w.b2, wave * SS_CPW, lg);             // linear2's bias: 2 float4 that ride through the chunk loop (issued here, ahead of the stream)
    acc_zero<MT>(acc);
    wave_gemm<MT>(acc, FIN + aoff, SS_PSTR, 0, 8, ws, b0, b1);
    ffn_pre(0);
    SS_STAMP(20);
    for (int j = 0; j < nj; j++) {
      if (j + 1 < nj) {
        acc_zero<MT>(acc);
        wave_gemm8_cb<MT>(acc, FIN + aoff, SS_PSTR, ws, b0, b1, [&](auto K) { ffn_tile(j, K); });
      } else {
        static_for<MT * SS_TPW>([&](auto K) { ffn_tile(j, K); });
      }
      SS_STAMP(21 + 4 * j);
      ss_barrier();
      bf16_t* HP = (j & 1) ? HP1 : HP0;
      panel_to_global<SS_D>(HP, SS_PSTR, L, w.h, p.ff, grow0, j * 512, tid);
      SS_STAMP(22 + 4 * j);
      wave_gemm<MT>(facc, HP + aoff, SS_PSTR, 0, 8, ws, b0, b1);
      if (j + 1 < nj) ffn_pre(j + 1);
      SS_STAMP(23 + 4 * j);
    }
    {                                                      // f = h W_2^T + b_2;  y = LN(x_in + drop(f)) [; y2 = LN_final(y)]
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int t = 0; t < SS_TPW; t++) res[m][t] = *reinterpret_cast<const BV4*>(FIN + (m * 16 + li) * SS_PSTR + ecol + t * 16);
      ss_barrier();                                        // the last chunk's copy and linear2 reads are done: the activation buffers become y / y2
      const Dropout dr = make_dropout(p.seed, w.site_n3, p.p_drop);
      const SsNorm nfl = p.nf;                             // (a pointer into the by-value argument would put all of it into scratch)
      epi_ln<MT>(facc, bv, res, nrm(w.n3), fin ? &nfl : nullptr, dr, grow0, L, FP, YP, Y2P, red, wave, li, lg);
      SS_STAMP(40);
    }
    ss_barrier();
    panel_to_global<SS_D>(FP, SS_PSTR, L, w.f, SS_D, grow0, 0, tid);
    panel_to_global<SS_D>(YP, SS_PSTR, L, w.n3.y, SS_D, grow0, 0, tid);
    if (fin) panel_to_global<SS_D>(Y2P, SS_PSTR, L, p.nf.y, SS_D, grow0, 0, tid);
    SS_STAMP(41);
    // the next layer's x is YP = R0 (rows >= L: finite LayerNorm outputs of finite rows)
  }
}

// ---- stream-order packing of weight blocks ---------------------------------------------------------------------------------------------
constexpr int SS_PACK_MAX = 48;
struct SsPackSeg { const bf16_t* w; long ldw; int nchunks; int dst_chunk; int tr; };
struct SsPackP { SsPackSeg seg[SS_PACK_MAX]; int nseg; bf16_t* dst; };

// one thread = one 16-byte vector of the stream: (chunk, wave, tile t, k-step s, lane) <- A[w*64 + t*16 + (lane & 15)][chunk*64 + s*32 + (lane >> 4)*8 ..],
// A = the segment's W rows (forward products) or W^T (transposed segments)
__global__ __launch_bounds__(256) void ss_pack_kernel(const SsPackP p) {
  const int sg = blockIdx.y;
  if (sg >= p.nseg) return;
  const SsPackSeg s = p.seg[sg];
  const long v = (long)blockIdx.x * 256 + threadIdx.x;     // vector index inside the segment
  if (v >= (long)s.nchunks * (SS_CHUNK / 8)) return;
  constexpr int FPW = 2 * SS_TPW;                            // fragments per wave and chunk
  const int lane = (int)(v & 63), frag = (int)((v >> 6) % FPW), w = (int)((v >> 6) / FPW % SS_NW);
  const long c = v >> 12;
  const int t = frag >> 1, ks = frag & 1;
  const int n = w * SS_CPW + t * 16 + (lane & 15);
  const long k = c * 64 + ks * 32 + (lane >> 4) * 8;
  BV8s val;
  if (s.tr) {                                                // block of W^T: element (n, k) = w[k][n] (the dX products of the backward)
#pragma unroll
    for (int j = 0; j < 8; j++) val.e[j] = s.w[(k + j) * s.ldw + n];
  } else {
    val = *reinterpret_cast<const BV8s*>(s.w + (long)n * s.ldw + k);
  }
  *reinterpret_cast<BV8s*>(p.dst + ((long)s.dst_chunk + c) * SS_CHUNK + (v & 4095) * 8) = val;
}

}  // namespace vct
using namespace vct;

extern "C" int vct_layer_ss_supported(int dtype, int d, int H, int ff, int L, int Lm) {
  if (dtype != VCT_BF16 || d != SS_D || H != SS_H) return 0;
  if (ff < 512 || (ff % 512) != 0 || ff > SS_FF_MAX) return 0;
  if (L < 1 || L > 32) return 0;
  if (Lm < 0 || Lm > 16) return 0;        // Lm = 0: encoder layer
  return 1;
}

extern "C" int64_t vct_layer_ss_stream_chunks(int ff, int cross) {
  // q|k|v 3 blocks + out_proj 1 [+ cross q 1 + cross k|v 2 + cross out_proj 1] + ff/512 x (linear1 block + linear2 K slice), 8 chunks each
  return (int64_t)8 * (4 + (cross ? 4 : 0) + 2 * (ff / 512));
}

extern "C" int vct_ss_pack(const vct_ss_pack_seg* segs, int nseg, void* dst, void* stream) {
  if (segs == nullptr || dst == nullptr || nseg < 1) return VCT_E_ARG;
  if (((uintptr_t)dst & 15)) return VCT_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < nseg; base += SS_PACK_MAX) {
    SsPackP p;
    p.nseg = nseg - base < SS_PACK_MAX ? nseg - base : SS_PACK_MAX;
    p.dst = reinterpret_cast<bf16_t*>(dst);
    int maxch = 0;
    for (int i = 0; i < p.nseg; i++) {
      const vct_ss_pack_seg& s = segs[base + i];
      if (s.w == nullptr || s.nchunks < 1 || s.dst_chunk < 0) return VCT_E_ARG;
      if ((s.ldw % 8) || ((uintptr_t)s.w & 15)) return VCT_E_ALIGN;
      p.seg[i].w = reinterpret_cast<const bf16_t*>(s.w); p.seg[i].ldw = s.ldw; p.seg[i].nchunks = s.nchunks; p.seg[i].dst_chunk = (int)s.dst_chunk; p.seg[i].tr = s.transposed != 0;
      maxch = s.nchunks > maxch ? s.nchunks : maxch;
    }
    for (int i = p.nseg; i < SS_PACK_MAX; i++) p.seg[i] = SsPackSeg{nullptr, 0, 0, 0, 0};
    const dim3 grid((unsigned)((long)maxch * (SS_CHUNK / 8) / 256), p.nseg);
    vct::launch(ss_pack_kernel, grid, dim3(256), 0, st, p);
    VCT_CHECK_LAUNCH();
  }
  return VCT_OK;
}

extern "C" int vct_layer_ss_fwd(const vct_layer_ss_desc* layers, int n_layers, void* stream) {
  if (layers == nullptr || n_layers < 1) return VCT_E_ARG;
  const vct_layer_ss_desc* q = layers;
  const int cross = q->mem != nullptr;
  if (!vct_layer_ss_supported(q->dtype, q->d, q->H, q->ff, q->L, cross ? q->Lm : 0) || q->B < 1) return VCT_E_SHAPE;
  if (cross && q->Lm < 1) return VCT_E_SHAPE;
  const int64_t per_layer = vct_layer_ss_stream_chunks(q->ff, cross);
  const int pro = q->pro;
  if (pro < 0 || pro > 2 || (pro == 1 && cross) || (pro == 2 && !cross)) return VCT_E_ARG;
  if (pro == 1 && (!q->feats || !q->b_unify || !q->pe_rows || q->L < 2 || (q->feats_dtype != VCT_F32 && q->feats_dtype != VCT_BF16) ||
                   (((uintptr_t)q->feats | (uintptr_t)q->x_in | (uintptr_t)q->b_unify | (uintptr_t)q->pe_rows) & 15)))
    return VCT_E_ARG;
  if (pro == 2 && (!q->emb_ids || !q->emb_table || !q->emb_pos || (((uintptr_t)q->emb_table | (uintptr_t)q->emb_pos) & 15))) return VCT_E_ARG;
  const int64_t pro_chunks = pro == 1 ? 8 : 0;            // the unify weight block in front of layer 0's stream
  if (!q->x || q->key_pad_shift < 0 || (q->key_pad != nullptr && q->key_pad_shift >= q->L)) return q->x ? VCT_E_SHAPE : VCT_E_ARG;
  auto norm_ok = [](const vct_ss_norm& n) { return n.gamma && n.beta && n.y && n.mean && n.rstd; };
  auto cvt = [](const vct_ss_norm& n) { return SsNorm{n.gamma, n.beta, reinterpret_cast<bf16_t*>(n.y), n.mean, n.rstd}; };
  for (int l = 0; l < n_layers; l++) {
    const vct_layer_ss_desc& d = layers[l];
    // one stack: every layer has the shape, masks, memory and dropout of the first; the packed streams lie back to back
    if (d.dtype != q->dtype || d.B != q->B || d.L != q->L || d.Lm != q->Lm || d.d != q->d || d.H != q->H || d.ff != q->ff || d.act != q->act ||
        d.causal != q->causal || d.key_pad_shift != q->key_pad_shift || d.mem != q->mem || d.key_pad != q->key_pad || d.key_ids != q->key_ids ||
        d.key_ids_bs != q->key_ids_bs || d.pad_id != q->pad_id || d.seed != q->seed || d.p_drop != q->p_drop)
      return VCT_E_ARG;
    if (d.nchunks != per_layer) return VCT_E_SHAPE;
    if (!d.wpk || (const char*)d.wpk != (const char*)q->wpk + (size_t)(l * per_layer + (l > 0 ? pro_chunks : 0)) * SS_CHUNK * 2) return VCT_E_ARG;
    if (l > 0 && d.pro != 0) return VCT_E_ARG;
    if (d.last && l != n_layers - 1) return VCT_E_ARG;
    if (!d.b_qkv || !d.b_o || !d.qkv || !d.o || !d.a || !d.b1 || !d.b2 || !d.hpre || !d.h || !d.f) return VCT_E_ARG;
    if (!norm_ok(d.n1) || !norm_ok(d.n3) || (d.last && !norm_ok(d.nf))) return VCT_E_ARG;
    if (cross && (!d.b_cq || !d.b_ckv || !d.b_co || !d.cq || !d.ckv || !d.co || !d.ca || !norm_ok(d.n2))) return VCT_E_ARG;
    const uintptr_t al = (uintptr_t)d.wpk | (uintptr_t)d.x | (uintptr_t)d.mem | (uintptr_t)d.qkv | (uintptr_t)d.o | (uintptr_t)d.a |
                         (uintptr_t)d.cq | (uintptr_t)d.ckv | (uintptr_t)d.co | (uintptr_t)d.ca | (uintptr_t)d.hpre | (uintptr_t)d.h |
                         (uintptr_t)d.f | (uintptr_t)d.n1.y | (uintptr_t)d.n2.y | (uintptr_t)d.n3.y | (uintptr_t)d.nf.y |
                         (uintptr_t)d.b_qkv | (uintptr_t)d.b_o | (uintptr_t)d.b_cq | (uintptr_t)d.b_ckv | (uintptr_t)d.b_co |
                         (uintptr_t)d.b1 | (uintptr_t)d.b2;
    if (al & 15) return VCT_E_ALIGN;
  }
  hipStream_t st = (hipStream_t)stream;
  // rows <= 16: one 16-row MFMA tile per product (half the epilogue work, LDS reads and MFMAs of the two-tile form)
  const int one = q->L <= 16 ? 1 : 0, which = cross * 2 + one;
  static vct::DynLdsOptIn optin[4];
  const void* fn = which == 0 ? (const void*)layer_ss_fwd_kernel<false, 2> : which == 1 ? (const void*)layer_ss_fwd_kernel<false, 1>
                 : which == 2 ? (const void*)layer_ss_fwd_kernel<true, 2> : (const void*)layer_ss_fwd_kernel<true, 1>;
  if (hipError_t e = optin[which].ensure(fn, SS_LDS); e != hipSuccess) return (int)e;
  for (int base = 0; base < n_layers; base += SS_MAXL) {       // SS_MAXL layers per launch; the hand-over between launches goes through y in HBM
    const int nl = n_layers - base < SS_MAXL ? n_layers - base : SS_MAXL;
    SsLayerP p;
    memset(&p, 0, sizeof(p));
    p.B = q->B; p.L = q->L; p.Lm = cross ? q->Lm : 0; p.ff = q->ff; p.act = q->act; p.causal = q->causal; p.nl = nl;
    p.last = layers[base + nl - 1].last;
    p.wpk = reinterpret_cast<const bf16_t*>(layers[base].wpk); p.nchunks = (int)(per_layer * nl + (base == 0 ? pro_chunks : 0));
    p.x = const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(base == 0 ? q->x : layers[base - 1].n3.y));
    p.pro = base == 0 ? pro : 0;
    p.feats = q->feats; p.feats_f32 = q->feats_dtype == VCT_F32; p.x_in = reinterpret_cast<bf16_t*>(q->x_in); p.b_u = q->b_unify; p.pe = q->pe_rows;
    p.emb_ids = q->emb_ids; p.emb_ids_bs = q->emb_ids_bs; p.emb_table = q->emb_table; p.emb_pos = q->emb_pos; p.site_emb = q->site_emb;
    p.mem = reinterpret_cast<const bf16_t*>(q->mem);
    p.nf = cvt(layers[base + nl - 1].nf);
    p.key_pad = q->key_pad; p.key_pad_shift = q->key_pad_shift; p.key_ids = q->key_ids; p.key_ids_bs = q->key_ids_bs; p.pad_id = q->pad_id;
    p.seed = q->seed; p.p_drop = q->p_drop;
    for (int l = 0; l < nl; l++) {
      const vct_layer_ss_desc& d = layers[base + l];
      SsLayerW& w = p.lw[l];
      w.b_qkv = d.b_qkv; w.b_o = d.b_o;
      w.qkv = reinterpret_cast<bf16_t*>(d.qkv); w.o = reinterpret_cast<bf16_t*>(d.o); w.a = reinterpret_cast<bf16_t*>(d.a);
      w.n1 = cvt(d.n1); w.n2 = cvt(d.n2); w.n3 = cvt(d.n3);
      w.b_cq = d.b_cq; w.b_ckv = d.b_ckv; w.b_co = d.b_co;
      w.cq = reinterpret_cast<bf16_t*>(d.cq); w.ckv = reinterpret_cast<bf16_t*>(d.ckv); w.co = reinterpret_cast<bf16_t*>(d.co);
      w.ca = reinterpret_cast<bf16_t*>(d.ca);
      w.b1 = d.b1; w.b2 = d.b2;
      w.hpre = reinterpret_cast<bf16_t*>(d.hpre); w.h = reinterpret_cast<bf16_t*>(d.h); w.f = reinterpret_cast<bf16_t*>(d.f);
      w.site_sa = d.site_sa; w.site_n1 = d.site_n1; w.site_ca = d.site_ca; w.site_n2 = d.site_n2; w.site_ff = d.site_ff; w.site_n3 = d.site_n3;
    }
#ifdef SS_STAMPS
    p.dbg = g_ss_dbg;
#endif
    if (which == 0) vct::launch(layer_ss_fwd_kernel<false, 2>, dim3(p.B), dim3(SS_NT), SS_LDS, st, p);
    else if (which == 1) vct::launch(layer_ss_fwd_kernel<false, 1>, dim3(p.B), dim3(SS_NT), SS_LDS, st, p);
    else if (which == 2) vct::launch(layer_ss_fwd_kernel<true, 2>, dim3(p.B), dim3(SS_NT), SS_LDS, st, p);
    else vct::launch(layer_ss_fwd_kernel<true, 1>, dim3(p.B), dim3(SS_NT), SS_LDS, st, p);
    VCT_CHECK_LAUNCH();
  }
  return VCT_OK;
}
