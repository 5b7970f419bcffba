// Persistent-tile bf16 GEMM for the LAYER products of the caption model on gfx950:  C[M,N] = epilogue(op(A)[M,K] op(B)[K,N]),
// forms NT (y = x W^T: every nn.Linear forward of the encoder / decoder layers, torch nn/modules/transformer.py:951-982,
// 1143-1199 as built at model/MMEncoder.py:236-238 and model/CapDecoder.py:18-20) and NN (dX = dY W: their autograd backward).
//
// Why.  These products are mid-size -- M = 3328 / 4864 token rows, N = 512 .. 2048, K = 512 .. 2048 -- and K = 512 means eight
// 64-deep K stages per output tile: with one tile per workgroup (vct_gemm_bf16_kernel.h) fill, drain and epilogue of every tile
// sit in the open and the L2 -> LDS stream of a CU runs at 17-26 B/clk where tools/l2_stream_probe.hip measures 54 B/clk for
// an L2-resident stream.  This kernel is the pipeline of the vocabulary kernel (vct_gemm256.hip: persistent workgroups walking
// a flat list of (tile, K stage) steps, ONE barrier per stage, the next stage's 1-KiB DMA instructions issued one at a time
// between the MFMA groups, the next tile's first stage in flight under the current tile's epilogue, accumulators transposed so
// that the epilogue moves whole output rows) at layer granularity: tiles of (32 TM) x (64 TN) -- 128 x 128 for wide outputs,
// 64 x 128 for the N = 512 ones -- 64 / 48 KB of LDS, TWO workgroups per CU, and the fused epilogues of the general kernel:
// bias, activation (+ saved pre-activation), counter-hash dropout, activation derivative, residual-gradient accumulate.
// Measured alone (tools/gemm_pt_bench.py, recorded replays, bias epilogue): decoder QKV 20.1 -> 14.9 us, FFN1 23.4 -> 19.8,
// encoder FFN1 19.3 -> 14.7, the batch-1024 FFN1 88 -> 65 us (625 TFLOP/s).
#include "vct_gemm_bf16_kernel.h"

namespace vct {

int persistent_grid(hipStream_t st);          // vct_gemm256.hip: compute units the stream may use, a multiple of 8

constexpr int pt_ipr(int tm, int cap) { int best = 1; for (int v = 1; v <= tm; v++) if (tm % v == 0 && v <= cap) best = v; return best; }

struct alignas(16) PV8 { bf16_t e[8]; };

// EPI = false: bias only, the row slab holds bf16 (one LDS pass per 128 rows); EPI = true: the slab holds fp32 and the store
// phase applies the full epilogue on 8 consecutive columns per thread.
template <int TA, int TB, int TM, int TN, bool EPI>
__global__ __launch_bounds__(512, 4) void gemm_pt_kernel(const GemmP p) {
  constexpr bool A_MC = (TA == 1), B_MC = (TB == 0), KSPLIT = A_MC && B_MC;
  constexpr int NW = 8, NT = 512;
  constexpr int BM = 32 * TM, BN = 64 * TN, WM = BM / 2, WNC = BN / 4;   // 2 x 4 waves, wave tile WM x WNC
  constexpr int STAGE = (BM + BN) * 128;                            // bytes per 64-deep K stage
  constexpr int NP = BM / 64 + BN / 64;                             // 1-KiB DMA instructions per wave and stage
  static_assert(NP <= 2 * TM, "more DMA pieces than issue slots");
  constexpr int ES = EPI ? 4 : 2;                                   // slab element size
  constexpr int IPR = pt_ipr(TM, STAGE / (BN * ES) / 32);           // MFMA tile rows per wave and slab round
  constexpr int RPR = IPR * 32;                                     // slab rows per round
  constexpr int CPRW = BN * ES / 16;                                // 16-byte chunks per slab row
  constexpr int SMASK = (CPRW < 32 ? CPRW : 32) - 1;                // chunk c of slab row r sits at c ^ (r & SMASK)
  constexpr int OPR = BN / 8;                                       // 8-column output vectors per row
  constexpr int OPT = RPR * OPR / NT;                               // ... per thread and round
  static_assert(RPR * OPR % NT == 0 && OPT >= 1, "slab does not divide over the workgroup");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int c16 = lane & 15, g4 = (lane >> 4) * 4;

  // ---- work items = output tiles; XCD x owns the contiguous run [x * per, (x + 1) * per), walked in groups of 8 tile columns with
  // N fastest inside a group (the group's weight panels stay in the XCD's L2 while M is swept); workgroup `slot` of the XCD takes
  // tiles slot, slot + nxw, ... of the run.  (Measured and dropped: PULLING the tiles from per-XCD device counters so that a
  // workgroup that becomes resident late does not hold a fixed share -- one returning atomic per tile costs every launch 3-4 us
  // even when issued a K stage ahead of its use, 19.0 vs 14.9 us for the decoder's QKV projection, and did not help the products
  // that run beside another stream's kernels; with work stealing between the XCDs' runs it was 40 us.) ----
  const int nitems = p.tiles_m * p.tiles_n;
  const int nxw = (int)gridDim.x >> 3;
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int w_begin = xcd * per, w_end = min(nitems, w_begin + per);
  const int nkt = (p.K + BK2 - 1) / BK2, kt_full = p.K / BK2;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

  auto item = [&](int w, int& m0, int& n0) {
    const int per_group = 8 * p.tiles_m;
    const int grp = w / per_group, rem = w - grp * per_group;
    const int gw = min(8, p.tiles_n - grp * 8);
    m0 = (rem / gw) * BM; n0 = (grp * 8 + rem % gw) * BN;
  };
  auto issue = [&](int m0, int n0, int kt, int buf) {
    unsigned char* nb = lds + buf * STAGE;
    if (kt < kt_full) {
      dma_tile<A_MC, BM, NW>(nb, A, p.lda, m0, p.M, kt * BK2, wave, lane);
      dma_tile<B_MC, BN, NW>(nb + BM * 128, B, p.ldb, n0, p.N, kt * BK2, wave, lane);
    } else {      // ragged last stage: zero-filling register path into the same swizzled images
      tail_tile<A_MC, BM, NT>(nb, A, p.lda, m0, p.M, kt * BK2, p.K, tid);
      tail_tile<B_MC, BN, NT>(nb + BM * 128, B, p.ldb, n0, p.N, kt * BK2, p.K, tid);
    }
  };

  // accumulators hold C TRANSPOSED per MFMA tile (operands swapped): lane = row i*16 + (lane & 15) of the wave's piece and FOUR
  // CONSECUTIVE columns j*16 + (lane >> 4)*4 + r
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};

  const Dropout dr = make_dropout(p.seed, p.site, p.p_drop);
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
  bf16_t* preact = reinterpret_cast<bf16_t*>(p.preact);
  const bf16_t* addend = reinterpret_cast<const bf16_t*>(p.addend);
  const bf16_t* dact = reinterpret_cast<const bf16_t*>(p.dact);

  // Loop control kept OUT of the K stages: cycle stamps of single waves (a development build) put a stage of the first version -- a
  // flat list of (tile, K stage) steps walked by two cursors -- at ~2100 cycles, of which the 16 MFMAs are 256: 170-260 cycles of
  // scalar bookkeeping in front of the fragment reads, 290-460 behind the MFMAs (cursor advance, 64-bit stage pointers, buffer
  // index), 90-600 at the barrier.  Now: nested loops, the next tile's coordinates and the lanes' DMA offsets once per tile,
  // running stage pointers, and a branch-free stage body (DMA = true: the next stage always exists and is a full one; the ragged
  // last K stage and the end of the work list take the DMA = false body): the hot loop is ONE basic block of 55 instructions per
  // stage.  Measured: the same speed (batch-1024 linear1 60.2 -> 59.1 us, QKV 13.5 -> 15.0 / 12.2 -> 12.3) -- a stage still takes
  // ~3400 cycles for two co-resident workgroups against 1024 of MFMA issue, ~800 of LDS traffic and ~1200 of L2 -> LDS stream: the
  // three do not overlap because every wave of a workgroup reads its fragments right behind the barrier and issues its MFMAs
  // right behind the reads (PMC: MFMA pipe 30 % busy, LDS 27 %, waves waiting 45 % of their cycles).  s_setprio around the MFMA
  // groups: no change.  (Three LDS stages with loads two steps ahead, measured on the first version, bought nothing either.)
  int w = w_begin + slot;
  if (w >= w_end) return;
  int m0 = 0, n0 = 0;
  item(w, m0, n0);
  uint32_t poff[NP];                                      // per-lane source offsets of the NP DMA pieces of a tile (dma_piece_offset)
  auto tile_offsets = [&](int tm0, int tn0) {
#pragma unroll
    for (int q = 0; q < NP; q++)
      poff[q] = q < BM / 64 ? dma_piece_offset<A_MC, BM, NW>(p.lda, tm0, p.M, wave, lane, q)
                            : dma_piece_offset<B_MC, BN, NW>(p.ldb, tn0, p.N, wave, lane, q - BM / 64);
  };
  tile_offsets(m0, n0);
  issue(m0, n0, 0, 0);
  const long a_step = A_MC ? (long)BK2 * p.lda : (long)BK2, b_step = B_MC ? (long)BK2 * p.ldb : (long)BK2;   // elements per K stage
  // one K stage: fragments of stage `buf`, MFMAs, and (DMA) the four 1-KiB pieces of the stage at a_k / b_k into the other buffer,
  // one after each row tile's MFMAs of the first k-step
  auto stage = [&](auto DMA_, const int buf, const bf16_t* a_k, const bf16_t* b_k) {
    constexpr bool DMA = decltype(DMA_)::value;
    unsigned char* nb = lds + (buf ^ 1) * STAGE;
    const unsigned char* la = lds + buf * STAGE;
    const unsigned char* lb = la + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) fa[i] = frag2<A_MC, BM, KSPLIT>(la, wm * WM + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < TN; j++) fb[j] = frag2<B_MC, BN, KSPLIT>(lb, wn * WNC + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        const int q = ks * TM + i;                         // DMA issue slot
        if constexpr (DMA) {
          if (q < NP) {
            if (q < BM / 64) dma_piece_at<NW>(nb, a_k, poff[q], wave, q);
            else dma_piece_at<NW>(nb + BM * 128, b_k, poff[q], wave, q - BM / 64);
          }
        }
      }
    }
  };
  auto stage_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of the current stage has landed
    __builtin_amdgcn_s_barrier();                          // ... everyone's has; everyone is done with the other buffer
    asm volatile("" ::: "memory");
  };
  int buf = 0;
  for (; w < w_end; w += nxw) {
    int m1 = 0, n1 = 0;
    const bool have_next = w + nxw < w_end;
    if (have_next) item(w + nxw, m1, n1);
    const bf16_t* a_k = A + a_step;                        // stage 1 of this tile
    const bf16_t* b_k = B + b_step;
    int kt = 0;
    const int kt_hot = min(nkt, kt_full) - 1;
    for (; kt < kt_hot; kt++) {                            // hot loop, ONE body: the next stage is this tile's and a full one
      stage_sync();
      stage(std::true_type{}, buf, a_k, b_k);
      a_k += a_step; b_k += b_step;
      buf ^= 1;
    }
    for (; kt + 1 < nkt; kt++) {                           // (at most once) the next stage is the ragged last one: register path, up front
      stage_sync();
      issue(m0, n0, kt + 1, buf ^ 1);
      stage(std::false_type{}, buf, a_k, b_k);
      buf ^= 1;
    }
    // the tile's last stage: the NEXT tile's first stage flies under it and under the epilogue
    stage_sync();
    if (have_next) {
      tile_offsets(m1, n1);
      if (kt_full > 0) stage(std::true_type{}, buf, A, B);
      else { issue(m1, n1, 0, buf ^ 1); stage(std::false_type{}, buf, A, B); }
    } else {
      stage(std::false_type{}, buf, A, B);
    }
    unsigned char* slab = lds + buf * STAGE;               // the stage just consumed is free until the next tile's second stage is issued
    buf ^= 1;
    // ---- epilogue ----
    float bj[TN][4];
    if constexpr (!EPI) {
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int col = n0 + wn * WNC + j * 16 + g4;
#pragma unroll
        for (int r = 0; r < 4; r++) bj[j][r] = p.bias != nullptr ? p.bias[min(col + r, p.N - 1)] : 0.0f;
      }
    }
    lds_barrier();                                           // every wave has finished its fragment reads of this stage
    static_for<TM / IPR>([&](auto RD) {
      constexpr int rd = decltype(RD)::value;
      if constexpr (rd > 0) lds_barrier();                   // the slab has been read out by everyone
#pragma unroll
      for (int ii = 0; ii < IPR; ii++) {
        const int sr = (wm * IPR + ii) * 16 + c16;           // slab row
#pragma unroll
        for (int j = 0; j < TN; j++) {
          const int e0 = wn * WNC + j * 16 + g4;             // first of this lane's four consecutive columns
          if constexpr (!EPI) {
            struct alignas(8) B4 { bf16_t e[4]; } v;
#pragma unroll
            for (int r = 0; r < 4; r++) v.e[r] = f2bf(acc[rd * IPR + ii][j][r] + bj[j][r]);
            const int g8 = e0 >> 2;                          // 8-byte granule
            *reinterpret_cast<B4*>(slab + sr * (BN * 2) + ((((g8 >> 1) ^ (sr & SMASK)) << 1) | (g8 & 1)) * 8) = v;
          } else {
            *reinterpret_cast<f32x4*>(slab + sr * (BN * 4) + (((e0 >> 2) ^ (sr & SMASK)) << 4)) = acc[rd * IPR + ii][j];
          }
        }
      }
      lds_barrier();
#pragma unroll
      for (int q = 0; q < OPT; q++) {
        const int oid = q * NT + tid;
        const int sr = oid / OPR, o = oid % OPR;             // slab row, 8-column vector
        const int row = m0 + (sr / (IPR * 16)) * WM + (rd * IPR + ((sr >> 4) % IPR)) * 16 + (sr & 15);
        const int col = n0 + o * 8;
        if constexpr (!EPI) {
          typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
          const u32x4 v = *reinterpret_cast<const u32x4*>(slab + sr * (BN * 2) + ((o ^ (sr & SMASK)) << 4));
          if (row < p.M && col < p.N) {
            bf16_t* dst = C + (size_t)row * p.ldc + col;
            if (col + 8 <= p.N) *reinterpret_cast<u32x4*>(dst) = v;
            else {
              const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
              for (int qq = 0; qq < 8; qq++) if (col + qq < p.N) dst[qq] = e[qq];
            }
          }
        } else {
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(slab + sr * (BN * 4) + (((2 * o) ^ (sr & SMASK)) << 4));
          const f32x4 t1 = *reinterpret_cast<const f32x4*>(slab + sr * (BN * 4) + (((2 * o + 1) ^ (sr & SMASK)) << 4));
          if (row < p.M && col < p.N) {
            const float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
            if (col + 8 <= p.N) {
              float bv[8];
#pragma unroll
              for (int qq = 0; qq < 8; qq++) bv[qq] = 0.0f;
              if (p.bias != nullptr) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col), b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
                bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
              }
              PV8 dv, av, ov, pv;
              const bool has_d = dact != nullptr, has_a = addend != nullptr;
              if (has_d) dv = *reinterpret_cast<const PV8*>(dact + (size_t)row * p.ld_dact + col);
              if (has_a) av = *reinterpret_cast<const PV8*>(addend + (size_t)row * p.ld_addend + col);
#pragma unroll
              for (int qq = 0; qq < 8; qq += 2) {            // pairs: packed fp32 arithmetic in the activation (vct_common.h)
                vf2 x = {v[qq] + bv[qq], v[qq + 1] + bv[qq + 1]};
                pv.e[qq] = f2bf(x[0]); pv.e[qq + 1] = f2bf(x[1]);
                x = act_fast_f2(p.act, x);
                if (has_d) x *= dact_fast_f2(p.dact_kind, vf2{bf2f(dv.e[qq]), bf2f(dv.e[qq + 1])});
                float dm2[2];
                drop_mults<2>(dr, (uint32_t)row * (uint32_t)p.N + (uint32_t)(col + qq), dm2);
                x *= vf2{dm2[0], dm2[1]};
                if (has_a) x += vf2{bf2f(av.e[qq]), bf2f(av.e[qq + 1])};
                ov.e[qq] = f2bf(x[0]); ov.e[qq + 1] = f2bf(x[1]);
              }
              *reinterpret_cast<PV8*>(C + (size_t)row * p.ldc + col) = ov;
              if (preact != nullptr) *reinterpret_cast<PV8*>(preact + (size_t)row * p.ld_preact + col) = pv;
            } else {
              for (int qq = 0; qq < 8; qq++) {
                if (col + qq >= p.N) break;
                float x = v[qq] + (p.bias != nullptr ? p.bias[col + qq] : 0.0f);
                if (preact != nullptr) preact[(size_t)row * p.ld_preact + col + qq] = f2bf(x);
                x = act_fast_f(p.act, x);
                if (dact != nullptr) x *= dact_fast_f(p.dact_kind, bf2f(dact[(size_t)row * p.ld_dact + col + qq]));
                x *= drop_mult(dr, (uint32_t)row * (uint32_t)p.N + (uint32_t)(col + qq));
                if (addend != nullptr) x += bf2f(addend[(size_t)row * p.ld_addend + col + qq]);
                C[(size_t)row * p.ldc + col + qq] = f2bf(x);
              }
            }
          }
        }
      }
    });
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};
    m0 = m1; n0 = n1;
    // the next stage's barrier (after its vmcnt wait) orders the last slab reads before the DMA that reuses this stage
  }
}

template <int TA, int TB, int TM, int TN, bool EPI> static int gpt_launch(const GemmP& p, hipStream_t st) {
  constexpr int LDSB = 2 * (32 * TM + 64 * TN) * 128;
  static vct::DynLdsOptIn optin;
  if (hipError_t e = optin.ensure((const void*)gemm_pt_kernel<TA, TB, TM, TN, EPI>, LDSB); e != hipSuccess) return (int)e;
  constexpr int per_cu = LDSB <= 80 * 1024 ? 2 : 1;
  vct::launch(gemm_pt_kernel<TA, TB, TM, TN, EPI>, dim3(per_cu * persistent_grid(st)), dim3(512), (size_t)LDSB, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

template <int TA, int TB> static int gpt_dispatch(const GemmP& p, int tile, bool epi, hipStream_t st) {
  if (tile == 0) return epi ? gpt_launch<TA, TB, 4, 2, true>(p, st) : gpt_launch<TA, TB, 4, 2, false>(p, st);     // 128 x 128
  return epi ? gpt_launch<TA, TB, 2, 2, true>(p, st) : gpt_launch<TA, TB, 2, 2, false>(p, st);                    // 64 x 128
}

// Eligibility + launch (called from vct_gemm after the skinny and the vocabulary kernels).  VCT_GEMM_PT: bit 0 = NT products,
// bit 1 = NN products, 0 = off; default 1.  Same-box A/B in the training step (bench.py, twice each): off 2.381 / 2.391 ms, NT
// 2.340 / 2.354 (forward bracket 0.586 -> 0.557 ms), NT + NN 2.366 / 2.369, NN alone 2.410 -- the dX through linear2 runs beside
// the second stream's grouped weight-gradient GEMMs, where two 512-thread workgroups with 64 KB of LDS each per CU co-schedule
// worse than the small tiles.  VCT_GEMM_PT_TILE: 0 = by shape, 1 = 128 x 128 always, 2 = 64 x 128 always.
int gemm_pt_try(const vct_gemm_desc* d, hipStream_t st, bool* used) {
  *used = false;
  static const char* env = getenv("VCT_GEMM_PT");
  int mask = env != nullptr ? atoi(env) : 1;
  // desc.reserved: 100 forces this kernel wherever it is eligible, 99 forbids it (tests / A-B runs inside one process); any other
  // non-zero value is a tile override of the general kernel
  if (d->reserved == 100) mask = 3;
  else if (d->reserved != 0) return VCT_OK;
  if (mask == 0 || d->dtype != VCT_BF16 || d->out_dtype != VCT_BF16) return VCT_OK;
  if (d->bias_grad != nullptr || d->split_k > 1) return VCT_OK;
  const int form = d->ta * 2 + d->tb;                        // 1 NT, 0 NN
  if (!((form == 1 && (mask & 1)) || (form == 0 && (mask & 2)))) return VCT_OK;
  if (d->M < 1024 || d->N < 256 || d->K < 128 || d->K > 4096) return VCT_OK;
  // by itself the kernel only takes the WIDE outputs (N >= 1024: QKV / cross K-V / linear1 forward, the dX through linear2):
  // for the N = 512 products (152 tiles of 128 x 128, or 304 of 64 x 128 with 32-64 K stages each) the small-tile kernel is
  // faster alone (22.8 vs 27-33 us for linear2) and co-schedules better beside the second stream's weight-gradient GEMMs
  if (d->reserved != 100 && d->N < 1024) return VCT_OK;
  if ((d->ldc % 8) || ((uintptr_t)d->C & 15)) return VCT_OK;
  {   // the DMA addresses are (wave-uniform base) + (32-bit per-lane byte offset)
    const int64_t a_rows = d->ta ? d->K : d->M, b_rows = d->tb ? d->N : d->K;
    if ((a_rows + 64) * d->lda * 2 >= (int64_t)1 << 32 || (b_rows + 64) * d->ldb * 2 >= (int64_t)1 << 32) return VCT_OK;
  }
  if (d->preact && ((d->ld_preact % 8) || ((uintptr_t)d->preact & 15))) return VCT_OK;
  if (d->addend && ((d->ld_addend % 8) || ((uintptr_t)d->addend & 15))) return VCT_OK;
  if (d->dact_src && ((d->ld_dact % 8) || ((uintptr_t)d->dact_src & 15))) return VCT_OK;
  GemmP p;
  p.A = d->A; p.B = d->B; p.C = d->C;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.kt_per_split = (d->K + 63) / 64;
  p.act = d->dact_src != nullptr ? VCT_ACT_NONE : d->act;
  p.dact_kind = d->dact_src != nullptr ? d->act : VCT_ACT_NONE;
  p.bias = d->bias;
  p.preact = d->preact; p.ld_preact = d->ld_preact;
  p.addend = d->addend; p.ld_addend = d->ld_addend;
  p.dact = d->dact_src; p.ld_dact = d->ld_dact;
  p.seed = d->seed; p.site = d->site; p.p_drop = d->p_drop;
  p.bias_grad = nullptr; p.partial = nullptr; p.bias_partial = nullptr; p.counters = nullptr;
  p.waves8 = 0; p.split = 1; p.nt_store = 0; p.nt_preact = 0; p.short_fast = 0;
  static const char* tenv = getenv("VCT_GEMM_PT_TILE");
  const int tsel = tenv != nullptr ? atoi(tenv) : 0;
  const int tile = tsel == 1 ? 0 : (tsel == 2 ? 1 : (d->N >= 1024 ? 0 : 1));
  const int bm = tile == 0 ? 128 : 64;
  p.tiles_m = (d->M + bm - 1) / bm; p.tiles_n = (d->N + 127) / 128;
  const bool epi = d->act != VCT_ACT_NONE || d->preact || d->addend || d->dact_src || (d->seed && d->p_drop > 0.0f);
  const int rc = form == 1 ? gpt_dispatch<0, 1>(p, tile, epi, st) : gpt_dispatch<0, 0>(p, tile, epi, st);
  if (rc == VCT_OK) *used = true;
  return rc;
}

}  // namespace vct
