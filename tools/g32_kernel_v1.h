// Persistent 256x256-tile bf16 GEMM, second generation (round 6): the K loop of gemm256_kernel (vct_gemm256.hip) software-pipelined at
// k-step granularity, on either MFMA opcode.
//
// Why.  Ablation of gemm256_kernel (round 3): MFMAs + fragment reads ALONE take 116 us on the vocabulary projection whose MFMA work is
// 61 us at peak -- after every stage barrier all eight waves first read fragments (nothing to issue on the matrix pipe), then all of them
// have MFMAs, then all wait at the next barrier.  Here:
//   * fragment registers are double-buffered per k-step: the reads of k-step j+1 are issued between the MFMAs of k-step j;
//   * the stage barrier sits IN FRONT OF THE LAST k-step of a stage: by then every fragment of the stage is in registers (the buffer is
//     free for the DMA of stage s+2) and stage s+1 has landed, so its first fragments are read under the last k-step's MFMAs and the
//     matrix pipe has work on both sides of the barrier;
//   * MF = 32: v_mfma_f32_32x32x16_bf16 (a wave's 128 x 64 piece = 4 x 2 tiles, 4 k-steps of 16 per stage, 8 MFMAs each -- half the MFMA
//     instructions of the 16x16x32 form for the same LDS bytes, 8 issue slots per MFMA instead of 4 for the reads / DMA / address VALU);
//     MF = 16: v_mfma_f32_16x16x32_bf16 in the same structure (2 k-steps of 32, 32 MFMAs each);
//   * operand DMA = buffer_load_dwordx4 ... lds: per-lane byte offset (row x leading dimension, swizzled chunk) fixed per tile in 8
//     VGPRs, the K offset in an SGPR -- no address arithmetic in the K loop;
//   * K-contiguous LDS image [rows][64] bf16, 16-byte chunk c of row r at c ^ f(r) with f = (r >> 1) & 7 for the 32-row fragments
//     (lanes 0-31 = 32 consecutive rows, one chunk: conflict-free over ds_read_b128's 16-lane groups), r & 7 for the 16-row ones.
#pragma once
#include "../video-captioning-transformer_amd/csrc/vct_gemm_bf16_kernel.h"

namespace vct {


template <int MF> struct MfmaShapeV1;
template <> struct MfmaShapeV1<16> { using acc_t = f32x4;  static constexpr int T = 16, NSTEP = 2; };
template <> struct MfmaShapeV1<32> { using acc_t = f32x16; static constexpr int T = 32, NSTEP = 4; };

template <int MF> __device__ __forceinline__ int kswz_v1(int row) { return MF == 32 ? ((row >> 1) & 7) : (row & 7); }

template <int MF> __device__ __forceinline__ typename MfmaShapeV1<MF>::acc_t mfma_bf16_v1(const bf16x8 a, const bf16x8 b, const typename MfmaShapeV1<MF>::acc_t c) {
  if constexpr (MF == 32) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// The caller's G256P (vct_gemm256.hip, which includes this file) is reused: same work-item order, same output conventions.

template <int TA, int TB, typename TO, int MF, int VAR>
__global__ __launch_bounds__(512, 2) void g32v1_kernel(const G256P p) {
  static_assert(TA == 0 && TB == 1, "K-contiguous operands (NT) only");
  using S = MfmaShapeV1<MF>;
  using acc_t = typename S::acc_t;
  constexpr int T = S::T, NSTEP = S::NSTEP, TM = 128 / T, TN = 64 / T;
  constexpr int NT = 512, WM = 128;
  constexpr int ES = (int)sizeof(TO);
  constexpr int STAGE = G256_STAGE, A_BYTES = G256_BM * 128;
  constexpr int RPR = STAGE / (G256_BN * ES);                      // slab rows per round: 128 (bf16) / 64 (fp32)
  constexpr int CPRW = G256_BN * ES / 16;                          // 16-byte chunks per slab row
  constexpr int CPT = RPR * CPRW / NT;                             // chunks per thread and round
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  if constexpr (VAR & 1) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }

  const int nitems = p.tiles_m * p.tiles_n * p.split;
  const int nxw = (int)gridDim.x >> 3;
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int w_begin = xcd * per, w_end = min(nitems, w_begin + per);
  const int nkt = p.K / BK2;                                       // (K % 64 == 0: checked by the launcher)

  auto item = [&](int w, int& m0, int& n0, int& z, int& k_lo, int& k_hi) {
    int tile;
    if (p.zmajor) { const int nt = p.tiles_m * p.tiles_n; z = w / nt; tile = w - z * nt; }
    else { tile = w / p.split; z = w - tile * p.split; }
    if (p.order == 0) {
      m0 = (tile % p.tiles_m) * G256_BM; n0 = (tile / p.tiles_m) * G256_BN;
    } else {
      const int per_group = 8 * p.tiles_m;
      const int grp = tile / per_group, rem = tile - grp * per_group;
      const int gw = min(8, p.tiles_n - grp * 8);
      m0 = (rem / gw) * G256_BM; n0 = (grp * 8 + rem % gw) * G256_BN;
    }
    k_lo = z * p.kt_per_split; k_hi = min(nkt, k_lo + p.kt_per_split);
  };

  // ---- operand DMA: 4 + 4 one-KiB pieces per wave and stage; voff = byte offset of the lane's 16 bytes at K offset 0 ----
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7fffffff, 0x00020000);
  int voff[8];
  const int lda32 = (int)p.lda, ldb32 = (int)p.ldb;               // (operands < 2 GiB: checked by the launcher)
  auto set_voff = [&](int m0, int n0) {
    int l = lane;
    asm volatile("" : "+v"(l));                                    // opaque: nothing of this is hoisted out of the item loop and kept alive
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int ci = (q & 3) * 8 + wave;
      const int row = ci * 8 + (l >> 3);
      const int c = (l & 7) ^ kswz_v1<MF>(row);
      if (q < 4) voff[q] = (min(m0 + row, p.M - 1) * lda32 + c * 8) * 2;
      else voff[q] = (min(n0 + row, p.N - 1) * ldb32 + c * 8) * 2;
    }
  };
  auto dma = [&](auto Q, unsigned char* stage_buf, int kt) {
    constexpr int q = decltype(Q)::value;
    const int ci = (q & 3) * 8 + wave;
    unsigned char* dst = stage_buf + (q >= 4 ? A_BYTES : 0) + ci * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(q >= 4 ? rsB : rsA, (__attribute__((address_space(3))) void*)dst, 16, voff[q], kt * 128, 0, 0);
  };

  // ---- fragments: per-lane LDS byte offsets at k-step 0 (the k-step flips bits 5.. of the chunk index, tiles add immediates) ----
  const int rl = (MF == 32) ? (lane & 31) : (lane & 15);
  const int hl = (MF == 32) ? (lane >> 5) : (lane >> 4);
  const int rowA = wm * WM + rl, rowB = wn * 64 + rl;
  const int offA = rowA * 128 + ((hl ^ kswz_v1<MF>(rowA)) << 4);
  const int offB = A_BYTES + rowB * 128 + ((hl ^ kswz_v1<MF>(rowB)) << 4);
  constexpr int SH = (MF == 32) ? 5 : 6;                           // chunk = step * (2 | 4) + hl
  bf16x8 fa[2][TM], fb[2][TN];
  auto read_frags = [&](auto SET, const unsigned char* sb, int step) {
    constexpr int st = decltype(SET)::value;
    const unsigned char* pa = sb + (offA ^ (step << SH));
    const unsigned char* pb = sb + (offB ^ (step << SH));
#pragma unroll
    for (int j = 0; j < TN; j++) fb[st][j] = *reinterpret_cast<const bf16x8*>(pb + j * T * 128);
#pragma unroll
    for (int i = 0; i < TM; i++) fa[st][i] = *reinterpret_cast<const bf16x8*>(pa + i * T * 128);
  };
  acc_t acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < T * T / 64; r++) acc[i][j][r] = 0.0f;
  };
  // accumulators hold C TRANSPOSED per MFMA tile (operands swapped): a lane owns consecutive COLUMNS of one row
  auto mfma_step = [&](auto SET) {
    constexpr int st = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) acc[i][j] = mfma_bf16_v1<MF>(fb[st][j], fa[st][i], acc[i][j]);
  };
  constexpr int NRD = TM + TN, NMF = TM * TN;                       // reads / MFMAs per k-step
  zero_acc();
  // bias of the lane's columns: groups of four consecutive columns (j, q), see the epilogue
  constexpr int NQ = T * T / 256, NBG = TN * NQ;
  f32x4 bj[NBG];
  int n0 = 0;
  auto load_bias = [&]() {
    int h = hl;
    asm volatile("" : "+v"(h));
#pragma unroll
    for (int g = 0; g < NBG; g++) {
      const int col = n0 + wn * 64 + (g / NQ) * T + (MF == 32 ? (g % NQ) * 8 + h * 4 : h * 4);
      if (p.bias == nullptr || p.partial != nullptr) bj[g] = f32x4{0, 0, 0, 0};
      else if (col + 4 <= p.N) bj[g] = *reinterpret_cast<const f32x4*>(p.bias + col);
      else {
#pragma unroll
        for (int r = 0; r < 4; r++) bj[g][r] = p.bias[min(col + r, p.N - 1)];
      }
    }
  };

  int w = w_begin + slot;
  int m0 = 0, z = 0, k_lo = 0, k_hi = 0;
  int buf = 0;
  if (w < w_end) {
    item(w, m0, n0, z, k_lo, k_hi);
    set_voff(m0, n0);
    static_for<8>([&](auto Q) { dma(Q, lds, k_lo); });
    static_for<8>([&](auto Q) { dma(Q, lds + STAGE, k_lo + 1); });
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(std::integral_constant<int, 0>{}, lds, 0);
  }
  for (; w < w_end; w += nxw) {
    int m1 = 0, n1 = 0, z1 = 0, k1_lo = 0, k1_hi = 0;
    const bool have_next = w + nxw < w_end;
    if (have_next) item(w + nxw, m1, n1, z1, k1_lo, k1_hi);
    for (int kt = k_lo; kt < k_hi; kt++) {
      unsigned char* sb = lds + buf * STAGE;
      unsigned char* nb = lds + (buf ^ 1) * STAGE;
      if (kt == k_hi - 2 && have_next) set_voff(m1, n1);           // every DMA from here on belongs to the next item
      const bool last = kt + 1 == k_hi;
      if (last) load_bias();                                        // in flight under this stage; the barrier's vmcnt(0) covers them
      // ---- k-steps 0 .. NSTEP-2: MFMAs of step j, reads of step j+1 between them ----
      static_for<NSTEP - 1>([&](auto J) {
        constexpr int j = decltype(J)::value;
        read_frags(std::integral_constant<int, (j + 1) & 1>{}, sb, j + 1);
        mfma_step(std::integral_constant<int, j & 1>{});
        static_for<NRD>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
        if constexpr (NMF > NRD) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
      });
      // ---- every fragment of this stage is in registers, the next stage has landed: barrier ----
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // ---- last k-step: first fragments of stage s+1, MFMAs, DMA of stage s+2 into this buffer between them.  ONE MFMA path: the
      // wave-uniform conditions only guard the small read / DMA blocks (three copies of the step made hipcc shuffle and spill whole
      // accumulators at the joins) ----
      const bool own2 = kt + 2 < k_hi;
      const int kt2 = own2 ? kt + 2 : k1_lo;
      const bool nx1 = !last || have_next;                          // a next stage exists (landed in the other buffer)
      const bool nx2 = !last && (own2 || have_next) && !(VAR & 8);  // stage s+2 goes out now (the last stage keeps its buffer for the slab)
      if (nx1) read_frags(std::integral_constant<int, 0>{}, nb, 0);
      static_for<TM>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int st = (NSTEP - 1) & 1;
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = mfma_bf16_v1<MF>(fb[st][j], fa[st][i], acc[i][j]);
        if (nx2) static_for<8 / TM>([&](auto U) { dma(std::integral_constant<int, i * (8 / TM) + decltype(U)::value>{}, sb, kt2); });
      });
      buf ^= 1;
    }
    // ---- epilogue: the stage just consumed (buf ^ 1) is free (its DMA was held back): the row slab lives there ----
    unsigned char* slab = lds + (buf ^ 1) * STAGE;
    const bool part = p.partial != nullptr;
    float* pc = part ? p.partial + (size_t)z * (size_t)p.M * (size_t)p.N : nullptr;
    const long ldo = part ? (long)p.N : p.ldc;
    constexpr int MTR = RPR / (2 * T);                              // MFMA tile rows per wave and round
    if constexpr (VAR & 4) {                                        // (ablation: no epilogue; the accumulators stay live)
      float t = 0.0f;                                               // (keeps the accumulators live at 128 adds per tile)
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < T * T / 64; r++) t += acc[i][j][r];
      if (t == 12345.678f) reinterpret_cast<float*>(p.C)[tid] = t;
    } else
    static_for<TM / MTR>([&](auto RD) {
      constexpr int rd = decltype(RD)::value;
      if constexpr (rd > 0) lds_barrier();
#pragma unroll
      for (int ii = 0; ii < MTR; ii++) {
        const int sr = (wm * MTR + ii) * T + rl;                    // slab row; 16-byte chunk c of row r sits at c ^ (r & 31)
#pragma unroll
        for (int j = 0; j < TN; j++) {
#pragma unroll
          for (int q = 0; q < T * T / 256; q++) {                   // groups of four consecutive columns
            const int e0 = wn * 64 + j * T + (MF == 32 ? q * 8 + hl * 4 : hl * 4);
            const f32x4 bv = bj[j * NQ + q];
            if constexpr (ES == 2) {
              struct alignas(8) B4 { bf16_t e[4]; } v;
#pragma unroll
              for (int r = 0; r < 4; r++) v.e[r] = f2bf(acc[rd * MTR + ii][j][q * 4 + r] + bv[r]);
              const int g8 = e0 >> 2;
              *reinterpret_cast<B4*>(slab + sr * (G256_BN * 2) + ((((g8 >> 1) ^ (sr & 31)) << 1) | (g8 & 1)) * 8) = v;
            } else {
              f32x4 v;
#pragma unroll
              for (int r = 0; r < 4; r++) v[r] = acc[rd * MTR + ii][j][q * 4 + r] + bv[r];
              *reinterpret_cast<f32x4*>(slab + sr * (G256_BN * 4) + (((e0 >> 2) ^ (sr & 31)) << 4)) = v;
            }
          }
        }
      }
      lds_barrier();
#pragma unroll 2
      for (int q = 0; q < CPT; q++) {
        const int cid = q * NT + tid;
        const int sr = cid / CPRW, c = cid % CPRW;
        const int row = m0 + (sr / (MTR * T)) * WM + (rd * MTR + ((sr / T) % MTR)) * T + (sr % T);
        constexpr int EPC = 16 / ES;
        const int col = n0 + c * EPC;
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab + sr * (G256_BN * ES) + ((c ^ (sr & 31)) << 4));
        if (row < p.M && col < p.N) {
          TO* dst = (part ? reinterpret_cast<TO*>(pc) : reinterpret_cast<TO*>(p.C)) + (size_t)row * ldo + col;
          if (col + EPC <= p.N && (ldo % EPC) == 0) *reinterpret_cast<u32x4*>(dst) = v;
          else {
            const TO* e = reinterpret_cast<const TO*>(&v);
            for (int qq = 0; qq < EPC; qq++) if (col + qq < p.N) dst[qq] = e[qq];
          }
        }
      }
    });
    zero_acc();
    // the slab has been read out by everyone: the held-back DMA (second stage of the next item) goes into it
    if (have_next) {
      lds_barrier();
      if (k1_lo + 1 < k1_hi && !(VAR & 8)) static_for<8>([&](auto Q) { dma(Q, slab, k1_lo + 1); });
    }
    m0 = m1; n0 = n1; z = z1; k_lo = k1_lo; k_hi = k1_hi;
  }
}

}  // namespace vct

namespace vct {
template <int TA, int TB, typename TO, int MF, int VAR> static int g32v1_launch(const G256P& p, hipStream_t st) {
  static vct::DynLdsOptIn optin;
  if (hipError_t e = optin.ensure((const void*)g32v1_kernel<TA, TB, TO, MF, VAR>, G256_LDS); e != hipSuccess) return (int)e;
  vct::launch(g32v1_kernel<TA, TB, TO, MF, VAR>, dim3(persistent_grid(st)), dim3(512), (size_t)G256_LDS, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}
}  // namespace vct
