// bf16 MFMA GEMM, second generation, for gfx950 (the throughput path; fp32 parity mode stays in
// vct_gemm.hip).   C[M,N] = epilogue(op(A)[M,K] * op(B)[K,N]),  v_mfma_f32_16x16x32_bf16.
//
//  * operands go HBM -> LDS directly (global_load_lds_dwordx4: 16 B per lane, no VGPR round trip),
//    double-buffered, ONE barrier per 64-deep K tile; the next tile's DMA flies under the MFMAs.
//  * LDS images are unpadded and lane-linear (what the DMA requires); bank conflicts are removed
//    by XOR-swizzling the per-lane SOURCE address and applying the same involution on the read:
//      K-contiguous operand : [rows][64] bf16 (128-B rows), 16-B chunk c of row r sits at c ^ (r & 7),
//                             fragments by ds_read_b128;
//      M/N-contiguous operand: [64 k][cols] bf16, 32-B block b of k-row r sits at b ^ (r & (NB-1)),
//                             fragments by the LDS transpose read ds_read_b64_tr_b16 (no transposed
//                             copies in HBM for dX = dY W and dW = dY^T X).
//  * rows beyond M/N are CLAMPED (their garbage only reaches outputs that are never stored); a
//    ragged last K tile goes through a zero-filling register path into the same swizzled image.
//  * workgroup -> tile map is XCD-aware: each XCD (private 4 MB L2) owns a contiguous run of tiles,
//    walked fastest along the dimension with fewer tiles so the smaller operand stays L2-resident
//    and the larger one streams once.
//  * epilogue: per-wave fp32 LDS transpose so every lane stores 16 contiguous bytes; bias /
//    activation (+ saved pre-activation) / dropout / residual-gradient accumulate / activation
//    derivative fused; bias gradient via one extra MFMA against a ones fragment; deterministic
//    split-K (partials + second pass in vct_gemm.hip).
#pragma once
#include <type_traits>
// Which forms interleave the next K tile's DMA with the MFMAs (see `compute`): the NN form (dX GEMMs) only.  Alone every form gains 0-11 %
// (tools/gemm_bench.py, same box); in the training step the NT / TN forms LOSE (the TN weight-gradient GEMMs run beside other kernels on
// the second stream and slow down more than they gain: step 2.54 vs 2.51 ms), the NN form wins (2.47 vs 2.49 ms).
#ifndef VCT_GEMM_DMA_IL
#define VCT_GEMM_DMA_IL(TA, TB, BM, BN, NW) ((TA) == 0 && (TB) == 0)
#endif
#include "vct_common.h"
#include "vct_adam_core.h"
#include "vct_gemm_params.h"

namespace vct {

constexpr int BK2 = 64;

// ---- LDS image addressing (bytes) --------------------------------------------------------------
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
template <int R> __device__ __forceinline__ int mc_off(int krow, int col) {  // col in elements, multiple of 4
  constexpr int NB = R / 16;  // 32-byte blocks per k-row
  const int b = col >> 4;
  return krow * (R * 2) + (((b ^ (krow & (NB - 1))) << 5) | ((col & 15) << 1));
}

// issue the DMA of one operand tile (full K tile, rows clamped)
template <bool MC, int R, int NW>
__device__ __forceinline__ void dma_tile(unsigned char* lds, const bf16_t* __restrict__ base, long ld, int r0, int r_ext,
                                         int k0, int wave, int lane) {
  constexpr int NI = R * 8 / (64 * NW);  // wave-instructions per wave (1 KiB each)
  static_assert(NI >= 1, "tile too small for this many waves");
#pragma unroll
  for (int q = 0; q < NI; q++) {
    const int ci = q * NW + wave;  // 1-KiB chunk index
    const bf16_t* src;
    if constexpr (!MC) {
      const int row = ci * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (row & 7);
      const int gr = min(r0 + row, r_ext - 1);
      src = base + (long)gr * ld + k0 + c * 8;
    } else {
      constexpr int CPRW = R / 8;         // 16-byte chunks per k-row
      constexpr int RPI = 64 / CPRW;      // k-rows per wave-instruction
      constexpr int NB = R / 16;
      const int krow = ci * RPI + lane / CPRW;
      const int p = lane % CPRW;
      const int b = (p >> 1) ^ (krow & (NB - 1));
      const int col = (b * 2 + (p & 1)) * 8;
      const int rlim = ((r_ext + 7) & ~7) - 8;  // last fully readable vector (ld covers the rounded-up extent)
      const int gc = min(r0 + col, rlim);
      src = base + (long)(k0 + krow) * ld + gc;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + ci * 1024), 16, 0, 0);
  }
}

// instruction q (of R * 8 / (64 * NW) per wave) of dma_tile: one 1-KiB global_load_lds of an operand tile
template <bool MC, int R, int NW, int AUX = 0>
__device__ __forceinline__ void dma_piece(unsigned char* lds, const bf16_t* __restrict__ base, long ld, int r0, int r_ext, int k0,
                                          int wave, int lane, int q) {
  const int ci = q * NW + wave;
  const bf16_t* src;
  if constexpr (!MC) {
    const int row = ci * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (row & 7);
    src = base + (long)min(r0 + row, r_ext - 1) * ld + k0 + c * 8;
  } else {
    constexpr int CPRW = R / 8, RPI = 64 / CPRW, NB = R / 16;
    const int krow = ci * RPI + lane / CPRW;
    const int pp = lane % CPRW;
    const int b = (pp >> 1) ^ (krow & (NB - 1));
    const int col = (b * 2 + (pp & 1)) * 8;
    const int rlim = ((r_ext + 7) & ~7) - 8;
    src = base + (long)(k0 + krow) * ld + min(r0 + col, rlim);
  }
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)(lds + ci * 1024), 16, 0, AUX);
}

// The same instruction with its per-lane source address split into a part that is fixed for a whole output tile (byte offset of the
// lane's 16 bytes at K offset 0: row clamp, row x leading dimension, swizzled chunk -- a 64-bit multiply) and a wave-uniform base
// that moves with the K stage.  Recomputing the address per piece and stage cost the K loop of the persistent-tile kernel eight
// vector instructions per piece, three of them quarter-rate multiplies: about as many VALU issue cycles per stage as its MFMAs.
template <bool MC, int R, int NW>
__device__ __forceinline__ uint32_t dma_piece_offset(long ld, int r0, int r_ext, int wave, int lane, int q) {
  const int ci = q * NW + wave;
  if constexpr (!MC) {
    const int row = ci * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (row & 7);
    return (uint32_t)(((long)min(r0 + row, r_ext - 1) * ld + c * 8) * 2);
  } else {
    constexpr int CPRW = R / 8, RPI = 64 / CPRW, NB = R / 16;
    const int krow = ci * RPI + lane / CPRW;
    const int pp = lane % CPRW;
    const int b = (pp >> 1) ^ (krow & (NB - 1));
    const int col = (b * 2 + (pp & 1)) * 8;
    const int rlim = ((r_ext + 7) & ~7) - 8;
    return (uint32_t)(((long)krow * ld + min(r0 + col, rlim)) * 2);
  }
}
// base_k: the operand advanced to the stage (K-contiguous: base + k0; M/N-contiguous: base + k0 * ld), wave-uniform
template <int NW>
__device__ __forceinline__ void dma_piece_at(unsigned char* lds, const bf16_t* __restrict__ base_k, uint32_t off, int wave, int q) {
  const int ci = q * NW + wave;
  const unsigned char* src = reinterpret_cast<const unsigned char*>(base_k) + (size_t)off;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)(lds + ci * 1024), 16, 0, 0);
}

// ragged last K tile: predicated 16-byte loads (zero fill) written into the same swizzled image
struct alignas(16) V16b { uint32_t w[4]; };
template <bool MC, int R, int NT>
__device__ __forceinline__ void tail_tile(unsigned char* lds, const bf16_t* __restrict__ base, long ld, int r0, int r_ext,
                                          int k0, int K, int tid) {
  constexpr int NV = R * 8 / NT;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int v = tid + i * NT;
    V16b val; val.w[0] = val.w[1] = val.w[2] = val.w[3] = 0u;
    if constexpr (!MC) {
      const int row = v >> 3, c = v & 7;
      const int gr = r0 + row, gk = k0 + c * 8;
      if (gr < r_ext && gk < K) {
        val = *reinterpret_cast<const V16b*>(base + (long)gr * ld + gk);
        if (gk + 8 > K) {                 // K % 8 != 0: elements past K are dropped here, whatever the buffer's padding holds
          const int keep = K - gk;        // 1..7 valid elements
#pragma unroll
          for (int q = 0; q < 4; q++) {
            if (2 * q >= keep) val.w[q] = 0u;
            else if (2 * q + 1 >= keep) val.w[q] &= 0xffffu;
          }
        }
      }
      *reinterpret_cast<V16b*>(lds + kc_off(row, c)) = val;
    } else {
      constexpr int CPRW = R / 8;
      const int krow = v / CPRW, col = (v % CPRW) * 8;
      const int gk = k0 + krow, gr = r0 + col;
      if (gk < K && gr < r_ext) val = *reinterpret_cast<const V16b*>(base + (long)gk * ld + gr);
      *reinterpret_cast<V16b*>(lds + mc_off<R>(krow, col)) = val;
    }
  }
}

// Split-K partials cross XCDs (each XCD has a private, mutually non-coherent L2): they are written and read with
// agent-scope relaxed ATOMIC accesses (write-through / L2-bypassing `sc1` stores and loads) so that no workgroup
// needs an agent-scope fence -- on a multi-XCD part that fence writes back and invalidates the whole L2.
__device__ __forceinline__ void coherent_store2(float* p, float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  const f32x2 v = {a, b};
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coherent_load2(const float* p, float& a, float& b) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  const f32x2 v = __builtin_bit_cast(f32x2, u);
  a = v[0]; b = v[1];
}
__device__ __forceinline__ void coherent_store1(float* p, float a) {
  __hip_atomic_store(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float coherent_load1(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// workgroup barrier that orders LDS traffic only: outstanding global loads / stores stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <bool MC, int R, bool KSPLIT>
__device__ __forceinline__ bf16x8 frag2(const unsigned char* lds, int r_base, int ks, int lane) {
  const int i = lane & 15, g = lane >> 4;
  if constexpr (!MC) {
    return *reinterpret_cast<const bf16x8*>(lds + kc_off(r_base + i, ks * 4 + g));
  } else {
    const int kb1 = ks * 32 + (KSPLIT ? g * 4 : g * 8);
    const int kb2 = KSPLIT ? ks * 32 + 16 + g * 4 : kb1 + 4;
    const int col = r_base + (i & 3) * 4;
    const s16x4 lo = lds_tr16(reinterpret_cast<const bf16_t*>(lds + mc_off<R>(kb1 + (i >> 2), col)));
    const s16x4 hi = lds_tr16(reinterpret_cast<const bf16_t*>(lds + mc_off<R>(kb2 + (i >> 2), col)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}

// One workgroup's tile of one GEMM: `bid` = workgroup index within the problem's tile grid, `zid` = K split.
template <typename TO, int TA, int TB, int BM, int BN, int NBUF, int WGM, int WGN>
__device__ __forceinline__ void gemm_bf16_v2_body(const GemmP& p, const int bid, const int zid, const bool direct = false) {
  constexpr bool A_MC = (TA == 1), B_MC = (TB == 0);
  constexpr bool KSPLIT = A_MC && B_MC;
  constexpr int NW = WGM * WGN, NT = 64 * NW;         // waves / threads per workgroup
  constexpr bool DMA_IL = VCT_GEMM_DMA_IL(TA, TB, BM, BN, NW);   // interleave the next tile's DMA with this tile's MFMAs
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;
  __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[NBUF * BUF_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // XCD-aware tile map (bijective for any tile count)
  const int nwg = p.tiles_m * p.tiles_n;
  int wgid;
  {
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    if (direct) wgid = bid;          // the caller (grouped launch, group-level map) already placed this tile on its XCD
  }
  // Within an XCD's run, tiles are walked in groups of GRP along the LONGER tile dimension: the GRP
  // tiles' operand slabs stay L2-resident while the shorter dimension is swept, so each operand
  // element is fetched from HBM/MALL once per group instead of once per tile.
  constexpr int GRP = 8;
  int tile_m, tile_n;
  {
    const bool n_long = p.tiles_n >= p.tiles_m;
    const int tl = n_long ? p.tiles_n : p.tiles_m, ts = n_long ? p.tiles_m : p.tiles_n;
    const int per_group = GRP * ts;
    const int grp = wgid / per_group, rem = wgid - grp * per_group;
    const int gsz = min(GRP, tl - grp * GRP);           // last group may be short
    int tshort = rem / gsz, tlong = grp * GRP + rem % gsz;
    // K-long operands (the vocabulary-deep dX): a tile's operand slabs are megabytes and never stay resident, so the only
    // reuse is between tiles that stream the SAME slab at the same time -- the ts tiles of one long-dimension index become
    // neighbours (same XCD, dispatched back to back) and the big operand is pulled from HBM once instead of once per XCD
    if (p.short_fast) { tlong = wgid / ts; tshort = wgid - tlong * ts; }
    tile_m = n_long ? tshort : tlong;
    tile_n = n_long ? tlong : tshort;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int nkt = (p.K + BK2 - 1) / BK2;
  const int kt_begin = zid * p.kt_per_split;
  const int kt_end = min(nkt, kt_begin + p.kt_per_split);
  const int kt_full_end = min(kt_end, p.K / BK2);   // tiles fully inside K go by DMA

  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

  f32x4 acc[TM][TN];
  // bias gradient = row sums of op(A): the WGN waves of a row group hold the same A fragments, so wave wn takes the tile rows
  // i == wn (mod WGN).  (Rounds 1-5: the wn == 0 wave took all TM rows -- TM extra MFMAs per k-step on one wave of four, +50 % on its
  // matrix-pipe work, and the tile, i.e. every tile of the first tile column, waited for it.)  Same MFMA sequence per row: same bits.
  constexpr int TMB = (TM + WGN - 1) / WGN;
  f32x4 accb[TMB];
#pragma unroll
  for (int u = 0; u < TMB; u++) accb[u] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < TM; i++) {
#pragma unroll
    for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};
  }
  // the bias gradient only exists for the weight-gradient form (dW = dY^T X, fp32 out): everywhere else its accumulators
  // and its branch are compiled out (16 VGPRs back, one basic block per K tile)
  constexpr bool BG = (TA == 1 && TB == 0 && sizeof(TO) == 4);
  const bool do_bias_grad = BG && (p.bias_grad != nullptr) && (tile_n == 0);
  const s16x8 ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);

  // One K tile (two 32-deep MFMA steps).  ALL fragment reads of the tile are issued before the first MFMA and the
  // bias-gradient MFMAs sit behind one branch at the END: a branch between the two k-steps (the former shape of this
  // loop) splits the basic block, and hipcc then issues "6 reads, wait for all, 8 MFMAs" twice per tile with the LDS
  // latency of the second group fully exposed.  As one block the k-step-1 reads fly under the k-step-0 MFMAs.
  // WITH_DMA: the next K tile (ktn, fully inside K) goes into `nb` by this wave's DMA instructions, placed by the scheduling hints
  // one at a time BETWEEN the MFMA groups of the first half of this tile (see vct_gemm256.hip: issued back to back, the DMA
  // instructions of a stage queue up behind the CU's vector-memory port and the wave cannot reach its fragment reads meanwhile).
  auto compute = [&](const unsigned char* la, const unsigned char* lb, auto WITH_DMA, unsigned char* nb, const int ktn) {
    constexpr bool DMA = decltype(WITH_DMA)::value;
    constexpr int KS = BK2 / 32;
    bf16x8 fa[KS][TM], fb[KS][TN];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
#pragma unroll
      for (int i = 0; i < TM; i++) fa[ks][i] = frag2<A_MC, BM, KSPLIT>(la, wm * WM + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < TN; j++) fb[ks][j] = frag2<B_MC, BN, KSPLIT>(lb, wn * WN + j * 16, ks, lane);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
    // schedule shape: every LDS fragment read of the tile first, then the MFMAs (the compiler inserts counted lgkmcnt
    // waits, so the k-step-0 MFMAs start as soon as their six fragments are back while the k-step-1 reads are in flight);
    // left alone hipcc minimises registers instead: 2-4 reads, wait for all, 2-4 MFMAs, five times per tile
    constexpr int N_DSREAD = KS * (TM * (A_MC ? 2 : 1) + TN * (B_MC ? 2 : 1));
    constexpr int NMF = KS * TM * TN;
    if constexpr (DMA) {
      constexpr int NP = BM * 8 / (64 * NW) + BN * 8 / (64 * NW);      // DMA instructions per wave and stage
      constexpr int GRP = (NMF / 2 / NP) > 0 ? (NMF / 2 / NP) : 1;     // MFMAs between two of them (all inside the first half)
      dma_tile<A_MC, BM, NW>(nb, A, p.lda, m0, p.M, ktn * BK2, wave, lane);
      dma_tile<B_MC, BN, NW>(nb + A_BYTES, B, p.ldb, n0, p.N, ktn * BK2, wave, lane);
      __builtin_amdgcn_sched_group_barrier(0x100, N_DSREAD, 0);
      static_for<NP>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, GRP, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      });
      if constexpr (NMF - NP * GRP > 0) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NP * GRP, 0);
    } else {
      __builtin_amdgcn_sched_group_barrier(0x100, N_DSREAD, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
    }
    if constexpr (BG) {
      if (do_bias_grad) {
        static_for<TM>([&](auto I) {
          constexpr int i = decltype(I)::value;
          if (wn == i % WGN) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) accb[i / WGN] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], ones, accb[i / WGN], 0, 0, 0);
          }
        });
      }
    }
  };

  // stage K tile `kt` into buffer `buf`: DMA when the tile lies fully inside K, else the zero-filling
  // register path (at most one ragged tile per GEMM, always the last)
  auto stage = [&](int kt, int buf) {
    unsigned char* nb = lds_raw + buf * BUF_BYTES;
    if (kt < kt_full_end) {
      dma_tile<A_MC, BM, NW>(nb, A, p.lda, m0, p.M, kt * BK2, wave, lane);
      dma_tile<B_MC, BN, NW>(nb + A_BYTES, B, p.ldb, n0, p.N, kt * BK2, wave, lane);
    } else {
      tail_tile<A_MC, BM, NT>(nb, A, p.lda, m0, p.M, kt * BK2, p.K, tid);
      tail_tile<B_MC, BN, NT>(nb + A_BYTES, B, p.ldb, n0, p.N, kt * BK2, p.K, tid);
    }
  };
  // NBUF = 1: stage -> wait -> barrier -> compute -> barrier (smallest LDS, most workgroups per CU)
  // NBUF = 2: next tile's DMA flies under the MFMAs, one barrier per tile (default)
  // (a 3-deep ring with counted vmcnt + raw s_barrier was measured slower at every shape: it costs occupancy and
  //  the kernel is bound by the per-CU fetch rate, not by exposed DMA latency)
  if constexpr (NBUF == 1) {
    for (int kt = kt_begin; kt < kt_end; kt++) {
      if (kt > kt_begin) __syncthreads();             // everyone finished reading the buffer
      stage(kt, 0);
      __syncthreads();                                // DMA landed (vmcnt(0)) for everyone
      compute(lds_raw, lds_raw + A_BYTES, std::false_type{}, nullptr, 0);
    }
  } else {
    static_assert(NBUF == 2, "NBUF is 1 or 2");
    if (kt_begin < kt_end) stage(kt_begin, 0);
    int cur = 0;
    int kt = kt_begin;
    if constexpr (DMA_IL) {
      // hot loop: the next tile is a full one -> its DMA rides between this tile's MFMAs; one basic block per K tile
      for (; kt + 1 < kt_full_end; kt++) {
        __syncthreads();
        const unsigned char* la = lds_raw + cur * BUF_BYTES;
        compute(la, la + A_BYTES, std::true_type{}, lds_raw + (cur ^ 1) * BUF_BYTES, kt + 1);
        cur ^= 1;
      }
    }
    for (; kt < kt_end; kt++) {                       // (DMA_IL: the last one or two tiles; ragged next tile by the register path)
      __syncthreads();
      if (kt + 1 < kt_end) stage(kt + 1, cur ^ 1);
      compute(lds_raw + cur * BUF_BYTES, lds_raw + cur * BUF_BYTES + A_BYTES, std::false_type{}, nullptr, 0);
      cur ^= 1;
    }
  }

  // ---- epilogue (see vct_gemm.hip for the rationale of the per-wave LDS transpose) ---------------
  const int c16 = lane & 15, g4 = (lane >> 4) * 4;
  if (do_bias_grad && c16 == 0) {
    float* bg = p.partial != nullptr ? p.bias_partial + (size_t)zid * p.M : p.bias_grad;
#pragma unroll
    for (int u = 0; u < TMB; u++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = wn + u * WGN;
        const int row = m0 + wm * WM + i * 16 + g4 + r;
        if (i < TM && row < p.M) {
          if (p.counters != nullptr) coherent_store1(bg + row, accb[u][r]);
          else bg[row] = accb[u][r];
        }
      }
  }
  __syncthreads();
  // The tile leaves through WORKGROUP-wide row slabs: for each 16-row MFMA tile row i every wave drops its 16 x WN piece
  // into a shared fp32 slab [WGM*16 rows][BN], then every lane picks up ONE 16-byte output chunk such that a row of the
  // slab is written by consecutive lanes -- a store instruction covers whole BN*sizeof(TO)-byte row segments (>= 128 B:
  // full cache lines).  The former per-wave transpose stored 16 rows x WN*sizeof(TO) = 64-byte HALF lines per instruction
  // (WN = 32 bf16), which the L2 had to merge: a second 20 MB output (the saved pre-activation) cost 12.5 us = 1.6 TB/s.
  float* stage_base = reinterpret_cast<float*>(lds_raw);
  constexpr int VO = 16 / (int)sizeof(TO);
  // two ways out, chosen per tile shape (measured on the same box: the slab form takes the FFN GEMM with activation +
  // saved pre-activation from 42 to 38 us, the per-wave form keeps the 128x128 vocabulary projection at 250 instead of 273 us):
  //   SLAB   every wave drops its 16 x WN piece of tile row i into a workgroup-wide fp32 slab [WGM*16][BN]; after a
  //          barrier each lane owns ONE 16-byte chunk such that consecutive lanes cover a whole output row of the tile
  //          (BN*sizeof(TO) >= 128 B: full cache lines per store instruction);
  //   !SLAB  every wave transposes through its private 16 x WN slab (no barrier; 16 rows x WN*sizeof(TO) per instruction).
  constexpr bool SLAB = (BM * BN != 128 * 128);
  constexpr int SLAB_ROWS = WGM * 16, CPRW = BN / VO;
  constexpr int CPRV = WN / VO;                                        // per-wave form: chunks per staged row
  constexpr int SSTR = SLAB ? BN + 4 : WN + 4;
  constexpr int STAGE_FLOATS = SLAB ? SLAB_ROWS * SSTR : NW * 16 * SSTR;
  constexpr int NSLAB = (SLAB && 2 * STAGE_FLOATS * 4 <= NBUF * BUF_BYTES) ? 2 : 1;
  static_assert(STAGE_FLOATS * 4 <= NBUF * BUF_BYTES, "epilogue staging must fit in the operand buffers");
  static_assert(!SLAB || (SLAB_ROWS * CPRW) % NT == 0, "slab chunks must divide over the workgroup");
  constexpr int CPL = SLAB ? SLAB_ROWS * CPRW / NT : (16 * CPRV + 63) / 64;
  const bool part = p.partial != nullptr;
  // this lane's chunk(s): staging offset e_lds (floats), row offset inside the tile (without i*16) e_row, tile column e_col
  int e_lds[CPL], e_row[CPL], e_col[CPL];
  bool e_act[CPL];
  float bvec[CPL][VO];
#pragma unroll
  for (int c = 0; c < CPL; c++) {
    if constexpr (SLAB) {
      const int chunk = c * NT + tid;
      const int sr = chunk / CPRW, sc = (chunk % CPRW) * VO;
      e_lds[c] = sr * SSTR + sc; e_row[c] = (sr >> 4) * WM + (sr & 15); e_col[c] = sc; e_act[c] = true;
    } else {
      const int chunk = c * 64 + lane;
      const int rr = (chunk / CPRV) & 15, cc = (chunk % CPRV) * VO;
      e_lds[c] = wave * 16 * SSTR + rr * SSTR + cc; e_row[c] = wm * WM + rr; e_col[c] = wn * WN + cc;
      e_act[c] = chunk < 16 * CPRV;
    }
#pragma unroll
    for (int q = 0; q < VO; q++) bvec[c][q] = 0.0f;
  }
  if (p.bias != nullptr && (!part || p.counters != nullptr)) {   // split-K: the in-kernel reduce applies the epilogue
#pragma unroll
    for (int c = 0; c < CPL; c++) {
#pragma unroll
      for (int q = 0; q < VO; q++) bvec[c][q] = p.bias[min(n0 + e_col[c] + q, p.N - 1)];
    }
  }

  const Dropout dr = make_dropout(p.seed, p.site, p.p_drop);
  TO* C = reinterpret_cast<TO*>(p.C);
  TO* preact = reinterpret_cast<TO*>(p.preact);
  const TO* addend = reinterpret_cast<const TO*>(p.addend);
  const TO* dact = reinterpret_cast<const TO*>(p.dact);
  float* partC = part ? p.partial + (size_t)zid * (size_t)p.M * (size_t)p.N : nullptr;
  const long ldo = part ? (long)p.N : p.ldc;
  const bool vec_ok = (ldo % VO == 0) && (part || (((uintptr_t)p.C & 15) == 0));
  const int nt_store = p.nt_store;
  struct alignas(16) OutV { TO e[VO]; };
  // Optimizer epilogue of the weight-gradient form (include/vct_hip.h, vct_gemm_adam): the gradient elements this lane would store
  // are consumed by torch.optim.Adam's update of the parameter elements they belong to (same layout, same offset in the flat
  // buffers), and the bf16 shadow / stream-order packed copy are written from the same registers -- the optimizer's 28 B per
  // parameter move inside this MFMA-bound kernel (other workgroups of the CU are in their K loops meanwhile) instead of forming
  // a serial HBM-bound tail of the step.  Shared by the direct path and the in-kernel split-K reduce, like `finish`.
  AdamConsts hc;
  bool adam_on = false;
  // every field of the descriptor is read HERE, once, into locals: a branch on p.adam.<field> inside the unrolled epilogue makes
  // hipcc copy the whole by-value group table (2.5 KB) into scratch in the grouped 128 x 128 kernel (the table is reached through the
  // run-time problem index there)
  float* ad_param = nullptr; float* ad_m = nullptr; float* ad_v = nullptr;
  uint16_t* ad_shadow = nullptr; uint16_t* ad_pk = nullptr;
  long ad_lds = 0; unsigned long long ad_chunks = 0ull;
  int ad_mode = 0, ad_row0 = 0;
  bool ad_store_grad = true;
  const int ad_M = p.M, ad_N = p.N;
  const long ad_ldc = p.ldc;
  if constexpr (BG) {
    adam_on = p.adam.param != nullptr;
    if (adam_on) {
      hc = adam_consts_uniform(p.adam.hyper, p.adam.step);
      ad_param = p.adam.param; ad_m = p.adam.m; ad_v = p.adam.v;
      ad_shadow = p.adam.shadow; ad_lds = p.adam.ld_shadow;
      ad_pk = p.adam.pk_stream; ad_chunks = p.adam.pk_chunks; ad_mode = p.adam.pk_mode; ad_row0 = p.adam.pk_row0;
      ad_store_grad = p.adam.store_grad != 0;
    }
  }
  // Two halves so that the direct epilogue can put the loads of ALL the chunks of a tile row in flight before the first update (a
  // chunk is three dependent HBM round trips otherwise: eight serial ones per wave and tile cost the vocabulary product +40 %):
  //   adam_load  : parameter / moment vectors of a full chunk (addresses clamped into the matrix: unconditional loads)
  //   adam_store : update + stores (parameters, moments, shadow, packed copy); ragged chunks take the element loop
  struct AdamRegs { float4 p, m, v; };
  auto adam_load = [&](const int row_in, const int col_in, AdamRegs& r) {
    if constexpr (BG) {
      // opaque copies: row / col depend on the thread index only, so without this every chunk's 64-bit addresses are computed at
      // kernel entry and kept alive (spilled) across the K loop
      int row = min(row_in, ad_M - 1), col = min(col_in, max(ad_N - 4, 0));
      asm volatile("" : "+v"(row), "+v"(col));
      const size_t e = (size_t)row * ad_ldc + col;
      r.p = *reinterpret_cast<const float4*>(ad_param + e);
      r.m = *reinterpret_cast<const float4*>(ad_m + e);
      r.v = *reinterpret_cast<const float4*>(ad_v + e);
    }
  };
  auto adam_store = [&](const int row_in, const int col_in, const float (&g)[VO], const bool full, AdamRegs& r) {
    if constexpr (BG) {
      int row = row_in, col = col_in;
      asm volatile("" : "+v"(row), "+v"(col));
      const size_t e = (size_t)row * ad_ldc + col;
      if (full) {
        static_assert(!BG || VO == 4, "fp32 output: four elements per lane chunk");
        float* pp = &r.p.x; float* mp = &r.m.x; float* vp = &r.v.x;
#pragma unroll
        for (int q = 0; q < 4; q++) adam_update(pp[q], g[q], mp[q], vp[q], hc);
        *reinterpret_cast<float4*>(ad_param + e) = r.p;
        *reinterpret_cast<float4*>(ad_m + e) = r.m;
        *reinterpret_cast<float4*>(ad_v + e) = r.v;
        if (ad_shadow != nullptr) {
          ushort4 o;
          o.x = f2bf(r.p.x); o.y = f2bf(r.p.y); o.z = f2bf(r.p.z); o.w = f2bf(r.p.w);
          *reinterpret_cast<ushort4*>(ad_shadow + (size_t)row * ad_lds + col) = o;
          if (ad_pk != nullptr) {
            const int64_t at = adam_pack_index(ad_row0 + row, col, ad_mode, ad_chunks);
            if (at >= 0) *reinterpret_cast<ushort4*>(ad_pk + at) = o;
          }
        }
      } else {
        for (int q = 0; q < VO; q++) {
          if (col + q >= ad_N) break;
          float pv = ad_param[e + q], mv = ad_m[e + q], vv = ad_v[e + q];
          adam_update(pv, g[q], mv, vv, hc);
          ad_param[e + q] = pv; ad_m[e + q] = mv; ad_v[e + q] = vv;
          if (ad_shadow != nullptr) {
            const uint16_t o = f2bf(pv);
            ad_shadow[(size_t)row * ad_lds + col + q] = o;
            if (ad_pk != nullptr) {
              const int64_t at = adam_pack_index(ad_row0 + row, col + q, ad_mode, ad_chunks);
              if (at >= 0) ad_pk[at] = o;
            }
          }
        }
      }
    }
  };
  auto adam_apply = [&](const int row, const int col, const float (&g)[VO], const bool full) {
    if constexpr (BG) {
      AdamRegs r;
      if (full) adam_load(row, col, r);
      adam_store(row, col, g, full, r);
    }
  };
  // bias / activation (+ saved pre-activation) / activation derivative / dropout / residual-gradient accumulate on
  // VO consecutive columns of one row, then the store -- shared by the direct path and the in-kernel split-K reduce
  auto finish_store = [&](const int row, const int col, const float (&v)[VO], const float (&bv)[VO], const bool full) {
    if (full) {
      OutV dv, av, ov, pv;
      const bool has_d = dact != nullptr, has_a = addend != nullptr;
      if (has_d) dv = *reinterpret_cast<const OutV*>(dact + (size_t)row * p.ld_dact + col);
      if (has_a) av = *reinterpret_cast<const OutV*>(addend + (size_t)row * p.ld_addend + col);
#pragma unroll
      for (int q = 0; q < VO; q += 2) {                      // pairs: packed fp32 arithmetic in the activation (vct_common.h)
        vf2 x = {v[q] + bv[q], v[q + 1] + bv[q + 1]};
        pv.e[q] = from_f<TO>(x[0]); pv.e[q + 1] = from_f<TO>(x[1]);
        x = act_fast_f2(p.act, x);
        if (has_d) x *= dact_fast_f2(p.dact_kind, vf2{to_f<TO>(dv.e[q]), to_f<TO>(dv.e[q + 1])});
        float dm2[2];
        drop_mults<2>(dr, (uint32_t)row * (uint32_t)p.N + (uint32_t)(col + q), dm2);
        x *= vf2{dm2[0], dm2[1]};
        if (has_a) x += vf2{to_f<TO>(av.e[q]), to_f<TO>(av.e[q + 1])};
        ov.e[q] = from_f<TO>(x[0]); ov.e[q + 1] = from_f<TO>(x[1]);
      }
      if (nt_store) {
        // experiment switch (VCT_GEMM_NT): agent-scope / non-temporal output stores -- off by default, see vct_gemm.hip
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        if (nt_store == 1) store_stream16(C + (size_t)row * p.ldc + col, __builtin_bit_cast(u32x4, ov));
        else __builtin_nontemporal_store(__builtin_bit_cast(u32x4, ov), reinterpret_cast<u32x4*>(C + (size_t)row * p.ldc + col));
      } else {
        *reinterpret_cast<OutV*>(C + (size_t)row * p.ldc + col) = ov;
      }
      if (preact != nullptr) {
        if ((p.ld_preact % VO) == 0) {
          typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
          if (p.nt_preact) store_stream16(preact + (size_t)row * p.ld_preact + col, __builtin_bit_cast(u32x4, pv));
          else *reinterpret_cast<OutV*>(preact + (size_t)row * p.ld_preact + col) = pv;
        }
        else for (int q = 0; q < VO; q++) preact[(size_t)row * p.ld_preact + col + q] = pv.e[q];
      }
    } else {
      for (int q = 0; q < VO; q++) {
        if (col + q >= p.N) break;
        float x = v[q] + bv[q];
        if (preact != nullptr) preact[(size_t)row * p.ld_preact + col + q] = from_f<TO>(x);
        x = act_fast_f(p.act, x);
        if (dact != nullptr) x *= dact_fast_f(p.dact_kind, to_f<TO>(dact[(size_t)row * p.ld_dact + col + q]));
        x *= drop_mult(dr, (uint32_t)row * (uint32_t)p.N + (uint32_t)(col + q));
        if (addend != nullptr) x += to_f<TO>(addend[(size_t)row * p.ld_addend + col + q]);
        C[(size_t)row * p.ldc + col + q] = from_f<TO>(x);
      }
    }
  };
  // (ONE call site of finish_store: a second one in the direct loop sent the grouped kernel's whole argument table to scratch)
  auto finish = [&](const int row, const int col, const float (&v)[VO], const float (&bv)[VO], const bool full, AdamRegs* pre) {
    if constexpr (BG) {
      if (adam_on) {
        if (pre != nullptr && full) adam_store(row, col, v, true, *pre);      // state already in flight (direct epilogue)
        else adam_apply(row, col, v, full);
        if (!ad_store_grad) return;
      }
    }
    finish_store(row, col, v, bv, full);
  };
  static_for<TM>([&](auto I) {
    constexpr int i = decltype(I)::value;
    float* stage = stage_base + (NSLAB == 2 ? (i & 1) * STAGE_FLOATS : 0);
    if constexpr (SLAB && NSLAB == 1 && i > 0) lds_barrier();    // the previous slab has been read by everyone
    static_for<TN>([&](auto J) {
      constexpr int j = decltype(J)::value;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if constexpr (SLAB) stage[(wm * 16 + g4 + r) * SSTR + wn * WN + j * 16 + c16] = acc[i][j][r];
        else stage[wave * 16 * SSTR + (g4 + r) * SSTR + j * 16 + c16] = acc[i][j][r];
      }
    });
    if constexpr (SLAB) lds_barrier();
    AdamRegs areg[CPL];
    const bool adam_direct = BG && adam_on && !part;
    if constexpr (BG) {
      if (adam_direct) {          // the optimizer state of every chunk of this tile row: in flight before the first update
        static_for<CPL>([&](auto CI) {
          constexpr int c = decltype(CI)::value;
          adam_load(m0 + e_row[c] + i * 16, n0 + e_col[c], areg[c]);
        });
      }
    }
    static_for<CPL>([&](auto CI) {
      constexpr int c = decltype(CI)::value;
      const int row = m0 + e_row[c] + i * 16, col = n0 + e_col[c];
      float v[VO];
#pragma unroll
      for (int q = 0; q < VO; q += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(stage + e_lds[c] + q);
        v[q] = t[0]; v[q + 1] = t[1]; v[q + 2] = t[2]; v[q + 3] = t[3];
      }
      if (!e_act[c]) return;
      if (row >= p.M || col >= p.N) return;
      const bool full = vec_ok && (col + VO <= p.N);
      if (part) {
        float* dst = partC + (size_t)row * p.N + col;
        if (p.counters != nullptr) {
          if (full) {
#pragma unroll
            for (int q = 0; q < VO; q += 2) coherent_store2(dst + q, v[q], v[q + 1]);
          } else {
            for (int q = 0; q < VO; q++) if (col + q < p.N) coherent_store1(dst + q, v[q]);
          }
        } else if (full) {
#pragma unroll
          for (int q = 0; q < VO; q += 4) *reinterpret_cast<f32x4*>(dst + q) = f32x4{v[q], v[q + 1], v[q + 2], v[q + 3]};
        } else {
          for (int q = 0; q < VO; q++) if (col + q < p.N) dst[q] = v[q];
        }
        return;
      }
      finish(row, col, v, bvec[c], full, adam_direct ? &areg[c] : nullptr);
    });
  });

  // ---- single-pass split-K: the LAST workgroup to finish a tile sums the partials in fixed z order ----
  // (deterministic: the order of the additions does not depend on which workgroup arrives last).  The tile
  // counters live in the caller's zero-initialised workspace head and are left zero again.
  if (part && p.counters != nullptr) {
    __shared__ int s_last;
    // this wave's write-through (sc1) partial stores must be acknowledged before the ticket: a workgroup-scope fence emits no vmcnt
    // wait on gfx950 (non-tgsplit), so wait explicitly (MI355X_MICROARCH "handoff-flag": sc1 payload -> vmcnt(0) -> flag)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      s_last = (__hip_atomic_fetch_add(p.counters + wgid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.split - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    const size_t zstride = (size_t)p.M * (size_t)p.N;
    static_for<TM>([&](auto I) {
      constexpr int i = decltype(I)::value;
      static_for<CPL>([&](auto CI) {
        constexpr int c = decltype(CI)::value;
        const int row = m0 + e_row[c] + i * 16, col = n0 + e_col[c];
        if (!e_act[c] || row >= p.M || col >= p.N) return;
        const float* src = p.partial + (size_t)row * p.N + col;
        const bool full = vec_ok && (col + VO <= p.N) && (p.ldc % VO == 0) && (((uintptr_t)p.C & 15) == 0);
        float s[VO];
#pragma unroll
        for (int q = 0; q < VO; q++) s[q] = 0.0f;
        if (full) {
          for (int z = 0; z < p.split; z++) {
#pragma unroll
            for (int q = 0; q < VO; q += 2) {
              float t0, t1;
              coherent_load2(src + (size_t)z * zstride + q, t0, t1);
              s[q] += t0; s[q + 1] += t1;
            }
          }
        } else {
          for (int q = 0; q < VO; q++) {
            if (col + q >= p.N) break;
            for (int z = 0; z < p.split; z++) s[q] += coherent_load1(src + (size_t)z * zstride + q);
          }
        }
        finish(row, col, s, bvec[c], full, nullptr);
      });
    });
    if (p.bias_grad != nullptr && tile_n == 0) {
      for (int r = tid; r < BM; r += NT) {
        const int row = m0 + r;
        if (row < p.M) {
          float a = 0.0f;
          for (int z = 0; z < p.split; z++) a += coherent_load1(p.bias_partial + (size_t)z * p.M + row);
          p.bias_grad[row] = a;
        }
      }
    }
    if (tid == 0) __hip_atomic_store(p.counters + wgid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next GEMM that uses this workspace
  }
}

// waves per SIMD the register allocator must leave room for: the eight-wave tiles are sized so that 2 (128x128) or
// 3 (128x64) workgroups share a CU -- one VGPR too many (130 instead of 128) silently halves that
constexpr int gemm_min_waves(int bm, int bn, int nw) { return nw == 8 ? (bm * bn == 128 * 128 ? 4 : (bm * bn == 128 * 64 ? 6 : 1)) : 1; }

template <typename TO, int TA, int TB, int BM, int BN, int NBUF, int WGM = 2, int WGN = 2>
__global__ __launch_bounds__(64 * WGM * WGN, gemm_min_waves(BM, BN, WGM * WGN)) void gemm_bf16_v2_kernel(const GemmP p) {
  gemm_bf16_v2_body<TO, TA, TB, BM, BN, NBUF, WGM, WGN>(p, blockIdx.x, blockIdx.z);
}

// Several independent GEMMs of the same layout / tile shape in ONE launch (a layer's weight gradients): the
// workgroup looks its problem up in a by-value table.
//   g.total > 0 (no problem is split): GROUP-level XCD map -- the tiles of all problems form ONE list (problem after problem, each in
//   its own walk order: blocks of 8 x (short dimension) tiles), and XCD x = blockIdx & 7 owns a contiguous eighth of it, so the ~32
//   tiles an XCD runs at the same time are one or two compact blocks of ONE or TWO problems that share their dY / X column panels
//   through that XCD's L2 (a decoder layer: 12-16 panels of 1.25 MB per XCD instead of 35 when every problem is dealt out over all
//   eight XCDs -- 366 MB of L2 misses per launch measured with the per-problem map, rocprofv3 TCC_EA0_RDREQ, round 5).
//   g.total == 0: each problem's run of workgroups starts at a multiple of 8 (so that `bid & 7` is still the XCD) and is mapped on its own.
template <typename TO, int TA, int TB, int BM, int BN, int NBUF, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, gemm_min_waves(BM, BN, WGM * WGN)) void gemm_bf16_v2_grouped_kernel(const GemmGroupP g) {
  const int b = blockIdx.x;
  int at = b;
  const bool direct = g.total > 0;
  if (direct) {
    const int xcd = b & 7, slot = b >> 3;
    const int q = g.total >> 3, r = g.total & 7;
    if (slot >= q + (xcd < r ? 1 : 0)) return;
    at = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  int gi = 0;
#pragma unroll
  for (int i = 1; i < VCT_GEMM_GROUP_MAX; i++) gi = (i < g.n && at >= g.start[i]) ? i : gi;
  const GemmP& p = g.p[gi];
  const int local = at - g.start[gi], nwg = p.tiles_m * p.tiles_n;
  if (local >= nwg * p.split) return;
  gemm_bf16_v2_body<TO, TA, TB, BM, BN, NBUF, WGM, WGN>(p, local % nwg, local / nwg, direct);
}

template <typename TO, int TA, int TB, int NBUF>
static int launch_tiles(const GemmP& p, int bm, int bn, dim3 grid, hipStream_t st) {
#define VCT_LAUNCH(BM_, BN_) vct::launch((gemm_bf16_v2_kernel<TO, TA, TB, BM_, BN_, NBUF>), grid, dim3(256), 0, st, p)
  if (p.waves8) {   // eight-wave variants: waves8 = variant id
#define VCT_LW(BM_, BN_, WM_, WN_) vct::launch((gemm_bf16_v2_kernel<TO, TA, TB, BM_, BN_, NBUF, WM_, WN_>), grid, dim3(64 * WM_ * WN_), 0, st, p)
    // measured (tools/gemm_bench.py): 16-wave 128x128, 8-wave 64x64 and every 256-wide tile (256x256 with 8 or 16 waves,
    // 256x128 with 8 waves, single or double buffered: 320-570 TF where 128x128 gives 550-780) lose to these two everywhere
    if (p.waves8 == 1 && bm == 128 && bn == 128) VCT_LW(128, 128, 2, 4);
    else if (p.waves8 == 4 && bm == 128 && bn == 64) VCT_LW(128, 64, 4, 2);
    else if constexpr (sizeof(TO) == 2 && TA == 0 && NBUF == 2) {
      // "cover" tiles for the layer GEMMs (bf16 out, NT / NN): ONE workgroup per CU and about one tile per CU, so that a
      // mid-size product (M = 3328 / 4864 rows) is a single round of workgroups at 85-128 FLOP per fetched byte instead of
      // 1.2-2.4 rounds at 32-43 -- every tile shape of this kernel is bound by the L2 -> LDS operand rate per CU
      if (p.waves8 == 6 && bm == 256 && bn == 128) VCT_LW(256, 128, 4, 2);
      else if (p.waves8 == 7 && bm == 320 && bn == 128) VCT_LW(320, 128, 4, 2);
      else if (p.waves8 == 9 && bm == 64 && bn == 128) VCT_LW(64, 128, 2, 4);
      else return VCT_E_SHAPE;
    }
    else return VCT_E_SHAPE;
#undef VCT_LW
  }
  else if (bm == 128 && bn == 128) VCT_LAUNCH(128, 128);
  else if (bm == 128 && bn == 64) VCT_LAUNCH(128, 64);
  else if (bm == 64 && bn == 128) VCT_LAUNCH(64, 128);
  else if (bm == 64 && bn == 64) VCT_LAUNCH(64, 64);
  else return VCT_E_SHAPE;
#undef VCT_LAUNCH
  return VCT_OK;
}

template <typename TO, int TA, int TB>
static int launch_nbuf(const GemmP& p, int bm, int bn, int nbuf, dim3 grid, hipStream_t st) {
  if (nbuf == 1) return launch_tiles<TO, TA, TB, 1>(p, bm, bn, grid, st);
  return launch_tiles<TO, TA, TB, 2>(p, bm, bn, grid, st);
}


// one translation unit per operand layout (parallel builds): see vct_gemm_bf16_{nt,nn,tn}.hip
template <int TA, int TB>
static int gemm_bf16_v2_layout(const vct_gemm_desc* d, const GemmP& p, int bm, int bn, int nbuf, dim3 grid, hipStream_t st) {
  return d->out_dtype == VCT_BF16 ? launch_nbuf<bf16_t, TA, TB>(p, bm, bn, nbuf, grid, st)
                                  : launch_nbuf<float, TA, TB>(p, bm, bn, nbuf, grid, st);
}

}  // namespace vct
