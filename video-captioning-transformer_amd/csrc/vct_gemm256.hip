// Persistent 256x256-tile bf16 GEMM for the vocabulary-sized products (gfx950):  C[M,N] = op(A)[M,K] op(B)[K,N] in the three
// forms of the generator nn.Linear and its autograd backward:
//     NT  logits = y W_g^T + b_g              4864 x 30522 x 512     (reference model/CapDecoder.py:25,55)
//     NN  dY     = dlogits W_g                4864 x 512 x 30522     (split over K, fp32 partials)
//     TN  dW_g   = dlogits^T y,  db_g = colsum(dlogits)   30522 x 512 x 4864 (fp32 out)
// = 55 % of the model's FLOPs, for which the 128x128-tile kernel (vct_gemm_bf16_kernel.h) sits at 600-820 TFLOP/s.
//
// Why another kernel.  The PMC passes of rounds 1-2 show every tile shape of the general kernel bounded by the rate at which a
// CU pulls operand tiles from its XCD's L2 into LDS (17-26 B/clk/CU), not by MFMA issue: throughput is FLOP-per-fetched-byte
// times that rate.  A 128x128 tile has 64 FLOP/B; 256x256 has 128.  A 256x256 tile needs the whole CU (128 KB of LDS for two
// 64-deep K stages, 128 accumulator registers per lane), so nothing else is resident to hide its fill, drain and epilogue --
// with K = 512 (8 K stages per output tile) those are a third of a tile's life.  Hence:
//   * PERSISTENT workgroups (one per CU) walk a flat stream of (work item, K stage) steps; the LDS double buffer never drains at
//     an item boundary: the first stage of the NEXT item is in flight while the current item's epilogue runs;
//   * one barrier per K stage: [own DMA landed] -> barrier -> issue the next stage's DMA -> 64 MFMAs per wave from the current one;
//   * 8 waves as 2 (M) x 4 (N): a wave owns 128 x 64 of C = 8 x 4 MFMA tiles (128 VGPRs); fragments are re-read per 32-deep
//     k-step (48 VGPRs) to stay inside the 256-register budget of two waves per SIMD;
//   * operands go HBM/L2 -> LDS by global_load_lds_dwordx4 into the swizzled images of the general kernel (dma_tile / frag2:
//     K-contiguous operands by ds_read_b128, M/N-contiguous ones by the LDS transpose read); a ragged last K stage takes the
//     zero-filling register path (tail_tile);
//   * accumulators hold C transposed per MFMA tile (operands swapped), i.e. four consecutive COLUMNS per lane: the epilogue
//     moves 8 / 16 bytes per LDS write into a row slab that lives in the stage just consumed, and every global store
//     instruction writes whole 512-byte / 1-KB output rows (plain write-back stores, see the note on the output store policy in gemm256_try);
//   * XCD-aware order: the 32 workgroups of an XCD walk a contiguous run of work items with M fastest.
// Measured at cfg-B (same box, tools/gen_fwd_bench.py): NT 176-186 us vs 234-250 us for the 128x128 kernel.  Ablation of the NT
// form (VCT_GEMM256_DBG): operand DMA alone 82 us (26 B/clk/CU), MFMA + fragment reads alone 116 us, epilogue ~45 us = the
// per-CU store issue rate (~14 B/clk): with one workgroup per CU nothing overlaps it, de-phasing the workgroups or draining
// the stores behind a counted vmcnt changes nothing.
#include "vct_gemm_bf16_kernel.h"
#include <cstring>
#include <cstdio>
#include <mutex>
#include <unordered_map>

namespace vct {

struct G256P {
  const bf16_t* A; const bf16_t* B; void* C;
  long lda, ldb, ldc;
  int M, N, K;
  int tiles_m, tiles_n, split, kt_per_split;
  const float* bias;            // NT form: + bias[n]
  float* bias_grad;             // TN form: [M] row sums of op(A) (db = column sums of dlogits), n-tile 0 only
  float* partial;               // split > 1: fp32 partials [split][M][N] instead of C
  int nt_store;
  int order;                    // tile order inside the flat work stream (see item())
  int zmajor;                   // split > 1: K split outermost (tiles that share the A rows of one K range are neighbours)
  int dbg;                      // experiments (VCT_GEMM256_DBG): 1 = no MFMA work, 2 = no operand DMA after the first stage, 4 = no epilogue
  AdamEpiP adam;                // weight-gradient form on g32_kernel only (vct_gemm_adam); param == nullptr: off
  int pf_dist;                  // g32_kernel: L2 prefetch of the A operand this many K stages ahead (0: off)
};

constexpr int G256_BM = 256, G256_BN = 256;
constexpr int G256_STAGE = (G256_BM + G256_BN) * 128;              // bytes per K stage (64-deep): 64 KB
constexpr int G256_LDS = 2 * G256_STAGE;                           // the epilogue's row slab lives in the stage that was just consumed

template <int TA, int TB, typename TO>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const G256P p) {
  constexpr bool A_MC = (TA == 1), B_MC = (TB == 0), KSPLIT = A_MC && B_MC;
  constexpr bool BG = (TA == 1 && TB == 0);                        // bias gradient exists in the weight-gradient form only
  constexpr int NW = 8, NT = 512, WM = 128, TM = 8, TN = 4;
  constexpr int ES = (int)sizeof(TO);                              // slab / output element size (2: bf16, 4: fp32)
  constexpr int RPR = G256_STAGE / (G256_BN * ES);                 // slab rows per round: 128 (bf16) / 64 (fp32)
  constexpr int IPR = RPR / 32;                                    // MFMA tile rows per wave and round: 4 / 2
  constexpr int CPRW = G256_BN * ES / 16;                          // 16-byte chunks per slab row: 32 / 64
  constexpr int CPT = RPR * CPRW / NT;                             // chunks per thread and round: 8
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int c16 = lane & 15, g4 = (lane >> 4) * 4;

  // ---- work items = (tile, K split), tile-major with M fastest; XCD x owns the contiguous run [x*per, (x+1)*per) ----
  const int nitems = p.tiles_m * p.tiles_n * p.split;
  const int nxw = (int)gridDim.x >> 3;                       // workgroups per XCD (grid is a multiple of 8)
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int w_begin = xcd * per, w_end = min(nitems, w_begin + per);
  const int nkt = (p.K + BK2 - 1) / BK2, kt_full = p.K / BK2;

  auto item = [&](int w, int& m0, int& n0, int& z, int& k_lo, int& k_hi) {
    int tile;
    if (p.zmajor) { const int nt = p.tiles_m * p.tiles_n; z = w / nt; tile = w - z * nt; }
    else { tile = w / p.split; z = w - tile * p.split; }
    if (p.order == 0) {                                      // M fastest
      m0 = (tile % p.tiles_m) * G256_BM; n0 = (tile / p.tiles_m) * G256_BN;
    } else {                                                 // groups of 8 tile columns, inside a group N fastest, then M: the 32
      const int per_group = 8 * p.tiles_m;                   // workgroups of an XCD hold 4 x 8 tiles = 1 MB of A + 2 MB of B at a
      const int grp = tile / per_group, rem = tile - grp * per_group;   // time and the 8 B panels stay for the whole sweep over M
      const int gw = min(8, p.tiles_n - grp * 8);
      m0 = (rem / gw) * G256_BM; n0 = (grp * 8 + rem % gw) * G256_BN;
    }
    k_lo = z * p.kt_per_split; k_hi = min(nkt, k_lo + p.kt_per_split);
  };
  auto issue = [&](int m0, int n0, int kt, int buf) {
    unsigned char* nb = lds + buf * G256_STAGE;
    if (kt < kt_full) {
      dma_tile<A_MC, G256_BM, NW>(nb, p.A, p.lda, m0, p.M, kt * BK2, wave, lane);
      dma_tile<B_MC, G256_BN, NW>(nb + G256_BM * 128, p.B, p.ldb, n0, p.N, kt * BK2, wave, lane);
    } else {      // ragged last stage: zero-filling register path into the same swizzled images
      tail_tile<A_MC, G256_BM, NT>(nb, p.A, p.lda, m0, p.M, kt * BK2, p.K, tid);
      tail_tile<B_MC, G256_BN, NT>(nb + G256_BM * 128, p.B, p.ldb, n0, p.N, kt * BK2, p.K, tid);
    }
  };

  // accumulators hold C TRANSPOSED per MFMA tile (operands swapped): lane = row i*16 + (lane & 15) of the wave's piece and
  // FOUR CONSECUTIVE columns j*16 + (lane >> 4)*4 + r
  f32x4 acc[TM][TN];
  float accb[BG ? TM : 1];          // bias gradient: per-lane partial row sums of the A fragments (VALU: 8 registers; an extra
                                    // MFMA against a ones fragment would cost 32 and spill)
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < (BG ? TM : 1); i++) accb[i] = 0.0f;

  int w = w_begin + slot;
  int m0 = 0, n0 = 0, z = 0, k_lo = 0, k_hi = 0;
  if (w < w_end) { item(w, m0, n0, z, k_lo, k_hi); issue(m0, n0, k_lo, 0); }
  int buf = 0;
  for (; w < w_end; w += nxw) {
    // the NEXT item's coordinates (its first stage is issued during this item's last step)
    int m1 = 0, n1 = 0, z1 = 0, k1_lo = 0, k1_hi = 0;
    const bool have_next = w + nxw < w_end;
    if (have_next) item(w + nxw, m1, n1, z1, k1_lo, k1_hi);
    const bool do_bg = BG && p.bias_grad != nullptr && n0 == 0 && wn == 0;
    for (int kt = k_lo; kt < k_hi; kt++) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the current stage has landed
      __builtin_amdgcn_s_barrier();                          // ... everyone's has; everyone is done with the other buffer
      asm volatile("" ::: "memory");
      // The next stage's operands (this item's, or the first stage of the NEXT item, which then flies under this item's tail and
      // epilogue).  A full stage goes by DMA, ONE instruction after each row tile's MFMAs of the first k-step: issued back to back
      // at the top of the step, the 64 wave-instructions of a stage (1 KiB each) queue up behind the CU's vector-memory issue port
      // and every wave sits in "issue" for the whole transfer before it gets to its fragment reads (K loop 14.1 us per tile with
      // 7 us of MFMA work in it).  A ragged stage takes the register path up front, as before.
      const bool nx_own = kt + 1 < k_hi;
      const int nx_m = nx_own ? m0 : m1, nx_n = nx_own ? n0 : n1, nx_kt = nx_own ? kt + 1 : k1_lo;
      const bool nx_any = (nx_own || have_next) && !(p.dbg & 2);
      const bool nx_dma = nx_any && nx_kt < kt_full;
      if (nx_any && !nx_dma) issue(nx_m, nx_n, nx_kt, buf ^ 1);
      unsigned char* nb = lds + (buf ^ 1) * G256_STAGE;
      const unsigned char* la = lds + buf * G256_STAGE;
      const unsigned char* lb = la + G256_BM * 128;
      if (!(p.dbg & 1))
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) fa[i] = frag2<A_MC, G256_BM, KSPLIT>(la, wm * WM + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < TN; j++) fb[j] = frag2<B_MC, G256_BN, KSPLIT>(lb, wn * 64 + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
          if (ks == 0 && nx_dma) {        // (one per TWO row tiles over both k-steps: no better)
            // (`nt` on either operand's loads: +10 us, same fetch traffic -- tools/g256_traffic.sh)
            if (i < TM / 2) dma_piece<A_MC, G256_BM, NW>(nb, p.A, p.lda, nx_m, p.M, nx_kt * BK2, wave, lane, i);
            else dma_piece<B_MC, G256_BN, NW>(nb + G256_BM * 128, p.B, p.ldb, nx_n, p.N, nx_kt * BK2, wave, lane, i - TM / 2);
          }
        }
        if constexpr (BG) {
          if (do_bg) {
#pragma unroll
            for (int i = 0; i < TM; i++) {
              const s16x8 v = __builtin_bit_cast(s16x8, fa[i]);
              float t = 0.0f;
#pragma unroll
              for (int u = 0; u < 8; u++) t += bf2f((bf16_t)v[u]);
              accb[i] += t;
            }
          }
        }
      }
      buf ^= 1;
    }
    // ---- epilogue.  The stage just consumed (buf ^ 1 after the flip) is free until the NEXT step issues into it, and the next
    // item's first stage is already in flight in the other one: the row slab lives there. ----
    unsigned char* slab = lds + (buf ^ 1) * G256_STAGE;
    const bool part = p.partial != nullptr;
    float bj[TN][4];
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int col = n0 + wn * 64 + j * 16 + g4;
#pragma unroll
      for (int r = 0; r < 4; r++) bj[j][r] = (p.bias != nullptr && !part) ? p.bias[min(col + r, p.N - 1)] : 0.0f;
    }
    if constexpr (BG) {
      if (do_bg) {                                           // fold the four k-groups of lanes that share a row
#pragma unroll
        for (int i = 0; i < TM; i++) {
          float t = accb[i];
          t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
          const int row = m0 + wm * WM + i * 16 + c16;
          if (g4 == 0 && row < p.M) (part ? p.partial + (size_t)p.split * p.M * p.N + (size_t)z * p.M : p.bias_grad)[row] = t;
        }
      }
#pragma unroll
      for (int i = 0; i < TM; i++) accb[i] = 0.0f;
    }
    float* pc = part ? p.partial + (size_t)z * (size_t)p.M * (size_t)p.N : nullptr;
    const long ldo = part ? (long)p.N : p.ldc;
    lds_barrier();                                           // every wave has finished its fragment reads of this stage
    if (!(p.dbg & 4))
    static_for<TM / IPR>([&](auto RD) {
      constexpr int rd = decltype(RD)::value;
      if constexpr (rd > 0) lds_barrier();                   // the slab has been read out by everyone
#pragma unroll
      for (int ii = 0; ii < IPR; ii++) {
        const int sr = (wm * IPR + ii) * 16 + c16;           // slab row; 16-byte chunk c of row r sits at c ^ (r & 31)
#pragma unroll
        for (int j = 0; j < TN; j++) {
          const int e0 = wn * 64 + j * 16 + g4;              // first of this lane's four consecutive columns
          if constexpr (ES == 2) {
            struct alignas(8) B4 { bf16_t e[4]; } v;
#pragma unroll
            for (int r = 0; r < 4; r++) v.e[r] = f2bf(acc[rd * IPR + ii][j][r] + bj[j][r]);
            const int g8 = e0 >> 2;                          // 8-byte granule
            *reinterpret_cast<B4*>(slab + sr * (G256_BN * 2) + ((((g8 >> 1) ^ (sr & 31)) << 1) | (g8 & 1)) * 8) = v;
          } else {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = acc[rd * IPR + ii][j][r] + bj[j][r];
            *reinterpret_cast<f32x4*>(slab + sr * (G256_BN * 4) + (((e0 >> 2) ^ (sr & 31)) << 4)) = v;
          }
        }
      }
      lds_barrier();
#pragma unroll 2
      for (int q = 0; q < CPT; q++) {
        const int cid = q * NT + tid;
        const int sr = cid / CPRW, c = cid % CPRW;
        const int row = m0 + (sr / (IPR * 16)) * WM + (rd * IPR + ((sr >> 4) % IPR)) * 16 + (sr & 15);
        constexpr int EPC = 16 / ES;                         // elements per chunk
        const int col = n0 + c * EPC;
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab + sr * (G256_BN * ES) + ((c ^ (sr & 31)) << 4));
        if (row < p.M && col < p.N) {
          TO* dst = (part ? reinterpret_cast<TO*>(pc) : reinterpret_cast<TO*>(p.C)) + (size_t)row * ldo + col;
          if (col + EPC <= p.N && (ldo % EPC) == 0) {
            if (p.nt_store == 1) store_stream16(dst, v);
            else if (p.nt_store == 2) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
            else *reinterpret_cast<u32x4*>(dst) = v;
          } else {
            const TO* e = reinterpret_cast<const TO*>(&v);
            for (int qq = 0; qq < EPC; qq++) if (col + qq < p.N) dst[qq] = e[qq];
          }
        }
      }
    });
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};
    m0 = m1; n0 = n1; z = z1; k_lo = k1_lo; k_hi = k1_hi;
    // the next step's barrier (after its vmcnt wait) orders the last slab reads before the DMA that reuses this stage
  }
}

// Persistent grid = the compute units the STREAM may use (one workgroup per CU, a multiple of 8 so that blockIdx & 7 stays the
// XCD): 256 on an unmasked MI355X stream, fewer on a CU-masked one (vct_stream_create_masked) -- a 256-workgroup grid there
// would queue two or more 128 KB-LDS workgroups behind each other on every allowed CU.  Looked up once per stream.
int persistent_grid(hipStream_t st) {
  static std::mutex mu;
  static std::unordered_map<void*, int> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find((void*)st);
  if (it != cache.end()) return it->second;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
  }
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (st != nullptr && hipExtStreamGetCUMask(st, 8, mask) == hipSuccess) {
    int n = 0;
    for (int i = 0; i < 8; i++) n += __builtin_popcount(mask[i]);
    if (n > 0 && n < cus) cus = n;
  }
  (void)hipGetLastError();                                   // a refused query must not surface at the next launch check
  const int grid = cus >= 8 ? (cus & ~7) : 8;
  cache.emplace((void*)st, grid);
  return grid;
}

}  // namespace vct
#include "vct_gemm32_kernel.h"      // g32_kernel: the same persistent tile with a software-pipelined K loop on 32x32x16 (round 6)
namespace vct {

// VCT_GEMM32: which forms run on the pipelined kernel (1 = NT, 2 = NN, 4 = TN, 8 = NT split over K).  Default 6: the vocabulary dX
// (NN, 0.218 -> 0.182 ms in the step) and the vocabulary dW (TN, with the optimizer epilogue: 0.333 -> 0.22 ms); the NT form is
// slower on the vocabulary projection, alone at sustained clocks and in the step.
// gemm256_kernel takes what the pipelined kernel does not (fewer than two full K stages per work item).
static bool g32_takes(int form_bit) {
  static const char* env32 = getenv("VCT_GEMM32");
  const int mask32 = env32 != nullptr ? atoi(env32) : 6;
  return (mask32 & form_bit) != 0;
}

template <int TA, int TB, typename TO> static int g256_launch(const G256P& p, hipStream_t st) {
  constexpr int form_bit = (TA == 1) ? 4 : (TB == 0 ? 2 : (sizeof(TO) == 4 ? 8 : 1));
  if (g32_takes(form_bit) && p.dbg == 0 && g32_eligible(p, TA == 1, TB == 0)) return g32_launch<TA, TB, TO, 0>(p, st);
  if (p.adam.param != nullptr) return VCT_E_ARG;             // (gemm256_try never gets here: the round-5 kernel has no optimizer epilogue)
  static vct::DynLdsOptIn optin;
  if (hipError_t e = optin.ensure((const void*)gemm256_kernel<TA, TB, TO>, G256_LDS); e != hipSuccess) return (int)e;
  vct::launch(gemm256_kernel<TA, TB, TO>, dim3(persistent_grid(st)), dim3(512), (size_t)G256_LDS, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

// Eligibility + launch (called from vct_gemm).  *used = true when the product was issued here; *reduce_split > 1 tells the
// caller to run its split-K reduce over `workspace` ([split][M][N] fp32 partials).
int gemm256_try(const vct_gemm_desc* d, hipStream_t st, bool* used, int* reduce_split) {
  *used = false;
  *reduce_split = 1;
  // VCT_GEMM256: 0 disables; else a mask 1 = NT, 2 = NN, 4 = TN, 8 = NT split over K (narrow output).  Default 15 (round 6: TN too).
  // With the DMA instructions spread between the MFMA groups (and their addresses no longer hoisted: no spills in any form) the same
  // box gives NT 184 vs 250, NN 223 vs 248, TN 182 vs 192 us against the 128x128 kernel; in the STEP the TN form (the weight gradient,
  // which runs beside the encoder backward on the other stream) makes things worse -- one workgroup per CU with 128 KB of LDS leaves
  // no room for the co-scheduled kernels: 2.52 vs 2.48 ms -- so it stays off.  History: measured on the same box (tools/gemm_bench.py
  // --auto, cfg-B vocabulary shapes) NT 214 vs 253 us on the 128x128 kernel, but NN 260 vs 249 and TN 222 vs 193 -- the forms
  // whose operands go through the LDS transpose read need twice the fragment reads and spill at 256 VGPRs; they stay available
  // (and tested) behind the mask.
  static const char* env = getenv("VCT_GEMM256");
  const int mask = env != nullptr ? atoi(env) : 15;
  if (mask == 0 || d->dtype != VCT_BF16 || (d->reserved != 0 && d->reserved < 99)) return VCT_OK;
  if (d->act != VCT_ACT_NONE || d->preact || d->addend || d->dact_src || (d->seed && d->p_drop > 0.0f)) return VCT_OK;
  const int form = d->ta * 2 + d->tb;                         // 1 NT, 0 NN, 2 TN
  if (d->K < 256) return VCT_OK;
  G256P p;
  memset(&p.adam, 0, sizeof(p.adam));
  p.A = reinterpret_cast<const bf16_t*>(d->A); p.B = reinterpret_cast<const bf16_t*>(d->B); p.C = d->C;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.tiles_m = (d->M + G256_BM - 1) / G256_BM; p.tiles_n = (d->N + G256_BN - 1) / G256_BN;
  p.bias = d->bias; p.bias_grad = d->bias_grad; p.partial = nullptr;
  p.split = 1;
  const int nkt = (d->K + 63) / 64;
  p.kt_per_split = nkt;
  // Output store policy of the 297 MB of logits (VCT_GEMM_NT: 0 plain = default, 1 agent scope `sc1`, 2 `nt`).  Round 4, PMC passes
  // (tools/g256_traffic.sh): plain and `nt` stores both allocate in the XCD's write-back L2, the logits evict the operand panels
  // between two rounds of tiles, and every operand line is fetched once per round: 257 MB of reads for 36 MB of operands (L2 hit rate
  // 0.69 = exactly "shared by the 4 / 8 workgroups of a round, nothing kept").  Agent-scope stores leave the panels alone: 141 MB of
  // reads (total fabric traffic 438 MB = 1.31 x algorithmic), same time back to back (183 vs 185 us) -- but inside the step the
  // launch then ends only when its write-through has drained: 0.207 ms against 0.178 (plain) / 0.183 (`nt`), whole step +20 us.
  // The re-fetches come from the 256 MB memory-side cache, not from HBM; the time is what counts: plain stores.
  p.nt_store = 0;
  { static const char* ntenv = getenv("VCT_GEMM_NT"); if (ntenv != nullptr) p.nt_store = atoi(ntenv); }
  { static const char* oenv = getenv("VCT_GEMM256_ORDER"); p.order = oenv != nullptr ? atoi(oenv) : 1; }
  static const char* denv = getenv("VCT_GEMM256_DBG");
  p.dbg = denv != nullptr ? atoi(denv) : 0;
  // L2 prefetch distance of the pipelined kernel (K stages; vct_gemm32_kernel.h): the NN form (vocabulary dX) gains 4-8 % alone at 2-4,
  // the TN form nothing (profiles/r06_g32_prefetch_probe.txt).  VCT_G32_PREFETCH="nn,tn" overrides.
  {
    static const char* penv = getenv("VCT_G32_PREFETCH");
    int pf_nn = 3, pf_tn = 0;
    if (penv != nullptr) sscanf(penv, "%d,%d", &pf_nn, &pf_tn);
    p.pf_dist = form == 0 ? pf_nn : (form == 2 ? pf_tn : 0);
  }
  const long tiles = (long)p.tiles_m * p.tiles_n;
  p.zmajor = 0;
  if (form == 1 && (mask & 8) && d->out_dtype == VCT_BF16 && !d->bias && !d->bias_grad && d->split_k != 1 && d->workspace != nullptr &&
      tiles <= 512 && nkt >= 128 && d->M >= 2048) {
    // NT with a vocabulary-long K and a narrow output (dX = dlogits W_g through the transposed weight shadow): split over K so that
    // every CU gets one item, fp32 partials + reduce.  K split outermost, N fastest inside: the tiles that read the same rows of
    // the 297 MB A operand sit on neighbouring CUs of one XCD and pull them from HBM once.
    // the split that minimises (rounds of items over the CUs) x (K stages per item) among those whose partials fit the workspace:
    // 38 tiles (cfg-B) -> 6 (228 items, one round of 80 stages); 152 tiles (batch 1024) -> 5 (760 items, three rounds of 96)
    int split = d->split_k > 1 ? d->split_k : 0;
    if (split == 0) {
      const long ncu = persistent_grid(st);
      long best = -1;
      for (int sp = 2; sp <= 16; sp++) {
        if ((int64_t)sp * d->M * d->N * 4 > d->workspace_bytes) break;
        const long per = (nkt + sp - 1) / sp, cost = ((tiles * sp + ncu - 1) / ncu) * per;
        if (best < 0 || cost < best) { best = cost; split = sp; }
      }
    }
    if (split >= 2) {
      p.kt_per_split = (nkt + split - 1) / split;
      p.split = (nkt + p.kt_per_split - 1) / p.kt_per_split;
      if ((int64_t)p.split * d->M * d->N * 4 <= d->workspace_bytes) {
        p.partial = reinterpret_cast<float*>(d->workspace);
        p.nt_store = 0; p.zmajor = 1; p.order = 1;
        const int rc = g256_launch<0, 1, float>(p, st);
        if (rc == VCT_OK) { *used = true; *reduce_split = p.split; }
        return rc;
      }
      p.split = 1; p.kt_per_split = nkt; p.partial = nullptr;
    }
  }
  if (form == 1) {             // NT: vocabulary projection forward (bf16 out, bias)
    if (!(mask & 1) || d->out_dtype != VCT_BF16 || d->bias_grad || d->split_k > 1) return VCT_OK;
    // vocabulary-wide outputs, or any product of >= 240 tiles with >= 16 K stages (one round of persistent workgroups at least; 4096^3:
    // 1190-1250 TF here against 960 on the 128x128 kernel) -- the layer products (<= 152 tiles of 256x256) stay where they are
    const bool wide = d->N >= 8192, big = tiles >= 240 && nkt >= 16;
    if (d->M < 2048 || !(wide || big) || (d->K % 8) || (d->ldc % 8) || ((uintptr_t)d->C & 15)) return VCT_OK;
    const int rc = g256_launch<0, 1, bf16_t>(p, st);
    if (rc == VCT_OK) *used = true;
    return rc;
  }
  if (form == 2) {             // TN: weight gradient (fp32 out, bias gradient), one item per tile, no split
    if (!(mask & 4) || d->out_dtype != VCT_F32 || d->bias || d->split_k > 1) return VCT_OK;
    if (tiles < 192 || tiles > 256 || nkt < 32 || (d->ldc % 4) || ((uintptr_t)d->C & 15)) return VCT_OK;
    memset(&p.adam, 0, sizeof(p.adam));
    if (d->adam != nullptr) {
      // optimizer epilogue (vct_gemm_adam): in the pipelined kernel only, whole 4-element chunks, no stream-order packed copy
      // (the vocabulary weight has none); anything else stays with the 128 x 128 kernel
      const vct_gemm_adam* a = d->adam;
      if (!g32_takes(4) || (d->N % 4) || a->pk_stream != nullptr || !g32_eligible(p, true, true)) return VCT_OK;
      p.adam.param = a->param; p.adam.m = a->exp_avg; p.adam.v = a->exp_avg_sq;
      p.adam.shadow = reinterpret_cast<uint16_t*>(a->shadow); p.adam.ld_shadow = (long)a->ld_shadow;
      p.adam.store_grad = a->store_grad; p.adam.hyper = a->hyper; p.adam.step = a->step;
    }
    const int rc = g256_launch<1, 0, float>(p, st);
    if (rc == VCT_OK) *used = true;
    return rc;
  }
  if (form == 0) {             // NN: dX = dY W over a vocabulary-long K: split so that every CU gets one item, fp32 partials
    if (!(mask & 2) || d->out_dtype != VCT_BF16 || d->bias || d->bias_grad || d->split_k == 1) return VCT_OK;
    if (tiles > 128 || nkt < 128 || d->workspace == nullptr) return VCT_OK;
    int split = (int)(256 / tiles);
    if (d->split_k > 1) split = d->split_k;
    if (split < 2) return VCT_OK;
    p.kt_per_split = (nkt + split - 1) / split;
    p.split = (nkt + p.kt_per_split - 1) / p.kt_per_split;
    if ((int64_t)p.split * d->M * d->N * 4 > d->workspace_bytes) return VCT_OK;
    p.partial = reinterpret_cast<float*>(d->workspace);
    p.nt_store = 0;
    {   // K split outermost, N fastest inside: the two N tiles that stream the same 2.6 MB slab of dlogits are neighbours on one XCD
      static const char* zenv = getenv("VCT_GEMM256_ZMAJOR");
      p.zmajor = zenv != nullptr ? (zenv[0] == '1') : 1;
      if (p.zmajor) p.order = 1;
    }
    const int rc = g256_launch<0, 0, float>(p, st);          // TO = float: the slab / partial element type
    if (rc == VCT_OK) { *used = true; *reduce_split = p.split; }
    return rc;
  }
  return VCT_OK;
}

}  // namespace vct
