// GEMM entry point of the C ABI (vct_gemm): planning (tile / wave / split-K choice), dispatch and the
// fp32 PARITY-MODE kernel.  C[M,N] = epilogue(op(A)[M,K] * op(B)[K,N]).
//
//  * bf16 inputs (throughput mode) go to gemm_bf16_v2_kernel (vct_gemm_bf16_kernel.h: LDS-DMA double buffer,
//    swizzled LDS images, LDS transpose reads, 4- or 8-wave tiles).
//  * f32 inputs: the kernel below -- v_mfma_f32_16x16x4_f32 (an exact fp32 fma chain), BK = 16, 256-thread
//    workgroups (2x2 waves), tile 128x128 or 64x64, register-staged prefetch of the next K tile.  Operands whose
//    contiguous dimension is the non-reduced one (A of dW = dY^T X, B of dX = dY W) are transposed while being
//    written to LDS, so the compute loop is layout independent.  Performance is secondary here: this mode exists
//    for <= 1e-3 parity with the CPU reference and bit-exact greedy-decode ids.
//  * both share the fused epilogue (bias, GELU/ReLU + saved pre-activation, counter-hash dropout, residual-gradient
//    accumulate, activation-derivative multiply, bias gradient via one extra MFMA against a ones fragment) staged
//    through a per-wave fp32 LDS transpose so that every lane stores 16 contiguous bytes, and the deterministic
//    split-K second pass (splitk_reduce_kernel).
#include <cstdlib>
#include <cstring>
#include "vct_common.h"
#include "vct_gemm_params.h"

namespace vct {

template <typename TI> struct GemmCfg;
template <> struct GemmCfg<float> { static constexpr int BK = 16, VEC = 4, KPAD = 4; };

// one global->register vector (16 bytes) ---------------------------------------------------------
struct alignas(16) Vec16 { uint32_t w[4]; };
__device__ __forceinline__ Vec16 vec_zero() { Vec16 v; v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0u; return v; }

// Operand staging.  KC = stored [rows][K] (K contiguous); MC = stored [K][rows] (rows contiguous).
// R = rows of the tile (BM or BN).  LDS image [R][BK+4] fp32 in both cases (MC is transposed while storing).
template <typename TI, bool MC, int R>
struct Stager {
  using Cfg = GemmCfg<TI>;
  static constexpr int BK = Cfg::BK, VEC = Cfg::VEC;
  static constexpr bool BF = sizeof(TI) == 2;
  static constexpr int NV = (R * BK / VEC) / 256;  // vectors per thread
  static constexpr int KC_STRIDE = BK + Cfg::KPAD;
  static constexpr int LDS_ELEMS = R * KC_STRIDE;
  static_assert(NV >= 1, "tile too small for 256 threads");

  // r_ext: valid rows; K: valid reduction length; reads along the contiguous dim are predicated per
  // 16-byte vector (extent rounded up to VEC -- the producer zero-pads / leading dim covers it).
  __device__ __forceinline__ static void load(Vec16 (&regs)[NV], const TI* __restrict__ base, long ld, int r0,
                                              int r_ext, int k0, int K) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int v = tid + i * 256;
      if constexpr (!MC) {
        constexpr int VPR = BK / VEC;
        const int row = v / VPR, kc = (v % VPR) * VEC;
        const int gr = r0 + row, gk = k0 + kc;
        regs[i] = (gr < r_ext && gk < K) ? *reinterpret_cast<const Vec16*>(base + (long)gr * ld + gk) : vec_zero();
      } else {
        constexpr int VPR = R / VEC;
        const int krow = v / VPR, rc = (v % VPR) * VEC;
        const int gk = k0 + krow, gr = r0 + rc;
        regs[i] = (gk < K && gr < r_ext) ? *reinterpret_cast<const Vec16*>(base + (long)gk * ld + gr) : vec_zero();
      }
    }
  }
  __device__ __forceinline__ static void store(const Vec16 (&regs)[NV], TI* lds) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int v = tid + i * 256;
      if constexpr (!MC) {
        constexpr int VPR = BK / VEC;
        const int row = v / VPR, kc = (v % VPR) * VEC;
        *reinterpret_cast<Vec16*>(lds + row * KC_STRIDE + kc) = regs[i];
      } else {
        constexpr int VPR = R / VEC;
        const int krow = v / VPR, rc = (v % VPR) * VEC;
#pragma unroll
        for (int j = 0; j < 4; j++) reinterpret_cast<uint32_t*>(lds)[(rc + j) * KC_STRIDE + krow] = regs[i].w[j];
      }
    }
  }
};

template <typename TI, typename TO, int TA, int TB, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
  using Cfg = GemmCfg<TI>;
  constexpr bool BF = sizeof(TI) == 2;
  constexpr int BK = Cfg::BK;
  constexpr bool A_MC = (TA == 1), B_MC = (TB == 0);
  using SA = Stager<TI, A_MC, BM>;
  using SB = Stager<TI, B_MC, BN>;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;

  // one LDS block: [A tile | B tile] during the K loop, then per-wave fp32 staging slabs for the epilogue
  constexpr int STG_STRIDE = WN + 4;                       // floats per staged row
  constexpr int STG_BYTES = 4 * 16 * STG_STRIDE * 4;       // 4 waves x 16 rows
  constexpr int TILE_BYTES = (SA::LDS_ELEMS + SB::LDS_ELEMS) * (int)sizeof(TI);
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[TILE_BYTES > STG_BYTES ? TILE_BYTES : STG_BYTES];
  TI* lds_a = reinterpret_cast<TI*>(lds_raw);
  TI* lds_b = lds_a + SA::LDS_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = (p.K + BK - 1) / BK;
  const int kt_begin = blockIdx.z * p.kt_per_split;
  const int kt_end = min(nkt, kt_begin + p.kt_per_split);

  const TI* A = reinterpret_cast<const TI*>(p.A);
  const TI* B = reinterpret_cast<const TI*>(p.B);

  f32x4 acc[TM][TN];
  f32x4 accb[TM];
#pragma unroll
  for (int i = 0; i < TM; i++) {
    accb[i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};
  }
  const bool do_bias_grad = (p.bias_grad != nullptr) && (tile_n == 0) && (wn == 0);

  Vec16 ra[SA::NV], rb[SB::NV];
  if (kt_begin < kt_end) {
    SA::load(ra, A, p.lda, m0, p.M, kt_begin * BK, p.K);
    SB::load(rb, B, p.ldb, n0, p.N, kt_begin * BK, p.K);
  }
  for (int kt = kt_begin; kt < kt_end; kt++) {
    __syncthreads();  // previous tile's fragment reads are done
    SA::store(ra, lds_a);
    SB::store(rb, lds_b);
    __syncthreads();
    if (kt + 1 < kt_end) {  // next tile's global loads fly under the MFMAs below
      SA::load(ra, A, p.lda, m0, p.M, (kt + 1) * BK, p.K);
      SB::load(rb, B, p.ldb, n0, p.N, (kt + 1) * BK, p.K);
    }
    {
      static_assert(!BF, "bf16 inputs use gemm_bf16_v2_kernel");
      const float* la = reinterpret_cast<const float*>(lds_a);
      const float* lb = reinterpret_cast<const float*>(lds_b);
      const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
      for (int kk = 0; kk < BK / 4; kk++) {
        float fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) fa[i] = la[(wm * WM + i * 16 + i16) * SA::KC_STRIDE + kk * 4 + g];
#pragma unroll
        for (int j = 0; j < TN; j++) fb[j] = lb[(wn * WN + j * 16 + i16) * SB::KC_STRIDE + kk * 4 + g];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (do_bias_grad) {
#pragma unroll
          for (int i = 0; i < TM; i++) accb[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], 1.0f, accb[i], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------
  // The C fragment of a 16x16 MFMA holds, per lane, ONE column (lane & 15) and four rows
  // ((lane >> 4) * 4 + r): storing it directly means 2/4-byte scattered stores.  Each wave instead
  // transposes one 16-row slab of its sub-tile through a private fp32 LDS slab and every lane then
  // owns VO consecutive columns of one row: bias / activation / dropout / addend / derivative are
  // applied on that vector and it is stored with ONE 16-byte instruction.
  const int c16 = lane & 15, g4 = (lane >> 4) * 4;
  if (do_bias_grad && c16 == 0) {
    float* bg = p.partial != nullptr ? p.bias_partial + (size_t)blockIdx.z * p.M : p.bias_grad;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = m0 + wm * WM + i * 16 + g4 + r;
        if (row < p.M) bg[row] = accb[i][r];
      }
  }
  __syncthreads();  // every wave is done with the operand tiles: LDS becomes staging space
  float* stg = reinterpret_cast<float*>(lds_raw) + wave * 16 * STG_STRIDE;
  const bool part = p.partial != nullptr;
  constexpr int VO = 16 / (int)sizeof(TO);          // output elements per 16-byte store
  constexpr int CPR = WN / VO;                      // 16-byte chunks per staged row
  constexpr int CPL = 16 * CPR / 64;                // chunks per lane
  static_assert(CPL >= 1, "staging geometry");
  const Dropout dr = make_dropout(p.seed, p.site, p.p_drop);
  TO* C = reinterpret_cast<TO*>(p.C);
  TO* preact = reinterpret_cast<TO*>(p.preact);
  const TO* addend = reinterpret_cast<const TO*>(p.addend);
  const TO* dact = reinterpret_cast<const TO*>(p.dact);
  float* partC = part ? p.partial + (size_t)blockIdx.z * (size_t)p.M * (size_t)p.N : nullptr;
  const long ldo = part ? (long)p.N : p.ldc;
  const bool vec_ok = (ldo % VO == 0) && (!part ? (((uintptr_t)p.C & 15) == 0) : true) && ((p.N % 4) == 0 || !part);
  // bias for this lane's columns: loaded ONCE, unconditionally (clamped), ahead of the staging loop --
  // a conditional per-element load would make hipcc branch + s_waitcnt vmcnt(0) around every element
  float bvec[CPL][VO];
#pragma unroll
  for (int c = 0; c < CPL; c++)
#pragma unroll
    for (int q = 0; q < VO; q++) bvec[c][q] = 0.0f;
  if (p.bias != nullptr && !part) {   // ONE uniform branch around all the loads
#pragma unroll
    for (int c = 0; c < CPL; c++) {
      const int col = n0 + wn * WN + ((c * 64 + lane) % CPR) * VO;
#pragma unroll
      for (int q = 0; q < VO; q++) bvec[c][q] = p.bias[min(col + q, p.N - 1)];
    }
  }
  struct alignas(16) OutV { TO e[VO]; };
  static_for<TM>([&](auto I) {
    constexpr int i = decltype(I)::value;
    static_for<TN>([&](auto J) {
      constexpr int j = decltype(J)::value;
#pragma unroll
      for (int r = 0; r < 4; r++) stg[(g4 + r) * STG_STRIDE + j * 16 + c16] = acc[i][j][r];
    });
    // same-wave LDS write -> read: ordered by the wave's own lgkmcnt, no barrier needed
    static_for<CPL>([&](auto CI) {
      constexpr int c = decltype(CI)::value;
      const int chunk = c * 64 + lane;
      const int rr = chunk / CPR, cc = (chunk % CPR) * VO;
      const int row = m0 + wm * WM + i * 16 + rr, col = n0 + wn * WN + cc;
      float v[VO];
#pragma unroll
      for (int q = 0; q < VO; q += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(stg + rr * STG_STRIDE + cc + q);
        v[q] = t[0]; v[q + 1] = t[1]; v[q + 2] = t[2]; v[q + 3] = t[3];
      }
      if (row >= p.M || col >= p.N) return;
      const bool full = vec_ok && (col + VO <= p.N);
      if (part) {
        float* dst = partC + (size_t)row * p.N + col;
        if (full) {
#pragma unroll
          for (int q = 0; q < VO; q += 4) *reinterpret_cast<f32x4*>(dst + q) = f32x4{v[q], v[q + 1], v[q + 2], v[q + 3]};
        } else {
          for (int q = 0; q < VO; q++) if (col + q < p.N) dst[q] = v[q];
        }
        return;
      }
      if (full) {
        // fast path: every global access is an unconditional 16-byte vector
        OutV dv, av, ov, pv;
        const bool has_d = dact != nullptr, has_a = addend != nullptr;
        if (has_d) dv = *reinterpret_cast<const OutV*>(dact + (size_t)row * p.ld_dact + col);
        if (has_a) av = *reinterpret_cast<const OutV*>(addend + (size_t)row * p.ld_addend + col);
#pragma unroll
        for (int q = 0; q < VO; q++) {
          float x = v[q] + bvec[c][q];
          pv.e[q] = from_f<TO>(x);
          x = act_f(p.act, x);
          if (has_d) x *= dact_f(p.dact_kind, to_f<TO>(dv.e[q]));
          x *= drop_mult(dr, (uint32_t)row * (uint32_t)p.N + (uint32_t)(col + q));
          if (has_a) x += to_f<TO>(av.e[q]);
          ov.e[q] = from_f<TO>(x);
        }
        *reinterpret_cast<OutV*>(C + (size_t)row * p.ldc + col) = ov;
        if (preact != nullptr) {
          if ((p.ld_preact % VO) == 0) *reinterpret_cast<OutV*>(preact + (size_t)row * p.ld_preact + col) = pv;
          else for (int q = 0; q < VO; q++) preact[(size_t)row * p.ld_preact + col + q] = pv.e[q];
        }
      } else {
        // ragged tail (last partial vector of a row, or unaligned output): element-wise
        for (int q = 0; q < VO; q++) {
          if (col + q >= p.N) break;
          float x = v[q] + bvec[c][q];
          if (preact != nullptr) preact[(size_t)row * p.ld_preact + col + q] = from_f<TO>(x);
          x = act_f(p.act, x);
          if (dact != nullptr) x *= dact_f(p.dact_kind, to_f<TO>(dact[(size_t)row * p.ld_dact + col + q]));
          x *= drop_mult(dr, (uint32_t)row * (uint32_t)p.N + (uint32_t)(col + q));
          if (addend != nullptr) x += to_f<TO>(addend[(size_t)row * p.ld_addend + col + q]);
          C[(size_t)row * p.ldc + col + q] = from_f<TO>(x);
        }
      }
    });
  });
}

// deterministic split-K second pass: C[i] = sum_z partial[z][i] (fixed z order)
template <typename TO>
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, TO* __restrict__ C, long ldc, int M, int N,
                                     int S, const float* __restrict__ bias_partial, float* __restrict__ bias_grad) {
  const size_t total = (size_t)M * N;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int z = 0; z < S; z++) s += partial[(size_t)z * total + idx];
    const int row = (int)(idx / N), col = (int)(idx % N);
    C[(size_t)row * ldc + col] = from_f<TO>(s);
  }
  if (bias_grad != nullptr) {
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < M; row += gridDim.x * blockDim.x) {
      float s = 0.0f;
      for (int z = 0; z < S; z++) s += bias_partial[(size_t)z * M + row];
      bias_grad[row] = s;
    }
  }
}

struct Plan { int bm, bn, nbuf, waves8; int split; int tiles_m, tiles_n, nkt, kt_per; };

static bool gemm_can_split(const vct_gemm_desc* d) {
  return d->bias == nullptr && d->act == VCT_ACT_NONE && d->preact == nullptr &&
         d->addend == nullptr && d->dact_src == nullptr && !(d->seed != nullptr && d->p_drop > 0.0f);
}

// overrides for experiments: desc.reserved = tile + 10 * nbuf; tile 1: 128x128, 2: 128x64, 3: 64x128, 4: 64x64 (0 auto); nbuf 1..3 (0 default)
// Single-pass (in-kernel) split-K reduce: the partials travel as write-through agent-scope stores, which are slow in
// bulk.  Measured on MI355X: a layer's weight gradients (<= 12 MB of partials) gain ~6 us per GEMM from dropping the
// second launch, the vocabulary dX (6 x 4864 x 512 fp32 = 60 MB of partials) LOSES 60 us -- so only small partial sets.
static bool use_counters(const vct_gemm_desc* d, const Plan& pl) {
  const int64_t partial_bytes = (int64_t)pl.split * d->M * d->N * 4;
  return d->tile_counters != nullptr && d->dtype == VCT_BF16 && (long)pl.tiles_m * pl.tiles_n <= (long)d->n_tile_counters &&
         partial_bytes <= ((int64_t)16 << 20);
}
static Plan gemm_plan(const vct_gemm_desc* d, bool have_ws) {
  Plan pl;
  pl.nbuf = 2;
  pl.waves8 = 0;
  const bool bf = d->dtype == VCT_BF16;
  const int BK = bf ? 64 : 16;
  pl.nkt = (d->K + BK - 1) / BK;
  auto tiles = [&](int bm, int bn) { return (long)((d->M + bm - 1) / bm) * ((d->N + bn - 1) / bn); };
  if (bf) {
    static const int cand[7][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {256, 128}, {320, 128}, {64, 128}};
    int pick = 3;
    const int rsv = d->reserved >= 99 ? 0 : d->reserved;      // 99 / 100: persistent-tile kernel off / forced (vct_gemm_pt.hip)
    const int tsel = rsv % 10;
    pl.nbuf = (rsv / 10) ? (rsv / 10) : 2;
    const bool cover_ok = d->out_dtype == VCT_BF16 && d->ta == 0;      // the cover tiles exist for bf16-out NT / NN
    if (tsel == 5) { pick = 0; pl.waves8 = 1; }        // 128x128, 8 waves
    else if (tsel == 8) { pick = 1; pl.waves8 = 4; }   // 128x64, 8 waves
    else if (tsel == 6 && cover_ok) { pick = 4; pl.waves8 = 6; pl.nbuf = 2; }   // 256x128, 8 waves (4 x 2)
    else if (tsel == 7 && cover_ok) { pick = 5; pl.waves8 = 7; pl.nbuf = 2; }   // 320x128, 8 waves (4 x 2)
    else if (tsel == 9 && cover_ok) { pick = 6; pl.waves8 = 9; pl.nbuf = 2; }   // 64x128, 8 waves (2 x 4)
    else if (tsel >= 1 && tsel <= 4) pick = tsel - 1;
    else {
      // measured on MI355X (tools/gemm_bench.py, cfg-B shapes).  The kernel is bound by the rate at which a CU
      // pulls operand tiles from L2 (~17 B/clk with 8 waves per CU issuing DMA, ~26 B/clk with 16-20), so:
      //  * big GEMMs (vocabulary projection and its two gradients): 128x128 tile (64 FLOP/B) with EIGHT waves;
      //  * wide-N, short-K layer GEMMs (QKV / FFN1 forward, FFN2 dX): 128x64 with eight waves;
      //  * everything else (N = 512 outputs, weight gradients): 64x64 with four waves, 5 workgroups per CU.
      pick = 3;
      static const bool legacy = getenv("VCT_GEMM_4WAVE_ONLY") != nullptr;   // A/B switch for the 8-wave variants
      if ((pl.nkt >= 64 && tiles(128, 128) >= 128) || (tiles(64, 64) >= 16384 && d->N >= 4096)) {
        pick = 0; pl.waves8 = legacy ? 0 : 1;
        if (legacy && pl.nkt < 64) pick = 2;
      } else if (!legacy && d->N >= 1024 && d->M >= 2048 && pl.nkt <= 16) { pick = 1; pl.waves8 = 4; }
    }
    pl.bm = cand[pick][0]; pl.bn = cand[pick][1];
  } else {
    pl.bm = pl.bn = (tiles(128, 128) >= 384) ? 128 : 64;
  }
  pl.tiles_m = (d->M + pl.bm - 1) / pl.bm;
  pl.tiles_n = (d->N + pl.bn - 1) / pl.bn;
  const long nt = (long)pl.tiles_m * pl.tiles_n;
  int split = 1;
  if (d->split_k > 1) split = d->split_k;
  else if (d->split_k == 0 && nt < 384 && pl.nkt >= 8) split = (int)((767 + nt) / nt);
  // a GEMM with an epilogue (bias, activation, dropout, addend, derivative) can only split when the reduction happens
  // inside the kernel, where the last workgroup of a tile applies the epilogue to the summed partials
  const bool epilogue = !gemm_can_split(d);
  if (!have_ws || (epilogue && (d->tile_counters == nullptr || !bf))) split = 1;
  // measured (tools/bench_decode.py): splitting the reduction of an epilogue GEMM pays for GEMV-like shapes (decode at
  // batch 1: 217 -> 198 us/token) and costs at batch 128 (241 -> 283 us/step: the write-through partials again)
  if (epilogue && d->split_k == 0 && d->M > 16) split = 1;
  if (split > pl.nkt / 2) split = pl.nkt / 2 > 0 ? pl.nkt / 2 : 1;
  if (split < 1) split = 1;
  pl.kt_per = (pl.nkt + split - 1) / split;
  pl.split = (pl.nkt + pl.kt_per - 1) / pl.kt_per;
  if ((epilogue || d->adam != nullptr) && pl.split > 1 && !use_counters(d, pl)) { pl.split = 1; pl.kt_per = pl.nkt; }   // (the optimizer epilogue lives in the producing kernel: no two-pass reduce)
  return pl;
}

template <typename TI, typename TO, int TA, int TB, int BM>
static void launch_gemm(const GemmP& p, dim3 grid, hipStream_t st) {
  vct::launch((gemm_kernel<TI, TO, TA, TB, BM, BM>), grid, dim3(256), 0, st, p);
}

template <typename TI, typename TO>
static int dispatch_layout(const vct_gemm_desc* d, const GemmP& p, int bm, dim3 grid, hipStream_t st) {
  const int key = d->ta * 2 + d->tb;
#define VCT_L(TA_, TB_)                                              \
  if (bm == 128) launch_gemm<TI, TO, TA_, TB_, 128>(p, grid, st);    \
  else launch_gemm<TI, TO, TA_, TB_, 64>(p, grid, st);               \
  return VCT_OK;
  switch (key) {
    case 1: { VCT_L(0, 1) }  // NT: x W^T
    case 0: { VCT_L(0, 0) }  // NN: dY W
    case 2: { VCT_L(1, 0) }  // TN: dY^T X
    default: return VCT_E_SHAPE;
  }
#undef VCT_L
}

int gemm_bf16_v2_dispatch(const vct_gemm_desc* d, const GemmP& p, int bm, int bn, int nbuf, dim3 grid, hipStream_t st);
int gemm256_try(const vct_gemm_desc* d, hipStream_t st, bool* used, int* reduce_split);   // persistent 256x256 kernel (vct_gemm256.hip)
int gemm_skinny_try(const vct_gemm_desc* d, hipStream_t st, bool* used);                    // M <= 256 rows (vct_gemm_skinny.hip)
int gemm_pt_try(const vct_gemm_desc* d, hipStream_t st, bool* used);                        // persistent 128x128 tiles for the layer GEMMs (vct_gemm256.hip)

}  // namespace vct

using namespace vct;

// workspace layout: [split x M x N partials][split x M bias partials]
static int64_t ws_bytes(const vct_gemm_desc* d, const Plan& pl) {
  if (pl.split <= 1) return 0;
  return (int64_t)pl.split * ((int64_t)d->M * d->N + d->M) * 4;
}

extern "C" int64_t vct_gemm_workspace_bytes(const vct_gemm_desc* d) {
  if (d == nullptr) return 0;
  return ws_bytes(d, gemm_plan(d, true));
}

static int check_desc(const vct_gemm_desc* d) {
  if (d == nullptr || d->A == nullptr || d->B == nullptr || d->C == nullptr) return VCT_E_ARG;
  if (d->dtype != VCT_F32 && d->dtype != VCT_BF16) return VCT_E_ARG;
  if (d->out_dtype != VCT_F32 && d->out_dtype != VCT_BF16) return VCT_E_ARG;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return VCT_E_SHAPE;
  if (d->ta == 1 && d->tb == 1) return VCT_E_SHAPE;
  const int vec = d->dtype == VCT_BF16 ? 8 : 4;
  if (d->lda % vec || d->ldb % vec) return VCT_E_ALIGN;
  if (((uintptr_t)d->A & 15) || ((uintptr_t)d->B & 15)) return VCT_E_ALIGN;
  if (d->dact_src != nullptr && d->act == VCT_ACT_NONE) return VCT_E_ARG;  // dact needs the activation kind
  // bf16 path: the bias gradient is fused into the weight-gradient form only (dW = A^T B, fp32 out)
  if (d->bias_grad != nullptr && d->dtype == VCT_BF16 && !(d->ta == 1 && d->tb == 0 && d->out_dtype == VCT_F32)) return VCT_E_ARG;
  if (d->adam != nullptr) {
    // optimizer epilogue: the bf16 weight-gradient form only, nothing else in the epilogue (the gradient is consumed as produced)
    const vct_gemm_adam* a = d->adam;
    if (!(d->dtype == VCT_BF16 && d->out_dtype == VCT_F32 && d->ta == 1 && d->tb == 0)) return VCT_E_ARG;
    if (d->bias != nullptr || d->act != VCT_ACT_NONE || d->preact != nullptr || d->addend != nullptr || d->dact_src != nullptr ||
        (d->seed != nullptr && d->p_drop > 0.0f))
      return VCT_E_ARG;
    if (!a->param || !a->exp_avg || !a->exp_avg_sq || !a->hyper || !a->step) return VCT_E_ARG;
    if (((uintptr_t)a->param | (uintptr_t)a->exp_avg | (uintptr_t)a->exp_avg_sq | (uintptr_t)d->C) & 15) return VCT_E_ALIGN;
    if (d->ldc % 4) return VCT_E_ALIGN;
    if (a->shadow != nullptr && (((uintptr_t)a->shadow & 7) || (a->ld_shadow % 4))) return VCT_E_ALIGN;
    if (a->pk_stream != nullptr && (a->shadow == nullptr || ((uintptr_t)a->pk_stream & 15) || a->pk_K < d->N || (a->pk_K % 8) ||
                                    a->pk_row0 < 0 || (a->pk_mode != 0 && a->pk_mode != 1)))
      return VCT_E_ARG;
    // the kernel carries the first chunk of each block as a 16-bit field, 0xffff = "not packed": a larger index cannot be expressed
    if (a->pk_stream != nullptr)
      for (int i = 0; i < 4; i++) if (a->pk_chunk0[i] >= 0xffff) return VCT_E_ARG;
  }
  return VCT_OK;
}

static void fill_params(const vct_gemm_desc* d, const Plan& pl, GemmP& p) {
  p.A = d->A; p.B = d->B; p.C = d->C;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.kt_per_split = pl.kt_per;
  p.tiles_m = pl.tiles_m;
  p.tiles_n = pl.tiles_n;
  p.act = d->dact_src != nullptr ? VCT_ACT_NONE : d->act;
  p.dact_kind = d->dact_src != nullptr ? d->act : VCT_ACT_NONE;
  p.bias = d->bias;
  p.preact = d->preact; p.ld_preact = d->ld_preact;
  p.addend = d->addend; p.ld_addend = d->ld_addend;
  p.dact = d->dact_src; p.ld_dact = d->ld_dact;
  p.seed = d->seed; p.site = d->site; p.p_drop = d->p_drop;
  p.bias_grad = d->bias_grad;
  p.partial = nullptr; p.bias_partial = nullptr; p.counters = nullptr;
  memset(&p.adam, 0, sizeof(p.adam));
  if (d->adam != nullptr) {
    const vct_gemm_adam* a = d->adam;
    p.adam.param = a->param; p.adam.m = a->exp_avg; p.adam.v = a->exp_avg_sq;
    p.adam.shadow = reinterpret_cast<uint16_t*>(a->shadow); p.adam.ld_shadow = (long)a->ld_shadow;
    p.adam.pk_stream = reinterpret_cast<uint16_t*>(a->pk_stream); p.adam.pk_K = a->pk_K; p.adam.pk_mode = a->pk_mode;
    {
      auto f = [](int c) { return (unsigned long long)(c < 0 || c >= 0xffff ? 0xffff : c); };
      p.adam.pk_chunks = f(a->pk_chunk0[0]) | (f(a->pk_chunk0[1]) << 16) | (f(a->pk_chunk0[2]) << 32) | (f(a->pk_chunk0[3]) << 48);
    }
    p.adam.pk_row0 = a->pk_row0; p.adam.store_grad = a->store_grad;
    p.adam.hyper = a->hyper; p.adam.step = a->step;
  }
  p.waves8 = pl.waves8;
  p.split = pl.split;
  {
    // Output store policy (A/B switch VCT_GEMM_NT): 0 = plain (default), 1 = agent-scope streaming store (vct_common.h), 2 = `nt`.
    // Round 4, same box, whole step: plain 2.243-2.252 ms, `nt` (the default until then for outputs > 64 MB) 2.252-2.265, agent scope
    // 2.263-2.264 -- see the note at gemm256_try.
    static const char* env = getenv("VCT_GEMM_NT");
    p.nt_store = env != nullptr ? atoi(env) : 0;
  }
  {
    static const char* env = getenv("VCT_GEMM_NT_PREACT");   // A/B switch
    p.nt_preact = env != nullptr ? (env[0] == '1') : 0;
  }
  {
    static const char* env = getenv("VCT_GEMM_ORDER");       // A/B switch: 0 = grouped order always, 1 = short dimension fastest always
    const int ts = pl.tiles_m < pl.tiles_n ? pl.tiles_m : pl.tiles_n;
    p.short_fast = env != nullptr ? (env[0] == '1') : (pl.split > 1 && ts <= 4);
  }
  if (pl.split > 1) {
    if (use_counters(d, pl)) p.counters = d->tile_counters;
    p.partial = reinterpret_cast<float*>(d->workspace);
    p.bias_partial = p.partial + (size_t)pl.split * d->M * d->N;
  }
}

extern "C" int vct_gemm(const vct_gemm_desc* d, void* stream) {
  const int ok = check_desc(d);
  if (ok != VCT_OK) return ok;
  hipStream_t st = (hipStream_t)stream;
  const bool general_only = d->adam != nullptr;      // the optimizer epilogue exists in the general bf16 kernel's dW form only
  if (!general_only) {
    bool used = false;
    const int rcs = gemm_skinny_try(d, st, &used);
    if (rcs != VCT_OK || used) return rcs;
  }
  if (!general_only || (d->ta == 1 && d->tb == 0)) {       // (the pipelined 256 x 256 kernel carries the optimizer epilogue in its TN form)
    bool used = false;
    int rsplit = 1;
    const int rc256 = gemm256_try(d, st, &used, &rsplit);
    if (rc256 != VCT_OK) return rc256;
    if (used) {
      if (rsplit > 1) {     // fp32 partials [rsplit][M][N] in the workspace -> C (bf16)
        const size_t total = (size_t)d->M * d->N;
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        vct::launch((splitk_reduce_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float*>(d->workspace),
                    reinterpret_cast<bf16_t*>(d->C), (long)d->ldc, d->M, d->N, rsplit, (const float*)nullptr, (float*)nullptr);
        VCT_CHECK_LAUNCH();
      }
      return VCT_OK;
    }
  }

  if (!general_only) {
    bool used = false;
    const int rcp = gemm_pt_try(d, st, &used);
    if (rcp != VCT_OK || used) return rcp;
  }

  const int64_t need = vct_gemm_workspace_bytes(d);
  const bool have_ws = d->workspace != nullptr && d->workspace_bytes >= need && need > 0;
  if (d->split_k > 1 && !have_ws) return VCT_E_WORKSPACE;
  const Plan pl = gemm_plan(d, have_ws);

  GemmP p;
  fill_params(d, pl, p);
  const dim3 grid(pl.tiles_m * pl.tiles_n, 1, pl.split);
  int rc;
  if (d->dtype == VCT_BF16) {
    rc = gemm_bf16_v2_dispatch(d, p, pl.bm, pl.bn, pl.nbuf, grid, st);
  } else {
    if (d->out_dtype != VCT_F32) return VCT_E_ARG;
    rc = dispatch_layout<float, float>(d, p, pl.bm, grid, st);
  }
  if (rc != VCT_OK) return rc;
  VCT_CHECK_LAUNCH();
  if (pl.split > 1 && p.counters == nullptr) {   // two-pass split-K (no tile counters given)
    const size_t total = (size_t)d->M * d->N;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (d->out_dtype == VCT_F32)
      vct::launch((splitk_reduce_kernel<float>), dim3(blocks), dim3(256), 0, st, p.partial,
                         reinterpret_cast<float*>(d->C), (long)d->ldc, d->M, d->N, pl.split, p.bias_partial, d->bias_grad);
    else
      vct::launch((splitk_reduce_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, p.partial,
                         reinterpret_cast<bf16_t*>(d->C), (long)d->ldc, d->M, d->N, pl.split, p.bias_partial, d->bias_grad);
    VCT_CHECK_LAUNCH();
  }
  return VCT_OK;
}

// ---- grouped weight-gradient launch --------------------------------------------------------------
namespace vct {
int gemm_bf16_v2_grouped_tn(const GemmGroupP& g, int bm, int bn, int total_wg, hipStream_t st);
}

// One tile shape for the whole group and a split per problem: a layer's weight gradients are 4-7 GEMMs with
// 16-128 output tiles each -- launched one by one each leaves most of the chip idle and pays its own ramp-up and
// split-K reduce.  descs[0].reserved overrides the tile for experiments (same codes as vct_gemm: 5, 8, 4).
struct GroupTile { int bm, bn, waves8; };
static long group_tiles(const vct_gemm_desc* descs, int n, const GroupTile& t) {
  long gt = 0;
  for (int k = 0; k < n; k++) gt += (long)((descs[k].M + t.bm - 1) / t.bm) * ((descs[k].N + t.bn - 1) / t.bn);
  return gt;
}
static GroupTile grouped_tile(const vct_gemm_desc* descs, int n) {
  static const int env_sel = [] { const char* e = getenv("VCT_GROUP_TILE"); return e ? atoi(e) : 0; }();      // A/B: 4 / 5 / 8 as below
  const int sel = descs[0].reserved % 10 ? descs[0].reserved % 10 : env_sel;
  if (sel == 4) return GroupTile{64, 64, 0};
  if (sel == 8) return GroupTile{128, 64, 4};
  if (sel == 5) return GroupTile{128, 128, 1};
  // measured (tools/gemm_bench.py --grouped, MI355X): a decoder layer (256 tiles of 128x128) 74 us with 128x128
  // tiles vs 88 us with 64x64; an encoder layer (192 tiles) 49 us vs 42 us -- one big tile per CU only pays
  // once (nearly) every CU gets one
  const GroupTile big{128, 128, 1};
  return group_tiles(descs, n, big) >= 160 ? big : GroupTile{64, 64, 0};      // (in the step, with the group-level XCD map, the 192-tile encoder layer also runs better on 128 x 128: 2.249 vs 2.255 ms, round 5)
}
static Plan grouped_plan(const vct_gemm_desc* d, const GroupTile& t, long gt) {
  Plan pl;
  pl.bm = t.bm; pl.bn = t.bn; pl.nbuf = 2; pl.waves8 = t.waves8;
  pl.nkt = (d->K + 63) / 64;
  pl.tiles_m = (d->M + t.bm - 1) / t.bm;
  pl.tiles_n = (d->N + t.bn - 1) / t.bn;
  // measured (tools/gemm_bench.py --grouped): every extra split costs ~10 us of coherent partial traffic on a layer's
  // worth of weight gradients, so split only when the group cannot give every CU a workgroup
  int split = gt >= 192 ? 1 : (int)((255 + gt) / gt);
  if (d->split_k >= 1) split = d->split_k;
  if (split > pl.nkt / 4) split = pl.nkt / 4 > 0 ? pl.nkt / 4 : 1;
  while (split > 1 && (int64_t)split * d->M * d->N * 4 > ((int64_t)16 << 20)) split--;   // in-kernel reduce: small partial sets only
  if (split < 1) split = 1;
  pl.kt_per = (pl.nkt + split - 1) / split;
  pl.split = (pl.nkt + pl.kt_per - 1) / pl.kt_per;
  return pl;
}

extern "C" int64_t vct_gemm_grouped_workspace_bytes(const vct_gemm_desc* descs, int32_t n, int32_t i) {
  if (descs == nullptr || n < 1 || n > VCT_GEMM_GROUP_MAX || i < 0 || i >= n) return 0;
  const GroupTile t = grouped_tile(descs, n);
  return ws_bytes(descs + i, grouped_plan(descs + i, t, group_tiles(descs, n, t)));
}

extern "C" int vct_gemm_grouped(const vct_gemm_desc* descs, int32_t n, void* stream) {
  if (descs == nullptr || n < 1 || n > VCT_GEMM_GROUP_MAX) return VCT_E_ARG;
  for (int i = 0; i < n; i++) {
    const vct_gemm_desc* d = descs + i;
    const int ok = check_desc(d);
    if (ok != VCT_OK) return ok;
    // the group kernel is the weight-gradient form: bf16 operands, dW[M,N] = A^T B, fp32 out, no epilogue but db
    if (d->dtype != VCT_BF16 || d->out_dtype != VCT_F32 || d->ta != 1 || d->tb != 0 || !gemm_can_split(d)) return VCT_E_ARG;
  }
  const GroupTile t = grouped_tile(descs, n);
  const long gt = group_tiles(descs, n, t);
  GemmGroupP g;
  g.n = n;
  int wg = 0;
  // group-level XCD map (vct_gemm_bf16_kernel.h, gemm_bf16_v2_grouped_kernel) whenever no problem is split; VCT_GROUP_MAP=0: A/B
  static const bool group_map_on = [] { const char* e = getenv("VCT_GROUP_MAP"); return !(e && e[0] == '0'); }();
  bool group_map = group_map_on;
  for (int i = 0; i < n && group_map; i++) group_map = grouped_plan(descs + i, t, gt).split == 1;
  for (int i = 0; i < n; i++) {
    const vct_gemm_desc* d = descs + i;
    const Plan pl = grouped_plan(d, t, gt);
    if (pl.split > 1) {   // the grouped form always reduces in-kernel
      if (!use_counters(d, pl)) return VCT_E_WORKSPACE;
      if (d->workspace == nullptr || d->workspace_bytes < ws_bytes(d, pl)) return VCT_E_WORKSPACE;
    }
    fill_params(d, pl, g.p[i]);
    g.start[i] = wg;
    wg += group_map ? pl.tiles_m * pl.tiles_n : (pl.tiles_m * pl.tiles_n * pl.split + 7) & ~7;
  }
  g.total = group_map ? wg : 0;
  for (int i = n; i < VCT_GEMM_GROUP_MAX; i++) { g.start[i] = wg; g.p[i] = g.p[0]; }
  wg = (wg + 7) & ~7;
  const int rc = gemm_bf16_v2_grouped_tn(g, t.bm, t.bn, wg, (hipStream_t)stream);
  if (rc != VCT_OK) return rc;
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}
