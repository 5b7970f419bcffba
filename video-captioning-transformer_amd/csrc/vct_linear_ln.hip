// Row-complete nn.Linear + dropout + residual + LayerNorm (+ a second LayerNorm) in ONE launch, bf16, gfx950:
//
//     a  = x W^T + b                          (saved in bf16: what the LayerNorm backward re-reads)
//     y  = LayerNorm(res + dropout(a))        (mean / rstd saved)
//     y2 = LayerNorm2(y)                      (optional: the stack-final norm behind the last layer)
//
// replaces, per attention block of nn.TransformerEncoderLayer / nn.TransformerDecoderLayer (torch nn/modules/transformer.py:
// 951-957, 1143-1167, built at model/MMEncoder.py:236-238 and model/CapDecoder.py:18-20): the out_proj nn.Linear
// (torch nn/functional.py:6637), dropout1/2, the residual add and norm1/2 -- two launches (vct_gemm, vct_add_ln_fwd) and one
// HBM round trip of a [tokens, d] activation on the unfused path; with linear2 as the Linear it is the tail of the feed-forward
// block (linear2 + dropout + add + norm2/3), and with gamma2 the stack-final LayerNorm (MMEncoder.py:238, CapDecoder.py:20) too.
//
// Shape of the kernel.  A LayerNorm needs whole rows, so a workgroup owns BM = 16 / 32 rows x ALL d output columns and streams
// the whole weight matrix [d, K] through LDS (global_load_lds, double-buffered 64-deep K stages in the swizzled images of
// vct_gemm_bf16_kernel.h).  Every workgroup reads the same W, which is L2-resident (out_proj: 512 KB): the kernel is bound by
// the L2 -> LDS rate of one CU times d*K*2 bytes, independent of BM -- affordable for K = d (out_proj), measured against the
// unfused pair for K = ff (linear2).  8 waves, wave w owns columns [w*d/8, (w+1)*d/8) of all BM rows; after the K loop the
// accumulators go through an fp32 row slab (aliasing the operand stages) and 512/BM threads finish each row: bias, saved
// activation, dropout (same counter stream as vct_add_ln_fwd: the unfused backward kernels run unchanged), residual, two-pass
// statistics by lane shuffles inside the row's thread group, normalise, 16-byte stores of whole 256-byte row segments.
#include "vct_gemm_bf16_kernel.h"

namespace vct {

struct LinLnP {
  const bf16_t* A; long lda;
  const bf16_t* W; long ldw;
  const float* bias;
  const bf16_t* res; long ld_res;
  const uint32_t* seed; uint32_t site; float p_drop;
  const float* gamma; const float* beta;
  bf16_t* aout; long ld_a;
  bf16_t* y; long ld_y;
  float* mean; float* rstd;
  const float* gamma2; const float* beta2;
  bf16_t* y2; long ld_y2;
  float* mean2; float* rstd2;
  int M, K;
};

struct alignas(16) LV8 { bf16_t e[8]; };

template <int TPR> __device__ __forceinline__ float row_group_sum(float v) {
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

template <int D, int TM>
__global__ __launch_bounds__(512, 2) void linear_ln_fwd_kernel(const LinLnP p) {
  constexpr int NW = 8, NT = 512, BM = 16 * TM, WN = D / NW, TN = WN / 16;
  static_assert(D % 128 == 0 && TN >= 1, "d_model must be a multiple of 128");
  constexpr int A_BYTES = BM * 128, B_BYTES = D * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int SSTR = D + 4;                                   // fp32 slab row stride
  static_assert(BM * SSTR * 4 <= 2 * STAGE, "row slab must fit in the operand stages");
  constexpr int TPR = NT / BM;                                  // threads per row in the epilogue (16 / 32)
  constexpr int CH = D / 8 / TPR;                               // 8-column chunks per thread
  static_assert(CH >= 1 && CH * TPR * 8 == D, "row does not divide over its thread group");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = (lane >> 4) * 4;
  const int m0 = blockIdx.x * BM;
  const int nkt = p.K / BK2;

  auto issue = [&](int kt, int buf) {
    unsigned char* nb = lds + buf * STAGE;
    if (wave < BM / 8) {                                        // A tile: BM rows x 128 bytes = BM / 8 one-KiB pieces
      const int row = wave * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (row & 7);
      const bf16_t* src = p.A + (long)min(m0 + row, p.M - 1) * p.lda + kt * BK2 + c * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(nb + wave * 1024), 16, 0, 0);
    }
    dma_tile<false, D, NW>(nb + A_BYTES, p.W, p.ldw, 0, D, kt * BK2, wave, lane);
  };

  // epilogue geometry of this thread: row `er` of the tile, chunks c -> columns (c * TPR + et) * 8
  const int er = tid / TPR, et = tid % TPR;
  const int grow = m0 + er;
  const bool rvalid = grow < p.M;
  const int crow = min(grow, p.M - 1);
  LV8 rres[CH];
  if (p.res != nullptr) {                                       // residual rows: issued now, consumed after the K loop
#pragma unroll
    for (int c = 0; c < CH; c++) rres[c] = *reinterpret_cast<const LV8*>(p.res + (long)crow * p.ld_res + (c * TPR + et) * 8);
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};

  issue(0, 0);
  int buf = 0;
  for (int kt = 0; kt < nkt; kt++) {
    __syncthreads();                                            // stage kt has landed for everyone; the other buffer is free
    if (kt + 1 < nkt) issue(kt + 1, buf ^ 1);
    const unsigned char* la = lds + buf * STAGE;
    const unsigned char* lb = la + A_BYTES;
    bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int i = 0; i < TM; i++) fa[ks][i] = frag2<false, BM, false>(la, i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < TN; j++) fb[ks][j] = frag2<false, D, false>(lb, wave * WN + j * 16, ks, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);
    buf ^= 1;
  }

  // ---- row slab (aliases the operand stages) -----------------------------------------------------------------------------
  __syncthreads();
  float* slab = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) slab[(i * 16 + g4 + r) * SSTR + wave * WN + j * 16 + c16] = acc[i][j][r];
  __syncthreads();

  const Dropout dr = make_dropout(p.seed, p.site, p.p_drop);
  float s[CH][8];
  float sum = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int col = (c * TPR + et) * 8;
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(slab + er * SSTR + col);
    const f32x4 t1 = *reinterpret_cast<const f32x4*>(slab + er * SSTR + col + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col), b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
    const float t[8] = {t0[0] + b0.x, t0[1] + b0.y, t0[2] + b0.z, t0[3] + b0.w, t1[0] + b1.x, t1[1] + b1.y, t1[2] + b1.z, t1[3] + b1.w};
    LV8 av;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      av.e[q] = f2bf(t[q]);                    // the saved activation is the bf16 value and the norm is built on it (= the unfused path)
      float v = bf2f(av.e[q]) * drop_mult(dr, (uint32_t)grow * (uint32_t)D + (uint32_t)(col + q));
      if (p.res != nullptr) v += bf2f(rres[c].e[q]);
      s[c][q] = v;
      sum += v;
    }
    if (p.aout != nullptr && rvalid) *reinterpret_cast<LV8*>(p.aout + (long)grow * p.ld_a + col) = av;
  }
  const float mean = row_group_sum<TPR>(sum) * (1.0f / (float)D);
  float sq = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int q = 0; q < 8; q++) { const float dlt = s[c][q] - mean; sq += dlt * dlt; }
  const float rstd = 1.0f / sqrtf(row_group_sum<TPR>(sq) * (1.0f / (float)D) + 1e-5f);
  if (et == 0 && rvalid) { p.mean[grow] = mean; p.rstd[grow] = rstd; }
  float sum2 = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int col = (c * TPR + et) * 8;
    const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + col), g1 = *reinterpret_cast<const float4*>(p.gamma + col + 4);
    const float4 e0 = *reinterpret_cast<const float4*>(p.beta + col), e1 = *reinterpret_cast<const float4*>(p.beta + col + 4);
    const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float bt[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    LV8 yv;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      yv.e[q] = f2bf((s[c][q] - mean) * rstd * gm[q] + bt[q]);
      s[c][q] = bf2f(yv.e[q]);                 // the second norm reads the bf16 rows the first one wrote (= the unfused path)
      sum2 += s[c][q];
    }
    if (rvalid) *reinterpret_cast<LV8*>(p.y + (long)grow * p.ld_y + col) = yv;
  }
  if (p.gamma2 == nullptr) return;
  const float m2 = row_group_sum<TPR>(sum2) * (1.0f / (float)D);
  float sq2 = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int q = 0; q < 8; q++) { const float dlt = s[c][q] - m2; sq2 += dlt * dlt; }
  const float r2 = 1.0f / sqrtf(row_group_sum<TPR>(sq2) * (1.0f / (float)D) + 1e-5f);
  if (et == 0 && rvalid) { p.mean2[grow] = m2; p.rstd2[grow] = r2; }
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int col = (c * TPR + et) * 8;
    const float4 g0 = *reinterpret_cast<const float4*>(p.gamma2 + col), g1 = *reinterpret_cast<const float4*>(p.gamma2 + col + 4);
    const float4 e0 = *reinterpret_cast<const float4*>(p.beta2 + col), e1 = *reinterpret_cast<const float4*>(p.beta2 + col + 4);
    const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float bt[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    LV8 yv;
#pragma unroll
    for (int q = 0; q < 8; q++) yv.e[q] = f2bf((s[c][q] - m2) * r2 * gm[q] + bt[q]);
    if (rvalid) *reinterpret_cast<LV8*>(p.y2 + (long)grow * p.ld_y2 + col) = yv;
  }
}

template <int D, int TM> static int linear_ln_launch(const LinLnP& p, hipStream_t st) {
  constexpr int BM = 16 * TM;
  constexpr int LDS = 2 * (BM * 128 + D * 128);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)linear_ln_fwd_kernel<D, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  vct::launch(linear_ln_fwd_kernel<D, TM>, dim3((p.M + BM - 1) / BM), dim3(512), (size_t)LDS, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

}  // namespace vct
using namespace vct;

// bf16, d_model 512 (one 64 KB weight stage per 64-deep K step, two in flight), K a multiple of 64
extern "C" int vct_linear_ln_supported(int dtype, int d, int K) {
  return (dtype == VCT_BF16 && d == 512 && K >= 64 && K % 64 == 0) ? 1 : 0;
}

extern "C" int vct_linear_ln_fwd(const vct_linear_ln_desc* d, void* stream) {
  if (d == nullptr || !d->x || !d->w || !d->bias || !d->gamma || !d->beta || !d->y || !d->mean || !d->rstd) return VCT_E_ARG;
  if ((d->gamma2 == nullptr) != (d->beta2 == nullptr)) return VCT_E_ARG;
  if (d->gamma2 != nullptr && (!d->y2 || !d->mean2 || !d->rstd2)) return VCT_E_ARG;
  if (d->M <= 0 || !vct_linear_ln_supported(d->dtype, d->d, d->K)) return VCT_E_SHAPE;
  if ((d->ldx % 8) || (d->ldw % 8) || (d->ld_y % 8) || (d->res && (d->ld_res % 8)) || (d->a_out && (d->ld_a % 8)) ||
      (d->y2 && (d->ld_y2 % 8)))
    return VCT_E_ALIGN;
  if (((uintptr_t)d->x | (uintptr_t)d->w | (uintptr_t)d->y | (uintptr_t)d->res | (uintptr_t)d->a_out | (uintptr_t)d->y2 |
       (uintptr_t)d->bias | (uintptr_t)d->gamma | (uintptr_t)d->beta | (uintptr_t)d->gamma2 | (uintptr_t)d->beta2) & 15)
    return VCT_E_ALIGN;
  LinLnP p;
  p.A = reinterpret_cast<const bf16_t*>(d->x); p.lda = d->ldx;
  p.W = reinterpret_cast<const bf16_t*>(d->w); p.ldw = d->ldw;
  p.bias = d->bias;
  p.res = reinterpret_cast<const bf16_t*>(d->res); p.ld_res = d->ld_res;
  p.seed = d->seed; p.site = d->site; p.p_drop = d->p_drop;
  p.gamma = d->gamma; p.beta = d->beta;
  p.aout = reinterpret_cast<bf16_t*>(d->a_out); p.ld_a = d->ld_a;
  p.y = reinterpret_cast<bf16_t*>(d->y); p.ld_y = d->ld_y;
  p.mean = d->mean; p.rstd = d->rstd;
  p.gamma2 = d->gamma2; p.beta2 = d->beta2;
  p.y2 = reinterpret_cast<bf16_t*>(d->y2); p.ld_y2 = d->ld_y2;
  p.mean2 = d->mean2; p.rstd2 = d->rstd2;
  p.M = d->M; p.K = d->K;
  hipStream_t st = (hipStream_t)stream;
  // 32 rows per workgroup (half the L2 -> LDS traffic of 16) unless that leaves most CUs without one
  const bool rows16 = d->rows_per_wg == 16 || (d->rows_per_wg == 0 && (d->M + 31) / 32 < 96);
  return rows16 ? linear_ln_launch<512, 1>(p, st) : linear_ln_launch<512, 2>(p, st);
}
