// Skinny bf16 GEMM for the token loop (gfx950):  C[M,N] = A[M,K] W[N,K]^T (+ bias, activation, residual) with M <= 256 rows
// -- one new token per caption: M = the decode batch (reference MMT4Caption.py:156-170 -> CapDecoder.decode_word: every
// nn.Linear of the decoder layers applied to [B, 1, d]).
//
// Why another kernel.  At M = 128, N = K = 512 the general kernel's smallest tile (64x64, LDS double buffer) gives 16
// workgroups that each walk 8 dependent K stages: 8.2 us per projection and 20 us for linear2 (K = 2048, 32 stages) on 16 of
// 256 CUs -- ten such products are 100 of the 188 us of a batch-128 decode step.  The products are latency-bound (67-270
// MFLOP, <= 2 MB of weights), so the kernel is shaped for the number of DEPENDENT memory round trips and for the width of
// the machine, not for reuse:
//   * a workgroup owns ONE 16-row MFMA tile x NTL (1 / 2 / 4) 16-column tiles, NTL chosen so that the launch has >= ~200
//     workgroups (128 x 512: 8 x 32 = 256); its 8 waves split K eight ways (wave w takes the 32-deep k-steps w, w+8, ...),
//     so K = 512 is two k-steps per wave and K = 2048 eight.  (A first version with 64 rows per workgroup ran 64 workgroups
//     that each pulled 64-128 KB of A through one CU's vector memory path: 6 us plain, 12-14 us with the LayerNorm prologue.)
//   * MFMA fragments are loaded straight from global memory / L2 in the operand layout of v_mfma_f32_16x16x32_bf16 (lane =
//     row (lane & 15), 8 consecutive k at (lane >> 4) * 8: 16 bytes per lane) -- no LDS staging, no barrier in the K loop,
//     all loads of up to four k-steps in flight before the first MFMA;
//   * the operands are swapped (mfma(b, a)) so a lane holds FOUR CONSECUTIVE columns of one output row;
//   * one LDS reduction over the eight waves, then bias / activation / residual and 8- or 16-byte stores;
//   * LayerNorm prologue (vct_decode_linear): the input rows are fp32 pre-norm sums; every workgroup sees whole rows (its
//     waves cover K), so the row statistics come from the fragments themselves with one LDS exchange, the fragments are
//     normalised in registers and rounded to bf16 for the MFMA; the workgroups of column block 0 store the normalised rows.
#include "vct_gemm_bf16_kernel.h"

namespace vct {

struct SkinnyP {
  const bf16_t* A; const bf16_t* B; void* C;
  long lda, ldb, ldc;
  int M, N, K;
  int act;
  const float* bias;
  const void* addend; long ld_addend;
  int add_f32;                  // residual rows: 0 = element type TO, 1 = fp32, 2 = bf16 (whatever TO is)
  // LayerNorm prologue (LNA): the input rows are LayerNorm(xpre; ln_g, ln_b) of fp32 pre-norm rows, normalised in registers
  const float* xpre; long ld_pre;
  const float* ln_g; const float* ln_b;
  float* xnorm; long ld_norm;   // optional: the normalised rows (fp32), written by the workgroups of output column block 0
  // embedding prologue (LNA kernel, ids != nullptr): input row r = xpre[ids[r * id_stride]] + ln_b (table row + positional row)
  const int64_t* ids; long id_stride;
};

template <int NTL, typename TO, bool LNA, int UNL = 4>
__global__ __launch_bounds__(512) void gemm_skinny_nt_kernel(const SkinnyP p) {
  constexpr int NW = 8, UN = LNA ? UNL : 4;                  // waves; k-steps whose loads are in flight together (LNA: K <= 256 UNL)
  __shared__ f32x4 red[NW][NTL][64];
  __shared__ float stat[LNA ? NW : 1][16][2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, kq = (lane >> 4) * 8, g4 = (lane >> 4) * 4;
  const int n0 = blockIdx.x * (16 * NTL), m0 = blockIdx.y * 16;
  const int ksteps = (p.K + 31) >> 5;
  const int arow_i = min(m0 + c16, p.M - 1);

  const bf16_t* brow[NTL];
#pragma unroll
  for (int j = 0; j < NTL; j++) brow[j] = p.B + (size_t)min(n0 + j * 16 + c16, p.N - 1) * p.ldb + kq;

  f32x4 acc[NTL];
#pragma unroll
  for (int j = 0; j < NTL; j++) acc[j] = f32x4{0, 0, 0, 0};
  const bf16x8 zero = __builtin_bit_cast(bf16x8, s16x8{0, 0, 0, 0, 0, 0, 0, 0});

  if constexpr (!LNA) {
    const bf16_t* arow = p.A + (size_t)arow_i * p.lda + kq;
    for (int s0 = wave; s0 < ksteps; s0 += NW * UN) {
      bf16x8 fa[UN], fb[UN][NTL];
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const int k = (s0 + u * NW) * 32;
        const bool in = (s0 + u * NW < ksteps) && (k + kq < p.K);      // K % 8 == 0: a lane's 8 k are all inside or all outside
        fa[u] = in ? *reinterpret_cast<const bf16x8*>(arow + k) : zero;
#pragma unroll
        for (int j = 0; j < NTL; j++) fb[u][j] = in ? *reinterpret_cast<const bf16x8*>(brow[j] + k) : zero;
      }
#pragma unroll
      for (int u = 0; u < UN; u++)
#pragma unroll
        for (int j = 0; j < NTL; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u][j], fa[u], acc[j], 0, 0, 0);
    }
  } else {
    // K <= 256 UNL: the wave's k-steps are all in registers
    bf16x8 fb[UN][NTL];
    f32x4 xa[UN][2], gg[UN][2], bb[UN][2];
    bool in[UN];
    const bool emb = p.ids != nullptr;
    const float* xr = p.xpre + (size_t)(emb ? p.ids[(size_t)arow_i * p.id_stride] : (int64_t)arow_i) * p.ld_pre + kq;
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int k = (wave + u * NW) * 32;
      in[u] = (wave + u * NW < ksteps) && (k + kq < p.K);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        xa[u][h] = in[u] ? *reinterpret_cast<const f32x4*>(xr + k + h * 4) : f32x4{0, 0, 0, 0};
        gg[u][h] = (in[u] && !emb) ? *reinterpret_cast<const f32x4*>(p.ln_g + k + kq + h * 4) : f32x4{1, 1, 1, 1};
        bb[u][h] = in[u] ? *reinterpret_cast<const f32x4*>(p.ln_b + k + kq + h * 4) : f32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int j = 0; j < NTL; j++) fb[u][j] = in[u] ? *reinterpret_cast<const bf16x8*>(brow[j] + k) : zero;
    }
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int u = 0; u < UN; u++)
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int e = 0; e < 4; e++) { const float x = xa[u][h][e]; s1 += x; s2 += x * x; }
    s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
    if (lane < 16) { stat[wave][c16][0] = s1; stat[wave][c16][1] = s2; }
    __syncthreads();
    s1 = 0.0f; s2 = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; w++) { s1 += stat[w][c16][0]; s2 += stat[w][c16][1]; }
    const float mean = emb ? 0.0f : s1 / (float)p.K;
    const float rstd = emb ? 1.0f : 1.0f / sqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.0f) + 1e-5f);
    const bool keep = p.xnorm != nullptr && blockIdx.x == 0 && m0 + c16 < p.M;
#pragma unroll
    for (int u = 0; u < UN; u++) {
      s16x8 pk;
      f32x4 y[2];
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          y[h][e] = (xa[u][h][e] - mean) * rstd * gg[u][h][e] + bb[u][h][e];
          pk[h * 4 + e] = in[u] ? (short)f2bf(y[h][e]) : (short)0;
        }
      if (keep && in[u]) {
        float* dst = p.xnorm + (size_t)(m0 + c16) * p.ld_norm + (wave + u * NW) * 32 + kq;
        *reinterpret_cast<f32x4*>(dst) = y[0];
        *reinterpret_cast<f32x4*>(dst + 4) = y[1];
      }
      const bf16x8 fa = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
      for (int j = 0; j < NTL; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u][j], fa, acc[j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < NTL; j++) red[wave][j][lane] = acc[j];
  __syncthreads();

  TO* C = reinterpret_cast<TO*>(p.C);
  const TO* addend = reinterpret_cast<const TO*>(p.addend);
  const float* addf = reinterpret_cast<const float*>(p.addend);
  const bf16_t* addh = reinterpret_cast<const bf16_t*>(p.addend);
  const bool add32 = p.add_f32 == 1, add16 = p.add_f32 == 2;
  const bool vec_ok = (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0) &&
                      (addend == nullptr || ((p.ld_addend % 4 == 0) && (((uintptr_t)p.addend & 15) == 0)));
  for (int j = wave; j < NTL; j += NW) {
    f32x4 v = red[0][j][lane];
#pragma unroll
    for (int w = 1; w < NW; w++) {                           // fixed order: bit-reproducible
      const f32x4 t = red[w][j][lane];
      v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    const int row = m0 + c16, col = n0 + j * 16 + g4;
    if (row >= p.M || col >= p.N) continue;
    const bool full = vec_ok && (col + 4 <= p.N);
    struct alignas(4 * sizeof(TO)) V4 { TO e[4]; };
    V4 av, ov;
    f32x4 af = {0, 0, 0, 0};
    if (full && addend != nullptr) {
      if (add32) af = *reinterpret_cast<const f32x4*>(addf + (size_t)row * p.ld_addend + col);
      else if (add16) {
        struct alignas(8) H4 { bf16_t e[4]; };
        const H4 hv = *reinterpret_cast<const H4*>(addh + (size_t)row * p.ld_addend + col);
#pragma unroll
        for (int r = 0; r < 4; r++) af[r] = bf2f(hv.e[r]);
      } else av = *reinterpret_cast<const V4*>(addend + (size_t)row * p.ld_addend + col);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (col + r >= p.N) break;
      float x = v[r] + (p.bias != nullptr ? p.bias[col + r] : 0.0f);
      x = act_fast_f(p.act, x);
      if (addend != nullptr) {
        if (add32) x += full ? af[r] : addf[(size_t)row * p.ld_addend + col + r];
        else if (add16) x += full ? af[r] : bf2f(addh[(size_t)row * p.ld_addend + col + r]);
        else x += full ? to_f<TO>(av.e[r]) : to_f<TO>(addend[(size_t)row * p.ld_addend + col + r]);
      }
      ov.e[r] = from_f<TO>(x);
      if (!full) C[(size_t)row * p.ldc + col + r] = ov.e[r];
    }
    if (full) *reinterpret_cast<V4*>(C + (size_t)row * p.ldc + col) = ov;
  }
}

// column tiles per workgroup: the widest of 4 / 2 / 1 that still leaves ~200 workgroups (or all there are)
static int skinny_ntl(int M, int N) {
  const int mt = (M + 15) / 16, nt = (N + 15) / 16;
  if (mt * ((nt + 3) / 4) >= 192) return 4;
  if (mt * ((nt + 1) / 2) >= 192) return 2;
  return 1;
}

template <typename TO, bool LNA, int UNL> static void skinny_launch(const SkinnyP& p, hipStream_t st) {
  const int ntl = skinny_ntl(p.M, p.N);
  const dim3 grid((p.N + 16 * ntl - 1) / (16 * ntl), (p.M + 15) / 16);
  if (ntl == 4) vct::launch(gemm_skinny_nt_kernel<4, TO, LNA, UNL>, grid, dim3(512), 0, st, p);
  else if (ntl == 2) vct::launch(gemm_skinny_nt_kernel<2, TO, LNA, UNL>, grid, dim3(512), 0, st, p);
  else vct::launch(gemm_skinny_nt_kernel<1, TO, LNA, UNL>, grid, dim3(512), 0, st, p);
}

// Eligibility + launch (called from vct_gemm).  VCT_GEMM_SKINNY=0 disables (A/B switch).
int gemm_skinny_try(const vct_gemm_desc* d, hipStream_t st, bool* used) {
  *used = false;
  static const char* env = getenv("VCT_GEMM_SKINNY");
  if (env != nullptr && env[0] == '0') return VCT_OK;
  if (d->dtype != VCT_BF16 || d->ta != 0 || d->tb != 1 || d->reserved != 0) return VCT_OK;
  if (d->M > 256 || d->N > 4096 || d->N < 16 || d->K < 64 || d->K > 8192 || (d->K % 8) != 0) return VCT_OK;
  if (d->preact || d->dact_src || d->bias_grad || d->split_k > 1 || (d->seed && d->p_drop > 0.0f)) return VCT_OK;
  SkinnyP p;
  p.A = reinterpret_cast<const bf16_t*>(d->A); p.B = reinterpret_cast<const bf16_t*>(d->B); p.C = d->C;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.act = d->act; p.bias = d->bias; p.addend = d->addend; p.ld_addend = d->ld_addend;
  p.add_f32 = 0;
  p.xpre = nullptr; p.ld_pre = 0; p.ln_g = p.ln_b = nullptr; p.xnorm = nullptr; p.ld_norm = 0; p.ids = nullptr; p.id_stride = 0;
  if (d->out_dtype == VCT_BF16) skinny_launch<bf16_t, false, 4>(p, st);
  else skinny_launch<float, false, 4>(p, st);
  VCT_CHECK_LAUNCH();
  *used = true;
  return VCT_OK;
}

// last layer's norm3 followed by decoder.norm on the batch rows of the current position: y = LN(LN(x; g1, b1); g2, b2), fp32 in,
// bf16 out (the vocabulary projection's input).  One wave per row, the row in registers (K <= 1024).
__global__ __launch_bounds__(256) void decode_ln2_kernel(int M, int K, const float* __restrict__ x, long ldx, const float* __restrict__ g1,
                                                         const float* __restrict__ b1, const float* __restrict__ g2,
                                                         const float* __restrict__ b2, bf16_t* __restrict__ y, long ldy) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  f32x4 v[4];
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int k = j * 256 + lane * 4;
    v[j] = k < K ? *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + k) : f32x4{0, 0, 0, 0};
    s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
  }
  auto wsum = [](float t) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    return t;
  };
  auto norm = [&](const float* g, const float* b) {
    const float mean = wsum(s) / (float)K;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int e = 0; e < 4; e++) { const float t = (j * 256 + lane * 4 + e < K) ? v[j][e] - mean : 0.0f; q += t * t; }
    const float rstd = 1.0f / sqrtf(wsum(q) / (float)K + 1e-5f);
    s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = j * 256 + lane * 4;
      if (k < K) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + k), bv = *reinterpret_cast<const f32x4*>(b + k);
#pragma unroll
        for (int e = 0; e < 4; e++) { v[j][e] = (v[j][e] - mean) * rstd * gv[e] + bv[e]; s += v[j][e]; }
      }
    }
  };
  norm(g1, b1);
  if (g2 != nullptr) norm(g2, b2);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int k = j * 256 + lane * 4;
    if (k < K) {
      struct alignas(8) B4 { bf16_t e[4]; } o;
#pragma unroll
      for (int e = 0; e < 4; e++) o.e[e] = f2bf(v[j][e]);
      *reinterpret_cast<B4*>(y + (size_t)row * ldy + k) = o;
    }
  }
}

}  // namespace vct

using namespace vct;

extern "C" int vct_decode_linear(const vct_decode_linear_desc* d, void* stream) {
  if (d == nullptr || d->W == nullptr || d->out == nullptr) return VCT_E_ARG;
  if ((d->x == nullptr) == (d->x_pre == nullptr)) return VCT_E_ARG;                 // exactly one input form
  if (d->out_dtype != VCT_BF16 && d->out_dtype != VCT_F32) return VCT_E_ARG;
  if (d->res != nullptr && d->res_dtype != VCT_F32 && d->res_dtype != VCT_BF16) return VCT_E_ARG;
  if (d->M <= 0 || d->M > 256 || d->N <= 0 || d->K < 32 || (d->K % 8) != 0) return VCT_E_SHAPE;
  if ((d->ldw % 8) || ((uintptr_t)d->W & 15)) return VCT_E_ALIGN;
  SkinnyP p;
  p.A = reinterpret_cast<const bf16_t*>(d->x); p.B = reinterpret_cast<const bf16_t*>(d->W); p.C = d->out;
  p.lda = d->ldx; p.ldb = d->ldw; p.ldc = d->ldo;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.act = d->act; p.bias = d->bias; p.addend = d->res; p.ld_addend = d->ld_res;
  p.add_f32 = d->res_dtype == VCT_F32 ? 1 : 2;
  p.xpre = d->x_pre; p.ld_pre = d->ld_pre; p.ln_g = d->ln_g; p.ln_b = d->ln_b; p.xnorm = d->x_norm; p.ld_norm = d->ld_norm;
  p.ids = d->ids; p.id_stride = d->id_stride;
  const bool lna = d->x_pre != nullptr;
  if (!lna && d->ids != nullptr) return VCT_E_ARG;
  if (lna) {
    if (d->K > 1024) return VCT_E_SHAPE;
    if (d->ln_b == nullptr || (d->ln_g == nullptr && d->ids == nullptr)) return VCT_E_ARG;
    if ((d->ld_pre % 4) || ((uintptr_t)d->x_pre & 15) || ((uintptr_t)d->ln_g & 15) || ((uintptr_t)d->ln_b & 15)) return VCT_E_ALIGN;
    if (d->x_norm != nullptr && ((d->ld_norm % 4) || ((uintptr_t)d->x_norm & 15))) return VCT_E_ALIGN;
  } else if ((d->ldx % 8) || ((uintptr_t)d->x & 15)) {
    return VCT_E_ALIGN;
  }
  hipStream_t st = (hipStream_t)stream;
  const bool ob = d->out_dtype == VCT_BF16;
  if (lna && d->K <= 512) {
    if (ob) skinny_launch<bf16_t, true, 2>(p, st); else skinny_launch<float, true, 2>(p, st);
  } else if (lna) {
    if (ob) skinny_launch<bf16_t, true, 4>(p, st); else skinny_launch<float, true, 4>(p, st);
  } else {
    if (ob) skinny_launch<bf16_t, false, 4>(p, st); else skinny_launch<float, false, 4>(p, st);
  }
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_decode_ln2(int M, int K, const float* x, int64_t ldx, const float* g1, const float* b1, const float* g2,
                              const float* b2, void* y, int64_t ldy, void* stream) {
  if (x == nullptr || g1 == nullptr || b1 == nullptr || y == nullptr || (g2 == nullptr) != (b2 == nullptr)) return VCT_E_ARG;
  if (M <= 0 || K <= 0 || K > 1024 || (K % 4)) return VCT_E_SHAPE;
  if ((ldx % 4) || (ldy % 4) || ((uintptr_t)x & 15) || ((uintptr_t)y & 7) || ((uintptr_t)g1 & 15) || ((uintptr_t)b1 & 15) ||
      ((uintptr_t)g2 & 15) || ((uintptr_t)b2 & 15)) return VCT_E_ALIGN;
  vct::launch(decode_ln2_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, M, K, x, (long)ldx, g1, b1, g2, b2,
              reinterpret_cast<bf16_t*>(y), (long)ldy);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}
