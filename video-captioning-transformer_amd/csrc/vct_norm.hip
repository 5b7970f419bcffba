// y = LayerNorm(res + dropout(x)) forward / backward for gfx950.  HBM-bound: one wave per row, the
// row lives in registers (16-byte loads, <= 16 values per lane => d <= 1024), statistics by wave
// shuffles, two-pass variance in fp32.  The backward recomputes the pre-norm sum from (x, res) and
// the dropout mask from its counter hash, writes ds (and the dropout-masked copy that feeds the
// producing GEMM's backward), and accumulates dgamma/dbeta as per-wave column partials that a
// second tiny kernel sums in a fixed order (deterministic, no float atomics).
#include "vct_common.h"

namespace vct {

constexpr int LN_ROWS_PER_WAVE = 1;   // 4 rows per 4-wave workgroup (round 5, in the step: 2.172 -> 2.164 ms against 2 rows per wave, 2.176 with 4)

template <typename T> struct LnCfg;
template <> struct LnCfg<float> { static constexpr int VEC = 4, MAXIT = 4; };
template <> struct LnCfg<bf16_t> { static constexpr int VEC = 8, MAXIT = 2; };

template <typename T, int VEC> struct alignas(16) RowVec { T v[VEC]; };

// second LayerNorm of the same row (the stack-final norm behind the last layer's norm: MMEncoder.py:238, CapDecoder.py:20):
// y2 = LayerNorm2(y) computed from the rows as STORED (rounded to T), i.e. what a separate launch would read back
struct Ln2P { const float* gamma; const float* beta; void* y; float* mean; float* rstd; };

template <typename T, bool TWO>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(int M, int d, const T* __restrict__ x, const T* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         T* __restrict__ y, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, const uint32_t* seed, uint32_t site,
                                                         float p_drop, const Ln2P n2) {
  constexpr int VEC = LnCfg<T>::VEC, MAXIT = LnCfg<T>::MAXIT;
  using RV = RowVec<T, VEC>;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const Dropout dr = make_dropout(seed, site, p_drop);
  const int nvec = d / VEC;
  float s[MAXIT][VEC];
  float sum = 0.0f;
#pragma unroll
  for (int it = 0; it < MAXIT; it++) {
    const int vi = it * 64 + lane;
    if (vi < nvec) {
      const RV xv = *reinterpret_cast<const RV*>(x + (size_t)row * d + vi * VEC);
      RV rv;
      if (res != nullptr) rv = *reinterpret_cast<const RV*>(res + (size_t)row * d + vi * VEC);
      float dmv[VEC];
      drop_mults<VEC>(dr, (uint32_t)row * (uint32_t)d + (uint32_t)(vi * VEC), dmv);
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        float v = to_f<T>(xv.v[j]) * dmv[j];
        if (res != nullptr) v += to_f<T>(rv.v[j]);
        s[it][j] = v;
        sum += v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; j++) s[it][j] = 0.0f;
    }
  }
  const float mean = wave_sum(sum) / (float)d;
  float sq = 0.0f;
#pragma unroll
  for (int it = 0; it < MAXIT; it++) {
    const int vi = it * 64 + lane;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < VEC; j++) { const float c = s[it][j] - mean; sq += c * c; }
    }
  }
  const float var = wave_sum(sq) / (float)d;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  float sum2 = 0.0f;
#pragma unroll
  for (int it = 0; it < MAXIT; it++) {
    const int vi = it * 64 + lane;
    if (vi < nvec) {
      RV out;
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const int c = vi * VEC + j;
        out.v[j] = from_f<T>((s[it][j] - mean) * rstd * gamma[c] + beta[c]);
        if constexpr (TWO) { s[it][j] = to_f<T>(out.v[j]); sum2 += s[it][j]; }
      }
      *reinterpret_cast<RV*>(y + (size_t)row * d + vi * VEC) = out;
    }
  }
  if constexpr (TWO) {
    const float m2 = wave_sum(sum2) / (float)d;
    float sq2 = 0.0f;
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
      const int vi = it * 64 + lane;
      if (vi < nvec) {
#pragma unroll
        for (int j = 0; j < VEC; j++) { const float c = s[it][j] - m2; sq2 += c * c; }
      }
    }
    const float r2 = 1.0f / sqrtf(wave_sum(sq2) / (float)d + 1e-5f);
    if (lane == 0) { n2.mean[row] = m2; n2.rstd[row] = r2; }
    T* y2 = reinterpret_cast<T*>(n2.y);
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
      const int vi = it * 64 + lane;
      if (vi < nvec) {
        RV out;
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          const int c = vi * VEC + j;
          out.v[j] = from_f<T>((s[it][j] - m2) * r2 * n2.gamma[c] + n2.beta[c]);
        }
        *reinterpret_cast<RV*>(y2 + (size_t)row * d + vi * VEC) = out;
      }
    }
  }
}

// TWO = true: the backward of a stack-final norm IN FRONT, same launch (the forward's add_ln_ln_fwd): dy is the gradient of
// y2 = LayerNorm_f(y), y = LayerNorm(res + drop(x)) as stored (fr.x); the gradient of y is rounded to T exactly where the two-launch
// schedule stores and re-reads it, so the result is bit-identical to vct_add_ln_bwd(final norm) followed by vct_add_ln_bwd(layer norm).
template <typename T> struct LnFront { const T* x; const float* gamma; const float* mean; const float* rstd; float* ws; };

template <typename T, bool TWO>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(int M, int d, const T* __restrict__ dy, const T* __restrict__ x,
                                                         const T* __restrict__ res, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                         T* __restrict__ ds, T* __restrict__ dxo, float* __restrict__ param_ws,
                                                         const uint32_t* seed, uint32_t site, float p_drop, const LnFront<T> fr) {
  constexpr int VEC = LnCfg<T>::VEC, MAXIT = LnCfg<T>::MAXIT;
  using RV = RowVec<T, VEC>;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);  // global wave = partial row in param_ws
  const Dropout dr = make_dropout(seed, site, p_drop);
  const int nvec = d / VEC;
  float pg[MAXIT][VEC], pb[MAXIT][VEC], g[MAXIT][VEC];
  float pgf[TWO ? MAXIT : 1][VEC], pbf[TWO ? MAXIT : 1][VEC], gf[TWO ? MAXIT : 1][VEC];
#pragma unroll
  for (int it = 0; it < MAXIT; it++) {
    const int vi = it * 64 + lane;
#pragma unroll
    for (int j = 0; j < VEC; j++) {
      pg[it][j] = 0.0f; pb[it][j] = 0.0f;
      g[it][j] = (vi < nvec) ? gamma[vi * VEC + j] : 0.0f;
      if constexpr (TWO) { pgf[it][j] = 0.0f; pbf[it][j] = 0.0f; gf[it][j] = (vi < nvec) ? fr.gamma[vi * VEC + j] : 0.0f; }
    }
  }
  for (int rr = 0; rr < LN_ROWS_PER_WAVE; rr++) {
    const int row = gw * LN_ROWS_PER_WAVE + rr;
    if (row >= M) break;
    const float mean = mean_in[row], rstd = rstd_in[row];
    float xh[MAXIT][VEC], dxh[MAXIT][VEC], dm[MAXIT][VEC];
    float dyin[MAXIT][VEC];                    // the gradient that enters the layer norm (TWO: the front norm's result, rounded to T)
    if constexpr (TWO) {
      const float meanf = fr.mean[row], rstdf = fr.rstd[row];
      float hf[MAXIT][VEC], dhf[MAXIT][VEC];
      float f1 = 0.0f, f2 = 0.0f;
#pragma unroll
      for (int it = 0; it < MAXIT; it++) {
        const int vi = it * 64 + lane;
        if (vi < nvec) {
          const RV xv = *reinterpret_cast<const RV*>(fr.x + (size_t)row * d + vi * VEC);
          const RV dv = *reinterpret_cast<const RV*>(dy + (size_t)row * d + vi * VEC);
#pragma unroll
          for (int j = 0; j < VEC; j++) {
            const float h = (to_f<T>(xv.v[j]) - meanf) * rstdf;
            const float dyv = to_f<T>(dv.v[j]);
            const float dh = dyv * gf[it][j];
            hf[it][j] = h; dhf[it][j] = dh;
            f1 += dh; f2 = __fmaf_rn(dh, h, f2);
            pgf[it][j] = __fmaf_rn(dyv, h, pgf[it][j]); pbf[it][j] += dyv;
          }
        }
      }
      f1 = wave_sum(f1) / (float)d;
      f2 = wave_sum(f2) / (float)d;
#pragma unroll
      for (int it = 0; it < MAXIT; it++)
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          float t = to_f<T>(from_f<T>(__fmul_rn(rstdf, __fsub_rn(__fsub_rn(dhf[it][j], f1), __fmul_rn(hf[it][j], f2)))));
          // opaque: HIP's __fmul_rn is a plain `*`, and a product feeding `pb += dy` / `dy * g` would otherwise be contracted into
          // an fma in THIS instantiation only (fp32: the two-launch schedule rounds the product when it stores it)
          asm volatile("" : "+v"(t));
          dyin[it][j] = t;
        }
    }
    float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
      const int vi = it * 64 + lane;
      if (vi < nvec) {
        const RV xv = *reinterpret_cast<const RV*>(x + (size_t)row * d + vi * VEC);
        RV dv;
        if constexpr (!TWO) dv = *reinterpret_cast<const RV*>(dy + (size_t)row * d + vi * VEC);
        RV rv;
        if (res != nullptr) rv = *reinterpret_cast<const RV*>(res + (size_t)row * d + vi * VEC);
        float dmv[VEC];
        drop_mults<VEC>(dr, (uint32_t)row * (uint32_t)d + (uint32_t)(vi * VEC), dmv);
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          const float m = dmv[j];
          float s = to_f<T>(xv.v[j]) * m;
          if (res != nullptr) s += to_f<T>(rv.v[j]);
          const float h = (s - mean) * rstd;
          float dyv;
          if constexpr (TWO) dyv = dyin[it][j]; else dyv = to_f<T>(dv.v[j]);
          const float dh = dyv * g[it][j];
          xh[it][j] = h; dxh[it][j] = dh; dm[it][j] = m;
          c1 += dh; c2 = __fmaf_rn(dh, h, c2);      // (pinned: the one- and two-norm instantiations must contract alike)
          pg[it][j] = __fmaf_rn(dyv, h, pg[it][j]); pb[it][j] += dyv;
        }
      }
    }
    c1 = wave_sum(c1) / (float)d;
    c2 = wave_sum(c2) / (float)d;
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
      const int vi = it * 64 + lane;
      if (vi < nvec) {
        RV o1, o2;
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          const float v = __fmul_rn(rstd, __fsub_rn(__fsub_rn(dxh[it][j], c1), __fmul_rn(xh[it][j], c2)));
          o1.v[j] = from_f<T>(v);
          o2.v[j] = from_f<T>(v * dm[it][j]);
        }
        *reinterpret_cast<RV*>(ds + (size_t)row * d + vi * VEC) = o1;
        if (dxo != nullptr && dxo != ds) *reinterpret_cast<RV*>(dxo + (size_t)row * d + vi * VEC) = o2;
      }
    }
  }
  // column partials: the 4 waves of the workgroup are summed through LDS (fixed order), then ONE
  // partial row per workgroup goes to param_ws[block][0][d] (dgamma) / [block][1][d] (dbeta)
  __shared__ float psum[3][2][1024];
  const int w = threadIdx.x >> 6;
  auto fold = [&](const float (&a)[TWO ? MAXIT : 1][VEC], const float (&bb)[TWO ? MAXIT : 1][VEC], float* ws, auto N) {
    constexpr int NIT = decltype(N)::value;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int vi = it * 64 + lane;
      if (vi < nvec && w > 0) {
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          psum[w - 1][0][vi * VEC + j] = a[it][j];
          psum[w - 1][1][vi * VEC + j] = bb[it][j];
        }
      }
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int vi = it * 64 + lane;
        if (vi < nvec) {
#pragma unroll
          for (int j = 0; j < VEC; j++) {
            const int c = vi * VEC + j;
            ws[((size_t)blockIdx.x * 2 + 0) * d + c] = a[it][j] + psum[0][0][c] + psum[1][0][c] + psum[2][0][c];
            ws[((size_t)blockIdx.x * 2 + 1) * d + c] = bb[it][j] + psum[0][1][c] + psum[1][1][c] + psum[2][1][c];
          }
        }
      }
    }
  };
  if constexpr (TWO) {
    fold(pgf, pbf, fr.ws, std::integral_constant<int, MAXIT>{});
    __syncthreads();
  }
  // (the layer norm's own partials: same code on arrays of MAXIT rows)
#pragma unroll
  for (int it = 0; it < MAXIT; it++) {
    const int vi = it * 64 + lane;
    if (vi < nvec && w > 0) {
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        psum[w - 1][0][vi * VEC + j] = pg[it][j];
        psum[w - 1][1][vi * VEC + j] = pb[it][j];
      }
    }
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
      const int vi = it * 64 + lane;
      if (vi < nvec) {
#pragma unroll
        for (int j = 0; j < VEC; j++) {
          const int c = vi * VEC + j;
          param_ws[((size_t)blockIdx.x * 2 + 0) * d + c] = pg[it][j] + psum[0][0][c] + psum[1][0][c] + psum[2][0][c];
          param_ws[((size_t)blockIdx.x * 2 + 1) * d + c] = pb[it][j] + psum[0][1][c] + psum[1][1][c] + psum[2][1][c];
        }
      }
    }
  }
}

// column sums of the per-workgroup partials: block = 16 columns x 16 row slices, fixed summation order
__global__ __launch_bounds__(256) void ln_param_finalize_kernel(int nws, int d, const float* __restrict__ ws,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[16][17];
  const int cx = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;   // column in [0, 2d): first d = dgamma, next d = dbeta
  float s = 0.0f;
  if (c < 2 * d) {
    const int which = c / d, col = c % d;
    for (int r = rg; r < nws; r += 16) s += ws[((size_t)r * 2 + which) * d + col];
  }
  red[rg][cx] = s;
  __syncthreads();
  if (rg == 0 && c < 2 * d) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += red[k][cx];
    (c < d ? dgamma : dbeta)[c % d] = t;
  }
}

// batched form: one launch finalizes the dgamma/dbeta of several LayerNorms (table of 4 x int64 per
// entry: partials pointer, dgamma pointer, dbeta pointer, number of partial rows)
__global__ __launch_bounds__(256) void ln_param_finalize_batched_kernel(const int64_t* __restrict__ table, int d) {
  // a workgroup owns 32 consecutive columns of the [rows][2 d] partial rows: eight lanes x 16 bytes = one whole 128-byte line per row and
  // load (round 6; the former 16-column x 4-byte form fetched half lines: 21 us for the decoder's 34 MB = 1.6 TB/s on the critical path)
  __shared__ float red[32][33];
  const int64_t* ent = table + (size_t)blockIdx.y * 4;
  const float* ws = reinterpret_cast<const float*>(ent[0]);
  float* dgamma = reinterpret_cast<float*>(ent[1]);
  float* dbeta = reinterpret_cast<float*>(ent[2]);
  const int nws = (int)ent[3];
  const int cx = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int c = blockIdx.x * 32 + cx * 4;                 // first of this lane's four columns (d % 4 == 0: never straddles gamma | beta)
  float4 s = {0.0f, 0.0f, 0.0f, 0.0f};
  if (c < 2 * d) {
    const float* src = ws + c;
    int r = rg;
    for (; r + 7 * 32 < nws; r += 8 * 32) {               // eight loads in flight per thread (the adds keep their order)
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(r + u * 32) * 2 * d);
#pragma unroll
      for (int u = 0; u < 8; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; r < nws; r += 32) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * 2 * d);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[rg][cx * 4 + 0] = s.x; red[rg][cx * 4 + 1] = s.y; red[rg][cx * 4 + 2] = s.z; red[rg][cx * 4 + 3] = s.w;
  __syncthreads();
  const int col = blockIdx.x * 32 + (int)threadIdx.x;
  if (threadIdx.x < 32 && col < 2 * d) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; k++) t += red[k][threadIdx.x];
    (col < d ? dgamma : dbeta)[col % d] = t;
  }
}

}  // namespace vct
using namespace vct;

extern "C" int vct_ln_ws_rows(int M) {   // partial rows = workgroups of the backward kernel
  const int waves = (M + LN_ROWS_PER_WAVE - 1) / LN_ROWS_PER_WAVE;
  return (waves + 3) / 4;
}

static int ln_check(int dtype, int M, int d) {
  if (dtype != VCT_F32 && dtype != VCT_BF16) return VCT_E_ARG;
  if (M <= 0 || d <= 0) return VCT_E_SHAPE;
  const int vec = dtype == VCT_BF16 ? 8 : 4;
  if (d % vec) return VCT_E_ALIGN;
  if (d > 1024) return VCT_E_SHAPE;
  return VCT_OK;
}

static int add_ln_fwd_launch(int dtype, int M, int d, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                             float* mean, float* rstd, const uint32_t* seed, uint32_t site, float p_drop, const Ln2P& n2, bool two,
                             hipStream_t st) {
  const dim3 grid((M + 3) / 4);
#define VCT_LN_GO(T_, TWO_)                                                                                                   \
  vct::launch((add_ln_fwd_kernel<T_, TWO_>), grid, dim3(256), 0, st, M, d, (const T_*)x, (const T_*)res, gamma, beta, (T_*)y, mean, \
              rstd, seed, site, p_drop, n2)
  if (dtype == VCT_BF16) { if (two) VCT_LN_GO(bf16_t, true); else VCT_LN_GO(bf16_t, false); }
  else { if (two) VCT_LN_GO(float, true); else VCT_LN_GO(float, false); }
#undef VCT_LN_GO
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_add_ln_fwd(int dtype, int M, int d, const void* x, const void* res, const float* gamma,
                              const float* beta, void* y, float* mean, float* rstd, const uint32_t* seed,
                              uint32_t site, float p_drop, void* stream) {
  int rc = ln_check(dtype, M, d);
  if (rc) return rc;
  if (!x || !gamma || !beta || !y || !mean || !rstd) return VCT_E_ARG;
  const Ln2P none = {nullptr, nullptr, nullptr, nullptr, nullptr};
  return add_ln_fwd_launch(dtype, M, d, x, res, gamma, beta, y, mean, rstd, seed, site, p_drop, none, false, (hipStream_t)stream);
}

extern "C" int vct_add_ln_ln_fwd(int dtype, int M, int d, const void* x, const void* res, const float* gamma, const float* beta,
                                 void* y, float* mean, float* rstd, const float* gamma2, const float* beta2, void* y2, float* mean2,
                                 float* rstd2, const uint32_t* seed, uint32_t site, float p_drop, void* stream) {
  int rc = ln_check(dtype, M, d);
  if (rc) return rc;
  if (!x || !gamma || !beta || !y || !mean || !rstd || !gamma2 || !beta2 || !y2 || !mean2 || !rstd2) return VCT_E_ARG;
  const Ln2P n2 = {gamma2, beta2, y2, mean2, rstd2};
  return add_ln_fwd_launch(dtype, M, d, x, res, gamma, beta, y, mean, rstd, seed, site, p_drop, n2, true, (hipStream_t)stream);
}

static int add_ln_bwd_launch(int dtype, int M, int d, const void* dy, const void* x, const void* res, const float* gamma, const float* mean,
                             const float* rstd, void* ds, void* dxo, float* dgamma, float* dbeta, float* param_ws, const uint32_t* seed,
                             uint32_t site, float p_drop, const void* xf, const float* gammaf, const float* meanf, const float* rstdf,
                             float* wsf, void* stream) {
  int rc = ln_check(dtype, M, d);
  if (rc) return rc;
  if (!dy || !x || !gamma || !mean || !rstd || !ds || !param_ws) return VCT_E_ARG;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return VCT_E_ARG;
  const bool two = xf != nullptr;
  if (two && (!gammaf || !meanf || !rstdf || !wsf || dgamma != nullptr)) return VCT_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nws = vct_ln_ws_rows(M);
  const dim3 grid(nws);
  if (dtype == VCT_BF16) {
    const LnFront<bf16_t> fr{(const bf16_t*)xf, gammaf, meanf, rstdf, wsf};
    if (two) vct::launch((add_ln_bwd_kernel<bf16_t, true>), grid, dim3(256), 0, st, M, d, (const bf16_t*)dy, (const bf16_t*)x,
                         (const bf16_t*)res, gamma, mean, rstd, (bf16_t*)ds, (bf16_t*)dxo, param_ws, seed, site, p_drop, fr);
    else vct::launch((add_ln_bwd_kernel<bf16_t, false>), grid, dim3(256), 0, st, M, d, (const bf16_t*)dy, (const bf16_t*)x,
                     (const bf16_t*)res, gamma, mean, rstd, (bf16_t*)ds, (bf16_t*)dxo, param_ws, seed, site, p_drop, fr);
  } else {
    const LnFront<float> fr{(const float*)xf, gammaf, meanf, rstdf, wsf};
    if (two) vct::launch((add_ln_bwd_kernel<float, true>), grid, dim3(256), 0, st, M, d, (const float*)dy, (const float*)x,
                         (const float*)res, gamma, mean, rstd, (float*)ds, (float*)dxo, param_ws, seed, site, p_drop, fr);
    else vct::launch((add_ln_bwd_kernel<float, false>), grid, dim3(256), 0, st, M, d, (const float*)dy, (const float*)x,
                     (const float*)res, gamma, mean, rstd, (float*)ds, (float*)dxo, param_ws, seed, site, p_drop, fr);
  }
  VCT_CHECK_LAUNCH();
  if (dgamma != nullptr && dbeta != nullptr) {   // NULL: the caller finalizes later with vct_ln_param_finalize_batched
    vct::launch(ln_param_finalize_kernel, dim3((2 * d + 15) / 16), dim3(256), 0, st, nws, d, param_ws, dgamma, dbeta);
    VCT_CHECK_LAUNCH();
  }
  return VCT_OK;
}

extern "C" int vct_add_ln_bwd(int dtype, int M, int d, const void* dy, const void* x, const void* res,
                              const float* gamma, const float* mean, const float* rstd, void* ds, void* dxo,
                              float* dgamma, float* dbeta, float* param_ws, const uint32_t* seed, uint32_t site,
                              float p_drop, void* stream) {
  return add_ln_bwd_launch(dtype, M, d, dy, x, res, gamma, mean, rstd, ds, dxo, dgamma, dbeta, param_ws, seed, site, p_drop, nullptr, nullptr,
                           nullptr, nullptr, nullptr, stream);
}

extern "C" int vct_add_ln_ln_bwd(int dtype, int M, int d, const void* dy2, const void* y, const float* gamma2, const float* mean2,
                                 const float* rstd2, float* param_ws2, const void* x, const void* res, const float* gamma,
                                 const float* mean, const float* rstd, void* ds, void* dxo, float* param_ws, const uint32_t* seed,
                                 uint32_t site, float p_drop, void* stream) {
  if (!y) return VCT_E_ARG;
  return add_ln_bwd_launch(dtype, M, d, dy2, x, res, gamma, mean, rstd, ds, dxo, nullptr, nullptr, param_ws, seed, site, p_drop, y, gamma2,
                           mean2, rstd2, param_ws2, stream);
}

extern "C" int vct_ln_param_finalize_batched(const int64_t* table_dev, int n_entries, int d, void* stream) {
  if (!table_dev) return VCT_E_ARG;
  if (n_entries <= 0 || d <= 0) return VCT_E_SHAPE;
  vct::launch(ln_param_finalize_batched_kernel, dim3((2 * d + 31) / 32, n_entries), dim3(256), 0, (hipStream_t)stream,
                     table_dev, d);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}
