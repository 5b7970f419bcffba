// torch.optim.Adam's single-tensor update (reference train.py:24-26,126) as ONE shared device function: the flat pass (adam_kernel),
// the 2-D pass that also transposes (adam2d_kernel), the multi-range pass (adam_ranges_kernel) and the optimizer epilogue of the
// weight-gradient GEMMs (vct_gemm_bf16_kernel.h) all inline exactly this expression tree, so a parameter gets bit-identical values
// whichever of them steps it (tests/test_kernels_gpu.py pins that).
//   m += (1-b1)(g-m); v = b2 v + (1-b2) g^2; p = p (1 - lr wd) - (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps),  bc = 1 - beta^t
#pragma once
#include "vct_common.h"

namespace vct {

struct AdamConsts { float b1, b2, eps, step_size, decay, bc2s; };

// hyper = {lr, beta1, beta2, eps, weight_decay} in DEVICE memory when given (a captured hipGraph / recorded launch list then follows
// the learning-rate schedule: kernel arguments are frozen at capture time); t = step[0] + 1 from a device counter
__device__ __forceinline__ AdamConsts adam_consts(float lr, float b1, float b2, float eps, float wd, const float* hyper, const int32_t* step) {
  if (hyper != nullptr) { lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; wd = hyper[4]; }
  const float t = (float)(step[0] + 1);
  const float bc1 = 1.0f - powf(b1, t);
  AdamConsts c;
  c.b1 = b1; c.b2 = b2; c.eps = eps;
  c.bc2s = sqrtf(1.0f - powf(b2, t));
  c.step_size = lr / bc1;
  c.decay = 1.0f - lr * wd;
  return c;
}

// the same constants pinned to SCALAR registers (they are wave-uniform): the GEMM epilogue that carries them has no vector register
// to spare (a 128 x 128 eight-wave tile sits at its 128-register occupancy limit)
__device__ __forceinline__ float adam_uniform(const float x) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
__device__ __forceinline__ AdamConsts adam_consts_uniform(const float* hyper, const int32_t* step) {
  AdamConsts c = adam_consts(0.0f, 0.0f, 0.0f, 0.0f, 0.0f, hyper, step);
  c.b1 = adam_uniform(c.b1); c.b2 = adam_uniform(c.b2); c.eps = adam_uniform(c.eps);
  c.step_size = adam_uniform(c.step_size); c.decay = adam_uniform(c.decay); c.bc2s = adam_uniform(c.bc2s);
  return c;
}

// Every rounding is pinned (explicit fma / mul / add intrinsics: no -ffp-contract decision left to the compiler): the same update
// inlined into four different kernels otherwise contracts differently from one to the next -- exp_avg_sq came out one ulp apart in a
// quarter of the elements between the flat pass and the GEMM epilogue (round 5, tools/_dbg_adam.py).
__device__ __forceinline__ void adam_update(float& p, const float g, float& m, float& v, const AdamConsts& c) {
  const float w0 = __fmul_rn(p, c.decay);
  m = __fmaf_rn(__fsub_rn(1.0f, c.b1), __fsub_rn(g, m), m);
  v = __fmaf_rn(__fmul_rn(__fsub_rn(1.0f, c.b2), g), g, __fmul_rn(c.b2, v));     // from v = 0: exactly ((1 - b2) g) g (tests pin that)
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), c.bc2s), c.eps);
  p = __fmaf_rn(-c.step_size, __fdiv_rn(m, denom), w0);
}

// Stream-order packed copy (vct_ss_pack layout) of a weight matrix [N, K]: where the 4 consecutive elements (r, c .. c+3) of the WHOLE
// matrix live in the stream.  mode 0 = packed as 512-ROW blocks (block r / 512 starts at chunk chunk0[r / 512]), mode 1 = a 512-row
// matrix packed as 512-COLUMN K slices (slice c / 512 at chunk0[c / 512]).  Returns the bf16 element index or -1 (block not packed).
// chunk = 64 columns of K; inside: wave nn / 64, tile (nn % 64) / 16, k-step, then lane = (k group, row % 16), 8 bf16 each.
__device__ __forceinline__ unsigned long long adam_pack_chunks(const int c0, const int c1, const int c2, const int c3) {
  auto f = [](int c) { return (unsigned long long)(c < 0 || c >= 0xffff ? 0xffff : c); };
  return f(c0) | (f(c1) << 16) | (f(c2) << 32) | (f(c3) << 48);
}
// chunks: first chunk of the four blocks as 16-bit fields (adam_pack_chunks; 0xffff = not packed) -- selected by a shift, never by an index
__device__ __forceinline__ int64_t adam_pack_index(const int r, const int c, const int mode, const unsigned long long chunks) {
  const int blk = mode == 0 ? (r >> 9) : (c >> 9);
  const int nn = mode == 0 ? (r & 511) : r, kk = mode == 0 ? c : (c & 511);
  if (blk >= 4) return -1;
  const int ch0 = (int)((chunks >> (16 * blk)) & 0xffffull);
  if (ch0 == 0xffff) return -1;
  const int64_t vec = (int64_t)(ch0 + (kk >> 6)) * 4096 + (((nn >> 6) * 8 + ((nn & 63) >> 4) * 2 + ((kk & 63) >> 5)) * 64 +
                                                            ((kk & 31) >> 3) * 16 + (nn & 15));
  return vec * 8 + (kk & 7);
}

}  // namespace vct
