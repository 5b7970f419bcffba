// Device-side helpers shared by the gfx950 kernels (wave64, MFMA fragment types, bf16 packing,
// counter-hash dropout, wave/block reductions).  CDNA4 only -- no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include <type_traits>
#include "../../include/vct_hip.h"
#include "vct_runtime.h"

namespace vct {

// Agent-scope (`sc1`) stores: written through the XCD's L2 without displacing what is resident there.  Measured on gfx950 with
// FETCH_SIZE / TCC_EA0_RDREQ (tools/g256_traffic.sh, the 4864 x 30522 x 512 vocabulary projection): a plain store and a non-temporal
// one (`nt`) both allocate in the write-back L2 and the 297 MB of logits evict the weight panels between two rounds of tiles (257 MB
// fetched for 36 MB of operands); with these stores 141 MB.  The launch then ends only when the write-through has drained, which costs
// more than the re-fetches (served by the memory-side cache) did -- an experiment switch (VCT_GEMM_NT=1), not a default.
typedef __attribute__((ext_vector_type(4))) uint32_t stream_u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t stream_u32x2;
__device__ __forceinline__ void store_stream16(void* dst, const stream_u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void store_stream8(void* dst, const stream_u32x2 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
}


typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;  // storage type of a bf16 element

constexpr int WAVE = 64;

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // hardware conversion (v_cvt_pk_bf16_f32 on gfx950: round-to-nearest-even, NaN quieted) -- the compiler
  // pairs adjacent conversions into one packed instruction; ~6 VALU ops cheaper than bit twiddling
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// ---- activation ------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float act_f(int act, float x) {
  return act == VCT_ACT_GELU ? gelu_f(x) : (act == VCT_ACT_RELU ? fmaxf(x, 0.0f) : x);
}
__device__ __forceinline__ float dact_f(int act, float x) {
  return act == VCT_ACT_GELU ? dgelu_f(x) : (act == VCT_ACT_RELU ? (x > 0.0f ? 1.0f : 0.0f) : 1.0f);
}

// bf16 throughput mode: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below a bf16 ulp): one v_exp, one v_rcp
// and a 5-term polynomial instead of libm's erff (~40 VALU ops + branches).  The GEMM epilogues apply GELU / GELU' to
// 16 elements per lane; with erff that epilogue cost as much as the K loop of the FFN GEMMs (34-40 us in the step
// against 20 us without the activation).  GELU' shares the exponential: exp(-(x/sqrt2)^2) is also the Gaussian pdf.
__device__ __forceinline__ void erf_cdf_pdf_fast(float x, float& cdf, float& pdf) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float e = __expf(-ax * ax);                                   // = exp(-x^2 / 2)
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);     // v_rcp_f32 (1 ulp); __frcp_rn expands to a ten-instruction correctly rounded division
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;                              // erf(|x| / sqrt2)
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  pdf = 0.3989422804014327f * e;
}
// The same for TWO elements per lane: the polynomial, the scalings and the sign transfer go through packed fp32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth per issue slot), only exp and rcp stay per element.  The epilogues of the FFN
// GEMMs spend their time in exactly this arithmetic (tools/ffn1_epi_bench.py: linear1 forward 19 us with a bias-only epilogue,
// 29 us with GELU).
typedef __attribute__((ext_vector_type(2))) float vf2;
__device__ __forceinline__ void erf_cdf_pdf_fast2(const vf2 x, vf2& cdf, vf2& pdf) {
  const vf2 ax = __builtin_elementwise_abs(x) * 0.70710678118654752f;
  const vf2 m = -ax * ax;
  vf2 e; e[0] = __expf(m[0]); e[1] = __expf(m[1]);
  const vf2 den = ax * 0.3275911f + 1.0f;
  vf2 t; t[0] = __builtin_amdgcn_rcpf(den[0]); t[1] = __builtin_amdgcn_rcpf(den[1]);
  const vf2 poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const vf2 erf_abs = 1.0f - poly * e;
  vf2 sg; sg[0] = copysignf(erf_abs[0], x[0]); sg[1] = copysignf(erf_abs[1], x[1]);
  cdf = sg * 0.5f + 0.5f;
  pdf = e * 0.3989422804014327f;
}
__device__ __forceinline__ vf2 act_fast_f2(int act, const vf2 x) {
  if (act == VCT_ACT_GELU) { vf2 c, d; erf_cdf_pdf_fast2(x, c, d); return x * c; }
  if (act == VCT_ACT_RELU) { vf2 r; r[0] = fmaxf(x[0], 0.0f); r[1] = fmaxf(x[1], 0.0f); return r; }
  return x;
}
__device__ __forceinline__ vf2 dact_fast_f2(int act, const vf2 x) {
  if (act == VCT_ACT_GELU) { vf2 c, d; erf_cdf_pdf_fast2(x, c, d); return c + x * d; }
  vf2 r = {1.0f, 1.0f};
  if (act == VCT_ACT_RELU) { r[0] = x[0] > 0.0f ? 1.0f : 0.0f; r[1] = x[1] > 0.0f ? 1.0f : 0.0f; }
  return r;
}
__device__ __forceinline__ float act_fast_f(int act, float x) {
  if (act == VCT_ACT_GELU) { float c, d; erf_cdf_pdf_fast(x, c, d); return x * c; }
  return act == VCT_ACT_RELU ? fmaxf(x, 0.0f) : x;
}
__device__ __forceinline__ float dact_fast_f(int act, float x) {
  if (act == VCT_ACT_GELU) { float c, d; erf_cdf_pdf_fast(x, c, d); return c + x * d; }
  return act == VCT_ACT_RELU ? (x > 0.0f ? 1.0f : 0.0f) : 1.0f;
}

// ---- dropout: counter hash (seed, site, element index) -> 32 random bits ----------------------
// Stateless so the backward kernels regenerate the forward mask.  Two multiply-xorshift rounds
// (murmur3 fmix32 with a golden-ratio pre-mix): plenty for a Bernoulli mask.
struct Dropout {
  uint32_t key;     // seed ^ site mix
  uint32_t thresh;  // drop if the element's 16-bit field < thresh (p rounded to 1 / 65536)
  float scale;      // 1/(1-p)
  bool on;
};
__device__ __forceinline__ Dropout make_dropout(const uint32_t* seed, uint32_t site, float p) {
  Dropout d;
  d.on = (seed != nullptr) && (p > 0.0f);
  d.key = d.on ? (seed[0] * 0x9E3779B1u) ^ (site * 0x85EBCA77u + 0x165667B1u) : 0u;
  d.thresh = d.on ? (uint32_t)fminf(p * 65536.0f + 0.5f, 65535.0f) : 0u;
  d.scale = d.on ? 1.0f / (1.0f - p) : 1.0f;
  return d;
}
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// One hash serves the element PAIR (2k, 2k + 1): low / high 16 bits.  The hash is three 32-bit integer multiplies (quarter rate)
// and eight shifts / xors -- as much VALU time per element as the fast GELU -- and the GEMM epilogues, the LayerNorm and the
// embedding kernels all walk consecutive elements: drop_mults below pays for it once per two elements.
// multiplier to apply to element `idx` (0 or 1/(1-p)); 1 when dropout is off
__device__ __forceinline__ float drop_mult(const Dropout& d, uint32_t idx) {
  if (!d.on) return 1.0f;
  const uint32_t h = hash32(((idx >> 1) * 0x9E3779B1u) ^ d.key);
  const uint32_t f = (idx & 1u) ? (h >> 16) : (h & 0xffffu);
  return f < d.thresh ? 0.0f : d.scale;
}
// ... of the N consecutive elements idx .. idx + N - 1 (N even; one hash per pair when idx is even, which it is wherever rows have
// an even length)
template <int N> __device__ __forceinline__ void drop_mults(const Dropout& d, uint32_t idx, float (&m)[N]) {
  static_assert(N % 2 == 0, "pairs");
  if (!d.on) {
#pragma unroll
    for (int j = 0; j < N; j++) m[j] = 1.0f;
    return;
  }
  if ((idx & 1u) == 0u) {
#pragma unroll
    for (int k = 0; k < N / 2; k++) {
      const uint32_t h = hash32((((idx >> 1) + (uint32_t)k) * 0x9E3779B1u) ^ d.key);
      m[2 * k] = (h & 0xffffu) < d.thresh ? 0.0f : d.scale;
      m[2 * k + 1] = (h >> 16) < d.thresh ? 0.0f : d.scale;
    }
  } else {
#pragma unroll
    for (int j = 0; j < N; j++) m[j] = drop_mult(d, idx + (uint32_t)j);
  }
}

// ---- reductions ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// block reductions over NW waves; `red` is LDS scratch of >= NW floats; all threads get the result
template <int NW> __device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NW; i++) s += red[i];
  return s;
}
template <int NW> __device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float s = red[0];
#pragma unroll
  for (int i = 1; i < NW; i++) s = fmaxf(s, red[i]);
  return s;
}

// ---- LDS transpose read (gfx950 ds_read_b64_tr_b16) -------------------------------------------
// Within each 16-lane group: lane i supplies the address of 4 contiguous bf16; lane i receives, for
// j = 0..3, element (i & 3) of the chunk supplied by lane 4*j + (i >> 2).  When lanes 4r..4r+3 point
// at the four 8-byte chunks of row r of a [4][16] block, lane i ends up with column i (rows 0..3).
// (Semantics confirmed on hardware by tools/hw_probe.hip.)
__device__ __forceinline__ s16x4 lds_tr16(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) -- guarantees that arrays
// indexed by the loop variable stay in registers (a runtime-indexed accumulator array goes to scratch)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <typename T> struct DTypeOf;
template <> struct DTypeOf<float> { static constexpr int v = VCT_F32; };
template <> struct DTypeOf<bf16_t> { static constexpr int v = VCT_BF16; };

}  // namespace vct

#define VCT_CHECK_LAUNCH()                               \
  do {                                                   \
    hipError_t e__ = hipGetLastError();                  \
    if (e__ != hipSuccess) return (int)e__;              \
  } while (0)
