// Fused Adam / AdamW over the flat fp32 parameter buffer (one launch for all 46 M parameters) that
// also refreshes the bf16 compute shadow, so no separate cast pass is needed after the step.
// HBM-bound: 16 B (p,g,m,v reads) + 12 B (p,m,v writes) + 2 B (shadow) per parameter.
// Arithmetic follows torch.optim.Adam's single-tensor path (reference train.py:24-26,126):
//   m += (1-b1)(g-m); v = b2 v + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// with bc = 1 - beta^t and t read from a DEVICE counter (the step stays hipGraph-capturable).
#include "vct_adam_core.h"

namespace vct {

// A weight matrix [N, K] at flat elements [begin, end) that has a stream-order packed copy (vct_ss_pack: the sample-stationary
// stack kernels' operand): mode 0 = packed as 512-ROW blocks (block r / 512 starts at chunk chunk0[r / 512]), mode 1 = packed as
// 512-COLUMN K slices of a 512-row matrix (slice c / 512 at chunk0[c / 512]).  The optimizer's pass writes the packed copy with the
// shadow: +2 B per parameter instead of a pack launch (read + write of the stream) behind it on the step's critical path.
struct AdamPackSeg { int64_t begin, end; int32_t K, mode; int32_t chunk0[4]; bf16_t* stream; };

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ shadow, int64_t n, float lr,
                                                   float b1, float b2, float eps, float wd, const int32_t* __restrict__ step,
                                                   int64_t skip_a, int64_t skip_b, const float* __restrict__ hyper,
                                                   const AdamPackSeg* __restrict__ segs, int nseg, int64_t base) {
  const AdamConsts hc = adam_consts(lr, b1, b2, eps, wd, hyper, step);
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; j++) adam_update(pp[j], gp[j], mp[j], vp[j], hc);
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
    const int64_t e = i << 2;
    if (shadow != nullptr && !(e >= skip_a && e < skip_b)) {
      ushort4 o;
      o.x = f2bf(pv.x); o.y = f2bf(pv.y); o.z = f2bf(pv.z); o.w = f2bf(pv.w);
      reinterpret_cast<ushort4*>(shadow)[i] = o;
      if (nseg > 0) {
        const int64_t ef = base + e;
        int lo = 0, hi = nseg;                               // last segment that begins at or before ef (sorted by begin)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (segs[mid].begin <= ef) lo = mid; else hi = mid; }
        const AdamPackSeg sg = segs[lo];
        if (ef >= sg.begin && ef < sg.end) {
          const int64_t rel = ef - sg.begin;
          const int r = (int)(rel / sg.K), c = (int)(rel - (int64_t)r * sg.K);
          const int64_t at = adam_pack_index(r, c, sg.mode, adam_pack_chunks(sg.chunk0[0], sg.chunk0[1], sg.chunk0[2], sg.chunk0[3]));
          if (at >= 0) *reinterpret_cast<ushort4*>(sg.stream + at) = o;
        }
      }
    }
  }
}

// The same step over ONE 2-D weight [rows, cols] (row-major, contiguous in the flat buffer) that ALSO writes the transposed bf16
// shadow WT[cols][ld_t] -- the operand of the NT form of dX = dY W (vocabulary projection: W_g^T).  A workgroup owns a 64 x 64
// tile: float4 accesses along the rows for p / g / m / v / shadow, then the bf16 tile goes through LDS and leaves as 16-byte
// pieces of the transposed rows (128 contiguous bytes per output row).  +2 B per parameter on a 30 B per parameter kernel,
// instead of a separate 62 MB transpose pass (35 us) behind the optimizer.
__global__ __launch_bounds__(256) void adam2d_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, bf16_t* __restrict__ shadow, bf16_t* __restrict__ shadow_t,
                                                     int rows, int cols, int64_t ld_t, float lr, float b1, float b2, float eps, float wd,
                                                     const int32_t* __restrict__ step, const float* __restrict__ hyper) {
  constexpr int STR = 72;                                    // LDS row stride in elements (16-byte aligned rows)
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * STR];
  const AdamConsts hc = adam_consts(lr, b1, b2, eps, wd, hyper, step);
  const int tiles_c = cols / 64;
  const int r0 = (blockIdx.x / tiles_c) * 64, c0 = (blockIdx.x % tiles_c) * 64, tid = threadIdx.x;
  float4 pv[4], gv[4], mv[4], vv[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {                              // all 16 loads of the thread first
    const int r = min(r0 + (tid >> 4) + 16 * i, rows - 1);
    const int64_t e = (int64_t)r * cols + c0 + (tid & 15) * 4;
    pv[i] = *reinterpret_cast<const float4*>(p + e); gv[i] = *reinterpret_cast<const float4*>(g + e);
    mv[i] = *reinterpret_cast<const float4*>(m + e); vv[i] = *reinterpret_cast<const float4*>(v + e);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int rl = (tid >> 4) + 16 * i, r = r0 + rl;
    float* pp = &pv[i].x; const float* gp = &gv[i].x; float* mp = &mv[i].x; float* vp = &vv[i].x;
#pragma unroll
    for (int j = 0; j < 4; j++) adam_update(pp[j], gp[j], mp[j], vp[j], hc);
    ushort4 o;
    o.x = f2bf(pv[i].x); o.y = f2bf(pv[i].y); o.z = f2bf(pv[i].z); o.w = f2bf(pv[i].w);
    *reinterpret_cast<ushort4*>(tile + rl * STR + (tid & 15) * 4) = o;
    if (r < rows) {
      const int64_t e = (int64_t)r * cols + c0 + (tid & 15) * 4;
      *reinterpret_cast<float4*>(p + e) = pv[i];
      *reinterpret_cast<float4*>(m + e) = mv[i];
      *reinterpret_cast<float4*>(v + e) = vv[i];
      if (shadow != nullptr) *reinterpret_cast<ushort4*>(shadow + e) = o;
    }
  }
  __syncthreads();
  struct alignas(16) V8 { bf16_t e[8]; };
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int cc = tid & 63, ch = (tid >> 6) + 4 * i;        // transposed row c0 + cc, its elements r0 + ch*8 .. +8
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o.e[j] = tile[(ch * 8 + j) * STR + cc];
    const int r = r0 + ch * 8;
    bf16_t* dst = shadow_t + (int64_t)(c0 + cc) * ld_t + r;
    if (r + 8 <= rows) *reinterpret_cast<V8*>(dst) = o;
    else for (int j = 0; j < 8; j++) if (r + j < rows) dst[j] = o.e[j];
  }
}

// The same step over a LIST of flat ranges in one launch: what is left of the parameter buffer when the weight matrices are stepped
// inside their own weight-gradient GEMMs (vct_gemm_adam) -- biases, LayerNorm parameters, the token-embedding table, and any matrix
// whose product did not take the epilogue.  ranges (device, sorted): [begin, end) flat elements (multiples of 4), blk0 = first
// workgroup of the range (ADAM_RANGE_EPB elements per workgroup), shadow != 0: the bf16 shadow (and packed copies) follow.
struct AdamRange { int64_t begin, end; int32_t blk0, shadow; };
constexpr int ADAM_RANGE_EPB = 4096;
__global__ __launch_bounds__(256) void adam_ranges_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, bf16_t* __restrict__ shadow,
                                                          const AdamRange* __restrict__ ranges, int nranges, float lr, float b1, float b2,
                                                          float eps, float wd, const int32_t* __restrict__ step,
                                                          const float* __restrict__ hyper, const AdamPackSeg* __restrict__ segs, int nseg) {
  const AdamConsts hc = adam_consts(lr, b1, b2, eps, wd, hyper, step);
  int lo = 0, hi = nranges;                                  // last range whose first workgroup is <= this one
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ranges[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const AdamRange rg = ranges[lo];
  const int64_t e0 = rg.begin + (int64_t)((int)blockIdx.x - rg.blk0) * ADAM_RANGE_EPB;
  const int64_t e1 = e0 + ADAM_RANGE_EPB < rg.end ? e0 + ADAM_RANGE_EPB : rg.end;
  for (int64_t e = e0 + (int64_t)threadIdx.x * 4; e < e1; e += 256 * 4) {
    float4 pv = *reinterpret_cast<float4*>(p + e);
    const float4 gv = *reinterpret_cast<const float4*>(g + e);
    float4 mv = *reinterpret_cast<float4*>(m + e);
    float4 vv = *reinterpret_cast<float4*>(v + e);
    float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; j++) adam_update(pp[j], gp[j], mp[j], vp[j], hc);
    *reinterpret_cast<float4*>(p + e) = pv;
    *reinterpret_cast<float4*>(m + e) = mv;
    *reinterpret_cast<float4*>(v + e) = vv;
    if (shadow != nullptr && rg.shadow) {
      ushort4 o;
      o.x = f2bf(pv.x); o.y = f2bf(pv.y); o.z = f2bf(pv.z); o.w = f2bf(pv.w);
      *reinterpret_cast<ushort4*>(shadow + e) = o;
      if (nseg > 0) {
        int slo = 0, shi = nseg;
        while (shi - slo > 1) { const int mid = (slo + shi) >> 1; if (segs[mid].begin <= e) slo = mid; else shi = mid; }
        const AdamPackSeg sg = segs[slo];
        if (e >= sg.begin && e < sg.end) {
          const int64_t rel = e - sg.begin;
          const int r = (int)(rel / sg.K), c = (int)(rel - (int64_t)r * sg.K);
          const int64_t at = adam_pack_index(r, c, sg.mode, adam_pack_chunks(sg.chunk0[0], sg.chunk0[1], sg.chunk0[2], sg.chunk0[3]));
          if (at >= 0) *reinterpret_cast<ushort4*>(sg.stream + at) = o;
        }
      }
    }
  }
}

__global__ void bump_step_kernel(int32_t* step) { step[0] += 1; }

}  // namespace vct
using namespace vct;

extern "C" int vct_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                             int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                             int32_t* step_dev, int64_t shadow_skip_begin, int64_t shadow_skip_end, int32_t bump_step,
                             const float* hyper_dev, void* stream) {
  return vct_adam_step_pk(param, grad, exp_avg, exp_avg_sq, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, step_dev, shadow_skip_begin,
                          shadow_skip_end, bump_step, hyper_dev, nullptr, 0, 0, stream);
}

extern "C" int vct_adam_step_pk(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                                int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                int32_t* step_dev, int64_t shadow_skip_begin, int64_t shadow_skip_end, int32_t bump_step,
                                const float* hyper_dev, const vct_adam_pack_seg* segs_dev, int32_t nseg, int64_t base, void* stream) {
  static_assert(sizeof(vct_adam_pack_seg) == sizeof(AdamPackSeg), "descriptor layout");
  if (nseg < 0 || (nseg > 0 && (segs_dev == nullptr || shadow_bf16 == nullptr || (base & 3)))) return VCT_E_ARG;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev) return VCT_E_ARG;
  if (n < 0 || (n & 3)) return VCT_E_SHAPE;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return VCT_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {   // bump only
    if (bump_step) { vct::launch(bump_step_kernel, dim3(1), dim3(1), 0, st, step_dev); VCT_CHECK_LAUNCH(); }
    return VCT_OK;
  }
  const int64_t want = ((n >> 2) + 255) / 256;
  const int blocks = (int)(want > 8192 ? 8192 : want);
  vct::launch(adam_kernel, dim3(blocks), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, (bf16_t*)shadow_bf16, n, lr,
                     beta1, beta2, eps, weight_decay, step_dev, shadow_skip_begin, shadow_skip_end, hyper_dev,
                     reinterpret_cast<const AdamPackSeg*>(segs_dev), (int)nseg, base);
  VCT_CHECK_LAUNCH();
  if (bump_step) {
    vct::launch(bump_step_kernel, dim3(1), dim3(1), 0, st, step_dev);
    VCT_CHECK_LAUNCH();
  }
  return VCT_OK;
}

extern "C" int vct_adam_step_ranges(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                                    const vct_adam_range* ranges_dev, int32_t nranges, int32_t total_blocks, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int32_t* step_dev, const float* hyper_dev,
                                    const vct_adam_pack_seg* segs_dev, int32_t nseg, void* stream) {
  static_assert(sizeof(vct_adam_range) == sizeof(AdamRange), "descriptor layout");
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev || !ranges_dev || nranges < 1 || total_blocks < 1) return VCT_E_ARG;
  if (nseg < 0 || (nseg > 0 && (segs_dev == nullptr || shadow_bf16 == nullptr))) return VCT_E_ARG;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return VCT_E_ALIGN;
  vct::launch(adam_ranges_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
              (bf16_t*)shadow_bf16, reinterpret_cast<const AdamRange*>(ranges_dev), (int)nranges, lr, beta1, beta2, eps, weight_decay,
              step_dev, hyper_dev, reinterpret_cast<const AdamPackSeg*>(segs_dev), (int)nseg);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_adam_step_2d(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                                void* shadow_t_bf16, int32_t rows, int32_t cols, int64_t ld_t, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int32_t* step_dev, const float* hyper_dev, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev || !shadow_t_bf16) return VCT_E_ARG;
  if (rows < 1 || cols < 64 || (cols % 64) || ld_t < rows || (ld_t % 8)) return VCT_E_SHAPE;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)shadow_t_bf16) & 15) return VCT_E_ALIGN;
  if (shadow_bf16 != nullptr && ((uintptr_t)shadow_bf16 & 7)) return VCT_E_ALIGN;
  const int blocks = ((rows + 63) / 64) * (cols / 64);
  vct::launch(adam2d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (bf16_t*)shadow_bf16,
              (bf16_t*)shadow_t_bf16, rows, cols, ld_t, lr, beta1, beta2, eps, weight_decay, step_dev, hyper_dev);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}
