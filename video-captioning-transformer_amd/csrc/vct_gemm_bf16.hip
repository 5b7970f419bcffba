// bf16 GEMM v2 dispatch over the three per-layout translation units (kernel: vct_gemm_bf16_kernel.h)
#include "vct_common.h"
#include "vct_gemm_params.h"
namespace vct {
int gemm_bf16_v2_nt(const vct_gemm_desc*, const GemmP&, int, int, int, dim3, hipStream_t);
int gemm_bf16_v2_nn(const vct_gemm_desc*, const GemmP&, int, int, int, dim3, hipStream_t);
int gemm_bf16_v2_tn(const vct_gemm_desc*, const GemmP&, int, int, int, dim3, hipStream_t);
// called by vct_gemm (vct_gemm.hip) for dtype == VCT_BF16
int gemm_bf16_v2_dispatch(const vct_gemm_desc* d, const GemmP& p, int bm, int bn, int nbuf, dim3 grid, hipStream_t st) {
  switch (d->ta * 2 + d->tb) {
    case 1: return gemm_bf16_v2_nt(d, p, bm, bn, nbuf, grid, st);
    case 0: return gemm_bf16_v2_nn(d, p, bm, bn, nbuf, grid, st);
    case 2: return gemm_bf16_v2_tn(d, p, bm, bn, nbuf, grid, st);
    default: return VCT_E_SHAPE;
  }
}
}  // namespace vct
