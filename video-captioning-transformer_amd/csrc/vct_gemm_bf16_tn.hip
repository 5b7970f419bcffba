// bf16 GEMM v2, operand layout TN (ta=1, tb=0); kernel in vct_gemm_bf16_kernel.h
#include "vct_gemm_bf16_kernel.h"
namespace vct {
int gemm_bf16_v2_tn(const vct_gemm_desc* d, const GemmP& p, int bm, int bn, int nbuf, dim3 grid, hipStream_t st) {
  return gemm_bf16_v2_layout<1, 0>(d, p, bm, bn, nbuf, grid, st);
}
// grouped weight gradients: dW = dY^T X, fp32 out (vct_gemm_grouped)
int gemm_bf16_v2_grouped_tn(const GemmGroupP& g, int bm, int bn, int total_wg, hipStream_t st) {
  if (bm == 128 && bn == 128)
    vct::launch((gemm_bf16_v2_grouped_kernel<float, 1, 0, 128, 128, 2, 2, 4>), dim3(total_wg), dim3(512), 0, st, g);
  else if (bm == 128 && bn == 64)
    vct::launch((gemm_bf16_v2_grouped_kernel<float, 1, 0, 128, 64, 2, 4, 2>), dim3(total_wg), dim3(512), 0, st, g);
  else if (bm == 64 && bn == 64)
    vct::launch((gemm_bf16_v2_grouped_kernel<float, 1, 0, 64, 64, 2, 2, 2>), dim3(total_wg), dim3(256), 0, st, g);
  else
    return VCT_E_SHAPE;
  return VCT_OK;
}
}  // namespace vct
