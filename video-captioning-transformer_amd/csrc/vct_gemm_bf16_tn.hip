// bf16 GEMM v2, operand layout TN (ta=1, tb=0); kernel in vct_gemm_bf16_kernel.h
#include "vct_gemm_bf16_kernel.h"
namespace vct {
int gemm_bf16_v2_tn(const vct_gemm_desc* d, const GemmP& p, int bm, int bn, int nbuf, dim3 grid, hipStream_t st) {
  return gemm_bf16_v2_layout<1, 0>(d, p, bm, bn, nbuf, grid, st);
}
}  // namespace vct
