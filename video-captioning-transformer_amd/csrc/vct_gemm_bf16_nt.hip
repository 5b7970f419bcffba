// bf16 GEMM v2, operand layout NT (ta=0, tb=1); kernel in vct_gemm_bf16_kernel.h
#include "vct_gemm_bf16_kernel.h"
namespace vct {
int gemm_bf16_v2_nt(const vct_gemm_desc* d, const GemmP& p, int bm, int bn, int nbuf, dim3 grid, hipStream_t st) {
  return gemm_bf16_v2_layout<0, 1>(d, p, bm, bn, nbuf, grid, st);
}
}  // namespace vct
