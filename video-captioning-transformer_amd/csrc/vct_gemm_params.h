// Kernel-side parameter block shared by the GEMM translation units.
#pragma once
#include <stdint.h>
#include "../../include/vct_hip.h"

namespace vct {

struct GemmP {
  const void* A; const void* B; void* C;
  long lda, ldb, ldc;
  int M, N, K;
  int kt_per_split;  // K tiles per blockIdx.z
  int tiles_m, tiles_n;
  int act;        // forward activation applied after bias
  int dact_kind;  // activation whose derivative multiplies the result (dact != nullptr)
  const float* bias;
  void* preact; long ld_preact;
  const void* addend; long ld_addend;
  const void* dact; long ld_dact;
  const uint32_t* seed; uint32_t site; float p_drop;
  float* bias_grad;
  float* partial;       // != nullptr: split-K mode, raw accumulators to partial[z][M*N]
  float* bias_partial;  // split-K mode: [z][M]
  int waves8;           // 128x128 tile with 8 waves (2x4) instead of 4 (2x2)
  int split;            // number of K splits (gridDim.z of the single launch)
  int* counters;        // != nullptr: per-tile arrival counters (zero on entry/exit): single-pass split-K
  int nt_store;         // output store policy: 0 plain, 1 agent-scope streaming (vct_common.h), 2 non-temporal (experiments; default 0)
  int nt_preact;        // the saved pre-activation is not read again before the backward: non-temporal stores
  int short_fast;       // tile order: the SHORTER tile dimension runs fastest (tiles sharing a K-long operand slab are neighbours)
};

struct GemmGroupP {
  int n;
  int start[VCT_GEMM_GROUP_MAX];   // first workgroup of problem i (multiple of 8)
  GemmP p[VCT_GEMM_GROUP_MAX];
};

}  // namespace vct
