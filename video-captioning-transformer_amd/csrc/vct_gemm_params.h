// Kernel-side parameter block shared by the GEMM translation units.
#pragma once
#include <stdint.h>
#include "../../include/vct_hip.h"

namespace vct {

// optimizer epilogue of the weight-gradient form (include/vct_hip.h, vct_gemm_adam); param == nullptr: off
struct AdamEpiP {
  float* param; float* m; float* v;
  uint16_t* shadow; long ld_shadow;
  uint16_t* pk_stream; int pk_K, pk_mode; int pk_row0; int store_grad;
  // first chunk of the four 512-row blocks / 512-column slices as 16-bit fields (0xffff: not packed), selected by a SHIFT: an array
  // member or four scalars + selects become an indexed load, and a run-time index into the by-value group table (reached through the
  // problem index) puts the whole 2.5 KB table into scratch, per lane
  unsigned long long pk_chunks;
  const float* hyper; const int32_t* step;
};

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
  AdamEpiP adam;        // weight-gradient form only
};

struct GemmGroupP {
  int n;
  int total;                       // > 0: group-level XCD map over `total` tiles, start[] = tile prefix sums; 0: per-problem map
  int start[VCT_GEMM_GROUP_MAX];   // first workgroup of problem i (a multiple of 8 in the per-problem map)
  GemmP p[VCT_GEMM_GROUP_MAX];
};

}  // namespace vct
