// Batch-1 greedy-decode step in THREE launches per decoder layer (+ generator + arg-max) on gfx950, bf16 weights.
//
// replaces (per generated token): CapDecoder.decode_word's module calls for one caption -- per nn.TransformerDecoderLayer the
// self-attention in-projection / SDPA / out-projection, norm1, the cross-attention q-projection / SDPA / out-projection, norm2,
// linear1 / GELU / linear2, norm3, then decoder.norm and the generator (reference model/CapDecoder.py:62-79, torch
// nn/modules/transformer.py:1143-1199): 14 launches on the matrix-vector kernels of vct_decode.hip (6 per layer), 8 here.
//
// At batch 1 every stage is a matrix-VECTOR product that needs the WHOLE output vector of the stage before it, so a launch per
// stage costs a dependent kernel boundary per stage (~5-7 us each for < 1 us of streaming).  Two of every three boundaries go
// away by splitting the reduction of the SECOND product of a block over the workgroups that own the FIRST one:
//   self block    workgroup h (one per head): q_h | k_h | v_h = W_in[rows of head h] x (+ bias) -> cache slot; attention of head h
//                 over the cached keys; PARTIAL out-projection  a_h[n] = sum_{c in head h} W_o[n, c] o_h[c]   (n = 0..d)
//   cross block   workgroup h: q_h = W_q[rows of head h] x1; attention over the memory's cached K/V; partial out-projection
//   feed-forward  workgroup c (one per 64 hidden units): g_c = act(W1[rows c] x2 + b1); partial  f_c[n] = sum_{j in c} W2[n, j] g_c[j]
// The H (or ff/64) partial vectors [d] are NOT reduced in the producing launch (that would be a grid-wide fan-in): every
// workgroup of the CONSUMING launch sums them in its prologue, in index order (deterministic), together with the residual, the
// bias and the LayerNorm(s) -- 16-64 KB of L2 reads per workgroup, a few dozen workgroups.  Activations between launches are fp32
// vectors; q / k / v are rounded to bf16 where they enter the cache (what the later steps and the batched kernels read).
#include "vct_common.h"

namespace vct {

constexpr int DB_THREADS = 512, DB_WAVES = 8, DB_DMAX = 512, DB_HD = 64, DB_LMAX = 64, DB_PMAX = 32;
constexpr int DB_EPT = DB_DMAX / DB_THREADS;      // vector elements per thread: 1 (d = 512: one thread per element, one 16-byte
                                                  // weight chunk per lane and row; wider models stay on vct_decode_gemv)

constexpr int NCH = 1;                              // 512-element chunks per weight row (d = 512)

struct DbVec {                    // the input vector x[d] of a launch (fp32), built by every workgroup
  const int64_t* id; const float* table; const float* pos_row;      // id != NULL: table[id[0]] + pos_row
  const float* res; const float* bias;                              // else: res + bias + sum_c part[c]  (each may be NULL)
  const float* part; int n_part;
  const float* g1; const float* b1; const float* g2; const float* b2;  // then up to two LayerNorms
  float* x_out;                                                     // workgroup 0 publishes x (the next block's residual)
};

struct DbP {
  int d, nch;                     // model width, 512-element chunks per row
  DbVec v;
  // attention blocks
  const bf16_t* w_in; long ld_in; const float* b_in; int n_proj;    // rows per head to project: 3 (q | k | v at row offsets 0, d, 2d) or 1
  bf16_t* slot;                                                    // self: q | k | v of the consumed token [3d] (cache row t - 1)
  const bf16_t* kc; const bf16_t* vc; long kv_ld; int Lk;          // cached key / value rows; Lk keys (self: the last one is the fresh one)
  const bf16_t* w_o; long ld_o;
  // feed-forward
  const bf16_t* w1; long ld1; const float* b1f; const bf16_t* w2; long ld2; int act;
  float* part_out;                                                 // [gridDim.x][d]
  // generator
  const bf16_t* wg; long ldg; const float* bg; float* logits; int V;
  // ... + greedy selection by the last workgroup to finish (sel_ws != nullptr)
  float* sel_ws;                  // [2 * gridDim.x] (value, index) per workgroup, then ONE int ticket counter (left zero)
  int64_t* tok_out; long long end_id; uint8_t* ended; int32_t* ended_count; unsigned long long* all_ended_at; int t;
};

__device__ __forceinline__ void unpack8(const uint4 v, float (&o)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; j++) { o[2 * j] = __uint_as_float(w[j] << 16); o[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
}

__device__ __forceinline__ float block_sum8(float v, float* red) {      // 8 waves; all threads get the sum; two barriers
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

// Every launch is a chain of phases that each need the previous one's result, and at batch 1 there is ONE workgroup per head /
// hidden-unit group: a phase that starts with its own global loads pays a full memory round trip (1-2 us) with nothing to hide
// it (a first version that loaded phase by phase ran 152 us per token against 97 for the launch-per-stage kernels).  So every
// kernel ISSUES ALL ITS GLOBAL LOADS FIRST -- vector parts, weight rows of the first product, cached keys and values, the weight
// segments of the second product -- and then works from registers and LDS only.

// The vector's inputs as 16-byte loads: thread = (group g of 128 threads, four consecutive elements); group g takes the partial
// vectors c = g, g + 4, ... (<= 8 each), group 0 also the residual and the bias.  (One 4-byte load per thread and partial -- 37
// vector-memory instructions per wave for 64 KB -- cost every launch ~2 us at the CU's memory-instruction issue port.)
constexpr int DB_PG = DB_PMAX / 4;                // partial vectors per thread group
struct VecRaw { float4 base, bias; float4 part[DB_PG]; float g1, b1, g2, b2; };

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ void vec_issue(const DbP& p, VecRaw& r) {
  const DbVec& s = p.v;
  const int tid = threadIdx.x, d = p.d;
  const int g = tid >> 7, e4 = (tid & 127) * 4;
  // unconditional loads (a predicated load in an unrolled loop costs a branch + vmcnt(0) each): absent inputs read a valid
  // stand-in address and are masked afterwards
  const float* safe = s.id != nullptr ? s.pos_row : s.res;
  if (s.id != nullptr) {
    r.base = ld4(s.table + s.id[0] * (long)d + e4);
    r.bias = ld4(s.pos_row + e4);
#pragma unroll
    for (int k = 0; k < DB_PG; k++) r.part[k] = float4{0.0f, 0.0f, 0.0f, 0.0f};
  } else {
    const float* bp = s.bias != nullptr ? s.bias : safe;
    const float* pp = s.n_part > 0 ? s.part : safe;
    const int np1 = max(s.n_part - 1, 0);
    r.base = ld4(s.res + e4);
    r.bias = ld4(bp + e4);
    if (s.bias == nullptr) r.bias = float4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < DB_PG; k++) {
      const int c = g + 4 * k;
      const float4 v = ld4(pp + (long)min(c, np1) * d + e4);
      r.part[k] = c < s.n_part ? v : float4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  }
  const int idx = min(tid, d - 1);
  const float* g1p = s.g1 != nullptr ? s.g1 : safe; const float* b1p = s.b1 != nullptr ? s.b1 : safe;
  const float* g2p = s.g2 != nullptr ? s.g2 : safe; const float* b2p = s.b2 != nullptr ? s.b2 : safe;
  r.g1 = g1p[idx]; r.b1 = b1p[idx]; r.g2 = g2p[idx]; r.b2 = b2p[idx];
}

// x[d] -> LDS (every workgroup), optionally published by workgroup 0.  psum: LDS [4][DB_DMAX]
__device__ __forceinline__ void vec_finish(const DbP& p, VecRaw& r, float* xs, float* psum, float* red) {
  const DbVec& s = p.v;
  const int tid = threadIdx.x, d = p.d;
  const int g = tid >> 7, e4 = (tid & 127) * 4;
  float4 a = g == 0 ? float4{r.base.x + r.bias.x, r.base.y + r.bias.y, r.base.z + r.bias.z, r.base.w + r.bias.w} : float4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < DB_PG; k++) { a.x += r.part[k].x; a.y += r.part[k].y; a.z += r.part[k].z; a.w += r.part[k].w; }   // fixed order
  *reinterpret_cast<float4*>(psum + g * DB_DMAX + e4) = a;
  __syncthreads();
  float v = ((psum[tid] + psum[DB_DMAX + tid]) + psum[2 * DB_DMAX + tid]) + psum[3 * DB_DMAX + tid];
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    if ((pass == 0 ? s.g1 : s.g2) == nullptr) break;
    const float mean = block_sum8(v, red) / (float)d;
    const float c = v - mean;
    const float rstd = 1.0f / sqrtf(block_sum8(c * c, red) / (float)d + 1e-5f);
    v = c * rstd * (pass == 0 ? r.g1 : r.g2) + (pass == 0 ? r.b1 : r.b2);
  }
  xs[tid] = v;
  if (s.x_out != nullptr && blockIdx.x == 0) s.x_out[tid] = v;
  __syncthreads();
}

template <int NR, int NCH>
__device__ __forceinline__ void rows_issue(const bf16_t* __restrict__ W, long ld, const int (&rows)[NR], uint4 (&w)[NR][NCH], int lane) {
#pragma unroll
  for (int r = 0; r < NR; r++)
#pragma unroll
    for (int c = 0; c < NCH; c++) w[r][c] = *reinterpret_cast<const uint4*>(W + (long)max(rows[r], 0) * ld + c * 512 + lane * 8);
}
// out[r] = W[rows[r]] . x, valid in every lane
template <int NR, int NCH>
__device__ __forceinline__ void rows_dot(const uint4 (&w)[NR][NCH], const float (&xr)[NCH][8], float (&out)[NR]) {
#pragma unroll
  for (int r = 0; r < NR; r++) {
    float a = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      float f[8];
      unpack8(w[r][c], f);
#pragma unroll
      for (int u = 0; u < 8; u++) a += f[u] * xr[c][u];
    }
    out[r] = a;
  }
#pragma unroll
  for (int r = 0; r < NR; r++) out[r] = wave_sum(out[r]);
}

// second product of a block, split over the workgroups: part_out[blockIdx.x][n] = sum_{j < 64} W[n, col0 + j] g[j], n = 0..d, read
// from the TRANSPOSED weight WT[col0 + j][n] (the caller keeps transposed copies of out_proj / linear2 for decoding): a
// workgroup's slice is 64 CONTIGUOUS 1-KB rows instead of 512 row segments of 128 bytes 1-4 KB apart (one DRAM page and one TLB
// entry each).  thread = (group cg of 8 input columns, 8 consecutive outputs); 8 loads, all issued up front.
constexpr int DB_OPASS = 8;
__device__ __forceinline__ void out_issue(const bf16_t* __restrict__ WT, long ld, int col0, int d, uint4 (&w)[DB_OPASS]) {
  const int cg = threadIdx.x >> 6, n8 = (threadIdx.x & 63) * 8;
#pragma unroll
  for (int q = 0; q < DB_OPASS; q++) w[q] = *reinterpret_cast<const uint4*>(WT + (long)(col0 + cg * 8 + q) * ld + n8);
}
// osum: LDS [8][DB_DMAX]
__device__ __forceinline__ void out_finish(const uint4 (&w)[DB_OPASS], const float* g /* LDS [64] */, float* osum, float* dst, int d) {
  const int tid = threadIdx.x, cg = tid >> 6, n8 = (tid & 63) * 8;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; u++) acc[u] = 0.0f;
#pragma unroll
  for (int q = 0; q < DB_OPASS; q++) {
    float f[8];
    unpack8(w[q], f);
    const float gq = g[cg * 8 + q];
#pragma unroll
    for (int u = 0; u < 8; u++) acc[u] += f[u] * gq;                     // column order inside the group
  }
  *reinterpret_cast<float4*>(osum + cg * DB_DMAX + n8) = float4{acc[0], acc[1], acc[2], acc[3]};
  *reinterpret_cast<float4*>(osum + cg * DB_DMAX + n8 + 4) = float4{acc[4], acc[5], acc[6], acc[7]};
  __syncthreads();
  float a = 0.0f;
#pragma unroll
  for (int c = 0; c < 8; c++) a += osum[c * DB_DMAX + tid];              // group order: deterministic
  if (tid < d) dst[tid] = a;
}

// ---- attention block: self (n_proj = 3) or cross (n_proj = 1); one workgroup per head -------------------------------------------
template <int NPROJ>
__global__ __launch_bounds__(DB_THREADS) void decode_attn_block_kernel(const DbP p) {
  constexpr int NR = NPROJ * DB_HD / DB_WAVES;                // projection rows per wave: 24 (self) / 8 (cross)
  __shared__ float xs[DB_DMAX];
  __shared__ __attribute__((aligned(16))) float psum[8 * DB_DMAX];      // vector partial sums [4][d], later the second product's [8][d]
  __shared__ float red[DB_WAVES];
  __shared__ float hv[3 * DB_HD];          // q | k | v of this head (bf16-rounded)
  __shared__ float sc[DB_LMAX];
  __shared__ float pj[DB_LMAX];
  __shared__ float ow[DB_WAVES][DB_HD];
  __shared__ float oh[DB_HD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, d = p.d;
  // ---- every global load of the kernel, up front ----
  VecRaw vr;
  vec_issue(p, vr);
  int rows[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int rr = wave * NR + r;                              // row within the head's projection rows
    rows[r] = (rr / DB_HD) * d + h * DB_HD + (rr % DB_HD);
  }
  uint4 w[NR][NCH];
  rows_issue<NR, NCH>(p.w_in, p.ld_in, rows, w, lane);
  float bias_r = 0.0f;                                         // lane r < NR finishes row r of this wave
  if (lane < NR) { int row = 0;
#pragma unroll
    for (int r = 0; r < NR; r++) row = lane == r ? rows[r] : row;
    bias_r = p.b_in[row]; }
  const int kj = tid >> 3, kc8 = tid & 7;                      // thread = (key, 8-element chunk of the head slice)
  const int ncached = NPROJ == 3 ? p.Lk - 1 : p.Lk;            // keys whose rows are in the cache (self: the newest is in LDS)
  const long koff = (long)min(kj, max(ncached - 1, 0)) * p.kv_ld + h * DB_HD + kc8 * 8;
  const uint4 kraw = *reinterpret_cast<const uint4*>(p.kc + koff);
  const uint4 vraw = *reinterpret_cast<const uint4*>(p.vc + koff);
  uint4 wo[DB_OPASS];
  out_issue(p.w_o, p.ld_o, h * DB_HD, d, wo);

  // ---- input vector, projection rows of this head ----
  vec_finish(p, vr, xs, psum, red);
  float xr[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int u = 0; u < 8; u++) xr[c][u] = xs[c * 512 + lane * 8 + u];
  {
    float out[NR];
    rows_dot<NR, NCH>(w, xr, out);
    if (lane < NR) {
      float mine = 0.0f; int row = 0;
#pragma unroll
      for (int r = 0; r < NR; r++) { mine = lane == r ? out[r] : mine; row = lane == r ? rows[r] : row; }
      const bf16_t q = f2bf(mine + bias_r);
      const int which = row / d, i = row - which * d - h * DB_HD;
      hv[which * DB_HD + i] = bf2f(q);
      if (NPROJ == 3) p.slot[row] = q;
    }
  }
  __syncthreads();
  // ---- scores: 8 threads per key ----
  {
    float kf[8];
    unpack8(kraw, kf);
    const bool fresh = NPROJ == 3 && kj == p.Lk - 1;
    float a = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; u++) a += (fresh ? hv[DB_HD + kc8 * 8 + u] : kf[u]) * hv[kc8 * 8 + u];
    a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
    if (kc8 == 0) sc[kj] = kj < p.Lk ? a * 0.125f : -INFINITY;
  }
  __syncthreads();
  if (wave == 0) {
    const float s0 = sc[lane];
    const float m = wave_max(s0);
    const float e = lane < p.Lk ? expf(s0 - m) : 0.0f;
    pj[lane] = e / wave_sum(e);
  }
  __syncthreads();
  // ---- o_h = sum_j p_j V_j: thread (j, chunk) scales its 8 values, the keys fold by shuffles (8 per wave) and through LDS ----
  {
    float vf[8];
    unpack8(vraw, vf);
    const bool fresh = NPROJ == 3 && kj == p.Lk - 1;
    const float pw = kj < p.Lk ? pj[kj] : 0.0f;
    float o[8];
#pragma unroll
    for (int u = 0; u < 8; u++) o[u] = pw * (fresh ? hv[2 * DB_HD + kc8 * 8 + u] : vf[u]);
#pragma unroll
    for (int off = 8; off < 64; off <<= 1)
#pragma unroll
      for (int u = 0; u < 8; u++) o[u] += __shfl_xor(o[u], off);
    if (lane < 8) {
#pragma unroll
      for (int u = 0; u < 8; u++) ow[wave][lane * 8 + u] = o[u];
    }
  }
  __syncthreads();
  if (tid < DB_HD) {
    float a = 0.0f;
#pragma unroll
    for (int w2 = 0; w2 < DB_WAVES; w2++) a += ow[w2][tid];                  // wave order = key order: deterministic
    oh[tid] = a;
  }
  __syncthreads();
  out_finish(wo, oh, psum, p.part_out + (long)h * d, d);
}

// ---- feed-forward block: one workgroup per 64 hidden units -------------------------------------------------------------------------
__global__ __launch_bounds__(DB_THREADS) void decode_ffn_block_kernel(const DbP p) {
  constexpr int NR = DB_HD / DB_WAVES;                        // 8 hidden units per wave
  __shared__ float xs[DB_DMAX];
  __shared__ __attribute__((aligned(16))) float psum[8 * DB_DMAX];      // vector partial sums [4][d], later the second product's [8][d]
  __shared__ float red[DB_WAVES];
  __shared__ float gc[DB_HD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * DB_HD, d = p.d;
  VecRaw vr;
  vec_issue(p, vr);
  int rows[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) rows[r] = c0 + wave * NR + r;
  uint4 w[NR][NCH];
  rows_issue<NR, NCH>(p.w1, p.ld1, rows, w, lane);
  const float bias_r = p.b1f[c0 + wave * NR + min(lane, NR - 1)];
  uint4 wo[DB_OPASS];
  out_issue(p.w2, p.ld2, c0, d, wo);
  vec_finish(p, vr, xs, psum, red);
  float xr[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int u = 0; u < 8; u++) xr[c][u] = xs[c * 512 + lane * 8 + u];
  float out[NR];
  rows_dot<NR, NCH>(w, xr, out);
  if (lane < NR) {
    float mine = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; r++) mine = lane == r ? out[r] : mine;
    gc[wave * NR + lane] = act_f(p.act, mine + bias_r);
  }
  __syncthreads();
  out_finish(wo, gc, psum, p.part_out + (long)blockIdx.x * d, d);
}

// ---- generator: logits[n] = W_g[n] . y + b_g[n]; 128 rows per workgroup ------------------------------------------------------------
constexpr int DB_GEN_ROWS = 128;
__global__ __launch_bounds__(DB_THREADS) void decode_gen_kernel(const DbP p) {
  constexpr int NR = DB_GEN_ROWS / DB_WAVES;                  // 16 rows per wave
  __shared__ float xs[DB_DMAX];
  __shared__ __attribute__((aligned(16))) float psum[8 * DB_DMAX];      // vector partial sums [4][d], later the second product's [8][d]
  __shared__ float red[DB_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base = blockIdx.x * DB_GEN_ROWS + wave * NR;
  VecRaw vr;
  vec_issue(p, vr);
  int rows[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) rows[r] = min(base + r, p.V - 1);
  uint4 w[NR][NCH];
  rows_issue<NR, NCH>(p.wg, p.ldg, rows, w, lane);
  const float bias_r = p.bg[min(base + min(lane, NR - 1), p.V - 1)];
  vec_finish(p, vr, xs, psum, red);
  float xr[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int u = 0; u < 8; u++) xr[c][u] = xs[c * 512 + lane * 8 + u];
  float out[NR];
  rows_dot<NR, NCH>(w, xr, out);
  float mine = 0.0f;
#pragma unroll
  for (int r = 0; r < NR; r++) mine = lane == r ? out[r] : mine;
  mine += bias_r;
  const int n = base + lane;
  const bool have = lane < NR && n < p.V;
  if (have) p.logits[n] = mine;
  if (p.sel_ws == nullptr) return;
  // ---- greedy selection (MMT4Caption.py:166-171; torch.max: the FIRST maximal index): workgroup arg-max -> (value, index) pair
  // written through the L2 -> ticket; the last workgroup reduces the pairs.  The separate arg-max launch over the 122 KB of
  // logits cost 6.7 us per token.  Same comparison as argmax_rows_kernel: larger value, then smaller index.
  float bv = have ? mine : -INFINITY;
  int bi = have ? n : 0x7fffffff;
  auto better = [](float v, int i, float bv_, int bi_) { return v > bv_ || (v == bv_ && i < bi_); };
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o);
    const int oi = __shfl_xor(bi, o);
    if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  __shared__ float s_v[DB_WAVES];
  __shared__ int s_i[DB_WAVES];
  __shared__ int s_last;
  if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
  __syncthreads();
  int* ticket = reinterpret_cast<int*>(p.sel_ws + 2 * gridDim.x);
  if (tid == 0) {
#pragma unroll
    for (int w = 1; w < DB_WAVES; w++) if (better(s_v[w], s_i[w], bv, bi)) { bv = s_v[w]; bi = s_i[w]; }
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const f32x2 pr = {bv, __int_as_float(bi)};
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.sel_ws + 2 * blockIdx.x), __builtin_bit_cast(unsigned long long, pr),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the pair is an sc1 (write-through) store and the last workgroup reads it with sc1 loads: what orders it before the ticket is
    // the store's acknowledgement, i.e. vmcnt(0) -- a workgroup-scope fence emits no wait on gfx950 (MI355X_MICROARCH: handoff-flag)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  bv = -INFINITY; bi = 0x7fffffff;
  for (int g = tid; g < (int)gridDim.x; g += DB_THREADS) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p.sel_ws + 2 * g), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    const f32x2 pr = __builtin_bit_cast(f32x2, u);
    const int gi = __float_as_int(pr[1]);
    if (better(pr[0], gi, bv, bi)) { bv = pr[0]; bi = gi; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o);
    const int oi = __shfl_xor(bi, o);
    if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  __syncthreads();                                       // s_v / s_i of the first reduction have been read
  if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int w = 1; w < DB_WAVES; w++) if (better(s_v[w], s_i[w], bv, bi)) { bv = s_v[w]; bi = s_i[w]; }
    const int tok = (bi == 0x7fffffff) ? 0 : bi;
    p.tok_out[0] = tok;
    if (p.ended != nullptr && tok == p.end_id && !p.ended[0]) {      // one caption: it completes the set by itself
      p.ended[0] = 1;
      if (atomicAdd(p.ended_count, 1) + 1 == 1) atomicMin(p.all_ended_at, (unsigned long long)p.t);
    }
    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next token
  }
}

}  // namespace vct
using namespace vct;

extern "C" int vct_decode_block_supported(int dtype, int d, int H, int ff, int Lk) {
  // ff / 64 feed-forward workgroups each publish one partial vector: the consumer's prologue sums at most DB_PMAX of them
  return (dtype == VCT_BF16 && d == DB_DMAX && H * DB_HD == d && H <= DB_PMAX && ff >= DB_HD && ff % DB_HD == 0 && ff <= DB_PMAX * DB_HD &&
          Lk >= 1 && Lk <= DB_LMAX) ? 1 : 0;
}

extern "C" int vct_decode_block(const vct_decode_block_desc* q, void* stream) {
  if (q == nullptr) return VCT_E_ARG;
  if (q->kind < 0 || q->kind > 3) return VCT_E_ARG;
  const int Lk = (q->kind == 0 || q->kind == 1) ? q->Lk : 1;
  if (!vct_decode_block_supported(VCT_BF16, q->d, q->d / DB_HD, q->kind == 2 ? q->ff : DB_HD, Lk)) return VCT_E_SHAPE;
  DbP p;
  p.d = q->d; p.nch = q->d / 512;
  p.v.id = q->id; p.v.table = q->table; p.v.pos_row = q->pos_row;
  p.v.res = q->res; p.v.bias = q->res_bias; p.v.part = q->part; p.v.n_part = q->n_part;
  p.v.g1 = q->g1; p.v.b1 = q->b1; p.v.g2 = q->g2; p.v.b2 = q->b2; p.v.x_out = q->x_out;
  if (p.v.id != nullptr && (!p.v.table || !p.v.pos_row)) return VCT_E_ARG;
  if (p.v.id == nullptr && !p.v.res && !(p.v.part && p.v.n_part > 0)) return VCT_E_ARG;
  if ((p.v.g1 == nullptr) != (p.v.b1 == nullptr) || (p.v.g2 == nullptr) != (p.v.b2 == nullptr) || (p.v.g2 && !p.v.g1)) return VCT_E_ARG;
  if (p.v.part == nullptr) p.v.n_part = 0;
  if (p.v.n_part > DB_PMAX) return VCT_E_SHAPE;
  p.w_in = reinterpret_cast<const bf16_t*>(q->w_a); p.ld_in = q->ld_a; p.b_in = q->b_a;
  p.n_proj = q->kind == 0 ? 3 : 1;
  p.slot = reinterpret_cast<bf16_t*>(q->slot);
  p.kc = reinterpret_cast<const bf16_t*>(q->kc); p.vc = reinterpret_cast<const bf16_t*>(q->vc); p.kv_ld = q->kv_ld; p.Lk = q->Lk;
  p.w_o = reinterpret_cast<const bf16_t*>(q->w_b); p.ld_o = q->ld_b;
  p.w1 = p.w_in; p.ld1 = p.ld_in; p.b1f = p.b_in; p.w2 = p.w_o; p.ld2 = p.ld_o; p.act = q->act;
  p.part_out = q->part_out;
  p.wg = p.w_in; p.ldg = p.ld_in; p.bg = p.b_in; p.logits = q->part_out; p.V = q->V;
  p.sel_ws = q->kind == 3 ? q->sel_ws : nullptr;
  p.tok_out = q->tok_out; p.end_id = q->end_id; p.ended = q->ended; p.ended_count = q->ended_count;
  p.all_ended_at = reinterpret_cast<unsigned long long*>(q->all_ended_at); p.t = q->t;
  if (p.sel_ws != nullptr && (!p.tok_out || !p.ended || !p.ended_count || !p.all_ended_at || q->t < 0)) return VCT_E_ARG;
  if (!q->w_a || !q->b_a || !q->part_out || (q->ld_a % 8) || ((uintptr_t)q->w_a & 15)) return VCT_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (q->kind == 0 || q->kind == 1) {
    if (!q->w_b || (q->ld_b % 8) || ((uintptr_t)q->w_b & 15) || !q->kc || !q->vc || (q->kv_ld % 8) || (((uintptr_t)q->kc | (uintptr_t)q->vc) & 15))
      return VCT_E_ARG;
    if (q->kind == 0 && !q->slot) return VCT_E_ARG;
    const dim3 grid(q->d / DB_HD);
    if (q->kind == 0) vct::launch(decode_attn_block_kernel<3>, grid, dim3(DB_THREADS), 0, st, p);
    else vct::launch(decode_attn_block_kernel<1>, grid, dim3(DB_THREADS), 0, st, p);
  } else if (q->kind == 2) {
    if (!q->w_b || (q->ld_b % 8) || ((uintptr_t)q->w_b & 15)) return VCT_E_ARG;
    const dim3 grid(q->ff / DB_HD);
    vct::launch(decode_ffn_block_kernel, grid, dim3(DB_THREADS), 0, st, p);
  } else {
    if (q->V < 1) return VCT_E_SHAPE;
    const dim3 grid((q->V + DB_GEN_ROWS - 1) / DB_GEN_ROWS);
    vct::launch(decode_gen_kernel, grid, dim3(DB_THREADS), 0, st, p);
  }
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}
