// Multi-head attention core for short sequences (Lq, Lk <= 64, head_dim <= 128) on gfx950.
// One wave per (batch, head); Q/K/V (and dO) head slices are staged once into LDS with 16-byte
// loads, every product is an MFMA tile (bf16 16x16x32 or exact-fp32 16x16x4), the softmax row
// reductions are wave shuffles, masks (causal / key padding) and dropout are applied in registers
// and the score matrix never touches HBM.
//
// Forward computes the scores TRANSPOSED (S^T = K Q^T): the C fragment of a 16x16 MFMA then holds,
// per lane, one query column and four keys -- exactly the A-operand shape of the P V product, so P
// feeds the second MFMA straight from registers.  V (and K, Q, dO in the backward) are consumed as
// B operands whose reduction index runs along LDS rows: bf16 uses the gfx950 LDS transpose read.
//
// Backward (one launch, same wave): phase A (per query tile) rebuilds S^T, P, dP^T = V dO^T,
// D = rowsum(dP*P), dS and dQ = dS K; phase B (per key tile) rebuilds S, P, dP in the un-transposed
// form whose C fragment is the A operand of the reductions over queries: dV = P^T dO, dK = dS^T Q.
#include "vct_common.h"

namespace vct {

struct AttnP {
  int B, H, Lq, Lk, hd, causal;
  const void* q; long ldq;
  const void* k; long ldk;
  const void* v; long ldv;
  void* o; long ldo;
  const uint8_t* key_pad; int key_pad_shift;
  const int64_t* key_ids; long key_ids_bs; long pad_id;
  const uint32_t* seed; uint32_t site; float p_drop;
  const void* d_o; long ld_do;
  void* dq; long ld_dq;
  void* dk; long ld_dk;
  void* dv; long ld_dv;
  long q_bs, k_bs, v_bs, o_bs;
};

template <typename T, int DT> struct AttnCfg {
  static constexpr bool BF = sizeof(T) == 2;
  static constexpr int HDP = DT * 16;                           // padded head dim (C/B tile columns)
  static constexpr int HDK = BF ? ((HDP + 31) / 32) * 32 : HDP;  // columns allocated (k-steps of 32 for bf16)
  static constexpr int STR = HDK + (BF ? 8 : 4);                // LDS row stride in elements
  static constexpr int VEC = BF ? 8 : 4;
  static constexpr int KS = BF ? HDK / 32 : 0;                  // bf16 k-steps over the head dim
};

// cooperative (one wave) copy of rows [0,L) x cols [0,hd) of NJ head slices into LDS, each zero padded to
// rows_alloc x HDK.  The 16-byte loads of ALL slices are issued back to back before the first LDS write (one
// HBM/L2 round trip per wave instead of one per slice -- the wave has nothing else to overlap it with), and
// UNCONDITIONALLY (row / column clamped into the slice, padding zeroed afterwards): a predicated load costs a
// branch + vmcnt(0) per vector.
struct alignas(16) AV16 { uint32_t w[4]; };
template <typename T> struct StageJob { T* lds; const T* g; long ld; int L; int rows_alloc; };

template <typename T, int DT>
__device__ __forceinline__ void stage_load4(AV16 (&val)[4], const StageJob<T>& j, int base, int hd, int lane) {
  using C = AttnCfg<T, DT>;
  constexpr int VPR = C::HDK / C::VEC;
  const int total = j.rows_alloc * VPR;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int idx = min(base + u * 64 + lane, total - 1);
    const int r = min(idx / VPR, j.L - 1), c = min((idx % VPR) * C::VEC, hd - C::VEC);
    val[u] = *reinterpret_cast<const AV16*>(j.g + (long)r * j.ld + c);
  }
}
template <typename T, int DT>
__device__ __forceinline__ void stage_commit4(AV16 (&val)[4], const StageJob<T>& j, int base, int hd, int lane) {
  using C = AttnCfg<T, DT>;
  constexpr int VPR = C::HDK / C::VEC;
  const int total = j.rows_alloc * VPR;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int idx = base + u * 64 + lane;
    if (idx < total) {
      const int r = idx / VPR, c = (idx % VPR) * C::VEC;
      if (r >= j.L || c >= hd) val[u].w[0] = val[u].w[1] = val[u].w[2] = val[u].w[3] = 0u;
      *reinterpret_cast<AV16*>(j.lds + r * C::STR + c) = val[u];
    }
  }
}
template <typename T, int DT, int NJ>
__device__ __forceinline__ void stage_multi(const StageJob<T> (&jobs)[NJ], int hd, int lane) {
  using C = AttnCfg<T, DT>;
  constexpr int VPR = C::HDK / C::VEC;
  AV16 val[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; j++) stage_load4<T, DT>(val[j], jobs[j], 0, hd, lane);
#pragma unroll
  for (int j = 0; j < NJ; j++) stage_commit4<T, DT>(val[j], jobs[j], 0, hd, lane);
#pragma unroll
  for (int j = 0; j < NJ; j++) {       // slices with more than 256 vectors (long sequences / wide heads)
    const int total = jobs[j].rows_alloc * VPR;
    for (int base = 256; base < total; base += 256) {
      stage_load4<T, DT>(val[0], jobs[j], base, hd, lane);
      stage_commit4<T, DT>(val[0], jobs[j], base, hd, lane);
    }
  }
}

// bf16 fragment helpers ------------------------------------------------------------------------
// operand whose reduction index runs along the head dim: rows = `row_base + (lane&15)`, 8 values at
// columns ks*32 + (lane>>4)*8
template <int STR> __device__ __forceinline__ bf16x8 frag_rowk(const bf16_t* lds, int row_base, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(lds + (row_base + (lane & 15)) * STR + ks * 32 + (lane >> 4) * 8);
}
// operand whose reduction index runs along LDS ROWS (pair of 16-row tiles r0, r1): k-slot (g, j<4)
// <-> row r0 + g*4 + j, (g, j>=4) <-> row r1 + g*4 + j - 4; column = col_base + (lane & 15)
template <int STR> __device__ __forceinline__ bf16x8 frag_colk(const bf16_t* lds, int r0, int r1, int col_base, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const s16x4 lo = lds_tr16(lds + (r0 + g * 4 + (i >> 2)) * STR + col_base + (i & 3) * 4);
  const s16x4 hi = lds_tr16(lds + (r1 + g * 4 + (i >> 2)) * STR + col_base + (i & 3) * 4);
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack_p(const f32x4& a, const f32x4& b) {
  s16x8 v;
#pragma unroll
  for (int j = 0; j < 4; j++) { v[j] = (short)f2bf(a[j]); v[4 + j] = (short)f2bf(b[j]); }
  return __builtin_bit_cast(bf16x8, v);
}

// Output tiles are produced TRANSPOSED (swap the MFMA operands: X^T = B^T A^T, and the per-lane register pattern of
// an A fragment equals that of a B fragment): the C fragment then holds, per lane, ONE sequence row (lane & 15) and
// FOUR consecutive head-dim columns ((lane >> 4) * 4 + r) -- one 8/16-byte store instead of four scattered 2/4-byte ones.
template <typename T> struct alignas(4 * sizeof(T)) Out4 { T e[4]; };
template <typename T>
__device__ __forceinline__ void store_row4(T* base, long ld, int row, int col, const f32x4& v, int nrows, int hd) {
  if (row < nrows && col < hd) {
    Out4<T> o;
#pragma unroll
    for (int r = 0; r < 4; r++) o.e[r] = from_f<T>(v[r]);
    *reinterpret_cast<Out4<T>*>(base + (long)row * ld + col) = o;
  }
}

// reduce over the 4 lane groups that share (lane & 15): lanes l, l^16, l^32, l^48
__device__ __forceinline__ float red4_sum(float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }
__device__ __forceinline__ float red4_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 32)); return v; }
// reduce over the 16 lanes of a group (same lane >> 4)
__device__ __forceinline__ float red16_sum(float v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}

// padmask: bit k set = key k is padded (built once per wave with a ballot over kp_row[lane]; a per-element
// byte load here would be a chain of dependent global loads inside the unrolled score loops)
__device__ __forceinline__ bool key_masked(const AttnP& p, unsigned long long padmask, int qq, int kk) {
  return kk >= p.Lk || (p.causal && kk > qq) || ((padmask >> kk) & 1ull);
}
__device__ __forceinline__ unsigned long long load_padmask(const AttnP& p, int b, int lane) {
  if (p.key_ids != nullptr) {
    const long id = p.key_ids[(long)b * p.key_ids_bs + min(lane, p.Lk - 1)];
    return __ballot(lane < p.Lk && id == p.pad_id);
  }
  if (p.key_pad == nullptr) return 0ull;
  const int w = p.Lk - p.key_pad_shift;                 // mask row width; keys below the shift are never padded
  const int j = min(max(lane - p.key_pad_shift, 0), w - 1);
  const uint8_t v = p.key_pad[(long)b * w + j];
  return __ballot(lane >= p.key_pad_shift && lane < p.Lk && v != 0);
}

// S^T tiles for query tile qt:  st[t][r] = scale * Q[q = qt*16 + i] . K[key = t*16 + g*4 + r]   (masked -> -inf)
template <typename T, int DT>
__device__ __forceinline__ void scores_T(f32x4 (&st)[4], const T* Ks, const T* Qs, int qt, int LKT, int hd4, float scale,
                                         const AttnP& p, unsigned long long kp_row, int lane) {
  using C = AttnCfg<T, DT>;
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    st[t] = f32x4{0, 0, 0, 0};
    if (t < LKT) {
      if constexpr (C::BF) {
#pragma unroll
        for (int ks = 0; ks < C::KS; ks++)
          st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rowk<C::STR>(Ks, t * 16, ks, lane),
                                                          frag_rowk<C::STR>(Qs, qt * 16, ks, lane), st[t], 0, 0, 0);
      } else {
        for (int k4 = 0; k4 < hd4; k4++)
          st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(t * 16 + i) * C::STR + k4 * 4 + g],
                                                       Qs[(qt * 16 + i) * C::STR + k4 * 4 + g], st[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int kk = t * 16 + g * 4 + r, qq = qt * 16 + i;
      st[t][r] = (t < LKT && !key_masked(p, kp_row, qq, kk)) ? st[t][r] * scale : -INFINITY;
    }
  }
}

template <typename T, int DT>
__global__ __launch_bounds__(64) void attn_fwd_kernel(const AttnP p) {
  using C = AttnCfg<T, DT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int LQT = (p.Lq + 15) / 16, LKT = (p.Lk + 15) / 16;
  const int RQ = LQT * 16, RK = ((LKT + 1) / 2) * 32;
  T* Qs = reinterpret_cast<T*>(smem);
  T* Ks = Qs + RQ * C::STR;
  T* Vs = Ks + RK * C::STR;
  const T* qg = reinterpret_cast<const T*>(p.q) + (long)b * p.q_bs + (long)h * p.hd;
  const T* kg = reinterpret_cast<const T*>(p.k) + (long)b * p.k_bs + (long)h * p.hd;
  const T* vg = reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + (long)h * p.hd;
  T* og = reinterpret_cast<T*>(p.o) + (long)b * p.o_bs + (long)h * p.hd;
  {
    const StageJob<T> jobs[3] = {{Qs, qg, p.ldq, p.Lq, RQ}, {Ks, kg, p.ldk, p.Lk, RK}, {Vs, vg, p.ldv, p.Lk, RK}};
    stage_multi<T, DT, 3>(jobs, p.hd, lane);
  }
  __syncthreads();
  const unsigned long long kp_row = load_padmask(p, b, lane);
  const Dropout dr = make_dropout(p.seed, p.site, p.p_drop);
  const float scale = 1.0f / sqrtf((float)p.hd);
  const int hd4 = (p.hd + 3) / 4;

  for (int qt = 0; qt < LQT; qt++) {
    f32x4 st[4];
    scores_T<T, DT>(st, Ks, Qs, qt, LKT, hd4, scale, p, kp_row, lane);
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) m = fmaxf(m, st[t][r]);
    m = red4_max(m);
    if (m == -INFINITY) m = 0.0f;
    float l = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) { st[t][r] = expf(st[t][r] - m); l += st[t][r]; }
    l = red4_sum(l);
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    const int qq = qt * 16 + i;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int kk = t * 16 + g * 4 + r;
        st[t][r] *= inv * drop_mult(dr, (uint32_t)((blockIdx.x * p.Lq + qq) * p.Lk + kk));
      }
    // O tile = P V
    f32x4 ot[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) ot[dt] = f32x4{0, 0, 0, 0};
    if constexpr (C::BF) {
#pragma unroll
      for (int kp = 0; kp < 2; kp++) {
        if (kp * 2 < LKT) {
          const bf16x8 pa = pack_p(st[kp * 2], st[kp * 2 + 1]);
#pragma unroll
          for (int dt = 0; dt < DT; dt++)
            ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_colk<C::STR>(Vs, kp * 32, kp * 32 + 16, dt * 16, lane), pa,
                                                             ot[dt], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; t++) {
        if (t < LKT) {
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int dt = 0; dt < DT; dt++)
              ot[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[(t * 16 + g * 4 + r) * C::STR + dt * 16 + i], st[t][r], ot[dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < DT; dt++) store_row4<T>(og, p.ldo, qt * 16 + i, dt * 16 + g * 4, ot[dt], p.Lq, p.hd);   // ot = O^T tile
  }
}

template <typename T, int DT>
__global__ __launch_bounds__(64) void attn_bwd_kernel(const AttnP p) {
  using C = AttnCfg<T, DT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int LQT = (p.Lq + 15) / 16, LKT = (p.Lk + 15) / 16;
  const int RQ = ((LQT + 1) / 2) * 32, RK = ((LKT + 1) / 2) * 32;
  T* Qs = reinterpret_cast<T*>(smem);
  T* dOs = Qs + RQ * C::STR;
  T* Ks = dOs + RQ * C::STR;
  T* Vs = Ks + RK * C::STR;
  float* stat_m = reinterpret_cast<float*>(Vs + RK * C::STR);  // [64] row max
  float* stat_i = stat_m + 64;                                  // [64] 1 / row sum
  float* stat_d = stat_i + 64;                                  // [64] D = rowsum(dP * P)
  const T* qg = reinterpret_cast<const T*>(p.q) + (long)b * p.Lq * p.ldq + (long)h * p.hd;
  const T* kg = reinterpret_cast<const T*>(p.k) + (long)b * p.Lk * p.ldk + (long)h * p.hd;
  const T* vg = reinterpret_cast<const T*>(p.v) + (long)b * p.Lk * p.ldv + (long)h * p.hd;
  const T* dog = reinterpret_cast<const T*>(p.d_o) + (long)b * p.Lq * p.ld_do + (long)h * p.hd;
  T* dqg = reinterpret_cast<T*>(p.dq) + (long)b * p.Lq * p.ld_dq + (long)h * p.hd;
  T* dkg = reinterpret_cast<T*>(p.dk) + (long)b * p.Lk * p.ld_dk + (long)h * p.hd;
  T* dvg = reinterpret_cast<T*>(p.dv) + (long)b * p.Lk * p.ld_dv + (long)h * p.hd;
  {
    const StageJob<T> jobs[4] = {{Qs, qg, p.ldq, p.Lq, RQ}, {dOs, dog, p.ld_do, p.Lq, RQ}, {Ks, kg, p.ldk, p.Lk, RK}, {Vs, vg, p.ldv, p.Lk, RK}};
    stage_multi<T, DT, 4>(jobs, p.hd, lane);
  }
  __syncthreads();
  const unsigned long long kp_row = load_padmask(p, b, lane);
  const Dropout dr = make_dropout(p.seed, p.site, p.p_drop);
  const float scale = 1.0f / sqrtf((float)p.hd);
  const int hd4 = (p.hd + 3) / 4;

  // ---------------- phase A: per query tile, transposed form -> stats + dQ ----------------
  for (int qt = 0; qt < LQT; qt++) {
    f32x4 st[4], dpt[4];
    scores_T<T, DT>(st, Ks, Qs, qt, LKT, hd4, scale, p, kp_row, lane);
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) m = fmaxf(m, st[t][r]);
    m = red4_max(m);
    if (m == -INFINITY) m = 0.0f;
    float l = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) { st[t][r] = expf(st[t][r] - m); l += st[t][r]; }
    l = red4_sum(l);
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    const int qq = qt * 16 + i;
    // dPd^T[key][q] = V[key] . dO[q]
    float dsum = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      dpt[t] = f32x4{0, 0, 0, 0};
      if (t < LKT) {
        if constexpr (C::BF) {
#pragma unroll
          for (int ks = 0; ks < C::KS; ks++)
            dpt[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rowk<C::STR>(Vs, t * 16, ks, lane),
                                                             frag_rowk<C::STR>(dOs, qt * 16, ks, lane), dpt[t], 0, 0, 0);
        } else {
          for (int k4 = 0; k4 < hd4; k4++)
            dpt[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[(t * 16 + i) * C::STR + k4 * 4 + g],
                                                          dOs[(qt * 16 + i) * C::STR + k4 * 4 + g], dpt[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int kk = t * 16 + g * 4 + r;
        st[t][r] *= inv;                                                       // P
        dpt[t][r] *= drop_mult(dr, (uint32_t)((blockIdx.x * p.Lq + qq) * p.Lk + kk));  // dP = dPd * M
        dsum += dpt[t][r] * st[t][r];
      }
    }
    dsum = red4_sum(dsum);
    if (g == 0) { stat_m[qq] = m; stat_i[qq] = inv; stat_d[qq] = dsum; }
    // dS^T = P * (dP - D) ; dQ tile = scale * dS K
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) st[t][r] = st[t][r] * (dpt[t][r] - dsum) * scale;
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) acc[dt] = f32x4{0, 0, 0, 0};
    if constexpr (C::BF) {
#pragma unroll
      for (int kp = 0; kp < 2; kp++) {
        if (kp * 2 < LKT) {
          const bf16x8 pa = pack_p(st[kp * 2], st[kp * 2 + 1]);
#pragma unroll
          for (int dt = 0; dt < DT; dt++)
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_colk<C::STR>(Ks, kp * 32, kp * 32 + 16, dt * 16, lane), pa,
                                                              acc[dt], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; t++) {
        if (t < LKT) {
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int dt = 0; dt < DT; dt++)
              acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(t * 16 + g * 4 + r) * C::STR + dt * 16 + i], st[t][r], acc[dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < DT; dt++) store_row4<T>(dqg, p.ld_dq, qt * 16 + i, dt * 16 + g * 4, acc[dt], p.Lq, p.hd);   // acc = dQ^T tile
  }
  __syncthreads();  // stats visible

  // ---------------- phase B: per key tile, un-transposed form -> dV, dK ----------------
  for (int t = 0; t < LKT; t++) {
    f32x4 av[DT], ak[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) { av[dt] = f32x4{0, 0, 0, 0}; ak[dt] = f32x4{0, 0, 0, 0}; }
    const int kk = t * 16 + i;  // this lane's key column in S[q][key]
#pragma unroll
    for (int qp = 0; qp < 2; qp++) {
      if (qp * 2 < LQT) {
        f32x4 pd[2], dsv[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int qt = qp * 2 + u;
          f32x4 s = f32x4{0, 0, 0, 0}, dp = f32x4{0, 0, 0, 0};
          if (qt < LQT) {
            if constexpr (C::BF) {
#pragma unroll
              for (int ks = 0; ks < C::KS; ks++) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rowk<C::STR>(Qs, qt * 16, ks, lane),
                                                            frag_rowk<C::STR>(Ks, t * 16, ks, lane), s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rowk<C::STR>(dOs, qt * 16, ks, lane),
                                                             frag_rowk<C::STR>(Vs, t * 16, ks, lane), dp, 0, 0, 0);
              }
            } else {
              for (int k4 = 0; k4 < hd4; k4++) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[(qt * 16 + i) * C::STR + k4 * 4 + g],
                                                         Ks[(t * 16 + i) * C::STR + k4 * 4 + g], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(dOs[(qt * 16 + i) * C::STR + k4 * 4 + g],
                                                          Vs[(t * 16 + i) * C::STR + k4 * 4 + g], dp, 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int qq = qt * 16 + g * 4 + r;  // row of S[q][key] held in register r
            float pv = 0.0f, dv = 0.0f, pdv = 0.0f;
            if (qt < LQT && qq < p.Lq && !key_masked(p, kp_row, qq, kk)) {
              const float mult = drop_mult(dr, (uint32_t)((blockIdx.x * p.Lq + qq) * p.Lk + kk));
              pv = expf(s[r] * scale - stat_m[qq]) * stat_i[qq];
              pdv = pv * mult;
              dv = pv * (dp[r] * mult - stat_d[qq]) * scale;
            }
            pd[u][r] = pdv;
            dsv[u][r] = dv;
          }
        }
        if constexpr (C::BF) {
          const bf16x8 pa = pack_p(pd[0], pd[1]);
          const bf16x8 da = pack_p(dsv[0], dsv[1]);
#pragma unroll
          for (int dt = 0; dt < DT; dt++) {
            av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_colk<C::STR>(dOs, qp * 32, qp * 32 + 16, dt * 16, lane), pa, av[dt], 0, 0, 0);
            ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_colk<C::STR>(Qs, qp * 32, qp * 32 + 16, dt * 16, lane), da, ak[dt], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int u = 0; u < 2; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int qrow = (qp * 2 + u) * 16 + g * 4 + r;
#pragma unroll
              for (int dt = 0; dt < DT; dt++) {
                av[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dOs[qrow * C::STR + dt * 16 + i], pd[u][r], av[dt], 0, 0, 0);
                ak[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[qrow * C::STR + dt * 16 + i], dsv[u][r], ak[dt], 0, 0, 0);
              }
            }
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {   // av / ak = dV^T / dK^T tiles
      store_row4<T>(dvg, p.ld_dv, t * 16 + i, dt * 16 + g * 4, av[dt], p.Lk, p.hd);
      store_row4<T>(dkg, p.ld_dk, t * 16 + i, dt * 16 + g * 4, ak[dt], p.Lk, p.hd);
    }
  }
}

template <typename T, int DT> static size_t attn_lds_bytes(int Lq, int Lk, bool bwd) {
  using C = AttnCfg<T, DT>;
  const int LQT = (Lq + 15) / 16, LKT = (Lk + 15) / 16;
  const int RK = ((LKT + 1) / 2) * 32;
  if (!bwd) return (size_t)(LQT * 16 + 2 * RK) * C::STR * sizeof(T);
  const int RQ = ((LQT + 1) / 2) * 32;
  return (size_t)(2 * RQ + 2 * RK) * C::STR * sizeof(T) + 3 * 64 * sizeof(float);
}

template <typename T, int DT> static int attn_launch(const AttnP& p, bool bwd, hipStream_t st) {
  const size_t lds = attn_lds_bytes<T, DT>(p.Lq, p.Lk, bwd);
  if (lds > 160 * 1024) return VCT_E_SHAPE;
  static int attr[2] = {0, 0};
  if (lds > 64 * 1024 && (int)lds > attr[bwd]) {
    hipError_t e = bwd ? hipFuncSetAttribute((const void*)attn_bwd_kernel<T, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                       : hipFuncSetAttribute((const void*)attn_fwd_kernel<T, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr[bwd] = (int)lds;
  }
  const dim3 grid(p.B * p.H);
  if (bwd) vct::launch((attn_bwd_kernel<T, DT>), grid, dim3(64), lds, st, p);
  else vct::launch((attn_fwd_kernel<T, DT>), grid, dim3(64), lds, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

template <typename T> static int attn_dispatch(const AttnP& p, bool bwd, hipStream_t st) {
  if (p.hd <= 16) return attn_launch<T, 1>(p, bwd, st);
  if (p.hd <= 32) return attn_launch<T, 2>(p, bwd, st);
  if (p.hd <= 64) return attn_launch<T, 4>(p, bwd, st);
  if (p.hd <= 96) return attn_launch<T, 6>(p, bwd, st);
  if (p.hd <= 128) return attn_launch<T, 8>(p, bwd, st);
  return VCT_E_SHAPE;
}

}  // namespace vct
using namespace vct;

static int attn_common(const vct_attn_desc* d, bool bwd, void* stream) {
  if (!d || !d->q || !d->k || !d->v) return VCT_E_ARG;
  if (d->dtype != VCT_F32 && d->dtype != VCT_BF16) return VCT_E_ARG;
  if (!bwd && !d->o) return VCT_E_ARG;
  if (bwd && (!d->d_o || !d->dq || !d->dk || !d->dv)) return VCT_E_ARG;
  if (d->B <= 0 || d->H <= 0 || d->Lq <= 0 || d->Lk <= 0 || d->hd <= 0) return VCT_E_SHAPE;
  if (d->Lq > 64 || d->Lk > 64 || d->hd > 128) return VCT_E_SHAPE;
  const int vec = d->dtype == VCT_BF16 ? 8 : 4;
  if (d->hd % vec || d->ldq % vec || d->ldk % vec || d->ldv % vec) return VCT_E_ALIGN;
  if (bwd && (d->ld_do % vec)) return VCT_E_ALIGN;
  // outputs are written four head-dim columns at a time
  const uintptr_t omask = d->dtype == VCT_BF16 ? 7 : 15;
  if (!bwd && ((d->ldo % 4) || ((uintptr_t)d->o & omask))) return VCT_E_ALIGN;
  if (bwd && ((d->ld_dq % 4) || (d->ld_dk % 4) || (d->ld_dv % 4) || (((uintptr_t)d->dq | (uintptr_t)d->dk | (uintptr_t)d->dv) & omask)))
    return VCT_E_ALIGN;
  AttnP p;
  p.B = d->B; p.H = d->H; p.Lq = d->Lq; p.Lk = d->Lk; p.hd = d->hd; p.causal = d->causal;
  p.q = d->q; p.ldq = d->ldq; p.k = d->k; p.ldk = d->ldk; p.v = d->v; p.ldv = d->ldv;
  p.o = d->o; p.ldo = d->ldo;
  p.key_pad = d->key_pad; p.key_pad_shift = d->key_pad_shift;
  p.key_ids = d->key_ids; p.key_ids_bs = d->key_ids_bs; p.pad_id = d->pad_id;
  if (d->key_pad_shift < 0 || (d->key_pad != nullptr && d->key_pad_shift >= d->Lk)) return VCT_E_SHAPE;
  p.seed = d->seed; p.site = d->site; p.p_drop = d->p_drop;
  p.d_o = d->d_o; p.ld_do = d->ld_do;
  p.dq = d->dq; p.ld_dq = d->ld_dq; p.dk = d->dk; p.ld_dk = d->ld_dk; p.dv = d->dv; p.ld_dv = d->ld_dv;
  p.q_bs = d->q_bs ? d->q_bs : (long)d->Lq * d->ldq;
  p.k_bs = d->k_bs ? d->k_bs : (long)d->Lk * d->ldk;
  p.v_bs = d->v_bs ? d->v_bs : (long)d->Lk * d->ldv;
  p.o_bs = d->o_bs ? d->o_bs : (long)d->Lq * d->ldo;
  if (bwd && (d->q_bs || d->k_bs || d->v_bs || d->o_bs)) return VCT_E_ARG;  // strided batches: forward (decode) only
  hipStream_t st = (hipStream_t)stream;
  return d->dtype == VCT_BF16 ? attn_dispatch<bf16_t>(p, bwd, st) : attn_dispatch<float>(p, bwd, st);
}

extern "C" int vct_attn_fwd(const vct_attn_desc* d, void* stream) { return attn_common(d, false, stream); }
extern "C" int vct_attn_bwd(const vct_attn_desc* d, void* stream) { return attn_common(d, true, stream); }
