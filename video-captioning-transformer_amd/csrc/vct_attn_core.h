// Shared device code of the attention kernels (vct_attn.hip: one to four waves per (batch, head); the sample-stationary layer kernels'
// in-LDS attention builds on the same helpers).  See vct_attn.hip for the algorithm notes.
#pragma once
#include "vct_common.h"

namespace vct {

// softmax exponential: libm's expf (range reduction + polynomial, ~25 vector instructions, on the critical path of a latency-bound
// one-wave kernel) only in the fp32 parity mode; the bf16 path takes v_exp_f32 (1 ulp of 2^x: far below a bf16 ulp of P)
template <typename T> __device__ __forceinline__ float attn_exp(float x) {
  if constexpr (sizeof(T) == 2) return __expf(x);
  else return expf(x);
}


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
__device__ __forceinline__ void stage_load4(AV16 (&val)[4], const StageJob<T>& j, int base, int hd, int lane, int nt = 64) {
  using C = AttnCfg<T, DT>;
  constexpr int VPR = C::HDK / C::VEC;
  const int total = j.rows_alloc * VPR;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int idx = min(base + u * nt + lane, total - 1);
    const int r = min(idx / VPR, j.L - 1), c = min((idx % VPR) * C::VEC, hd - C::VEC);
    val[u] = *reinterpret_cast<const AV16*>(j.g + (long)r * j.ld + c);
  }
}
template <typename T, int DT>
__device__ __forceinline__ void stage_commit4(AV16 (&val)[4], const StageJob<T>& j, int base, int hd, int lane, int nt = 64) {
  using C = AttnCfg<T, DT>;
  constexpr int VPR = C::HDK / C::VEC;
  const int total = j.rows_alloc * VPR;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int idx = base + u * nt + lane;
    if (idx < total) {
      const int r = idx / VPR, c = (idx % VPR) * C::VEC;
      if (r >= j.L || c >= hd) val[u].w[0] = val[u].w[1] = val[u].w[2] = val[u].w[3] = 0u;
      *reinterpret_cast<AV16*>(j.lds + r * C::STR + c) = val[u];
    }
  }
}
// lane / nt: the staging thread's index and the number of threads that stage together (one wave: lane, 64; a whole multi-wave
// workgroup: threadIdx.x, blockDim.x -- then a 64 x 128 slice is one round trip instead of four)
template <typename T, int DT, int NJ>
__device__ __forceinline__ void stage_multi(const StageJob<T> (&jobs)[NJ], int hd, int lane, int nt = 64) {
  using C = AttnCfg<T, DT>;
  constexpr int VPR = C::HDK / C::VEC;
  AV16 val[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; j++) stage_load4<T, DT>(val[j], jobs[j], 0, hd, lane, nt);
#pragma unroll
  for (int j = 0; j < NJ; j++) stage_commit4<T, DT>(val[j], jobs[j], 0, hd, lane, nt);
#pragma unroll
  for (int j = 0; j < NJ; j++) {       // slices with more than 4 vectors per staging thread (long sequences / wide heads)
    const int total = jobs[j].rows_alloc * VPR;
    for (int base = 4 * nt; base < total; base += 4 * nt) {
      stage_load4<T, DT>(val[0], jobs[j], base, hd, lane, nt);
      stage_commit4<T, DT>(val[0], jobs[j], base, hd, lane, nt);
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

// Forward of ONE (batch b, head h) by ONE wave: stages its Q/K/V slices into `smem` (attn_lds_bytes<T, DT>(Lq, Lk,
// false) bytes, private to the wave), writes O rows to global memory and -- when o_lds is given -- also into a
// workgroup-shared LDS panel [rows][o_lds_stride] at columns h*hd.. (rows >= Lq of the last tile are written as
// computed from zero-padded Q: finite values the consumer ignores).  bh = b*H + h indexes the dropout stream.
template <typename T, int DT>
__device__ __forceinline__ void attn_fwd_wave(const AttnP& p, const int b, const int h, const int bh, unsigned char* smem,
                                              const int lane_in, T* o_lds, const long o_lds_stride, const int nwv = 1) {
  // nwv > 1: `smem` is shared by the nwv waves of the workgroup (lane_in = threadIdx.x): they stage together and take the query
  // tiles in turn
  const int lane = lane_in & 63, wv = lane_in >> 6;
  using C = AttnCfg<T, DT>;
  const int i = lane & 15, g = lane >> 4;
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
    stage_multi<T, DT, 3>(jobs, p.hd, lane_in, nwv * 64);
  }
  __syncthreads();
  const unsigned long long kp_row = load_padmask(p, b, lane);
  const Dropout dr = make_dropout(p.seed, p.site, p.p_drop);
  const float scale = 1.0f / sqrtf((float)p.hd);
  const int hd4 = (p.hd + 3) / 4;

  for (int qt = wv; qt < LQT; qt += nwv) {
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
    for (int t = 0; t < 4; t++) {           // key tiles beyond Lk hold -inf: nothing to exponentiate, hash or scale there
      if (t < LKT) {
#pragma unroll
        for (int r = 0; r < 4; r++) { st[t][r] = attn_exp<T>(st[t][r] - m); l += st[t][r]; }
      } else {
        st[t] = f32x4{0, 0, 0, 0};
      }
    }
    l = red4_sum(l);
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    const int qq = qt * 16 + i;
#pragma unroll
    for (int t = 0; t < 4; t++)
      if (t < LKT) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int kk = t * 16 + g * 4 + r;
          st[t][r] *= inv * drop_mult(dr, (uint32_t)((bh * p.Lq + qq) * p.Lk + kk));
        }
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
    for (int dt = 0; dt < DT; dt++) {   // ot = O^T tile
      store_row4<T>(og, p.ldo, qt * 16 + i, dt * 16 + g * 4, ot[dt], p.Lq, p.hd);
      if (o_lds != nullptr) store_row4<T>(o_lds + h * p.hd, o_lds_stride, qt * 16 + i, dt * 16 + g * 4, ot[dt], LQT * 16, p.hd);
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

}  // namespace vct
