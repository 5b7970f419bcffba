// ARCHIVED PROBE (round 6), not part of the library; result at the end of this comment.
// Persistent 256x256-tile bf16 GEMM, ring variant of g32_kernel (vct_gemm32_kernel.h, which this file follows line by line outside the K
// loop): the same 128 KB of LDS as FOUR slots of HALF a 64-deep stage (32 k: two k-steps of v_mfma_f32_32x32x16_bf16) instead of two
// whole stages.  Why: g32_kernel can issue the DMA of stage s+2 only when stage s has been read, one stage (~3000 cycles) before it is
// consumed, its last pieces one k-step before -- and an LDS-DMA takes ~1500 cycles to land even from L2: every stage barrier waited
// 600-1000 cycles (profiles/r06_g32_stamps.txt).  With four half-stage slots one slot is being consumed and THREE are ahead: a slot is
// refilled four half-stages (~2 stages) before it is read again; the barrier's wait is a COUNTED vmcnt (the two younger half-stages'
// 4 + 4 pieces stay in flight; vmcnt retires in order).  Price: one barrier per half-stage.  Item boundaries are simple instead of
// overlapped: the ring drains, the epilogue's row slab takes slots 0-1, the next item starts with a cold prologue -- this variant is for
// the long-K forms (vocabulary dX / dW: one item per CU).
//   K-contiguous image of a slot: [256 rows][32 k] = 64 bytes per row, 16-byte chunk c of row r at c ^ ((r >> 2) & 3): the 16 lanes of
//   one ds_read_b128 pass (16 consecutive rows at one chunk) cover 4 rows x 4 chunk positions = all 64 banks; a DMA piece = 16 rows.
//   M/N-contiguous image: [32 k][256 cols], as in g32_kernel (64-byte unit u of k-row r at u ^ (r & 3)).
// RESULT (profiles/r06_g32_ring_probe.txt, tools/g32_probe.hip): correct and bit-identical on every probe shape; vocabulary dX 153 vs
// 155 us (equal), vocabulary dW 174 vs 151 us, NT-through-transpose dX 170 vs 159 us, projection 194 vs 177 us: the longer DMA lead
// buys nothing once the barriers double -- the stage barrier's wait is not (only) DMA latency; the LDS itself (192 KB of fragment
// reads + 64 KB of DMA writes per 64-deep stage = 2048 cycles at 128 B/clk, the stage's MFMA time at peak) is the co-limit.
#pragma once
namespace vct {

__device__ __forceinline__ int kch_swz(int row) { return (row >> 2) & 3; }

// ragged last half-stage: predicated 16-byte loads (zero fill beyond K / beyond the operand's rows) into the swizzled image
template <bool MC, int NT>
__device__ __forceinline__ void tail_tile32h(unsigned char* img, const bf16_t* __restrict__ base, long ld, int r0, int r_ext, int k0, int K, int tid) {
  constexpr int NV = 256 * 4 / NT;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int v = tid + i * NT;
    V16b val; val.w[0] = val.w[1] = val.w[2] = val.w[3] = 0u;
    if constexpr (!MC) {
      const int row = v >> 2, c = v & 3;
      const int gr = r0 + row, gk = k0 + c * 8;
      if (gr < r_ext && gk < K) {
        val = *reinterpret_cast<const V16b*>(base + (long)gr * ld + gk);
        if (gk + 8 > K) {
          const int keep = K - gk;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            if (2 * q >= keep) val.w[q] = 0u;
            else if (2 * q + 1 >= keep) val.w[q] &= 0xffffu;
          }
        }
      }
      *reinterpret_cast<V16b*>(img + row * 64 + ((c ^ kch_swz(row)) << 4)) = val;
    } else {
      const int krow = v >> 5, p = v & 31;                     // 32 sixteen-byte pieces per k-row, 32 k-rows
      const int col = p * 8;
      const int gk = k0 + krow, gr = r0 + col;
      if (gk < K && gr < r_ext) val = *reinterpret_cast<const V16b*>(base + (long)gk * ld + gr);
      *reinterpret_cast<V16b*>(img + krow * 512 + ((((p >> 2) ^ mc32_swz(krow)) << 2) | (p & 3)) * 16) = val;
    }
  }
}

// The caller's G256P (vct_gemm256.hip, which includes this file) is reused: same work-item order, same output conventions.
template <int TA, int TB, typename TO, int VAR>
__global__ __launch_bounds__(512, 2) void g32r_kernel(const G256P p) {
  constexpr bool A_MC = (TA == 1), B_MC = (TB == 0);
  constexpr bool BG = (TA == 1 && TB == 0);
  constexpr int T = 32, NSTEP = 2, TM = 4, TN = 2;                 // a ring slot = HALF a 64-deep stage: two k-steps of 16
  constexpr int NSLOT = 4, BKH = 32;
  constexpr int NT = 512, WM = 128;
  constexpr int ES = (int)sizeof(TO);
  constexpr int SLOT = G256_STAGE / 2, A_BYTES = G256_BM * 64;      // 32 KB per slot: A image 16 KB + B image 16 KB
  constexpr int STAGE = G256_STAGE;                                  // (the epilogue's row slab: two adjacent slots)
  constexpr int RPR = STAGE / (G256_BN * ES);                      // slab rows per round: 128 (bf16) / 64 (fp32)
  constexpr int CPRW = G256_BN * ES / 16;                          // 16-byte chunks per slab row
  constexpr int CPT = RPR * CPRW / NT;                             // chunks per thread and round
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  if constexpr (VAR & 1) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }

  const int nitems = p.tiles_m * p.tiles_n * p.split;
  const int nxw = (int)gridDim.x >> 3;
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int w_begin = xcd * per, w_end = min(nitems, w_begin + per);
  const int nkt = (p.K + BKH - 1) / BKH, kt_full = p.K / BKH;     // in half-stages

  auto item = [&](int w, int& m0, int& n0, int& z, int& k_lo, int& k_hi) {
    int tile;
    if (p.zmajor) { const int nt = p.tiles_m * p.tiles_n; z = w / nt; tile = w - z * nt; }
    else { tile = w / p.split; z = w - tile * p.split; }
    if (p.order == 0) {
      m0 = (tile % p.tiles_m) * G256_BM; n0 = (tile / p.tiles_m) * G256_BN;
    } else {
      const int per_group = 8 * p.tiles_m;
      const int grp = tile / per_group, rem = tile - grp * per_group;
      const int gw = min(8, p.tiles_n - grp * 8);
      m0 = (rem / gw) * G256_BM; n0 = (grp * 8 + rem % gw) * G256_BN;
    }
    k_lo = 2 * z * p.kt_per_split; k_hi = min(nkt, k_lo + 2 * p.kt_per_split);   // (kt_per_split counts 64-deep stages)
  };

  // ---- operand DMA: 4 + 4 one-KiB pieces per wave and stage; voff = byte offset of the lane's 16 bytes at K offset 0 ----
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7fffffff, 0x00020000);
  const int lda32 = (int)p.lda, ldb32 = (int)p.ldb;               // (operands < 2 GiB: g32_eligible)
  const int kstrA = A_MC ? 64 * lda32 : 64, kstrB = B_MC ? 64 * ldb32 : 64;       // bytes per half-stage
  int voff[4];                                                     // 2 + 2 one-KiB pieces per wave and slot
  int pf_off = 0;                                                  // L2 prefetch of A (below): byte offset of the thread's dword at K offset 0
  auto set_voff = [&](int m0, int n0) {
    int l = lane;
    asm volatile("" : "+v"(l));                                    // opaque: nothing of this is hoisted out of the item loop and kept alive
    if (p.pf_dist > 0) {                                           // one dword per 64 bytes of the 32 KB A stage = one per thread
      int t = tid;
      asm volatile("" : "+v"(t));
      t &= 255;                                                    // (a half-stage of A is 256 x 64 bytes: the upper half of the threads repeats)
      if constexpr (!A_MC) pf_off = (min(m0 + t, p.M - 1) * lda32) * 2;
      else pf_off = ((t >> 3) * lda32 + min(m0 + (t & 7) * 32, ((p.M + 7) & ~7) - 8)) * 2;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ci = (q & 1) * 8 + wave;
      const bool mc = q < 2 ? A_MC : B_MC;
      const int r0 = q < 2 ? m0 : n0, ext = q < 2 ? p.M : p.N, ld = q < 2 ? lda32 : ldb32;
      if (!mc) {                                                   // [256 rows][32 k]: a piece = 16 rows x 64 bytes
        const int row = ci * 16 + (l >> 2);
        const int c = (l & 3) ^ kch_swz(row);
        voff[q] = (min(r0 + row, ext - 1) * ld + c * 8) * 2;
      } else {
        const int krow = ci * 2 + (l >> 5), pp = l & 31;
        const int col = ((((pp >> 2) ^ mc32_swz(krow)) << 2) | (pp & 3)) * 8;
        const int rlim = ((ext + 7) & ~7) - 8;                     // last fully readable vector (ld covers the rounded-up extent)
        voff[q] = (krow * ld + min(r0 + col, rlim)) * 2;
      }
    }
  };
  auto dma = [&](auto Q, unsigned char* stage_buf, int kt) {
    constexpr int q = decltype(Q)::value;
    const int ci = (q & 1) * 8 + wave;
    unsigned char* dst = stage_buf + (q >= 2 ? A_BYTES : 0) + ci * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(q >= 2 ? rsB : rsA, (__attribute__((address_space(3))) void*)dst, 16, voff[q],
                                             kt * (q >= 2 ? kstrB : kstrA), 0, 0);
  };
  // L2 prefetch of the A operand's stage `kt`: the vocabulary gradients stream 297 MB of dlogits from HBM once, a stage's DMA is issued
  // one stage (~1.4 us) before it is consumed and HBM answers in about that time under load -- every stage barrier waited 600-1000
  // cycles of 3600 for it (tools/g32_probe.hip stamps, profiles/r06_g32_stamps.txt).  A 4-byte LDS-DMA per thread into a dump area
  // behind the stages touches every 64 bytes of the stage p.pf_dist stages ahead: the real DMA then hits in L2.  It is issued as the
  // LAST vector-memory instruction in front of the stage barrier, whose wait is vmcnt(1): it never waits for the prefetch itself.
  auto prefetch_a = [&](int kt) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(lds + NSLOT * SLOT + wave * 256), 4, pf_off,
                                             kt * kstrA, 0, 0);
  };
  auto tail_stage = [&](unsigned char* stage_buf, int m0, int n0, int kt) {      // ragged stage: register path, every thread
    tail_tile32h<A_MC, NT>(stage_buf, A, p.lda, m0, p.M, kt * BKH, p.K, tid);
    tail_tile32h<B_MC, NT>(stage_buf + A_BYTES, B, p.ldb, n0, p.N, kt * BKH, p.K, tid);
  };

  // ---- fragments: per-lane LDS byte offsets at k-step 0, tile 0 ----
  // K-contiguous: lane = row (lane & 31), chunk = step * 2 + (lane >> 5); the k-step flips bits 5-6 of the offset, tiles add immediates.
  // M/N-contiguous: two transpose reads per fragment (k-rows +0..3, +4..7 of the lane half's eight); within a 16-lane group lane 4r + c
  // points at columns 4c..4c+3 of k-row r; the k-step adds 8 KiB, the tile flips bits 6-7 (block index + 2 per 32 columns).
  const int rl = lane & 31, hl = lane >> 5;
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  int offA, offA2 = 0, offB, offB2 = 0;
  if constexpr (!A_MC) { const int r = wm * WM + rl; offA = r * 64 + ((hl ^ kch_swz(r)) << 4); }
  else {
    const int kr = hl * 8 + (i16 >> 2), unit = wm * 4;
    offA = kr * 512 + ((unit ^ mc32_swz(kr)) << 6) + g16 * 32 + (i16 & 3) * 8;
    offA2 = offA + 4 * 512;
  }
  if constexpr (!B_MC) { const int r = wn * 64 + rl; offB = A_BYTES + r * 64 + ((hl ^ kch_swz(r)) << 4); }
  else {
    const int kr = hl * 8 + (i16 >> 2), unit = wn * 2;
    offB = A_BYTES + kr * 512 + ((unit ^ mc32_swz(kr)) << 6) + g16 * 32 + (i16 & 3) * 8;
    offB2 = offB + 4 * 512;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  bf16x8 fa[2][TM], fb[2][TN];
  // the asm transpose reads are not tracked by hipcc's lgkmcnt bookkeeping: before a fragment set is consumed, wait for every LDS read in
  // flight and tie the wait to the registers (a register-only MFMA may otherwise be hoisted above it)
  auto wait_frags0 = [&]() {
    if constexpr (A_MC || B_MC)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fb[0][0]), "+v"(fb[0][1]) :: "memory");
  };
  auto wait_frags1 = [&]() {
    if constexpr (A_MC || B_MC)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fb[1][0]), "+v"(fb[1][1]) :: "memory");
  };
  auto read_one = [&](auto MCT, const unsigned char* sb, int o1, int o2, int step, int t) -> bf16x8 {
    if constexpr (!decltype(MCT)::value) {
      return *reinterpret_cast<const bf16x8*>(sb + (o1 ^ (step << 5)) + t * T * 64);
    } else {
      // INLINE ASM, on purpose: behind a pending LDS-DMA hipcc puts `s_waitcnt vmcnt(0)` in front of the transpose-read INTRINSIC (it
      // does not for plain ds_read_b128 loads) -- every k-step then waits for the DMA of the NEXT stage to land, the DMA never overlaps
      // the MFMAs, and the NN / TN forms pay the whole DMA time on top of the compute loop (+55-78 us on the vocabulary products, in
      // gemm256_kernel too).  The asm is invisible to that pass; completion is ordered by wait_frags() below.
      const uint32_t ad = (uint32_t)(sb - lds) + lds_base + (uint32_t)(o1 ^ (t << 6));
      s16x4 lo, hi;
      if (step == 0) asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(lo), "=&v"(hi) : "v"(ad) : "memory");
      else asm volatile("ds_read_b64_tr_b16 %0, %2 offset:8192\n\tds_read_b64_tr_b16 %1, %2 offset:10240" : "=&v"(lo), "=&v"(hi) : "v"(ad) : "memory");
      (void)o2;
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8, v);
    }
  };
  auto read_frags = [&](auto SET, const unsigned char* sb, int step) {
    constexpr int st = decltype(SET)::value;
    int oa = offA, oa2 = offA2, ob = offB, ob2 = offB2;
    asm volatile("" : "+v"(oa), "+v"(oa2), "+v"(ob), "+v"(ob2));  // (per-step / per-tile variants are recomputed, not hoisted and spilled)
#pragma unroll
    for (int j = 0; j < TN; j++) fb[st][j] = read_one(std::integral_constant<bool, B_MC>{}, sb, ob, ob2, step, j);
#pragma unroll
    for (int i = 0; i < TM; i++) fa[st][i] = read_one(std::integral_constant<bool, A_MC>{}, sb, oa, oa2, step, i);
  };
  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
  };
  // accumulators hold C TRANSPOSED per MFMA tile (operands swapped): lane = row i * 32 + (lane & 31) of the wave's piece, register
  // 4q + r = column j * 32 + q * 8 + (lane >> 5) * 4 + r
  auto mfma_step = [&](auto SET) {
    constexpr int st = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[st][j], fa[st][i], acc[i][j], 0, 0, 0);
  };
  constexpr int NRD = TM * (A_MC ? 2 : 1) + TN * (B_MC ? 2 : 1), NMF = TM * TN;   // LDS reads / MFMAs per k-step
  zero_acc();
  // bias of the lane's columns (groups of four consecutive columns, index j * 4 + q)
  constexpr int NBG = TN * 4;
  constexpr bool HAS_BIAS = (TA == 0 && TB == 1 && ES == 2);       // the plain NT form only (gemm256_try): 32 registers back elsewhere
  f32x4 bj[HAS_BIAS ? NBG : 1];
  // bias gradient (weight-gradient form) = row sums of op(A).  The four waves of a row group hold the same A fragments: wave wn sums the
  // fragments of k-step j == wn of every stage (all four tile rows: static register indices -- a wave-uniform SELECT of the tile row
  // makes hipcc park the fragments in scratch and wait vmcnt(0) per k-step, 580 us), the four partial sums meet in LDS at the tile's
  // end in fixed order.  (gemm256_kernel: the wn == 0 wave did all of it, 64 VALU per k-step, and the other six waves waited for it
  // at every stage barrier: the TN compute loop took 150 us with the bias gradient and 106 without.)
  float accb[BG ? TM : 1];
#pragma unroll
  for (int i = 0; i < (BG ? TM : 1); i++) accb[i] = 0.0f;
  int n0 = 0;
  auto load_bias = [&]() {
    if constexpr (!HAS_BIAS) return;
    int h = hl;
    asm volatile("" : "+v"(h));
#pragma unroll
    for (int g = 0; g < (HAS_BIAS ? NBG : 1); g++) {
      const int col = n0 + wn * 64 + (g >> 2) * T + (g & 3) * 8 + h * 4;
      if (p.bias == nullptr || p.partial != nullptr) bj[g] = f32x4{0, 0, 0, 0};
      else if (col + 4 <= p.N) bj[g] = *reinterpret_cast<const f32x4*>(p.bias + col);
      else {
#pragma unroll
        for (int r = 0; r < 4; r++) bj[g][r] = p.bias[min(col + r, p.N - 1)];
      }
    }
  };
  auto bias_grad_step = [&](auto SET) {
    if constexpr (BG) {
      constexpr int st = decltype(SET)::value;
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const s16x8 v = __builtin_bit_cast(s16x8, fa[st][i]);
        float t = 0.0f;
#pragma unroll
        for (int u = 0; u < 8; u++) t += bf2f((bf16_t)v[u]);
        accb[i] += t;
      }
    }
  };

  bool adam_on = false;
  AdamConsts hc = {};
  if constexpr (BG && ES == 4) {
    adam_on = p.adam.param != nullptr && p.partial == nullptr;
    if (adam_on) hc = adam_consts_uniform(p.adam.hyper, p.adam.step);
  }
  // (VAR & 32, tools/g32_probe.hip: wave 0 of workgroup 0 logs (label, shader clock) pairs into the buffer p.bias_grad points at)
  int sidx = 0;
  auto stamp = [&](int label) {
    if constexpr ((VAR & 32) != 0) {
      if (blockIdx.x == 0 && tid == 0 && sidx < 400) {
        long long* sp = reinterpret_cast<long long*>(p.bias_grad);
        sp[2 * sidx] = label; sp[2 * sidx + 1] = (long long)clock64(); sidx++;
      }
    }
  };
  int m0 = 0, z = 0, k_lo = 0, k_hi = 0;
  for (int w = w_begin + slot; w < w_end; w += nxw) {
    // ---- item prologue: the ring is empty (the previous item drained it): four half-stages in flight, the first one awaited ----
    item(w, m0, n0, z, k_lo, k_hi);
    set_voff(m0, n0);
    static_for<NSLOT>([&](auto U) {                                // (every item starts with four full half-stages: g32_eligible)
      constexpr int u = decltype(U)::value;
      static_for<4>([&](auto Q) { dma(Q, lds + u * SLOT, k_lo + u); });
    });
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(std::integral_constant<int, 0>{}, lds, 0);
    const bool do_bg = BG && p.bias_grad != nullptr && n0 == 0;
    const int dma_end = min(k_hi, kt_full);                         // half-stages [k_lo, dma_end) of this item arrive by DMA
    int pend_kt = -1;                                               // half-stage whose DMA pieces 1 and 3 are still to be issued ...
    unsigned char* pend_sb = lds;                                   // ... into this slot
    for (int kt = k_lo; kt < k_hi; kt++) {
      unsigned char* sb = lds + ((kt - k_lo) & (NSLOT - 1)) * SLOT;
      unsigned char* nb = lds + ((kt - k_lo + 1) & (NSLOT - 1)) * SLOT;
      const bool last = kt + 1 == k_hi;
      stamp(0);
      if (last) load_bias();
      // ---- k-step 0: MFMAs of step 0, reads of step 1 between them; the second half of the DMA group opened in the previous k-step ----
      if (pend_kt >= 0) { dma(std::integral_constant<int, 1>{}, pend_sb, pend_kt); dma(std::integral_constant<int, 3>{}, pend_sb, pend_kt); pend_kt = -1; }
      wait_frags0();
      read_frags(std::integral_constant<int, 1>{}, sb, 1);
      mfma_step(std::integral_constant<int, 0>{});
      static_for<NMF>([&](auto I) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int lo = decltype(I)::value * NRD / NMF, hi = (decltype(I)::value + 1) * NRD / NMF;
        if constexpr (hi > lo) __builtin_amdgcn_sched_group_barrier(0x100, hi - lo, 0);
      });
      if (do_bg && ((2 * (kt - k_lo)) & 3) == wn) bias_grad_step(std::integral_constant<int, 0>{});
      // ---- every fragment of this half-stage is in registers; the NEXT one must have landed.  In flight behind it: the (up to two)
      // younger half-stages' 4 pieces each (+ the prefetch): a counted wait, vmcnt retires in order ----
      stamp(1);
      {
        const int younger = max(0, min(2, dma_end - (kt + 2)));
        if (p.pf_dist > 0) {
          prefetch_a(min(kt + p.pf_dist, dma_end - 1));
          if (younger == 2) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
          else if (younger == 1) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        } else {
          if (younger == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
          else if (younger == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
      }
      stamp(2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stamp(3);
      // ---- k-step 1: first fragments of the next half-stage, MFMAs, and this slot (now read by everyone) refilled four half-stages ahead ----
      const int kt4 = kt + NSLOT;
      const bool refill = kt4 < k_hi && !(VAR & 8);
      const bool refill_dma = refill && kt4 < kt_full;
      if (refill && !refill_dma) tail_stage(sb, m0, n0, kt4);
      wait_frags1();
      if (!last) read_frags(std::integral_constant<int, 0>{}, nb, 0);
      static_for<TM>([&](auto I) {
        constexpr int i = decltype(I)::value;
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
        if constexpr (i == 1) { if (refill_dma) dma(std::integral_constant<int, 0>{}, sb, kt4); }
        if constexpr (i == 3) { if (refill_dma) dma(std::integral_constant<int, 2>{}, sb, kt4); }
      });
      if (refill_dma) { pend_kt = kt4; pend_sb = sb; }
      if (do_bg && ((2 * (kt - k_lo) + 1) & 3) == wn) bias_grad_step(std::integral_constant<int, 1>{});
    }
    // ---- epilogue: nothing is in flight (the last refills were consumed), every slot has been read: the row slab takes slots 0-1 ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    lds_barrier();
    stamp(4);
    unsigned char* slab = lds;
    const bool part = p.partial != nullptr;
    if constexpr (BG) {
      if (do_bg) {     // (workgroup-uniform) lane halves hold the two halves of a k-step's k's, the four wn waves the four k-steps
        float* red = reinterpret_cast<float*>(slab);                // [wm][wn][128 rows]: 4 KB of the free stage
#pragma unroll
        for (int i = 0; i < TM; i++) {
          float t = accb[i];
          t += __shfl_xor(t, 32);
          if (hl == 0) red[(wm * 4 + wn) * WM + i * T + rl] = t;
        }
        lds_barrier();
        if (wn == 0) {
#pragma unroll
          for (int h2 = 0; h2 < 2; h2++) {
            const int r = h2 * 64 + lane;
            const float t = ((red[(wm * 4 + 0) * WM + r] + red[(wm * 4 + 1) * WM + r]) + red[(wm * 4 + 2) * WM + r]) + red[(wm * 4 + 3) * WM + r];
            const int row = m0 + wm * WM + r;
            if (row < p.M) (part ? p.partial + (size_t)p.split * p.M * p.N + (size_t)z * p.M : p.bias_grad)[row] = t;
          }
        }
        lds_barrier();                                              // the slab rounds below reuse these bytes
      }
#pragma unroll
      for (int i = 0; i < TM; i++) accb[i] = 0.0f;
    }
    float* pc = part ? p.partial + (size_t)z * (size_t)p.M * (size_t)p.N : nullptr;
    const long ldo = part ? (long)p.N : p.ldc;
    constexpr int MTR = RPR / (2 * T);                              // MFMA tile rows per wave and round: 2 (bf16) / 1 (fp32)
    // opaque copies of the lane coordinates: the slab offsets / output coordinates below depend on the lane only, and hipcc otherwise
    // computes all of them at kernel entry, spills them across the K loops and reloads them (10 scratch loads, a round trip each) here
    int e_rl = rl, e_hl = hl, e_tid = tid;
    asm volatile("" : "+v"(e_rl), "+v"(e_hl), "+v"(e_tid));
    if constexpr (VAR & 4) {                                        // (ablation: no epilogue; the accumulators stay live at 128 adds per tile)
      float t = 0.0f;
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) t += acc[i][j][r];
      if (t == 12345.678f) reinterpret_cast<float*>(p.C)[tid] = t;
    } else
    static_for<TM / MTR>([&](auto RD) {
      constexpr int rd = decltype(RD)::value;
      if constexpr (rd > 0) { lds_barrier(); stamp(7); }            // the slab has been read out by everyone
#pragma unroll
      for (int ii = 0; ii < MTR; ii++) {
        const int sr = (wm * MTR + ii) * T + e_rl;                    // slab row; 16-byte chunk c of row r sits at c ^ (r & 31)
#pragma unroll
        for (int j = 0; j < TN; j++) {
#pragma unroll
          for (int q = 0; q < 4; q++) {                             // groups of four consecutive columns
            const int e0 = wn * 64 + j * T + q * 8 + e_hl * 4;
            const f32x4 bv = HAS_BIAS ? bj[HAS_BIAS ? j * 4 + q : 0] : f32x4{0, 0, 0, 0};
            if constexpr (ES == 2) {
              struct alignas(8) B4 { bf16_t e[4]; } v;
#pragma unroll
              for (int r = 0; r < 4; r++) v.e[r] = f2bf(acc[rd * MTR + ii][j][q * 4 + r] + bv[r]);
              const int g8 = e0 >> 2;
              *reinterpret_cast<B4*>(slab + sr * (G256_BN * 2) + ((((g8 >> 1) ^ (sr & 31)) << 1) | (g8 & 1)) * 8) = v;
            } else {
              f32x4 v;
#pragma unroll
              for (int r = 0; r < 4; r++) v[r] = acc[rd * MTR + ii][j][q * 4 + r] + bv[r];
              *reinterpret_cast<f32x4*>(slab + sr * (G256_BN * 4) + (((e0 >> 2) ^ (sr & 31)) << 4)) = v;
            }
          }
        }
      }
      stamp(5);
      lds_barrier();
      stamp(6);
      if constexpr (BG && ES == 4) {
        if (adam_on) {
          // Optimizer epilogue (include/vct_hip.h, vct_gemm_adam; the same shared update as the 128 x 128 kernel's, csrc/vct_adam_core.h):
          // the gradient chunk this lane would store is consumed by torch.optim.Adam's update of its four parameters; parameter / moment
          // vectors of TWO chunks are in flight before the first update (four: 88 spilled registers beside the live accumulators) (a chunk is three dependent HBM round trips otherwise).
          static_for<CPT / 2>([&](auto H) {
            constexpr int h = decltype(H)::value;
            float4 ap[2], am[2], av[2];
            int ae[2]; bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
              const int cid = (h * 2 + u) * NT + e_tid;
              const int sr = cid / CPRW, c = cid % CPRW;
              const int row = m0 + (sr / (MTR * T)) * WM + (rd * MTR + ((sr / T) % MTR)) * T + (sr % T), col = n0 + c * 4;
              ok[u] = row < p.M && col + 4 <= p.N;
              ae[u] = min(row, p.M - 1) * (int)p.ldc + min(col, max(p.N - 4, 0));      // (M x ldc < 2^31: gemm256_try)
              ap[u] = *reinterpret_cast<const float4*>(p.adam.param + ae[u]);
              am[u] = *reinterpret_cast<const float4*>(p.adam.m + ae[u]);
              av[u] = *reinterpret_cast<const float4*>(p.adam.v + ae[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
              const int cid = (h * 2 + u) * NT + e_tid;
              const int sr = cid / CPRW, c = cid % CPRW;
              const f32x4 g = *reinterpret_cast<const f32x4*>(slab + sr * (G256_BN * 4) + ((c ^ (sr & 31)) << 4));
              if (ok[u]) {
                float* pp = &ap[u].x; float* mp = &am[u].x; float* vp = &av[u].x;
#pragma unroll
                for (int e = 0; e < 4; e++) adam_update(pp[e], g[e], mp[e], vp[e], hc);
                *reinterpret_cast<float4*>(p.adam.param + ae[u]) = ap[u];
                *reinterpret_cast<float4*>(p.adam.m + ae[u]) = am[u];
                *reinterpret_cast<float4*>(p.adam.v + ae[u]) = av[u];
                if (p.adam.shadow != nullptr) {
                  const int row = ae[u] / (int)p.ldc, col = ae[u] - row * (int)p.ldc;
                  ushort4 o;
                  o.x = f2bf(ap[u].x); o.y = f2bf(ap[u].y); o.z = f2bf(ap[u].z); o.w = f2bf(ap[u].w);
                  *reinterpret_cast<ushort4*>(p.adam.shadow + (size_t)row * p.adam.ld_shadow + col) = o;
                }
                if (p.adam.store_grad) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + ae[u]) = g;
              }
            }
          });
        }
      }
      if (!adam_on)
#pragma unroll 2
      for (int q = 0; q < CPT; q++) {
        const int cid = q * NT + e_tid;
        const int sr = cid / CPRW, c = cid % CPRW;
        const int row = m0 + (sr / (MTR * T)) * WM + (rd * MTR + ((sr / T) % MTR)) * T + (sr % T);
        constexpr int EPC = 16 / ES;
        const int col = n0 + c * EPC;
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab + sr * (G256_BN * ES) + ((c ^ (sr & 31)) << 4));
        if (row < p.M && col < p.N) {
          TO* dst = (part ? reinterpret_cast<TO*>(pc) : reinterpret_cast<TO*>(p.C)) + (size_t)row * ldo + col;
          if (col + EPC <= p.N && (ldo % EPC) == 0) *reinterpret_cast<u32x4*>(dst) = v;
          else {
            const TO* e = reinterpret_cast<const TO*>(&v);
            for (int qq = 0; qq < EPC; qq++) if (col + qq < p.N) dst[qq] = e[qq];
          }
        }
      }
    });
    stamp(8);
    zero_acc();
    lds_barrier();                                                  // the slab has been read out by everyone: the next item's prologue refills the ring
    stamp(9);
  }
}

template <int TA, int TB, typename TO, int VAR> static int g32r_launch(const G256P& p, hipStream_t st) {
  static vct::DynLdsOptIn optin;
  if (hipError_t e = optin.ensure((const void*)g32r_kernel<TA, TB, TO, VAR>, G32_LDS); e != hipSuccess) return (int)e;
  vct::launch(g32r_kernel<TA, TB, TO, VAR>, dim3(persistent_grid(st)), dim3(512), (size_t)G32_LDS, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

}  // namespace vct
