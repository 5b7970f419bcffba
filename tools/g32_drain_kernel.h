// Dev probe (round 6, tools/g32_probe.hip): register-drain epilogue of the pipelined 256x256 NT kernel.  Measured (profiles/r06_g32_probe_drain.txt):
// correct, 165-170 us on the vocabulary projection against 161-169 us for the LDS-slab epilogue of g32_kernel and 171-173 us for the
// round-5 kernel -- the stores block the issuing wave whether they sit in the epilogue or between the next tile's MFMAs, and the
// kernel as a whole moves 560 MB + through the fabric: not adopted, kept here for the record.
#pragma once

namespace vct {

// ---- NT, bf16 out: the tile leaves STRAIGHT FROM REGISTERS, under the next tile's K loop ------------------------------------------------
// gemm256_kernel's epilogue (accumulators -> LDS slab -> whole-row stores, four barriers) costs the vocabulary projection 33-47 us of
// its 170: one workgroup owns the CU, so while its stores issue (~14 B/clk per CU) the matrix pipe idles, nine tiles in a row.  Here
//   * at tile end the accumulators (+ bias) are packed to bf16 in registers (64 VGPRs for the wave's 128 x 64 piece); one
//     v_permlane32_swap per dword pairs the half-waves' column groups so that a lane owns 16 contiguous bytes of its row (guide T21);
//   * tile rows i < TM - DI are stored at once, the last DI tile rows stay in registers and leave as buffer_store_dwordx4 between the
//     MFMAs of the NEXT tile's first stages (two or four store instructions per stage);
//   * no LDS slab: the operand ring never pauses at a tile boundary (the held-back DMA of g32_kernel is gone).
// Row-per-lane 16-byte stores: a wave's four instructions of one tile row complete whole 128-byte lines of 32 rows.
template <int DI, int NDR, int VAR>
__global__ __launch_bounds__(512, 2) void g32d_kernel(const G256P p) {
  constexpr int MF = 32, T = 32, NSTEP = 4, TM = 4, TN = 2;
  constexpr int WM = 128;
  constexpr int STAGE = G256_STAGE, A_BYTES = G256_BM * 128;
  static_assert(DI >= 0 && DI <= TM && NDR >= 1 && (DI * 4) % NDR == 0, "stores per drain stage");
  using acc_t = f32x16;
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  if constexpr (VAR & 1) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }

  const int nitems = p.tiles_m * p.tiles_n;
  const int nxw = (int)gridDim.x >> 3;
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int w_begin = xcd * per, w_end = min(nitems, w_begin + per);
  const int nkt = p.K / BK2;

  auto item = [&](int w, int& m0, int& n0) {
    const int tile = w;
    if (p.order == 0) {
      m0 = (tile % p.tiles_m) * G256_BM; n0 = (tile / p.tiles_m) * G256_BN;
    } else {
      const int per_group = 8 * p.tiles_m;
      const int grp = tile / per_group, rem = tile - grp * per_group;
      const int gw = min(8, p.tiles_n - grp * 8);
      m0 = (rem / gw) * G256_BM; n0 = (grp * 8 + rem % gw) * G256_BN;
    }
  };

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7fffffff, 0x00020000);
  // rows beyond M are dropped by the range check of the store descriptor
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)min((long)p.M * p.ldc * 2, 0x7fffffffL), 0x00020000);
  const int lda32 = (int)p.lda, ldb32 = (int)p.ldb, ldc32 = (int)p.ldc;
  int voff[8];
  auto set_voff = [&](int m0, int n0) {
    int l = lane;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int ci = (q & 3) * 8 + wave;
      const int row = ci * 8 + (l >> 3);
      const int c = (l & 7) ^ kswz<MF>(row);
      if (q < 4) voff[q] = (min(m0 + row, p.M - 1) * lda32 + c * 8) * 2;
      else voff[q] = (min(n0 + row, p.N - 1) * ldb32 + c * 8) * 2;
    }
  };
  auto dma = [&](auto Q, unsigned char* stage_buf, int kt) {
    constexpr int q = decltype(Q)::value;
    const int ci = (q & 3) * 8 + wave;
    unsigned char* dst = stage_buf + (q >= 4 ? A_BYTES : 0) + ci * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(q >= 4 ? rsB : rsA, (__attribute__((address_space(3))) void*)dst, 16, voff[q], kt * 128, 0, 0);
  };

  const int rl = lane & 31, hl = lane >> 5;
  const int rowA = wm * WM + rl, rowB = wn * 64 + rl;
  const int offA = rowA * 128 + ((hl ^ kswz<MF>(rowA)) << 4);
  const int offB = A_BYTES + rowB * 128 + ((hl ^ kswz<MF>(rowB)) << 4);
  bf16x8 fa[2][TM], fb[2][TN];
  auto read_frags = [&](auto SET, const unsigned char* sb, int step) {
    constexpr int st = decltype(SET)::value;
    int oa = offA, ob = offB;
    asm volatile("" : "+v"(oa), "+v"(ob));                   // (the per-step variants are recomputed -- one v_xor each -- not hoisted and spilled)
    const unsigned char* pa = sb + (oa ^ (step << 5));
    const unsigned char* pb = sb + (ob ^ (step << 5));
#pragma unroll
    for (int j = 0; j < TN; j++) fb[st][j] = *reinterpret_cast<const bf16x8*>(pb + j * T * 128);
#pragma unroll
    for (int i = 0; i < TM; i++) fa[st][i] = *reinterpret_cast<const bf16x8*>(pa + i * T * 128);
  };
  acc_t acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
  };
  auto mfma_step = [&](auto SET) {
    constexpr int st = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[st][j], fa[st][i], acc[i][j], 0, 0, 0);
  };
  constexpr int NRD = TM + TN, NMF = TM * TN;
  zero_acc();

  // ---- the packed tile: pk[i][j][pair] = 16 bytes of row (i * 32 + rl): columns j * 32 + (2 * pair + hl) * 8 .. + 8 ----
  u32x4 pk[TM][TN][2];
  int st_m0 = 0, st_n0 = 0;                 // tile the packed registers belong to
  bool have_pk = false;
  f32x4 bj[TN * 4];
  int n0 = 0;
  auto load_bias = [&]() {
    int h = hl;
    asm volatile("" : "+v"(h));
#pragma unroll
    for (int g = 0; g < TN * 4; g++) {
      const int col = n0 + wn * 64 + (g >> 2) * T + (g & 3) * 8 + h * 4;
      if (p.bias == nullptr) bj[g] = f32x4{0, 0, 0, 0};
      else if (col + 4 <= p.N) bj[g] = *reinterpret_cast<const f32x4*>(p.bias + col);
      else {
#pragma unroll
        for (int r = 0; r < 4; r++) bj[g][r] = p.bias[min(col + r, p.N - 1)];
      }
    }
  };
  auto pack_row = [&](auto I) {
    constexpr int i = decltype(I)::value;
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int pr = 0; pr < 2; pr++) {
        uint32_t w[2][2];                  // [q0 | q1][dword]
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const int q = pr * 2 + s;
          const f32x4 bv = bj[j * 4 + q];
#pragma unroll
          for (int d = 0; d < 2; d++) {
            const uint32_t lo = f2bf(acc[i][j][q * 4 + d * 2] + bv[d * 2]), hi = f2bf(acc[i][j][q * 4 + d * 2 + 1] + bv[d * 2 + 1]);
            w[s][d] = lo | (hi << 16);
          }
        }
        // vdst = group q0, src = group q1 (guide T21): lanes 0-31 end up with [own q0 | upper's q0], lanes 32-63 with [lower's q1 | own q1]
        const auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
        pk[i][j][pr] = u32x4{r0[0], r1[0], r0[1], r1[1]};
      }
  };
  // store instruction s (0 .. 15) of the packed tile: tile row i = s / 4, then (j, pair)
  const int st_lane = rl * ldc32 * 2 + hl * 16;          // per-lane byte offset inside a tile row block
  auto store_piece = [&](auto SI) {
    constexpr int s = decltype(SI)::value, i = s >> 2, j = (s >> 1) & 1, pr = s & 1;
    const int row0 = st_m0 + wm * WM + i * T;              // (wave-uniform)
    const int col = st_n0 + wn * 64 + j * T + pr * 16;     // first column of the lane pair's 32 bytes
    if (col + 16 <= p.N) {
      int sl = st_lane;
      asm volatile("" : "+v"(sl));
      __builtin_amdgcn_raw_buffer_store_b128(pk[i][j][pr], rsC, sl, (row0 * ldc32 + col) * 2, 0);
    } else {                                               // ragged right edge: element stores of the valid columns
      int l = lane;
      asm volatile("" : "+v"(l));                          // (cold path: nothing of it is hoisted to the kernel entry and kept alive)
      const int c0 = col + (l >> 5) * 8, row = row0 + (l & 31);
      if (row < p.M) {
        bf16_t* dst = reinterpret_cast<bf16_t*>(p.C) + (size_t)row * p.ldc;
#pragma unroll
        for (int e = 0; e < 8; e++)
          if (c0 + e < p.N) dst[c0 + e] = (bf16_t)((pk[i][j][pr][e >> 1] >> ((e & 1) * 16)) & 0xffffu);
      }
    }
  };

  int w = w_begin + slot;
  int m0 = 0;
  int buf = 0;
  if (w < w_end) {
    item(w, m0, n0);
    set_voff(m0, n0);
    static_for<8>([&](auto Q) { dma(Q, lds, 0); });
    static_for<8>([&](auto Q) { dma(Q, lds + STAGE, 1); });
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(std::integral_constant<int, 0>{}, lds, 0);
  }
  constexpr int SPS = NDR > 0 ? DI * 4 / NDR : 0;           // drain stores per drain stage
  for (; w < w_end; w += nxw) {
    int m1 = 0, n1 = 0;
    const bool have_next = w + nxw < w_end;
    if (have_next) item(w + nxw, m1, n1);
    // one K stage; DR >= 0: drain step DR of the previous tile's packed rows rides between the MFMAs of the middle k-steps
    auto stage_body = [&](const int kt, auto DRS) {
      constexpr int dr = decltype(DRS)::value;
      unsigned char* sb = lds + buf * STAGE;
      unsigned char* nb = lds + (buf ^ 1) * STAGE;
      if (kt == nkt - 2 && have_next) set_voff(m1, n1);
      const bool last = kt + 1 == nkt;
      if (last) load_bias();
      static_for<NSTEP - 1>([&](auto J) {
        constexpr int j = decltype(J)::value;
        read_frags(std::integral_constant<int, (j + 1) & 1>{}, sb, j + 1);
        mfma_step(std::integral_constant<int, j & 1>{});
        static_for<NRD>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
        if constexpr (NMF > NRD) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
        if constexpr (dr >= 0 && j < SPS) {                 // (one store per k-step; SPS <= 3)
          if (have_pk) store_piece(std::integral_constant<int, (TM - DI) * 4 + dr * SPS + j>{});
        }
      });
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const bool own2 = kt + 2 < nkt;
      const int kt2 = own2 ? kt + 2 : kt + 2 - nkt;          // (the next tile's stage 0 or 1)
      const bool nx1 = !last || have_next;
      const bool nx2 = (own2 || have_next) && !(last && !have_next);
      if (nx1) read_frags(std::integral_constant<int, 0>{}, nb, 0);
      static_for<TM>([&](auto I) {
        constexpr int i = decltype(I)::value;
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
        if (nx2) static_for<2>([&](auto U) { dma(std::integral_constant<int, i * 2 + decltype(U)::value>{}, sb, kt2); });
      });
      if constexpr (dr >= 0 && SPS > 3) {
        if (have_pk) store_piece(std::integral_constant<int, (TM - DI) * 4 + dr * SPS + 3>{});
      }
      buf ^= 1;
    };
    static_for<NDR>([&](auto U) { stage_body((int)decltype(U)::value, U); });
    for (int kt = NDR; kt < nkt; kt++) stage_body(kt, std::integral_constant<int, -1>{});
    // ---- tile end: pack (+ bias), store the first TM - DI tile rows now, keep the rest for the next tile's first stages ----
    // (row by row, in this order: a packed row's accumulators are dead before the next row is packed)
    st_m0 = m0; st_n0 = n0; have_pk = true;
    static_for<TM>([&](auto I) {
      constexpr int i = decltype(I)::value;
      pack_row(I);
      if constexpr (i < TM - DI) static_for<4>([&](auto S4) { store_piece(std::integral_constant<int, i * 4 + decltype(S4)::value>{}); });
      __builtin_amdgcn_sched_barrier(0);
    });
    zero_acc();
    m0 = m1; n0 = n1;
  }
  if (have_pk) static_for<DI * 4>([&](auto SI) { store_piece(std::integral_constant<int, (TM - DI) * 4 + decltype(SI)::value>{}); });
}

template <int DI, int NDR, int VAR> static int g32d_launch(const G256P& p, hipStream_t st) {
  static vct::DynLdsOptIn optin;
  if (hipError_t e = optin.ensure((const void*)g32d_kernel<DI, NDR, VAR>, G256_LDS); e != hipSuccess) return (int)e;
  vct::launch(g32d_kernel<DI, NDR, VAR>, dim3(persistent_grid(st)), dim3(512), (size_t)G256_LDS, st, p);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

}  // namespace vct
