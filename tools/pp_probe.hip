// K-stage schedules of a 128 x 128 x 64 bf16 stage on one CU, without a GEMM around them (dev tool): every workgroup (512 threads,
// 2 x 32 KB LDS stages, two per CU) streams 32 KB per stage from an L2-resident window by LDS-DMA, reads twelve 1-KiB fragments
// per wave and issues 16 MFMAs per wave.  mode 0: the persistent-tile kernel's schedule (vmcnt(0), barrier, every wave reads then
// multiplies).  mode 1: ping-pong -- waves 0-3 multiply while waves 4-7 read and vice versa, two barriers per stage.
// mode 2: like 0 without the DMA, mode 3: like 1 without the DMA.
// build: hipcc --offload-arch=gfx950 -O3 -w tools/pp_probe.hip -o tools/bin/pp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int STAGE = 32768;

// row-strided source like a K-contiguous GEMM operand: a 1-KiB piece = 8 rows x 128 B at leading dimension ldb bytes (the k0 column
// block `kb` of the stage selects the 128-byte segment of each row)
template <bool SWZ>
__device__ __forceinline__ void dma4_rows(unsigned char* lds_stage, const unsigned char* A, const unsigned char* B, long ldb, int kb,
                                          int wave, int lane) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int ci = (q & 1) * 8 + wave;                   // 16 pieces of 8 rows = 128 rows per operand
    const int row = ci * 8 + (lane >> 3);
    const unsigned char* s = (q < 2 ? A : B) + (long)row * ldb + (long)kb * 128 + (((lane & 7) ^ (SWZ ? (row & 7) : 0))) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                     (__attribute__((address_space(3))) void*)(lds_stage + (q * 8 + wave) * 1024), 16, 0, 0);
  }
}
// pieces 0-1 (16 KB, the "A" half) from srcA, pieces 2-3 (the "B" half) from srcB
__device__ __forceinline__ void dma4(unsigned char* lds_stage, const unsigned char* srcA, const unsigned char* srcB, int wave, int lane) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int ci = q * 8 + wave;
    const unsigned char* s = q < 2 ? srcA + ci * 1024 : srcB + ci * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + lane * 16),
                                     (__attribute__((address_space(3))) void*)(lds_stage + ci * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ void read12(const unsigned char* st, int wave, int lane, bf16x8 (&f)[12]) {
  // conflict-free: every 16-lane group reads 16 consecutive 16-byte chunks
#pragma unroll
  for (int q = 0; q < 12; q++) f[q] = *reinterpret_cast<const bf16x8*>(st + ((q * 8 + wave) & 31) * 1024 + lane * 16);
}
// the GEMM kernels' own fragment addressing: 128-byte rows, 16-byte chunk c of row r at c ^ (r & 7); wave (wm, wn) of a 2 x 4 grid
// reads 4 A row tiles of its 64 rows and 2 B row tiles of its 32 rows per 32-deep k-step
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ void read12_real(const unsigned char* st, int wave, int lane, bf16x8 (&f)[12]) {
  const int wm = wave >> 2, wn = wave & 3, i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
#pragma unroll
    for (int t = 0; t < 4; t++) f[ks * 6 + t] = *reinterpret_cast<const bf16x8*>(st + kc_off(wm * 64 + t * 16 + i, ks * 4 + g));
#pragma unroll
    for (int t = 0; t < 2; t++) f[ks * 6 + 4 + t] = *reinterpret_cast<const bf16x8*>(st + 16384 + kc_off(wn * 32 + t * 16 + i, ks * 4 + g));
  }
}
__device__ __forceinline__ void mfma16(const bf16x8 (&f)[12], f32x4 (&acc)[8]) {
#pragma unroll
  for (int ks = 0; ks < 2; ks++)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[ks * 6 + 4 + j], f[ks * 6 + i], acc[i * 2 + j], 0, 0, 0);
}

template <int MODE, bool STREAM>
__global__ __launch_bounds__(512, 4) void pp_kernel(const unsigned char* __restrict__ src, const unsigned char* __restrict__ big, float* out, int stages) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const unsigned char* win = src + (size_t)(blockIdx.x & 7) * (8 * STAGE);    // one 256 KB window per XCD, shared by its workgroups: L2-resident
  constexpr bool DMA = MODE < 2, PP = (MODE & 1) != 0;
  // STREAM: the A half of every stage comes from a region nobody has touched (beyond the L2: 1 GB ring), shared by the 8 neighbouring
  // workgroups of the XCD like an A row panel shared by 8 N tiles; the B half stays in the XCD's resident window
  const size_t panel = ((size_t)(blockIdx.x & 7) * 64 + (blockIdx.x >> 6)) * 4096;       // stages of 16 KB per panel
  auto a_of = [&](int st) { return STREAM ? big + ((panel + (size_t)st) * 16384) % ((size_t)1 << 30) : win + (st & 7) * STAGE; };
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 f[12];
  if constexpr (!PP) {
    dma4(lds, a_of(0), win, wave, lane);
    for (int s = 0; s < stages; s++) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (DMA) dma4(lds + ((s + 1) & 1) * STAGE, a_of(s + 1), win + ((s + 1) & 7) * STAGE, wave, lane);
      read12(lds + (s & 1) * STAGE, wave, lane, f);
      mfma16(f, acc);
    }
  } else {
    dma4(lds, a_of(0), win, wave, lane);
    dma4(lds + STAGE, a_of(1), win + STAGE, wave, lane);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp == 0) read12(lds, wave, lane, f);
    for (int s = 0; s < stages; s++) {
      // phase A: group 0 multiplies stage s, group 1 reads stage s
      if (grp == 0) mfma16(f, acc); else read12(lds + (s & 1) * STAGE, wave, lane, f);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stage s + 1 has landed (this wave's share)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // phase B: stage s + 2 goes into the buffer of stage s; group 0 reads stage s + 1, group 1 multiplies stage s
      if (DMA) dma4(lds + (s & 1) * STAGE, a_of(s + 2), win + ((s + 2) & 7) * STAGE, wave, lane);
      if (grp == 0) read12(lds + ((s + 1) & 1) * STAGE, wave, lane, f); else mfma16(f, acc);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  float sum = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (sum == 123.456f) out[blockIdx.x * 512 + tid] = sum;
}

template <int EPI, bool REAL>
__global__ __launch_bounds__(512, 4) void rows_kernel(const unsigned char* __restrict__ src, float* out, int stages, long ldb, int nkb, unsigned char* cout) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // per XCD: 8 "A panels" and 8 "B panels" of 128 rows x nkb*128 bytes; workgroup slot -> (panel a, panel b)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const unsigned char* A = src + ((size_t)xcd * 16 + (slot & 7)) * 128 * ldb;
  const unsigned char* B = src + ((size_t)xcd * 16 + 8 + ((slot >> 3) & 7)) * 128 * ldb;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 f[12];
  dma4_rows<REAL>(lds, A, B, ldb, 0, wave, lane);
  for (int s = 0; s < stages; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    dma4_rows<REAL>(lds + ((s + 1) & 1) * STAGE, A, B, ldb, (s + 1) % nkb, wave, lane);
    if (REAL) read12_real(lds + (s & 1) * STAGE, wave, lane, f); else read12(lds + (s & 1) * STAGE, wave, lane, f);
    mfma16(f, acc);
    if (EPI && (s & 7) == 7) {
      // a tile is done: 128 x 128 bf16 = 32 KB through the just-consumed stage as a row slab, then whole-row 16-byte stores to a
      // fresh place in global memory (EPI 2: no global stores, slab only)
      unsigned char* slab = lds + (s & 1) * STAGE;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < 8; i++) {
        typedef __attribute__((ext_vector_type(2))) unsigned int u2;
        const u2 v = {__float_as_uint(acc[i][0]) >> 16 | (__float_as_uint(acc[i][1]) & 0xffff0000u),
                      __float_as_uint(acc[i][2]) >> 16 | (__float_as_uint(acc[i][3]) & 0xffff0000u)};
        *reinterpret_cast<u2*>(slab + ((i * 8 + wave) * 64 + lane) * 8) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      unsigned char* dst = cout + ((size_t)blockIdx.x * 64 + ((s >> 3) & 63)) * 32768;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u4;
        const u4 v = *reinterpret_cast<const u4*>(slab + (q * 512 + tid) * 16);
        if (EPI == 1) *reinterpret_cast<u4*>(dst + (q * 512 + tid) * 16) = v;
        else if (v[0] == 0x12345678u) *reinterpret_cast<u4*>(dst) = v;
      }
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = f32x4{0, 0, 0, 0};
    }
  }
  float sum = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (sum == 123.456f) out[blockIdx.x * 512 + tid] = sum;
}
template <int EPI, bool REAL> static void run_rows(const unsigned char* src, float* out, int stages, long ldb, int nkb, const char* name, unsigned char* cout) {
  hipFuncSetAttribute((const void*)rows_kernel<EPI, REAL>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {256, 512}) {
    rows_kernel<EPI, REAL><<<grid, 512, 2 * STAGE>>>(src, out, 64, ldb, nkb, cout);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      rows_kernel<EPI, REAL><<<grid, 512, 2 * STAGE>>>(src, out, stages, ldb, nkb, cout);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double flop = (double)grid * stages * 8 * 16 * 16384.0;
    printf("%-44s %d workgroups per CU: %7.1f us, %5.0f clk per stage, %6.0f TFLOP/s equivalent\n", name, grid / 256, best * 1e3,
           best * 1e-3 * 2.4e9 / stages, flop / (best * 1e-3) / 1e12);
  }
}
template <int MODE, bool STREAM> static void run(const unsigned char* src, const unsigned char* big, float* out, int stages, const char* name) {
  hipFuncSetAttribute((const void*)pp_kernel<MODE, STREAM>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {256, 512}) {
    pp_kernel<MODE, STREAM><<<grid, 512, 2 * STAGE>>>(src, big, out, 64);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      pp_kernel<MODE, STREAM><<<grid, 512, 2 * STAGE>>>(src, big, out, stages);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double flop = (double)grid * stages * 8 * 16 * 16384.0;
    printf("%-34s %d workgroups per CU: %7.1f us, %5.0f clk per stage, %6.0f TFLOP/s equivalent\n", name, grid / 256, best * 1e3,
           best * 1e-3 * 2.4e9 / stages, flop / (best * 1e-3) / 1e12);
  }
}
int main() {
  unsigned char* src; unsigned char* big; float* out;
  hipMalloc(&src, 512L * 8 * STAGE); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&big, (size_t)1 << 30);
  hipMemset(src, 0x3c, 512L * 8 * STAGE); hipMemset(big, 0x3c, (size_t)1 << 30);
  const int stages = getenv("STAGES") ? atoi(getenv("STAGES")) : 2000;
  run<0, false>(src, big, out, stages, "all read, all multiply (+ DMA, L2)");
  run<0, true>(src, big, out, stages, "all read, all multiply (+ DMA, A from HBM)");
  run<1, false>(src, big, out, stages, "ping-pong halves (+ DMA, L2)");
  run<1, true>(src, big, out, stages, "ping-pong halves (+ DMA, A from HBM)");
  run<2, false>(src, big, out, stages, "all read, all multiply (no DMA)");
  // row-strided operands: 128 panels x 128 rows; K = 512 (ld 1024 B, 8 column blocks), the same padded by 128 / 64 B, K = 2048
  unsigned char* cout; hipMalloc(&cout, 512L * 64 * 32768);       // 1 GB of tile outputs
  run_rows<0, false>(big, out, stages, 1024, 8, "rows, K = 512, no epilogue", cout);
  run_rows<2, false>(big, out, stages, 1024, 8, "rows, K = 512, slab epilogue, no stores", cout);
  run_rows<1, false>(big, out, stages, 1024, 8, "rows, K = 512, slab epilogue + 32 KB stores", cout);
  run_rows<0, true>(big, out, stages, 1024, 8, "GEMM fragment addressing, no epilogue", cout);
  run_rows<1, true>(big, out, stages, 1024, 8, "GEMM fragment addressing, epilogue + stores", cout);
  return 0;
}
