// L2 -> CU operand-stream probe (dev tool): how many bytes per clock a CU can pull from an L2-resident buffer, by load flavour.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction) into an LDS ring, DEPTH instructions in flight per wave
//   mode 1: global_load_dwordx4 into registers (16 B per lane), DEPTH loads in flight per wave
// Every workgroup streams the SAME `span` bytes (a weight matrix all CUs read: L2-resident after the first touch) `reps` times, or
// (shared = 0) its own slice of a large buffer (activations: L2 misses served by the Infinity Cache / HBM).
// build: hipcc --offload-arch=gfx950 -O3 tools/l2_stream_probe.hip -o tools/bin/l2_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE, int DEPTH, int NT>
__global__ __launch_bounds__(NT) void stream_kernel(const unsigned char* __restrict__ buf, long span, int reps, int shared, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NT / 64;
  const unsigned char* base = buf + (shared ? 0 : (long)blockIdx.x * span);
  const long per_wave = span / NW;                       // bytes each wave streams per repetition
  const unsigned char* wb = base + wave * per_wave;
  const int iters = (int)(per_wave / 1024);              // 1 KiB wave-instructions
  u32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; r++) {
    for (int i = 0; i < iters; i += DEPTH) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int u = 0; u < DEPTH; u++)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + (long)(i + u) * 1024 + lane * 16),
                                           (__attribute__((address_space(3))) void*)(lds + (wave * DEPTH + u) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        u32x4 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; u++) v[u] = *reinterpret_cast<const u32x4*>(wb + (long)(i + u) * 1024 + lane * 16);
#pragma unroll
        for (int u = 0; u < DEPTH; u++) acc ^= v[u];
      }
    }
  }
  if (MODE == 0) acc[0] = *reinterpret_cast<unsigned*>(lds + tid * 4);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int DEPTH, int NT>
static void run(const unsigned char* buf, long span, int reps, int shared, int wgs, unsigned* sink, hipEvent_t e0, hipEvent_t e1) {
  const size_t ldsb = MODE == 0 ? (size_t)(NT / 64) * DEPTH * 1024 : 0;
  hipFuncSetAttribute((const void*)stream_kernel<MODE, DEPTH, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  for (int i = 0; i < 2; i++) stream_kernel<MODE, DEPTH, NT><<<wgs, NT, ldsb>>>(buf, span, reps, shared, sink);
  hipDeviceSynchronize(); hipEventRecord(e0);
  for (int i = 0; i < 5; i++) stream_kernel<MODE, DEPTH, NT><<<wgs, NT, ldsb>>>(buf, span, reps, shared, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)wgs * span * reps;
  printf("%s depth %2d threads %4d wgs %4d %s span %7ld KB: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU\n", MODE ? "regs  " : "ldsdma", DEPTH, NT, wgs,
         shared ? "shared " : "private", span >> 10, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / 256.0 / (ms * 1e-3 * 2.4e9));
}

int main() {
  unsigned char* buf; unsigned* sink;
  const long big = 256L << 20;
  hipMalloc(&buf, big); hipMalloc(&sink, 4);
  hipMemset(buf, 1, big);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // a weight matrix every CU reads (512 KB / 2 MB), 16 passes
  for (long span : {512L << 10, 2048L << 10}) {
    for (int wgs : {256, 512}) {
      run<0, 4, 512>(buf, span, 16, 1, wgs, sink, e0, e1);
      run<0, 8, 512>(buf, span, 16, 1, wgs, sink, e0, e1);
      run<0, 16, 512>(buf, span, 16, 1, wgs, sink, e0, e1);
      run<0, 8, 256>(buf, span, 16, 1, wgs, sink, e0, e1);
      run<1, 4, 512>(buf, span, 16, 1, wgs, sink, e0, e1);
      run<1, 8, 512>(buf, span, 16, 1, wgs, sink, e0, e1);
      run<1, 16, 512>(buf, span, 16, 1, wgs, sink, e0, e1);
      run<1, 8, 256>(buf, span, 16, 1, wgs, sink, e0, e1);
    }
  }
  // private slices of a 128 MB buffer (activation-like: 512 KB per workgroup), one pass
  for (int wgs : {256}) {
    run<0, 8, 512>(buf, 512L << 10, 1, 0, wgs, sink, e0, e1);
    run<0, 16, 512>(buf, 512L << 10, 1, 0, wgs, sink, e0, e1);
    run<1, 8, 512>(buf, 512L << 10, 1, 0, wgs, sink, e0, e1);
    run<1, 16, 512>(buf, 512L << 10, 1, 0, wgs, sink, e0, e1);
  }
  return 0;
}
