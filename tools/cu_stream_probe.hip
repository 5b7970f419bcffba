// Per-CU stream rate from BEYOND the L2 (Infinity Cache / HBM) as a function of how many workgroups stream at once (dev tool):
// `wgs` workgroups of 512 threads each read a private `span`-byte slice (16-byte loads, 16 in flight per wave); between two timed
// launches a flush kernel streams 96 MB of other data through every L2.  Batch-1 decode blocks run 8-32 workgroups: is a block
// bound by the per-CU rate or by the chip's?
// build: hipcc --offload-arch=gfx950 -O3 -w tools/cu_stream_probe.hip -o tools/bin/cu_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ __launch_bounds__(512) void stream_kernel(const unsigned char* __restrict__ buf, long span, unsigned* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned char* wb = buf + (long)blockIdx.x * span + wave * (span / 8);
  const int iters = (int)(span / 8 / 1024);
  u32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; i += 16) {
    u32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = *reinterpret_cast<const u32x4*>(wb + (long)min(i + u, iters - 1) * 1024 + lane * 16);
#pragma unroll
    for (int u = 0; u < 16; u++) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}
int main() {
  unsigned char* buf; unsigned* sink;
  const long big = 512L << 20;
  hipMalloc(&buf, big); hipMalloc(&sink, 4);
  hipMemset(buf, 1, big);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned char* flush = buf + (256L << 20);
  for (long span : {128L << 10, 256L << 10, 1024L << 10}) {
    for (int wgs : {1, 8, 16, 32, 64, 128, 256}) {
      float best = 1e9f, fl = 1e9f;
      for (int rep = 0; rep < 6; rep++) {
        stream_kernel<<<256, 512>>>(flush, 384L << 10, sink);     // 96 MB through the L2s
        hipDeviceSynchronize();
        hipEventRecord(e0);
        stream_kernel<<<wgs, 512>>>(buf, span, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double bytes = (double)wgs * span;
      printf("span %5ld KB  wgs %3d: %7.1f us  %6.2f TB/s  %6.1f GB/s per workgroup  %5.1f B/clk/CU\n", span >> 10, wgs, best * 1e3,
             bytes / (best * 1e-3) / 1e12, (double)span / (best * 1e-3) / 1e9, (double)span / (best * 1e-3 * 2.4e9));
    }
  }
  return 0;
}
