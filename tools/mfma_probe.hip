// Sustained MFMA issue rate (dev tool): W waves per SIMD, each looping over 32 independent v_mfma_f32_16x16x32_bf16 accumulators.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC>
__global__ __launch_bounds__(512) void mfma_kernel(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {256, 512}) {
    for (int iters : {2000, 20000}) {
      mfma_kernel<32><<<256, threads>>>(out, 100);
      hipDeviceSynchronize(); hipEventRecord(e0);
      mfma_kernel<32><<<256, threads>>>(out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = 256.0 * (threads / 64) * iters * 32 * 16384.0;
      printf("waves/SIMD %d, %6d iterations: %8.1f us  %7.1f TFLOP/s\n", threads / 256, iters, ms * 1e3, flop / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
