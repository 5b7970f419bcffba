// Store-rate probe (dev tool): one 512-thread workgroup per CU, each storing `bytes_per_wg` as 512-byte row segments of a
// [rows][30528] bf16 matrix (the epilogue pattern of the 256x256 GEMM), with only the first `active` workgroups storing.
// build: hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ __launch_bounds__(512) void store_kernel(unsigned short* out, long ld, int tiles_per_wg, int active, int nt) {
  if ((int)blockIdx.x >= active) return;
  const int tid = threadIdx.x;
  const u32x4 v = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, (unsigned)tid};
  for (int t = 0; t < tiles_per_wg; t++) {
    const int tile = blockIdx.x * tiles_per_wg + t;          // tile = 256 rows x 256 cols
    const int m0 = (tile % 19) * 256, n0 = (tile / 19) * 256;
    for (int q = 0; q < 16; q++) {                           // 16 x 512 threads x 16 B = 128 KB
      const int cid = q * 512 + tid, row = cid >> 5, c = cid & 31;
      u32x4* dst = reinterpret_cast<u32x4*>(out + (size_t)(m0 + row) * ld + n0 + c * 8);
      if (nt) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
  }
}
int main() {
  const long ld = 30528, rows = 4864;
  unsigned short* out; hipMalloc(&out, rows * ld * 2 + (1 << 20));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nt = 0; nt < 2; nt++)
    for (int active : {256, 128, 64, 32, 8}) {
      const int tiles = 8;
      for (int i = 0; i < 3; i++) store_kernel<<<256, 512>>>(out, ld, tiles, active, nt);
      hipDeviceSynchronize(); hipEventRecord(e0);
      for (int i = 0; i < 10; i++) store_kernel<<<256, 512>>>(out, ld, tiles, active, nt);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
      const double bytes = (double)active * tiles * 131072.0;
      printf("nt=%d active=%3d: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", nt, active, ms * 1e3, bytes / (ms * 1e-3) / 1e12,
             bytes / active / (ms * 1e-3 * 2.4e9));
    }
  return 0;
}
