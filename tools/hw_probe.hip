// Hardware probe (dev tool, not product): confirms gfx950 MFMA fragment layouts and
// ds_read_b64_tr_b16 semantics that csrc/ kernels rely on. Run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/hw_probe.hip -o /tmp/hw_probe && /tmp/hw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k_tr(const int* addr_elems, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(&lds[addr_elems[l]]));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = v[j];
}

// A[16][32], B[32][16] (B given as Bt[n][k]) -> C[16][16], bf16 16x16x32
__global__ void k_mfma_bf16(const float* A, const float* Bt, float* C) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; j++) {
    a[j] = (__bf16)A[(l & 15) * 32 + (l >> 4) * 8 + j];
    b[j] = (__bf16)Bt[(l & 15) * 32 + (l >> 4) * 8 + j];
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// A[16][4], Bt[16][4] -> C[16][16], f32 16x16x4
__global__ void k_mfma_f32(const float* A, const float* Bt, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 15) * 4 + (l >> 4)];
  float b = Bt[(l & 15) * 4 + (l >> 4)];
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// A[32][16], Bt[32][16] -> C[32][32], bf16 32x32x16
__global__ void k_mfma_bf16_32(const float* A, const float* Bt, float* C) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; j++) {
    a[j] = (__bf16)A[(l & 31) * 16 + (l >> 5) * 8 + j];
    b[j] = (__bf16)Bt[(l & 31) * 16 + (l >> 5) * 8 + j];
  }
  f32x16 c;
  for (int i = 0; i < 16; i++) c[i] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    C[row * 32 + (l & 31)] = c[r];
  }
}

static int check(const char* name, const float* A, const float* Bt, const float* C, int M, int N, int K) {
  int bad = 0;
  for (int i = 0; i < M; i++)
    for (int j = 0; j < N; j++) {
      float s = 0;
      for (int k = 0; k < K; k++) s += A[i * K + k] * Bt[j * K + k];
      if (fabsf(s - C[i * N + j]) > 1e-3f) bad++;
    }
  printf("%s: %s (%d mismatches)\n", name, bad ? "FAIL" : "OK", bad);
  return bad;
}

int main() {
  int* d_addr; short* d_out;
  hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 256 * 2);
  int addr[64]; short out[256];
  for (int mode = 0; mode < 3; mode++) {
    for (int l = 0; l < 64; l++) {
      int i = l & 15, g = l >> 4;
      if (mode == 0) addr[l] = l * 4;                                   // lane-linear
      else if (mode == 1) addr[l] = g * 256 + (i >> 2) * 64 + (i & 3) * 4; // 4 rows of 64-elem stride, 16 cols
      else addr[l] = g * 4 * 40 + (i >> 2) * 40 + (i & 3) * 4;             // row stride 40 elems
    }
    hipMemcpy(d_addr, addr, sizeof(addr), hipMemcpyHostToDevice);
    k_tr<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
    printf("tr16_b64 mode %d: lane: got[0..3]\n", mode);
    for (int l = 0; l < 64; l++) {
      printf(" L%02d(a=%4d):%4d %4d %4d %4d%s", l, addr[l], out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    // hypothesis: within a 16-lane group, lane i gets element (i&3) of the 8B chunk supplied by lane (j*4 + (i>>2)), j=0..3
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) {
      int g = l >> 4, i = l & 15;
      int src = g * 16 + j * 4 + (i >> 2);
      int expect = addr[src] + (i & 3);
      if (out[l * 4 + j] != expect) bad++;
    }
    printf("tr16 hypothesis H1 (lane i elem j <- lane (4j + i>>2) elem (i&3)): %s (%d)\n", bad ? "FAIL" : "OK", bad);
  }
  float *dA, *dB, *dC;
  hipMalloc(&dA, 4096 * 4); hipMalloc(&dB, 4096 * 4); hipMalloc(&dC, 4096 * 4);
  float A[1024], Bt[1024], C[1024];
  srand(1);
  for (int i = 0; i < 1024; i++) { A[i] = (float)(rand() % 7 - 3); Bt[i] = (float)(rand() % 5 - 2); }
  hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, Bt, sizeof(Bt), hipMemcpyHostToDevice);
  k_mfma_bf16<<<1, 64>>>(dA, dB, dC); hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
  check("mfma_f32_16x16x32_bf16 layout", A, Bt, C, 16, 16, 32);
  k_mfma_f32<<<1, 64>>>(dA, dB, dC); hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
  check("mfma_f32_16x16x4f32 layout", A, Bt, C, 16, 16, 4);
  k_mfma_bf16_32<<<1, 64>>>(dA, dB, dC); hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
  check("mfma_f32_32x32x16_bf16 layout", A, Bt, C, 32, 32, 16);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device: %s CUs=%d clock=%d MHz L2=%d MB sharedPerBlock=%zu\n", p.name, p.multiProcessorCount, p.clockRate / 1000, p.l2CacheSize >> 20, p.sharedMemPerBlock);
  return 0;
}
