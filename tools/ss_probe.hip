// Sample-stationary layer probe (dev tool, round 4).  One 512-thread workgroup per SAMPLE keeps that sample's rows (<= 32) in LDS for a
// whole Transformer layer and streams the layer's bf16 weights from L2 straight into MFMA fragments -- no LDS staging of weights, no
// inter-workgroup dependency.  Question: how close does the weight stream get to the 54-57 B/clk/CU of tools/l2_stream_probe.hip once
// MFMAs, A-fragment reads, epilogues and product boundaries are in the loop, and with a stream (8.4 MB per decoder layer) that does
// not fit one XCD's 4 MB L2?
//   LAYOUT 0: W [N, K] row-major, lane (li, lg) loads 16 B at row n0 + li, k = s*32 + lg*8           (64-B segments of 16 rows / instr)
//   LAYOUT 1: same memory, K consumed in permuted order: lane loads 32 contiguous bytes (k = lg*16 .. +16) as two instructions
//   LAYOUT 2: W pre-packed fragment-major ([N/16][K/32][64 lanes][8]): every wave instruction reads 1 KiB contiguous
// build: hipcc --offload-arch=gfx950 -O3 tools/ss_probe.hip -o tools/bin/ss_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef unsigned short bf16_t;

template <class F, int... Is> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

constexpr int MAXP = 8;
struct Prod { long w_off; int N, K; };
struct Layer { Prod p[MAXP]; int np; };
constexpr int NLAY = 4;
struct Net { Layer l[NLAY]; int nl; };

constexpr int ASTR = 2048 + 8;   // LDS row stride of the activation panel (bf16 elements)

// one wave's cursor over (layer, product, n-block of 64 columns, 64-deep K chunk)
struct Cur { int l, p, nb, c; };

template <int LAYOUT, int MT, bool DO_MFMA, bool DO_EPI, int NBUF>
__global__ __launch_bounds__(512, 2) void ss_kernel(const Net net, const bf16_t* __restrict__ wts, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* panel = reinterpret_cast<bf16_t*>(smem);                 // [32][ASTR]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 32 * ASTR / 2; i += 512) reinterpret_cast<uint32_t*>(panel)[i] = 0x3c003c00u + (uint32_t)(i & 7);
  __syncthreads();

  bf16x8 bq[NBUF][4][2];
  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[m][t] = f32x4{0, 0, 0, 0};

  auto valid = [&](const Cur& q) { return q.l < net.nl; };
  auto advance = [&](Cur q) {
    const Prod& pr = net.l[q.l].p[q.p];
    q.c++;
    if (q.c == pr.K / 64) { q.c = 0; q.nb++;
      if (q.nb == pr.N / 512) { q.nb = 0; q.p++;
        if (q.p == net.l[q.l].np) { q.p = 0; q.l++; } } }
    return q;
  };
  auto issue = [&](const Cur& q, auto SLOT) {
    constexpr int slot = decltype(SLOT)::value;
    const Prod& pr = net.l[q.l].p[q.p];
    const int n0 = (q.nb * 8 + wave) * 64;                         // this wave's 64 columns of the n-block
    const bf16_t* w = wts + pr.w_off;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const bf16_t* src;
        if constexpr (LAYOUT == 0) src = w + (long)(n0 + t * 16 + li) * pr.K + q.c * 64 + s * 32 + lg * 8;
        else if constexpr (LAYOUT == 1) src = w + (long)(n0 + t * 16 + li) * pr.K + q.c * 64 + lg * 16 + s * 8;
        else if constexpr (LAYOUT == 2) src = w + ((long)((n0 >> 4) + t) * (pr.K / 32) + q.c * 2 + s) * 512 + lane * 8;
        else src = w + ((((long)q.nb * (pr.K / 64) + q.c) * 8 + wave) * 8 + t * 2 + s) * 512 + lane * 8;   // stream order: the workgroup reads 64 KB contiguous per chunk step
        bq[slot][t][s] = *reinterpret_cast<const bf16x8*>(src);
      }
  };
  auto compute = [&](const Cur& q, auto SLOT) {
    constexpr int slot = decltype(SLOT)::value;
    if constexpr (DO_MFMA) {
      bf16x8 af[MT][2];
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const int k = LAYOUT == 1 ? q.c * 64 + lg * 16 + s * 8 : q.c * 64 + s * 32 + lg * 8;
          af[m][s] = *reinterpret_cast<const bf16x8*>(panel + (m * 16 + li) * ASTR + k);
        }
#pragma unroll
      for (int s = 0; s < 2; s++)
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
          for (int t = 0; t < 4; t++) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[slot][t][s], af[m][s], acc[m][t], 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const f32x4 v = __builtin_bit_cast(f32x4, bq[slot][t][s]);
          acc[0][t] += v;
        }
    }
  };
  auto tail = [&](const Cur& q) {
    const Prod& pr = net.l[q.l].p[q.p];
    if (q.c + 1 != pr.K / 64) return;
    // end of this wave's 64-column block: the accumulators hold out^T (lane: 4 consecutive columns of sample row li)
    const bool last_nb = (q.nb + 1 == pr.N / 512);
    if constexpr (DO_EPI) {
      if (last_nb) __syncthreads();                               // everyone is done READING the panel as this product's A
      const int n0 = ((q.nb * 8 + wave) * 64) & 2047;
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = (__bf16)(acc[m][t][r] * 1e-3f);
          if (last_nb) *reinterpret_cast<bf16x4*>(panel + (m * 16 + li) * ASTR + n0 + t * 16 + lg * 4) = o;
          else if (o[0] == (__bf16)123.0f) sink[1] = 1.0f;
          acc[m][t] = f32x4{0, 0, 0, 0};
        }
      if (last_nb) __syncthreads();                               // the next product's A is complete
    }
  };

  Cur cur{0, 0, 0, 0};
  Cur ahead = cur;                                                // the cursor NBUF - 1 chunks ahead
  sfor<NBUF - 1>([&](auto I) { if (valid(ahead)) { issue(ahead, I); ahead = advance(ahead); } });
  bool go = valid(cur);
  while (go) {
    sfor<NBUF>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if (!go) return;
      if (valid(ahead)) { issue(ahead, std::integral_constant<int, (i + NBUF - 1) % NBUF>{}); ahead = advance(ahead); }
      compute(cur, I);
      tail(cur);
      cur = advance(cur);
      go = valid(cur);
    });
  }
  float s = 0;
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int t = 0; t < 4; t++) s += acc[m][t][0] + acc[m][t][1] + acc[m][t][2] + acc[m][t][3];
  s += (float)panel[tid];
  if (s == 12345.678f) sink[0] = s;
}

static Layer dec_layer(long& off) {
  Layer L; L.np = 7;
  const int NK[7][2] = {{1536, 512}, {512, 512}, {512, 512}, {1024, 512}, {512, 512}, {2048, 512}, {512, 2048}};
  for (int i = 0; i < 7; i++) { L.p[i] = Prod{off, NK[i][0], NK[i][1]}; off += (long)NK[i][0] * NK[i][1]; }
  return L;
}
static Layer enc_layer(long& off) {
  Layer L; L.np = 4;
  const int NK[4][2] = {{1536, 512}, {512, 512}, {2048, 512}, {512, 2048}};
  for (int i = 0; i < 4; i++) { L.p[i] = Prod{off, NK[i][0], NK[i][1]}; off += (long)NK[i][0] * NK[i][1]; }
  return L;
}

template <int LAYOUT, int MT, bool DO_MFMA, bool DO_EPI, int NBUF = 2>
static void run(const char* what, const Net& net, long elems, const bf16_t* wts, float* sink, int wgs, hipEvent_t e0, hipEvent_t e1) {
  const size_t lds = (size_t)32 * ASTR * 2;
  hipFuncSetAttribute((const void*)ss_kernel<LAYOUT, MT, DO_MFMA, DO_EPI, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int i = 0; i < 2; i++) ss_kernel<LAYOUT, MT, DO_MFMA, DO_EPI, NBUF><<<wgs, 512, lds>>>(net, wts, sink);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
  hipEventRecord(e0);
  const int R = 5;
  for (int i = 0; i < R; i++) ss_kernel<LAYOUT, MT, DO_MFMA, DO_EPI, NBUF><<<wgs, 512, lds>>>(net, wts, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= R;
  const double bytes = (double)elems * 2 * wgs;
  const double flops = 2.0 * elems * 16 * MT * wgs;   // padded rows
  printf("%-34s layout %d nbuf %d MT %d mfma %d epi %d wgs %4d: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU  %7.1f TF(padded)\n", what, LAYOUT, NBUF, MT, (int)DO_MFMA,
         (int)DO_EPI, wgs, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / 256.0 / (ms * 1e-3 * 2.4e9) * (256.0 / (wgs < 256 ? wgs : 256)), flops / (ms * 1e-3) / 1e12);
}

int main() {
  bf16_t* wts; float* sink;
  const long cap = 64L << 20;   // elements
  hipMalloc(&wts, cap * 2); hipMalloc(&sink, 16);
  {
    bf16_t* h = (bf16_t*)malloc(cap * 2);
    unsigned s = 12345;
    for (long i = 0; i < cap; i++) { s = s * 1664525u + 1013904223u; h[i] = (bf16_t)(0x3c00 + ((s >> 20) & 0x7f) + ((s >> 31) << 15)); }
    hipMemcpy(wts, h, cap * 2, hipMemcpyHostToDevice); free(h);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // one decoder layer (8.4 MB), two decoder layers, the whole stack (2 enc + 2 dec = 29.4 MB)
  Net one; { long off = 0; one.nl = 1; one.l[0] = dec_layer(off); }
  long e_one = 0; for (int i = 0; i < 7; i++) e_one += (long)one.l[0].p[i].N * one.l[0].p[i].K;
  Net enc1; { long off = 0; enc1.nl = 1; enc1.l[0] = enc_layer(off); }
  long e_enc = 0; for (int i = 0; i < 4; i++) e_enc += (long)enc1.l[0].p[i].N * enc1.l[0].p[i].K;
  Net full; long e_full = 0; { long off = 0; full.nl = 4; full.l[0] = enc_layer(off); full.l[1] = enc_layer(off); full.l[2] = dec_layer(off); full.l[3] = dec_layer(off); e_full = off; }
  // a single small product repeated: L2-resident ceiling in this access pattern (512 x 512 = 0.5 MB)
  Net small; { small.nl = 4; for (int l = 0; l < 4; l++) { small.l[l].np = 8; for (int i = 0; i < 8; i++) small.l[l].p[i] = Prod{0, 512, 512}; } }
  const long e_small = 32L * 512 * 512;

#define RUNS(MT, MF, EP, what, net, el, wgs) \
  run<2, MT, MF, EP, 2>(what, net, el, wts, sink, wgs, e0, e1); run<3, MT, MF, EP, 2>(what, net, el, wts, sink, wgs, e0, e1); \
  run<3, MT, MF, EP, 3>(what, net, el, wts, sink, wgs, e0, e1); run<3, MT, MF, EP, 4>(what, net, el, wts, sink, wgs, e0, e1);
  RUNS(2, false, false, "small 0.5MB x32 loads only", small, e_small, 256)
  RUNS(2, true, false, "small 0.5MB x32 +mfma", small, e_small, 256)
  RUNS(2, false, false, "dec layer 8.4MB loads only", one, e_one, 256)
  RUNS(2, true, false, "dec layer +mfma", one, e_one, 256)
  RUNS(2, true, true, "dec layer +mfma +epi/barriers", one, e_one, 256)
  RUNS(1, true, true, "enc layer MT1 +mfma +epi", enc1, e_enc, 256)
  RUNS(2, true, true, "stack 2enc+2dec MT2 full", full, e_full, 256)
  run<0, 2, true, true, 2>("dec layer row-major (reference)", one, e_one, wts, sink, 256, e0, e1);
  return 0;
}
