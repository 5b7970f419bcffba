// Dev probe (round 6): the persistent 256x256 bf16 GEMM with a software-pipelined K loop, on BOTH MFMA opcodes, against the shipped
// gemm256_kernel on the same data in the same process.
//   * fragments of k-step j+1 are read from LDS while the MFMAs of k-step j issue (double-buffered fragment registers); the stage barrier
//     sits in front of the LAST k-step of a stage, so the first fragments of the next stage are in flight under that step's MFMAs;
//   * operand DMA = buffer_load_dwordx4 ... lds with the per-lane offset fixed per tile and the K offset in an SGPR;
//   * MF = 16: v_mfma_f32_16x16x32_bf16 (2 k-steps of 32 per 64-deep stage, 32 MFMAs each);  MF = 32: v_mfma_f32_32x32x16_bf16
//     (4 k-steps of 16, 8 MFMAs each: half the MFMA instructions, the same LDS bytes).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/g32_probe.hip -o tools/bin/g32_probe
#include "../video-captioning-transformer_amd/csrc/vct_gemm256.hip"
#include <vector>
#include <cstdlib>
#include <cstring>
#include <cmath>

namespace vct {
thread_local CmdList* g_rec = nullptr;
void rec_push(hipStream_t, std::function<void(hipStream_t)>&&) {}
void replay_note_error(int) {}
}  // namespace vct


#include "g32_kernel_v1.h"
#include "g32w4_kernel.h"
#include "g32r_kernel.h"
using namespace vct;

static uint16_t f2bf_host(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f_host(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

// the round-5 kernel, directly (g256_launch dispatches to the pipelined one by default)
template <int TA, int TB, typename TO> static void base_launch(const G256P& p, hipStream_t st) {
  static bool once = false;
  if (!once) { hipFuncSetAttribute((const void*)gemm256_kernel<TA, TB, TO>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS); once = true; }
  hipLaunchKernelGGL((gemm256_kernel<TA, TB, TO>), dim3(persistent_grid(st)), dim3(512), (size_t)G256_LDS, st, p);
}

struct Shape { const char* name; int ta, tb, M, N, K, split; };

template <typename TO> static double cmp(const std::vector<TO>& a, const std::vector<TO>& b, long rows, long cols, long ld, long* nbad) {
  double maxd = 0; *nbad = 0;
  for (long r = 0; r < rows; r++)
    for (long c = 0; c < cols; c++) {
      double x, y;
      if constexpr (sizeof(TO) == 2) { x = bf2f_host(a[r * ld + c]); y = bf2f_host(b[r * ld + c]); }
      else { x = a[r * ld + c]; y = b[r * ld + c]; }
      const double d = fabs(x - y);
      if (!(d <= 0.02 * (1.0 + fabs(x)))) (*nbad)++;
      if (d > maxd) maxd = d;
    }
  return maxd;
}

template <int TA, int TB, typename TO> static void run_shape(const Shape& s, hipStream_t st, int iters, int rounds) {
  const int M = s.M, N = s.N, K = s.K;
  const long Kp = (K + 7) / 8 * 8, Mp = (M + 7) / 8 * 8, Np = (N + 7) / 8 * 8;
  const long lda = TA ? Mp : Kp, ldb = TB ? Kp : Np, ldc = Np;
  const long arows = TA ? K : M, brows = TB ? N : K;
  std::vector<uint16_t> ha((size_t)arows * lda), hb((size_t)brows * ldb);
  uint32_t seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : ha) v = f2bf_host(rnd());                        // (padding columns hold random, finite values too)
  for (auto& v : hb) v = f2bf_host(rnd() * 0.05f);
  std::vector<float> hbias(N);
  for (auto& v : hbias) v = rnd();
  const int split = s.split;
  const size_t out_elems = split > 1 ? (size_t)split * M * N : (size_t)M * ldc;
  uint16_t *dA, *dB; TO *dC0, *dC1; float *dbias, *dbg0, *dbg1;
  hipMalloc(&dA, ha.size() * 2 + 4096); hipMalloc(&dB, hb.size() * 2 + 4096); hipMalloc(&dC0, out_elems * sizeof(TO)); hipMalloc(&dC1, out_elems * sizeof(TO));
  hipMalloc(&dbias, N * 4); hipMalloc(&dbg0, M * 4); hipMalloc(&dbg1, M * 4);
  hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice);
  G256P p; memset(&p, 0, sizeof(p));
  p.A = dA; p.B = dB; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
  const int nkt = (K + 63) / 64;
  p.kt_per_split = (nkt + split - 1) / split; p.split = (nkt + p.kt_per_split - 1) / p.kt_per_split;
  p.order = 1; p.zmajor = split > 1 ? 1 : 0;
  const bool nt_plain = (TA == 0 && TB == 1 && split == 1);
  p.bias = nt_plain ? dbias : nullptr;
  struct Var { const char* name; std::function<void(TO*, float*)> run; };
  std::vector<Var> vars;
  auto prep = [&](TO* c, float* bg) { G256P q = p; if (split > 1) q.partial = reinterpret_cast<float*>(c); else q.C = c; if (TA == 1 && TB == 0) q.bias_grad = bg; return q; };
  vars.push_back({"round-5 gemm256", [&](TO* c, float* bg) { base_launch<TA, TB, TO>(prep(c, bg), st); }});
  vars.push_back({"pipelined 32x32x16", [&](TO* c, float* bg) { g32_launch<TA, TB, TO, 0>(prep(c, bg), st); }});
  vars.push_back({"ring of 4 half-stages", [&](TO* c, float* bg) { g32r_launch<TA, TB, TO, 0>(prep(c, bg), st); }});
  vars.push_back({"ring + L2 prefetch 6", [&](TO* c, float* bg) { G256P q = prep(c, bg); q.pf_dist = 6; g32r_launch<TA, TB, TO, 0>(q, st); }});
  vars.push_back({"pipelined, 4 waves x 128x128", [&](TO* c, float* bg) { G256P q = prep(c, bg); q.pf_dist = (TA == 0 && TB == 0) ? 3 : 0; g32w4_launch<TA, TB, TO, 128>(q, st); }});
  vars.push_back({"pipelined, DMA in one k-step", [&](TO* c, float* bg) { g32_launch<TA, TB, TO, 2>(prep(c, bg), st); }});
  if constexpr (TA == 0 && TB == 1 && sizeof(TO) == 2)
    vars.push_back({"pipelined v1 (first probe)", [&](TO* c, float* bg) { g32v1_launch<0, 1, bf16_t, 32, 0>(prep(c, bg), st); }});
  else
  vars.push_back({"pipelined, prio", [&](TO* c, float* bg) { g32_launch<TA, TB, TO, 1>(prep(c, bg), st); }});
  if constexpr (!(TA == 0 && TB == 1)) {
    vars.push_back({"pipelined + L2 prefetch 2", [&](TO* c, float* bg) { G256P q = prep(c, bg); q.pf_dist = 2; g32_launch<TA, TB, TO, 0>(q, st); }});
    vars.push_back({"pipelined + L2 prefetch 3", [&](TO* c, float* bg) { G256P q = prep(c, bg); q.pf_dist = 3; g32_launch<TA, TB, TO, 0>(q, st); }});
    vars.push_back({"pipelined + L2 prefetch 4", [&](TO* c, float* bg) { G256P q = prep(c, bg); q.pf_dist = 4; g32_launch<TA, TB, TO, 0>(q, st); }});
    vars.push_back({"pipelined 8+0+0 + L2 prefetch 3", [&](TO* c, float* bg) { G256P q = prep(c, bg); q.pf_dist = 3; g32_launch<TA, TB, TO, 2>(q, st); }});
  }
  if constexpr (TA == 1) {
    vars.push_back({"r5, no bias grad", [&](TO* c, float* bg) { G256P q = prep(c, bg); q.bias_grad = nullptr; base_launch<TA, TB, TO>(q, st); }});
    vars.push_back({"pipelined, no bias grad", [&](TO* c, float* bg) { G256P q = prep(c, bg); q.bias_grad = nullptr; g32_launch<TA, TB, TO, 0>(q, st); }});
    vars.push_back({"pipelined, no bias grad, no epi no DMA", [&](TO* c, float* bg) { G256P q = prep(c, bg); q.bias_grad = nullptr; g32_launch<TA, TB, TO, 12>(q, st); }});
  }
  vars.push_back({"ablate r5: no epilogue", [&](TO* c, float* bg) { G256P q = prep(c, bg); q.dbg = 4; base_launch<TA, TB, TO>(q, st); }});
  vars.push_back({"ablate pipelined: no epilogue", [&](TO* c, float* bg) { g32_launch<TA, TB, TO, 4>(prep(c, bg), st); }});
  vars.push_back({"ablate r5: no epi no DMA", [&](TO* c, float* bg) { G256P q = prep(c, bg); q.dbg = 6; base_launch<TA, TB, TO>(q, st); }});
  vars.push_back({"ablate pipelined: no epi no DMA", [&](TO* c, float* bg) { g32_launch<TA, TB, TO, 12>(prep(c, bg), st); }});
  printf("[%s] eligible for the pipelined kernel: %d (split %d x %d stages)\n", s.name, (int)g32_eligible(prep(dC0, dbg0), TA == 1, TB == 0), p.split, p.kt_per_split);
  // ---- correctness: the first three against the round-5 kernel everywhere and a host fp64 reference on a lattice ----
  hipMemset(dC0, 0, out_elems * sizeof(TO)); hipMemset(dbg0, 0, M * 4);
  vars[0].run(dC0, dbg0); hipStreamSynchronize(st);
  std::vector<TO> c0(out_elems), c1(out_elems);
  std::vector<float> g0(M), g1(M);
  hipMemcpy(c0.data(), dC0, out_elems * sizeof(TO), hipMemcpyDeviceToHost); hipMemcpy(g0.data(), dbg0, M * 4, hipMemcpyDeviceToHost);
  for (size_t v = 0; v < (TA == 0 && TB == 1 ? 7 : 11); v++) {
    hipMemset(dC1, 0xff, out_elems * sizeof(TO)); hipMemset(dbg1, 0xff, M * 4);
    vars[v].run(dC1, dbg1); hipStreamSynchronize(st);
    const hipError_t e = hipGetLastError();
    hipMemcpy(c1.data(), dC1, out_elems * sizeof(TO), hipMemcpyDeviceToHost); hipMemcpy(g1.data(), dbg1, M * 4, hipMemcpyDeviceToHost);
    long nbad = 0; double maxd;
    if (split > 1) maxd = cmp<TO>(c0, c1, (long)split * M, N, N, &nbad); else maxd = cmp<TO>(c0, c1, M, N, ldc, &nbad);
    double maxe = 0, maxg = 0;
    for (int r = 0; r < M; r += 397)
      for (int c = 0; c < N; c += 61) {
        double acc = nt_plain ? hbias[c] : 0.0;
        for (int k = 0; k < K; k++) {
          const double a = bf2f_host(TA ? ha[(size_t)k * lda + r] : ha[(size_t)r * lda + k]);
          const double b = bf2f_host(TB ? hb[(size_t)c * ldb + k] : hb[(size_t)k * ldb + c]);
          acc += a * b;
        }
        double got = 0;
        if (split > 1) { for (int z = 0; z < p.split; z++) got += (double)*reinterpret_cast<float*>(&c1[((size_t)z * M + r) * N + c]); }
        else if constexpr (sizeof(TO) == 2) got = bf2f_host(c1[(size_t)r * ldc + c]);
        else got = c1[(size_t)r * ldc + c];
        const double d = fabs(acc - got) / (1.0 + fabs(acc));
        if (d > maxe) maxe = d;
      }
    if (TA == 1 && TB == 0)
      for (int r = 0; r < M; r += 97) {
        double acc = 0;
        for (int k = 0; k < K; k++) acc += bf2f_host(ha[(size_t)k * lda + r]);
        const double d = fabs(acc - g1[r]) / (1.0 + fabs(acc));
        if (d > maxg) maxg = d;
      }
    printf("[%s] %-32s err=%d  vs round 5: max|d| %.4g bad %ld   vs fp64 lattice: max rel %.4g   bias-grad max rel %.3g\n", s.name, vars[v].name, (int)e, maxd, nbad, maxe, maxg);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < rounds; r++)
    for (size_t v = 0; v < vars.size(); v++) {
      for (int i = 0; i < 3; i++) vars[v].run(dC1, dbg1);
      hipEventRecord(e0, st);
      for (int i = 0; i < iters; i++) vars[v].run(dC1, dbg1);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / iters;
      printf("[%s] round %d %-32s %8.1f us  %7.1f TF\n", s.name, r, vars[v].name, us, 2.0 * M * N * K / us / 1e6);
    }
  if constexpr (!(TA == 1 && TB == 0)) {
    // where a tile's time goes: (label, shader clock) pairs from wave 0 of workgroup 0 (g32_kernel, VAR & 32)
    long long* dst; hipMalloc(&dst, 800 * 8); 
    for (int rep = 0; rep < 50; rep++) {
      hipMemset(dst, 0, 800 * 8);
      G256P q = prep(dC1, dbg1); q.bias_grad = reinterpret_cast<float*>(dst);
      if (const char* e = getenv("VCT_PROBE_PF")) q.pf_dist = atoi(e);
      g32_launch<TA, TB, TO, 32>(q, st); hipStreamSynchronize(st);
    }
    std::vector<long long> hs(800); hipMemcpy(hs.data(), dst, 800 * 8, hipMemcpyDeviceToHost);
    printf("[%s] stamps of workgroup 0 (label:delta cycles since the previous stamp; 0 stage top, 1 k-steps 0-2 done, 2 vmcnt(0) passed, 3 barrier passed, 4 K loop done, 5 slab written, 6 barrier, 7 read-out + stores issued + barrier, 8 stores issued, 9 next stage DMA issued)\n", s.name);
    long long prev = hs[1], t_tile = hs[1], t4 = 0; int ntile = 0;
    printf("stamps:\n");
    for (int i = 0; i < 400 && hs[2 * i + 1] != 0; i++) {
      if (i < 64 || hs[2 * i] >= 4) printf(" %lld:%lld", hs[2 * i], hs[2 * i + 1] - prev);
      prev = hs[2 * i + 1];
      if (hs[2 * i] == 4) t4 = prev;
      if (hs[2 * i] == 9) { printf("   | K loop %lld epilogue %lld\n", t4 - t_tile, prev - t4); t_tile = prev; ntile++; }
    }
    printf("\n");
    hipFree(dst);
  }
  hipFree(dA); hipFree(dB); hipFree(dC0); hipFree(dC1); hipFree(dbias); hipFree(dbg0); hipFree(dbg1);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const int rounds = argc > 2 ? atoi(argv[2]) : 3;
  const char* only = argc > 3 ? argv[3] : "";
  Shape shapes[] = {
      {"NT square 4096^3", 0, 1, 4096, 4096, 4096, 1}, {"NT gen_fwd 4864x30522x512", 0, 1, 4864, 30522, 512, 1},
      {"NT ragged 1000x3001x520", 0, 1, 1000, 3001, 520, 1},
      {"NN gen_dx 4864x512x30522 split 6", 0, 0, 4864, 512, 30522, 6}, {"NN ragged 700x500x5000 split 3", 0, 0, 700, 500, 5000, 3},
      {"TN gen_dw 30522x512x4864", 1, 0, 30522, 512, 4864, 1}, {"TN ragged 3001x500x1000", 1, 0, 3001, 500, 1000, 1},
      {"NTs dx-nt 4864x512x30528 split 6", 0, 1, 4864, 512, 30528, 6},
  };
  hipStream_t st; hipStreamCreate(&st);
  for (const Shape& s : shapes) {
    if (only[0] && !strstr(s.name, only)) continue;
    if (s.ta == 0 && s.tb == 1 && s.split == 1) run_shape<0, 1, bf16_t>(s, st, iters, rounds);
    else if (s.ta == 0 && s.tb == 1) run_shape<0, 1, float>(s, st, iters, rounds);
    else if (s.ta == 0 && s.tb == 0) run_shape<0, 0, float>(s, st, iters, rounds);
    else run_shape<1, 0, float>(s, st, iters, rounds);
  }
  return 0;
}
