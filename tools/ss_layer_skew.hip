// Wave skew inside the sample-stationary layer kernel (dev tool): every wave stamps the phase boundaries; prints, per phase, the median
// over workgroups of (first wave to arrive, last wave to arrive) relative to the workgroup's start.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ss_layer_skew.hip video-captioning-transformer_amd/csrc/vct_runtime.hip -o tools/bin/ss_layer_skew
#define SS_STAMPS 1
#define SS_STAMPS_ALLWAVES 1
#include <cstring>
#include <cstdlib>
unsigned long long* g_ss_dbg = nullptr;
#include "../video-captioning-transformer_amd/csrc/vct_layer_ss.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
static void* dalloc(size_t bytes, int fill) { void* p; hipMalloc(&p, bytes); hipMemset(p, fill, bytes); return p; }
static float* fvec(int n, float v) { std::vector<float> h(n, v); float* p; hipMalloc(&p, n * 4); hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p; }
static void* bfrand(size_t n, float scale) {
  std::vector<unsigned short> h(n); unsigned s = 777;
  for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; float v = ((int)(s >> 9) % 2001 - 1000) * 1e-3f * scale; unsigned u; memcpy(&u, &v, 4); h[i] = (unsigned short)(u >> 16); }
  void* p; hipMalloc(&p, n * 2); hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice); return p;
}
int main() {
  const int cross = 1, B = 256, L = 19, Lm = 13, d = 512, ff = 2048;
  vct_layer_ss_desc q; memset(&q, 0, sizeof(q));
  q.dtype = VCT_BF16; q.B = B; q.L = L; q.Lm = Lm; q.d = d; q.H = 8; q.ff = ff; q.act = VCT_ACT_GELU; q.last = 1; q.causal = 1;
  q.nchunks = vct_layer_ss_stream_chunks(ff, cross);
  q.wpk = bfrand((size_t)q.nchunks * 32768, 0.05f);
  const size_t M = (size_t)B * L;
  q.x = bfrand(M * d, 1.0f); q.mem = bfrand((size_t)B * Lm * d, 1.0f);
  q.b_qkv = fvec(3 * d, 0.01f); q.b_o = fvec(d, 0.01f); q.b_cq = fvec(d, 0.01f); q.b_ckv = fvec(2 * d, 0.01f); q.b_co = fvec(d, 0.01f);
  q.b1 = fvec(ff, 0.01f); q.b2 = fvec(d, 0.01f);
  q.qkv = dalloc(M * 3 * d * 2, 0); q.o = dalloc(M * d * 2, 0); q.a = dalloc(M * d * 2, 0);
  q.cq = dalloc(M * d * 2, 0); q.ckv = dalloc((size_t)B * 16 * 2 * d * 2, 0); q.co = dalloc(M * d * 2, 0); q.ca = dalloc(M * d * 2, 0);
  q.hpre = dalloc(M * ff * 2, 0); q.h = dalloc(M * ff * 2, 0); q.f = dalloc(M * d * 2, 0);
  vct_ss_norm* ns[4] = {&q.n1, &q.n2, &q.n3, &q.nf};
  for (auto* n : ns) { n->gamma = fvec(d, 1.0f); n->beta = fvec(d, 0.0f); n->y = dalloc(M * d * 2, 0); n->mean = (float*)dalloc(M * 4, 0); n->rstd = (float*)dalloc(M * 4, 0); }
  uint32_t* seed = (uint32_t*)dalloc(4, 1); q.seed = seed; q.p_drop = 0.3f;
  q.site_sa = 1; q.site_n1 = 2; q.site_ca = 3; q.site_n2 = 4; q.site_ff = 5; q.site_n3 = 6;
  hipMalloc(&g_ss_dbg, (size_t)B * 8 * 64 * 8); hipMemset(g_ss_dbg, 0, (size_t)B * 8 * 64 * 8);
  for (int i = 0; i < 4; i++) { int rc = vct_layer_ss_fwd(&q, 1, nullptr); if (rc) { printf("rc %d\n", rc); return 1; } }
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)B * 8 * 64);
  hipMemcpy(h.data(), g_ss_dbg, h.size() * 8, hipMemcpyDeviceToHost);
  printf("phase: median over workgroups of [first wave, wave 0, last wave] arrival (k cycles since the workgroup's stamp 0), spread\n");
  for (int i = 1; i < 42; i++) {
    std::vector<double> lo, w0, hi;
    for (int b = 0; b < B; b++) {
      unsigned long long t0 = ~0ull, mn = ~0ull, mx = 0; bool ok = true;
      for (int w = 0; w < 8; w++) { unsigned long long a = h[((size_t)b * 8 + w) * 64 + 0]; if (a && a < t0) t0 = a; }
      for (int w = 0; w < 8; w++) { unsigned long long c = h[((size_t)b * 8 + w) * 64 + i]; if (!c) { ok = false; break; } mn = std::min(mn, c); mx = std::max(mx, c); }
      if (!ok) continue;
      lo.push_back((double)(mn - t0)); hi.push_back((double)(mx - t0)); w0.push_back((double)(h[((size_t)b * 8) * 64 + i] - t0));
    }
    if (lo.empty()) continue;
    std::sort(lo.begin(), lo.end()); std::sort(hi.begin(), hi.end()); std::sort(w0.begin(), w0.end());
    printf("  %2d  first %7.1f  wave0 %7.1f  last %7.1f   spread %6.1f\n", i, lo[lo.size() / 2] * 1e-3, w0[w0.size() / 2] * 1e-3, hi[hi.size() / 2] * 1e-3,
           (hi[hi.size() / 2] - lo[lo.size() / 2]) * 1e-3);
  }
  printf("per-wave mean arrival minus the workgroup's first arrival (k cycles), phases 2 (qkv), 8 (out_proj), 12 (cross q + kv), 21, 23 (ffn)\n");
  for (int i : {2, 8, 12, 21, 23, 27}) {
    double sum[8] = {0}; int n = 0;
    for (int b = 0; b < B; b++) {
      unsigned long long mn = ~0ull; bool ok = true;
      for (int w = 0; w < 8; w++) { unsigned long long c = h[((size_t)b * 8 + w) * 64 + i]; if (!c) ok = false; mn = std::min(mn, c); }
      if (!ok) continue;
      for (int w = 0; w < 8; w++) sum[w] += (double)(h[((size_t)b * 8 + w) * 64 + i] - mn);
      n++;
    }
    printf("  phase %2d:", i);
    for (int w = 0; w < 8; w++) printf(" w%d %5.2f", w, sum[w] / n * 1e-3);
    printf("\n");
  }
  return 0;
}
