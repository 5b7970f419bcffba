// Phase timing of the sample-stationary layer kernel (dev tool): compiles csrc/vct_layer_ss.hip with SS_STAMPS (wave 0 of every
// workgroup writes s_memtime stamps at the phase boundaries) and prints the median phase durations over the workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ss_layer_stamps.hip video-captioning-transformer_amd/csrc/vct_runtime.hip -o tools/bin/ss_layer_stamps
#define SS_STAMPS 1
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
  std::vector<unsigned short> h(n);
  unsigned s = 777;
  for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; float v = ((int)(s >> 9) % 2001 - 1000) * 1e-3f * scale; unsigned u; memcpy(&u, &v, 4); h[i] = (unsigned short)(u >> 16); }
  void* p; hipMalloc(&p, n * 2); hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice); return p;
}

static void run(int cross, int B, int L, int Lm, float p_drop) {
  const int d = 512, ff = 2048;
  vct_layer_ss_desc q; memset(&q, 0, sizeof(q));
  q.dtype = VCT_BF16; q.B = B; q.L = L; q.Lm = cross ? Lm : 0; q.d = d; q.H = 8; q.ff = ff; q.act = VCT_ACT_GELU; q.last = 1; q.causal = cross;
  q.nchunks = vct_layer_ss_stream_chunks(ff, cross);
  q.wpk = bfrand((size_t)q.nchunks * 32768, 0.05f);
  const size_t M = (size_t)B * L;
  q.x = bfrand(M * d, 1.0f); q.mem = cross ? bfrand((size_t)B * Lm * d, 1.0f) : nullptr;
  q.b_qkv = fvec(3 * d, 0.01f); q.b_o = fvec(d, 0.01f); q.b_cq = fvec(d, 0.01f); q.b_ckv = fvec(2 * d, 0.01f); q.b_co = fvec(d, 0.01f);
  q.b1 = fvec(ff, 0.01f); q.b2 = fvec(d, 0.01f);
  q.qkv = dalloc(M * 3 * d * 2, 0); q.o = dalloc(M * d * 2, 0); q.a = dalloc(M * d * 2, 0);
  q.cq = dalloc(M * d * 2, 0); q.ckv = dalloc((size_t)B * 16 * 2 * d * 2, 0); q.co = dalloc(M * d * 2, 0); q.ca = dalloc(M * d * 2, 0);
  q.hpre = dalloc(M * ff * 2, 0); q.h = dalloc(M * ff * 2, 0); q.f = dalloc(M * d * 2, 0);
  vct_ss_norm* ns[4] = {&q.n1, &q.n2, &q.n3, &q.nf};
  for (auto* n : ns) { n->gamma = fvec(d, 1.0f); n->beta = fvec(d, 0.0f); n->y = dalloc(M * d * 2, 0); n->mean = (float*)dalloc(M * 4, 0); n->rstd = (float*)dalloc(M * 4, 0); }
  uint32_t* seed = (uint32_t*)dalloc(4, 1);
  if (p_drop > 0) { q.seed = seed; q.p_drop = p_drop; }
  q.site_sa = 1; q.site_n1 = 2; q.site_ca = 3; q.site_n2 = 4; q.site_ff = 5; q.site_n3 = 6;
  hipMalloc(&g_ss_dbg, (size_t)B * 64 * 8); hipMemset(g_ss_dbg, 0, (size_t)B * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) { int rc = vct_layer_ss_fwd(&q, 1, nullptr); if (rc) { printf("rc %d\n", rc); return; } }
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int R = 10;
  for (int i = 0; i < R; i++) vct_layer_ss_fwd(&q, 1, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)B * 64);
  hipMemcpy(h.data(), g_ss_dbg, h.size() * 8, hipMemcpyDeviceToHost);
  printf("== %s layer  B %d L %d Lm %d p_drop %.1f: %.1f us per launch (events)\n", cross ? "decoder" : "encoder", B, L, Lm, p_drop, ms * 1e3 / R);
  int prev = 0;
  const char* names[64] = {0};
  names[1] = "load x/mem + barrier"; names[2] = "qkv gemm x3 + epi + res loads"; names[3] = "barrier"; names[4] = "copy qkv"; names[5] = "self attention";
  names[6] = "barrier"; names[7] = "copy o"; names[8] = "out_proj gemm"; names[9] = "LN1 epilogue"; names[10] = "barrier + copy a, x1";
  names[11] = "cross q gemm + epi"; names[12] = "cross kv gemm x2 + epi"; names[13] = "barrier + copy q, kv"; names[14] = "cross attention";
  names[15] = "barrier + copy o2"; names[16] = "cross out_proj gemm"; names[17] = "LN2 epilogue"; names[18] = "barrier + copy a2, x2";
  for (int j = 0; j < 4; j++) { names[20 + 4 * j] = "  lin1 gemm"; names[21 + 4 * j] = "  ffn epilogue + barriers"; names[22 + 4 * j] = "  copy hpre, h"; names[23 + 4 * j] = "  lin2 gemm"; }
  names[40] = "final LN epilogue(s)"; names[41] = "barrier + copies";
  double total = 0;
  for (int i = 1; i < 64; i++) {
    if (!names[i]) continue;
    std::vector<double> v;
    for (int b = 0; b < B; b++) { unsigned long long a = h[(size_t)b * 64 + prev], c = h[(size_t)b * 64 + i]; if (c && a) v.push_back((double)(c - a)); }
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2];
    printf("  %2d %-34s %8.0f ticks  (min %6.0f max %6.0f)\n", i, names[i], med, v.front(), v.back());
    total += med; prev = i;
  }
  printf("  total %.0f ticks of s_memtime (100 MHz => %.1f us if ticks are 10 ns)\n", total, total * 0.01);
}

int main() {
  run(0, 256, 13, 0, 0.3f);
  run(1, 256, 19, 13, 0.3f);
  run(1, 256, 19, 13, 0.0f);
  return 0;
}
