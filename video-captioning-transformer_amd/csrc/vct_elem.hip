// HBM-bound pieces of the caption path on gfx950: encoder front end (mean token + temporal
// encoding), token embedding gather / deterministic scatter-add, the symmetric-cross-entropy loss
// with its logits gradient (row resident in LDS: one HBM read + one HBM write per logit), casts,
// first-index arg-max, dropout seed advance.  All loads/stores are 16-byte vectors where the
// layout allows; reductions use wave shuffles + a few LDS words.
#include "vct_common.h"

namespace vct {

template <typename T> struct EV { static constexpr int VEC = 16 / sizeof(T); };
template <typename T, int VEC> struct alignas(sizeof(T) * VEC) PackT { T v[VEC]; };

// ---------------------------------------------------------------------------------------------
// encoder front end: z[b,0,:] = mean_t u[b,t,:]; z[b,t+1,:] = u[b,t,:] + pe[t+1,:]
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void enc_frontend_fwd_kernel(int B, int Tn, int d, const T* __restrict__ u, const float* __restrict__ pe,
                                        T* __restrict__ z) {
  constexpr int VEC = EV<T>::VEC;
  using P = PackT<T, VEC>;
  const int b = blockIdx.x;
  const int nvec = d / VEC;
  for (int vi = threadIdx.x; vi < nvec; vi += blockDim.x) {
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) acc[j] = 0.0f;
    for (int t = 0; t < Tn; t++) {
      const P uv = *reinterpret_cast<const P*>(u + ((size_t)b * Tn + t) * d + vi * VEC);
      P o;
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const float f = to_f<T>(uv.v[j]);
        acc[j] += f;
        o.v[j] = from_f<T>(f + pe[(size_t)(t + 1) * d + vi * VEC + j]);
      }
      *reinterpret_cast<P*>(z + ((size_t)b * (Tn + 1) + t + 1) * d + vi * VEC) = o;
    }
    P m;
#pragma unroll
    for (int j = 0; j < VEC; j++) m.v[j] = from_f<T>(acc[j] / (float)Tn + pe[vi * VEC + j]);
    *reinterpret_cast<P*>(z + ((size_t)b * (Tn + 1)) * d + vi * VEC) = m;
  }
}

template <typename T>
__global__ void enc_frontend_bwd_kernel(int B, int Tn, int d, const T* __restrict__ dz, T* __restrict__ du) {
  constexpr int VEC = EV<T>::VEC;
  using P = PackT<T, VEC>;
  const int b = blockIdx.x;
  const int nvec = d / VEC;
  const float inv = 1.0f / (float)Tn;
  for (int vi = threadIdx.x; vi < nvec; vi += blockDim.x) {
    const P g0 = *reinterpret_cast<const P*>(dz + ((size_t)b * (Tn + 1)) * d + vi * VEC);
    for (int t = 0; t < Tn; t++) {
      const P g = *reinterpret_cast<const P*>(dz + ((size_t)b * (Tn + 1) + t + 1) * d + vi * VEC);
      P o;
#pragma unroll
      for (int j = 0; j < VEC; j++) o.v[j] = from_f<T>(to_f<T>(g.v[j]) + to_f<T>(g0.v[j]) * inv);
      *reinterpret_cast<P*>(du + ((size_t)b * Tn + t) * d + vi * VEC) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// token embedding (fp32 master table gathered directly: no bf16 copy of the 62.5 MB table needed)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(int N, int S, int d, const int64_t* __restrict__ ids,
                                                        int64_t id_bstride, const float* __restrict__ table,
                                                        const float* __restrict__ pos, T* __restrict__ x,
                                                        const uint32_t* seed, uint32_t site, float p_drop) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const Dropout dr = make_dropout(seed, site, p_drop);
  const int b = n / S, s = n % S;
  const int64_t id = ids[(size_t)b * id_bstride + s];
  const float4* trow = reinterpret_cast<const float4*>(table + (size_t)id * d);
  const float4* prow = reinterpret_cast<const float4*>(pos + (size_t)s * d);
  for (int vi = lane; vi < d / 4; vi += 64) {
    const float4 tv = trow[vi], pv = prow[vi];
    const float e[4] = {tv.x + pv.x, tv.y + pv.y, tv.z + pv.z, tv.w + pv.w};
    PackT<T, 4> o;
    float dmv[4];
    drop_mults<4>(dr, (uint32_t)n * (uint32_t)d + (uint32_t)(vi * 4), dmv);
#pragma unroll
    for (int j = 0; j < 4; j++) o.v[j] = from_f<T>(e[j] * dmv[j]);
    *reinterpret_cast<PackT<T, 4>*>(x + (size_t)n * d + vi * 4) = o;
  }
}

// Deterministic scatter-add without float atomics.
//   index pass   (one thread per token position): integer atomics build first_pos[id] = smallest position holding that id and
//                cnt[id] = occurrences (both order-independent); only the FIRST occurrence owns the table row.
//   sum pass     ONE launch with two roles.  Light: one WAVE per position; a token seen once (the common case) is a straight row
//                copy, a token seen 2..8 times scans the later positions (1024 ids per trip, ballots) and adds its occurrences in
//                increasing position order.  Heavy (the first 32 workgroups, 1024 threads): tokens seen more often, found by the
//                workgroup itself in the count table: ordered compaction of the matches among 8192 positions per round, 16
//                column-parallel slots over fixed subsequences of the ordered occurrence list, slot partials added in slot
//                order.  ([CLS] sits in every caption: as a 256-thread workgroup among the 4 864 of the first version's
//                one-workgroup-per-position kernel its chain of scan rounds and row fetches was the whole 44 us.)
// Bitwise reproducible; the rows to write are zeroed by embed_prep_kernel first.
__global__ void embed_index_kernel(int N, int S, const int64_t* __restrict__ ids, int64_t id_bstride, int64_t pad_id,
                                   int32_t* __restrict__ first_pos, int32_t* __restrict__ cnt, int32_t* __restrict__ hdr,
                                   int32_t* __restrict__ flat_cur) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n == 0) hdr[1] = N;                       // this call's positions: what the NEXT call's prep kernel zeroes
  if (n >= N) return;
  const int64_t id = ids[(size_t)(n / S) * id_bstride + (n % S)];
  flat_cur[n] = (int32_t)id;  // position-major copy: the passes behind scan it without an integer division per element
  if (id == pad_id) return;
  atomicMin(&first_pos[id], n);
  atomicAdd(&cnt[id], 1);
}

constexpr int EMB_LIGHT_MAX = 8;

template <typename T>
__device__ __forceinline__ void embed_bwd_light(int n, int N, int d, const int32_t* __restrict__ flat, int32_t pad_id,
                                                const T* __restrict__ dx, float* __restrict__ dtable,
                                                const int32_t* __restrict__ first_pos, const int32_t* __restrict__ cnt,
                                                const uint32_t* seed, uint32_t site, float p_drop) {
  constexpr int VEC = EV<T>::VEC, MAXCH = 1024 / (64 * VEC) > 0 ? 1024 / (64 * VEC) : 1;      // d <= 1024: 2 (bf16) / 4 (fp32) chunks per lane
  using P = PackT<T, VEC>;
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const int32_t id = flat[n];
  if (id == pad_id || first_pos[id] != n) return;       // not the owner of this table row
  const int occurrences = cnt[id];
  if (occurrences > EMB_LIGHT_MAX) return;              // the heavy workgroups of this launch own it
  const Dropout dr = make_dropout(seed, site, p_drop);
  const int nch = d / (64 * VEC);                        // full chunks per lane (d is a multiple of 64 * VEC or smaller than it)
  const int tailv = (d - nch * 64 * VEC) / VEC;          // lanes with one more vector (d % (64 * VEC) != 0)
  float acc[MAXCH + 1][VEC];
#pragma unroll
  for (int c = 0; c <= MAXCH; c++)
#pragma unroll
    for (int j = 0; j < VEC; j++) acc[c][j] = 0.0f;
  auto add_row = [&](int pos) {
#pragma unroll
    for (int c = 0; c <= MAXCH; c++) {
      const bool on = c < nch || (c == nch && lane < tailv);
      if (on) {
        const int col = (c * 64 + lane) * VEC;
        const P v = *reinterpret_cast<const P*>(dx + (size_t)pos * d + col);
        float dmv[VEC];
        drop_mults<VEC>(dr, (uint32_t)pos * (uint32_t)d + (uint32_t)col, dmv);
#pragma unroll
        for (int j = 0; j < VEC; j++) acc[c][j] += to_f<T>(v.v[j]) * dmv[j];
      }
    }
  };
  add_row(n);
  int found = 1;
  // 16 x 64 ids per trip, all loads independent: a token whose next occurrence is far away walks the id array in N / 1024
  // dependent round trips (with 256 ids per trip the slowest wave of the kernel took 19 of them: 60 us)
  constexpr int TRIP = 16;
  for (int base = n + 1; base < N && found < occurrences; base += TRIP * 64) {
    unsigned long long bal[TRIP];
#pragma unroll
    for (int u = 0; u < TRIP; u++) {
      const int j = base + u * 64 + lane;
      const bool mt = flat[min(j, N - 1)] == id && j < N;
      bal[u] = __ballot(mt);
    }
#pragma unroll
    for (int u = 0; u < TRIP; u++) {
      unsigned long long mk = bal[u];
      while (mk != 0ull) {                               // increasing position order
        const int bit = __ffsll((long long)mk) - 1;
        mk &= mk - 1ull;
        add_row(base + u * 64 + bit);
        found++;
      }
    }
  }
#pragma unroll
  for (int c = 0; c <= MAXCH; c++) {
    const bool on = c < nch || (c == nch && lane < tailv);
    if (on) {
      const int col = (c * 64 + lane) * VEC;
#pragma unroll
      for (int j = 0; j < VEC; j += 4)
        *reinterpret_cast<float4*>(dtable + (size_t)id * d + col + j) = float4{acc[c][j], acc[c][j + 1], acc[c][j + 2], acc[c][j + 3]};
    }
  }
}

constexpr int EMB_HEAVY_WGS = 32;

template <typename T>
__global__ __launch_bounds__(1024) void embed_bwd_kernel(int N, int d, int V, const int32_t* __restrict__ flat, int32_t pad_id,
                                                         const T* __restrict__ dx, float* __restrict__ dtable,
                                                         const int32_t* __restrict__ first_pos, const int32_t* __restrict__ cnt,
                                                         const uint32_t* seed, uint32_t site, float p_drop) {
  constexpr int VEC = EV<T>::VEC, NT = 1024, NWV = NT / 64, SUBS = 8;
  using P = PackT<T, VEC>;
  if (blockIdx.x >= EMB_HEAVY_WGS) {                    // light role: one wave per position
    embed_bwd_light<T>((blockIdx.x - EMB_HEAVY_WGS) * NWV + (threadIdx.x >> 6), N, d, flat, pad_id, dx, dtable, first_pos, cnt, seed,
                       site, p_drop);
    return;
  }
  // heavy role (the FIRST workgroups of the grid, so the longest chains start first): workgroup h owns the tokens with more than
  // EMB_LIGHT_MAX occurrences in its slices of the id range; it finds them itself (a list appended to by the light waves would need
  // a second launch behind them: measured 17 us after the light pass instead of beside it)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* s_list = reinterpret_cast<int*>(smem);                          // [SUBS * NT] ordered matches of a round
  int* s_wcnt = s_list + SUBS * NT;                                    // [SUBS * NWV]
  int* s_off = s_wcnt + SUBS * NWV;                                    // [SUBS * NWV + 1] exclusive prefix (+ total), padded to 256
  int* s_own = s_off + 256;                                            // [NT] owner positions found in a slice of NT positions
  int* s_oid = s_own + NT;                                             // [NT] their ids
  int* s_onum = s_oid + NT;                                            // [NT] their occurrence counts
  int* s_ocnt = s_onum + NT;                                           // [NWV + 1], padded to 32
  float* s_part = reinterpret_cast<float*>(s_ocnt + 32);               // [slots][d]
  const Dropout dr = make_dropout(seed, site, p_drop);
  const int chunks = d / VEC, slots = NT / chunks;                     // thread = (occurrence slot, 16-byte column chunk)
  const int chunk = threadIdx.x % chunks, slot = threadIdx.x / chunks;
  const bool active = slot < slots;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int ibase = blockIdx.x * NT; ibase < V; ibase += EMB_HEAVY_WGS * NT) {
   // this workgroup's heavy tokens among ids [ibase, ibase + NT): ONE coalesced look at the count table (walking the positions
   // instead cost five dependent id -> table round trips), their owner positions compacted into s_own
   {
    const int pid = ibase + (int)threadIdx.x;
    const int fp = first_pos[min(pid, V - 1)];
    const int pc = cnt[min(pid, V - 1)];
    const bool own = pid < V && pc > EMB_LIGHT_MAX;                     // the padding id is never counted
    const unsigned long long ob = __ballot(own);
    __syncthreads();                                    // the previous slice's readers of s_own are done
    if (l == 0) s_ocnt[w] = __popcll(ob);
    __syncthreads();
    int off = 0;
    for (int i = 0; i < w; i++) off += s_ocnt[i];
    if (own) {
      const int slot_o = off + __popcll(ob & ((1ull << l) - 1ull));
      s_own[slot_o] = fp; s_oid[slot_o] = pid; s_onum[slot_o] = pc;
    }
    if (threadIdx.x == NT - 1) s_ocnt[NWV] = off + __popcll(ob);
    __syncthreads();
   }
   const int nown = s_ocnt[NWV];
   for (int e = 0; e < nown; e++) {
    const int n = s_own[e];
    const int32_t id = s_oid[e];
    const int occurrences = s_onum[e];
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) acc[j] = 0.0f;
    int seen = 0;                                        // occurrences consumed so far (global order index)
    for (int base = n; base < N && seen < occurrences; base += SUBS * NT) {
      // ordered compaction of the matches among positions [base, base + SUBS * NT): the SUBS id loads of a thread are issued
      // together, one ballot per wave and NT-position slice, ONE barrier, prefix over (slice, wave)
      bool match[SUBS];
      unsigned long long bal[SUBS];
#pragma unroll
      for (int sub = 0; sub < SUBS; sub++) {
        match[sub] = flat[min(base + sub * NT + (int)threadIdx.x, N - 1)] == id;
      }
      __syncthreads();                                  // the previous round's readers of s_list / s_wcnt are done
#pragma unroll
      for (int sub = 0; sub < SUBS; sub++) {
        match[sub] = match[sub] && (base + sub * NT + (int)threadIdx.x < N);
        bal[sub] = __ballot(match[sub]);
        if (l == 0) s_wcnt[sub * NWV + w] = __popcll(bal[sub]);
      }
      __syncthreads();
      // exclusive prefix over the SUBS * NWV = 128 (slice, wave) counts by ONE wave (two entries per lane, shuffle scan); a serial
      // walk over them by every thread cost the 16 waves ~20 us
      if (w == 0) {
        const int c0 = s_wcnt[2 * l], c1 = s_wcnt[2 * l + 1];
        int incl = c0 + c1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (l >= off) incl += t; }
        const int excl = incl - (c0 + c1);
        s_off[2 * l] = excl; s_off[2 * l + 1] = excl + c0;
        if (l == 63) s_off[SUBS * NWV] = incl;
      }
      __syncthreads();
#pragma unroll
      for (int sub = 0; sub < SUBS; sub++)
        if (match[sub]) s_list[s_off[sub * NWV + w] + __popcll(bal[sub] & ((1ull << l) - 1ull))] = base + sub * NT + threadIdx.x;
      const int found = s_off[SUBS * NWV];
      __syncthreads();
      if (active) {
        // slot s takes list entries whose GLOBAL order index is congruent to s (mod slots); eight rows are fetched before the
        // first one is added (the loads are independent, the ADDS keep their order)
        constexpr int DEPTH = 8;
        for (int q = ((slot - seen) % slots + slots) % slots; q < found; q += DEPTH * slots) {
          P v[DEPTH];
          int pos[DEPTH];
#pragma unroll
          for (int u = 0; u < DEPTH; u++) {
            const int qi = q + u * slots;
            pos[u] = qi < found ? s_list[qi] : -1;
            v[u] = *reinterpret_cast<const P*>(dx + (size_t)(pos[u] >= 0 ? pos[u] : n) * d + chunk * VEC);
          }
#pragma unroll
          for (int u = 0; u < DEPTH; u++) {
            if (pos[u] >= 0) {
              float dmv[VEC];
              drop_mults<VEC>(dr, (uint32_t)pos[u] * (uint32_t)d + (uint32_t)(chunk * VEC), dmv);
#pragma unroll
              for (int j = 0; j < VEC; j++) acc[j] += to_f<T>(v[u].v[j]) * dmv[j];
            }
          }
        }
      }
      seen += found;
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < VEC; j++) s_part[slot * d + chunk * VEC + j] = acc[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += NT) {
      float sm = 0.0f;
      for (int sl = 0; sl < slots; sl++) sm += s_part[sl * d + c];
      dtable[(size_t)id * d + c] = sm;
    }
   }
  }
}

// ---------------------------------------------------------------------------------------------
// SCE loss + dlogits.  One 1024-thread workgroup per row; the row sits in LDS as fp32.
// ---------------------------------------------------------------------------------------------
__global__ void count_valid_kernel(int N, int S, const int64_t* __restrict__ labels, int64_t lbstride, int64_t pad_id,
                                   float* __restrict__ out) {
  __shared__ float red[16];
  float c = 0.0f;
  for (int n = threadIdx.x; n < N; n += blockDim.x) c += (labels[(size_t)(n / S) * lbstride + (n % S)] != pad_id) ? 1.0f : 0.0f;
  c = block_sum<16>(c, red);
  if (threadIdx.x == 0) out[0] = c;
}

constexpr float SCE_C = 9.210340371976184f;  // -log(1e-4): loss.py:86-88 off-target log(clamp(onehot))

// One NT-thread workgroup per row; the row lives in REGISTERS (IT 16-byte vectors per thread, loaded
// once from HBM), so every logit costs one HBM read, one exp and one HBM write (the gradient), and
// LDS only carries the block reductions.  NT*IT vectors must cover the row: 512 x 8 for bf16 rows up to 4096 vectors (measured best; 256 x 16: few waves per
// barrier, 16 loads in flight per thread, 3 rows per CU) or 1024 x 8 for very wide vocabularies.
template <typename T, int IT, int NT>
__global__ __launch_bounds__(NT, NT / 128) void sce_loss_kernel(int N, int S, int V, const T* __restrict__ logits, int64_t ldl,
                                                        const int64_t* __restrict__ labels, int64_t lbstride,
                                                        int64_t pad_id, float alpha, T* __restrict__ dlogits, int64_t ld_dl,
                                                        float* __restrict__ row_ws) {
  // one LDS array per block reduction: each costs ONE barrier (no guard barrier before reusing a shared scratch)
  __shared__ float red_m[16], red_s[16], red_q[16], red_c[16];
  constexpr int VEC = EV<T>::VEC, NW = NT / 64;
  using P = PackT<T, VEC>;
  const int n = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const T* x = logits + (size_t)n * ldl;
  const int64_t y_in = labels[(size_t)(n / S) * lbstride + (n % S)];
  const bool valid = (y_in != pad_id);
  // a label outside the vocabulary (torch raises a device assert there) must not become a stray read / write: the row is
  // computed against column 0 and, when it counts towards the loss, poisons it with NaN so the step fails loudly
  const bool oob = (y_in < 0 || y_in >= V);
  const int64_t y = oob ? 0 : y_in;
  const float nvalid = row_ws[2 * N];
  const float xy = to_f<T>(x[y]);
  const int nv = (V + VEC - 1) / VEC;                 // vectors holding valid columns (ldl covers the rounded-up row)
  // all IT loads are issued back to back, unconditionally (index clamped, out-of-row vectors masked after)
  P pk[IT];
#pragma unroll
  for (int it = 0; it < IT; it++) pk[it] = *reinterpret_cast<const P*>(x + (size_t)min(it * NT + tid, nv - 1) * VEC);
  float e[IT][VEC];
  float mx = -INFINITY;
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int vi = it * NT + tid;
#pragma unroll
    for (int j = 0; j < VEC; j++) e[it][j] = to_f<T>(pk[it].v[j]);
    if (vi >= nv - 1) {                                // only the last vector of the row (and beyond) needs masking
#pragma unroll
      for (int j = 0; j < VEC; j++)
        if (vi >= nv || vi * VEC + j >= V) e[it][j] = -INFINITY;
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) mx = fmaxf(mx, e[it][j]);
  }
  mx = wave_max(mx);
  if (ln == 0) red_m[wv] = mx;
  __syncthreads();
  mx = red_m[0];
#pragma unroll
  for (int i = 1; i < NW; i++) mx = fmaxf(mx, red_m[i]);
  float se = 0.0f;
#pragma unroll
  for (int it = 0; it < IT; it++)
#pragma unroll
    for (int j = 0; j < VEC; j++) { e[it][j] = __expf(e[it][j] - mx); se += e[it][j]; }   // padding: exp(-inf) = 0
  se = wave_sum(se);
  if (ln == 0) red_s[wv] = se;
  __syncthreads();
  se = 0.0f;
#pragma unroll
  for (int i = 0; i < NW; i++) se += red_s[i];
  const float inv = 1.0f / se;
  const float beta = 1.0f - alpha;
  const float py = __expf(xy - mx) * inv;
  // over ALL columns: Qa = sum_{p_j >= 1e-7} p_j, ca = #{p_j < 1e-7} (padding has p = 0 and is counted: fixed below);
  // the label column is then taken out analytically -- no per-element (j != y) tests
  float q = 0.0f, cnt = 0.0f;
  if (alpha != 1.0f) {
#pragma unroll
    for (int it = 0; it < IT; it++)
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const float pj = e[it][j] * inv;
        const bool big = pj >= 1e-7f;
        q += big ? pj : 0.0f;
        cnt += big ? 0.0f : 1.0f;
      }
    q = wave_sum(q);
    cnt = wave_sum(cnt);
    if (ln == 0) { red_q[wv] = q; red_c[wv] = cnt; }
    __syncthreads();
    q = 0.0f; cnt = 0.0f;
#pragma unroll
    for (int i = 0; i < NW; i++) { q += red_q[i]; cnt += red_c[i]; }
    cnt -= (float)(IT * NT * VEC - V);                 // masked / padding slots were counted as "small"
    if (py >= 1e-7f) q -= py; else cnt -= 1.0f;        // remove the label column
  }
  if (tid == 0) {
    row_ws[n] = valid ? (oob ? __builtin_nanf("") : (mx + __logf(se)) - xy) : 0.0f;
    row_ws[N + n] = SCE_C * (q + 1e-7f * cnt);
  }
  if (dlogits == nullptr) return;
  // gradient for j != y:  p_j * (a + bn*(G_j - c*Q)),  G_j = c*[p_j >= 1e-7]; padding has p = 0 -> 0
  const float a = valid ? alpha / nvalid : 0.0f;
  const float bn = (alpha != 1.0f) ? beta / (float)N : 0.0f;
  const float k_small = a - bn * SCE_C * q;            // multiplier when p_j <  1e-7
  const float k_big = k_small + bn * SCE_C;            // multiplier when p_j >= 1e-7
  T* dx = dlogits + (size_t)n * ld_dl;
  const int nvo = (int)(ld_dl / VEC);
#pragma unroll
  for (int it = 0; it < IT; it++) {
    const int vi = it * NT + tid;
    if (vi < nvo) {
      P o;
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        const float pj = e[it][j] * inv;
        o.v[j] = from_f<T>(pj * (pj >= 1e-7f ? k_big : k_small));
      }
      *reinterpret_cast<P*>(dx + vi * VEC) = o;
    }
  }
  // the label column: a*(p_y - 1) + bn*p_y*(0 - c*Q)   (written after the row's vector stores)
  __syncthreads();
  if (tid == 0) dx[y] = from_f<T>(a * (py - 1.0f) - bn * py * SCE_C * q);
}

__global__ void sce_finalize_kernel(int N, float alpha, const float* __restrict__ row_ws, float* __restrict__ loss) {
  __shared__ float red[16];
  float ce = 0.0f, rce = 0.0f;
  for (int n = threadIdx.x; n < N; n += blockDim.x) { ce += row_ws[n]; rce += row_ws[N + n]; }
  ce = block_sum<16>(ce, red);
  rce = block_sum<16>(rce, red);
  if (threadIdx.x == 0) {
    const float nvalid = row_ws[2 * N];
    const float l_ce = ce / nvalid;
    loss[0] = (alpha == 1.0f) ? l_ce : alpha * l_ce + (1.0f - alpha) * (rce / (float)N);
  }
}

// ---------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t n) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const PackT<TS, 4> s = reinterpret_cast<const PackT<TS, 4>*>(src)[i];
    PackT<TD, 4> o;
#pragma unroll
    for (int j = 0; j < 4; j++) o.v[j] = from_f<TD>(to_f<TS>(s.v[j]));
    reinterpret_cast<PackT<TD, 4>*>(dst)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    dst[i] = from_f<TD>(to_f<TS>(src[i]));
  }
}

// ---------------------------------------------------------------------------------------------
// Batch assembly from a device-resident feature store: one wave per output row (b, t).  Row t of clip idx[b] is
// copied (and cast) when t < len(clip), zero-filled otherwise; lane 0 writes the padding-mask byte.
template <typename TD>
__global__ __launch_bounds__(256) void gather_pad_rows_kernel(const float* __restrict__ store, const int64_t* __restrict__ offsets,
                                                              const int64_t* __restrict__ idx, int B, int Tmax, int E,
                                                              TD* __restrict__ out, uint8_t* __restrict__ mask) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * Tmax) return;
  const int b = row / Tmax, t = row % Tmax;
  const int64_t clip = idx[b], r0 = offsets[clip], len = offsets[clip + 1] - r0;
  const bool valid = t < len;
  if (lane == 0) mask[row] = valid ? 0 : 1;
  const float* src = store + (size_t)(r0 + (valid ? t : 0)) * E;
  TD* dst = out + (size_t)row * E;
  const int e4 = (E % 4 == 0) ? E / 4 : 0;     // rows are 16-byte aligned only then
  for (int i = lane; i < e4; i += 64) {
    PackT<float, 4> v = reinterpret_cast<const PackT<float, 4>*>(src)[i];    // unconditional load, zeroed after
    PackT<TD, 4> o;
#pragma unroll
    for (int j = 0; j < 4; j++) o.v[j] = from_f<TD>(valid ? v.v[j] : 0.0f);
    reinterpret_cast<PackT<TD, 4>*>(dst)[i] = o;
  }
  for (int i = e4 * 4 + lane; i < E; i += 64) dst[i] = from_f<TD>(valid ? src[i] : 0.0f);
}

constexpr int AMX_THREADS = 1024;
template <typename T>
__global__ __launch_bounds__(AMX_THREADS) void argmax_rows_kernel(int cols, const T* __restrict__ x, int64_t ldx,
                                                          int64_t* __restrict__ out, int64_t out_stride,
                                                          int64_t end_id, uint8_t* __restrict__ ended,
                                                          int32_t* __restrict__ ended_count,
                                                          unsigned long long* __restrict__ all_ended_at, int t) {
  __shared__ float s_v[AMX_THREADS / 64];
  __shared__ int s_i[AMX_THREADS / 64];
  const T* r = x + (size_t)blockIdx.x * ldx;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  // 16-byte loads when the row allows it (a 30522-column row is 4 vector loads per thread instead of 30 scalar round trips:
  // the one-workgroup-per-row scan took 37 us per decoded token)
  constexpr int VEC = 16 / (int)sizeof(T);
  const bool vec_ok = (((uintptr_t)r) & 15) == 0;
  const int nv = vec_ok ? cols / VEC : 0;
  struct alignas(16) V { T e[VEC]; };
  for (int j = threadIdx.x; j < nv; j += AMX_THREADS) {
    const V v = reinterpret_cast<const V*>(r)[j];
#pragma unroll
    for (int u = 0; u < VEC; u++) {
      const float f = to_f<T>(v.e[u]);
      const int c = j * VEC + u;
      if (f > best || (f == best && c < bi)) { best = f; bi = c; }
    }
  }
  for (int j = nv * VEC + threadIdx.x; j < cols; j += AMX_THREADS) {
    const float v = to_f<T>(r[j]);
    if (v > best || (v == best && j < bi)) { best = v; bi = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = best; s_i[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < AMX_THREADS / 64; w++)
      if (s_v[w] > best || (s_v[w] == best && s_i[w] < bi)) { best = s_v[w]; bi = s_i[w]; }
    const int tok = (bi == 0x7fffffff) ? 0 : bi;
    out[(size_t)blockIdx.x * out_stride] = tok;
    // greedy-decode bookkeeping (MMT4Caption.py:166-171): sticky per-row end flag; the row that completes the set
    // records the step.  Integer atomics only: the result does not depend on arrival order.
    if (ended != nullptr && tok == end_id && !ended[blockIdx.x]) {
      ended[blockIdx.x] = 1;
      if (atomicAdd(ended_count, 1) + 1 == (int)gridDim.x) atomicMin(all_ended_at, (unsigned long long)t);
    }
  }
}

__global__ void advance_seed_kernel(uint32_t* seed) { seed[0] += 1u; }

}  // namespace vct
using namespace vct;

static bool dt_ok(int dt) { return dt == VCT_F32 || dt == VCT_BF16; }
static int vec_of(int dt) { return dt == VCT_BF16 ? 8 : 4; }

extern "C" int vct_enc_frontend_fwd(int dtype, int B, int T, int d, const void* u, const float* pe_rows, void* z,
                                    void* stream) {
  if (!dt_ok(dtype) || !u || !pe_rows || !z) return VCT_E_ARG;
  if (B <= 0 || T <= 0 || d <= 0) return VCT_E_SHAPE;
  if (d % vec_of(dtype)) return VCT_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int threads = 128;
  if (dtype == VCT_BF16)
    vct::launch((enc_frontend_fwd_kernel<bf16_t>), dim3(B), dim3(threads), 0, st, B, T, d, (const bf16_t*)u, pe_rows, (bf16_t*)z);
  else
    vct::launch((enc_frontend_fwd_kernel<float>), dim3(B), dim3(threads), 0, st, B, T, d, (const float*)u, pe_rows, (float*)z);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_enc_frontend_bwd(int dtype, int B, int T, int d, const void* dz, void* du, void* stream) {
  if (!dt_ok(dtype) || !dz || !du) return VCT_E_ARG;
  if (B <= 0 || T <= 0 || d <= 0) return VCT_E_SHAPE;
  if (d % vec_of(dtype)) return VCT_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VCT_BF16)
    vct::launch((enc_frontend_bwd_kernel<bf16_t>), dim3(B), dim3(128), 0, st, B, T, d, (const bf16_t*)dz, (bf16_t*)du);
  else
    vct::launch((enc_frontend_bwd_kernel<float>), dim3(B), dim3(128), 0, st, B, T, d, (const float*)dz, (float*)du);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_embed_fwd(int dtype, int B, int S, int d, const int64_t* ids, int64_t id_batch_stride,
                             const void* table, const float* pos, void* x, const uint32_t* seed, uint32_t site,
                             float p_drop, void* stream) {
  if (!dt_ok(dtype) || !ids || !table || !pos || !x) return VCT_E_ARG;
  if (B <= 0 || S <= 0 || d <= 0) return VCT_E_SHAPE;
  if (d % 4) return VCT_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int N = B * S;
  if (dtype == VCT_BF16)
    vct::launch((embed_fwd_kernel<bf16_t>), dim3((N + 3) / 4), dim3(256), 0, st, N, S, d, ids, id_batch_stride,
                       (const float*)table, pos, (bf16_t*)x, seed, site, p_drop);
  else
    vct::launch((embed_fwd_kernel<float>), dim3((N + 3) / 4), dim3(256), 0, st, N, S, d, ids, id_batch_stride,
                       (const float*)table, pos, (float*)x, seed, site, p_drop);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

// ONE launch in front of the scatter (it was three hipMemsetAsync calls = five fill kernels, 44 us): the occurrence tables
// are reset and the dense gradient is zeroed -- all V rows, or (incremental) only the rows the PREVIOUS call wrote: the rows of the
// ids of its batch, which that call left in its position-order id buffer (4864 rows of 2 KB instead of 62.5 MB; a row that occurs
// k times is zeroed k times).  (A compacted list of the written rows, appended to by every owner wave, was measured first: 4 600
// atomic increments of ONE counter serialise at ~11 ns each -- it took the scatter kernel from 44 to 65 us.)
namespace vct {
__global__ __launch_bounds__(256) void embed_prep_kernel(int V, int d, float* __restrict__ dtable, int32_t* __restrict__ first_pos,
                                                         int32_t* __restrict__ cnt, const int32_t* __restrict__ flat_prev,
                                                         const int32_t* __restrict__ hdr, int incremental) {
  const int gt = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
  for (int i = gt; i < V; i += nth) { first_pos[i] = 0x7f7f7f7f; cnt[i] = 0; }
  const float4 z = {0.0f, 0.0f, 0.0f, 0.0f};
  if (!incremental) {
    float4* dst = reinterpret_cast<float4*>(dtable);
    const size_t n4 = (size_t)V * d / 4;
    for (size_t i = gt; i < n4; i += nth) dst[i] = z;
  } else {
    const int nprev = hdr[1], d4 = d / 4;
    const int w = gt >> 6, nw = nth >> 6, lane = threadIdx.x & 63;
    for (int r = w; r < nprev; r += nw) {                  // one wave per position of the previous batch
      const int id = flat_prev[r];
      if (id < 0 || id >= V) continue;
      float4* dst = reinterpret_cast<float4*>(dtable + (size_t)id * d);
      for (int c = lane; c < d4; c += 64) dst[c] = z;
    }
  }
}
}  // namespace vct

extern "C" int vct_embed_bwd(int dtype, int B, int S, int d, int V, const int64_t* ids, int64_t id_batch_stride,
                             int64_t pad_id, const void* dx, float* dtable, int32_t* id_ws, int64_t id_ws_ints, int incremental,
                             const uint32_t* seed, uint32_t site, float p_drop, void* stream) {
  if (!dt_ok(dtype) || !ids || !dx || !dtable || !id_ws) return VCT_E_ARG;
  if (B <= 0 || S <= 0 || d <= 0 || V <= 0) return VCT_E_SHAPE;
  if (d % vec_of(dtype) || d / vec_of(dtype) > 256 || (d % 4) || d > 1024) return VCT_E_SHAPE;   // one 16-byte column chunk per thread
  if (((uintptr_t)dtable & 15)) return VCT_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int N = B * S;
  int32_t* first_pos = id_ws;
  int32_t* cnt = id_ws + V;
  // [first_pos V | count V | hdr 4 (hdr[1] = positions of the last call) | ids of the last call, position order]: the prep kernel
  // runs FIRST and zeroes the rows of the ids the previous call left, then the index kernel overwrites them with this call's --
  // the same buffer every call, so a recorded launch list replays correctly any number of times
  int32_t* hdr = id_ws + 2 * (size_t)V;
  int32_t* flat = hdr + 4;
  if (id_ws_ints < 2 * (int64_t)V + 4 + N) return VCT_E_WORKSPACE;
  vct::launch(embed_prep_kernel, dim3(incremental ? 512 : 2048), dim3(256), 0, st, V, d, dtable, first_pos, cnt, flat, hdr,
              incremental ? 1 : 0);
  VCT_CHECK_LAUNCH();
  vct::launch(embed_index_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, S, ids, id_batch_stride, pad_id, first_pos, cnt, hdr, flat);
  VCT_CHECK_LAUNCH();
  const int vecb = vec_of(dtype);
  const int slots = 1024 / (d / vecb);
  const size_t heavy_lds = (size_t)(8 * 1024 + 8 * 16 + 256 + 3 * 1024 + 32) * sizeof(int) + (size_t)slots * d * sizeof(float);
  const dim3 grid(EMB_HEAVY_WGS + (N + 15) / 16);
  if (dtype == VCT_BF16) {
    static vct::DynLdsOptIn optin;
    if (optin.ensure((const void*)embed_bwd_kernel<bf16_t>, 160 * 1024) != hipSuccess) return VCT_E_SHAPE;
    vct::launch((embed_bwd_kernel<bf16_t>), grid, dim3(1024), heavy_lds, st, N, d, V, flat, (int32_t)pad_id, (const bf16_t*)dx, dtable,
                first_pos, cnt, seed, site, p_drop);
  } else {
    static vct::DynLdsOptIn optin;
    if (optin.ensure((const void*)embed_bwd_kernel<float>, 160 * 1024) != hipSuccess) return VCT_E_SHAPE;
    vct::launch((embed_bwd_kernel<float>), grid, dim3(1024), heavy_lds, st, N, d, V, flat, (int32_t)pad_id, (const float*)dx, dtable,
                first_pos, cnt, seed, site, p_drop);
  }
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_sce_loss(int dtype, int N, int S, int V, const void* logits, int64_t ldl, const int64_t* labels,
                            int64_t label_batch_stride, int64_t pad_id, float alpha, float* loss_out, void* dlogits,
                            int64_t ld_dl, float* row_ws, void* stream) {
  if (!dt_ok(dtype) || !logits || !labels || !loss_out || !row_ws) return VCT_E_ARG;
  if (N <= 0 || S <= 0 || V <= 0 || N % S) return VCT_E_SHAPE;
  const int vec = vec_of(dtype);
  const int64_t vround = ((int64_t)V + vec - 1) / vec * vec;
  if (ldl < vround || ldl % vec || ((uintptr_t)logits & 15)) return VCT_E_ALIGN;       // rows are read as 16-byte vectors
  if (dlogits && (ld_dl < vround || ld_dl % vec || ((uintptr_t)dlogits & 15))) return VCT_E_ALIGN;
  if (ld_dl > (int64_t)8 * 1024 * vec || vround > (int64_t)8 * 1024 * vec) return VCT_E_SHAPE;   // row must fit the register tile
  hipStream_t st = (hipStream_t)stream;
  vct::launch(count_valid_kernel, dim3(1), dim3(1024), 0, st, N, S, labels, label_batch_stride, pad_id, row_ws + 2 * (size_t)N);
  VCT_CHECK_LAUNCH();
  const int64_t width = dlogits ? (ld_dl > vround ? ld_dl : vround) : vround;
  const bool small = width <= (int64_t)4 * 1024 * vec;
#define VCT_SCE(T_, IT_, NT_) vct::launch((sce_loss_kernel<T_, IT_, NT_>), dim3(N), dim3(NT_), 0, st, N, S, V, (const T_*)logits, ldl, \
                                                 labels, label_batch_stride, pad_id, alpha, (T_*)dlogits, ld_dl, row_ws)
  // bf16, rows up to 4096 vectors: 512 threads x 8 vectors (4 workgroups per CU) -- measured in the step at V = 30522: 114 us against
  // 136 us for 1024 x 4 and 126 us for 256 x 16 (5.2 TB/s of HBM traffic)
  if (dtype == VCT_BF16) { if (small) VCT_SCE(bf16_t, 8, 512); else VCT_SCE(bf16_t, 8, 1024); }
  else { if (small) VCT_SCE(float, 4, 1024); else VCT_SCE(float, 8, 1024); }
#undef VCT_SCE
  VCT_CHECK_LAUNCH();
  vct::launch(sce_finalize_kernel, dim3(1), dim3(1024), 0, st, N, alpha, row_ws, loss_out);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

// read-only pass over a buffer (16 bytes per thread and step): pulls it into the memory-side cache.  The xor of everything read is
// compared with a value it cannot practically have, so that the loads are not dead code; nothing is ever written.
namespace vct {
__global__ __launch_bounds__(256) void warm_kernel(const uint4* __restrict__ src, const int64_t n16, unsigned int* sink) {
  uint4 a = {0u, 0u, 0u, 0u};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    const uint4 v = src[i];
    a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
  }
  if ((a.x ^ a.y) == 0x9e3779b9u && (a.z ^ a.w) == 0x7f4a7c15u && a.x == 0x85ebca6bu && sink != nullptr) *sink = a.w;
}
}  // namespace vct
extern "C" int vct_warm(const void* src, int64_t bytes, void* stream) {
  if (!src || ((uintptr_t)src & 15)) return VCT_E_ARG;
  if (bytes < 16) return VCT_E_SHAPE;
  const int64_t n16 = bytes / 16;
  const int64_t want = (n16 + 256 * 8 - 1) / (256 * 8);
  const int blocks = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  vct::launch(vct::warm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, n16, (unsigned int*)nullptr);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream) {
  if (!dt_ok(src_dtype) || !dt_ok(dst_dtype) || !src || !dst) return VCT_E_ARG;
  if (n <= 0) return VCT_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t want = (n / 4 + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  if (src_dtype == VCT_F32 && dst_dtype == VCT_BF16)
    vct::launch((cast_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == VCT_BF16 && dst_dtype == VCT_F32)
    vct::launch((cast_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == VCT_F32)
    vct::launch((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)src, (float*)dst, n);
  else
    vct::launch((cast_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_argmax_rows(int dtype, int rows, int cols, const void* x, int64_t ldx, int64_t* out, int64_t out_stride,
                               void* stream) {
  if (!dt_ok(dtype) || !x || !out) return VCT_E_ARG;
  if (rows <= 0 || cols <= 0 || out_stride <= 0) return VCT_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VCT_BF16)
    vct::launch((argmax_rows_kernel<bf16_t>), dim3(rows), dim3(AMX_THREADS), 0, st, cols, (const bf16_t*)x, ldx, out, out_stride,
                       (int64_t)0, (uint8_t*)nullptr, (int32_t*)nullptr, (unsigned long long*)nullptr, 0);
  else
    vct::launch((argmax_rows_kernel<float>), dim3(rows), dim3(AMX_THREADS), 0, st, cols, (const float*)x, ldx, out, out_stride,
                       (int64_t)0, (uint8_t*)nullptr, (int32_t*)nullptr, (unsigned long long*)nullptr, 0);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_greedy_select(int dtype, int rows, int cols, const void* x, int64_t ldx, int64_t* out, int64_t out_stride,
                                 int64_t end_id, uint8_t* ended, int32_t* ended_count, int64_t* all_ended_at, int32_t t,
                                 void* stream) {
  if (!dt_ok(dtype) || !x || !out || !ended || !ended_count || !all_ended_at) return VCT_E_ARG;
  if (rows <= 0 || cols <= 0 || out_stride <= 0 || t < 0) return VCT_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* at = reinterpret_cast<unsigned long long*>(all_ended_at);
  if (dtype == VCT_BF16)
    vct::launch((argmax_rows_kernel<bf16_t>), dim3(rows), dim3(AMX_THREADS), 0, st, cols, (const bf16_t*)x, ldx, out, out_stride,
                       end_id, ended, ended_count, at, (int)t);
  else
    vct::launch((argmax_rows_kernel<float>), dim3(rows), dim3(AMX_THREADS), 0, st, cols, (const float*)x, ldx, out, out_stride,
                       end_id, ended, ended_count, at, (int)t);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_gather_pad_rows(int out_dtype, int B, int Tmax, int E, const float* store, const int64_t* offsets,
                                   const int64_t* idx, void* out, uint8_t* mask, void* stream) {
  if (!dt_ok(out_dtype) || !store || !offsets || !idx || !out || !mask) return VCT_E_ARG;
  if (B <= 0 || Tmax <= 0 || E <= 0) return VCT_E_SHAPE;
  if ((E % 4 == 0) && ((((uintptr_t)store) & 15) || (((uintptr_t)out) & 15))) return VCT_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (B * Tmax + 3) / 4;
  if (out_dtype == VCT_BF16)
    vct::launch((gather_pad_rows_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, store, offsets, idx, B, Tmax, E, (bf16_t*)out, mask);
  else
    vct::launch((gather_pad_rows_kernel<float>), dim3(blocks), dim3(256), 0, st, store, offsets, idx, B, Tmax, E, (float*)out, mask);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_advance_seed(uint32_t* seed, void* stream) {
  if (!seed) return VCT_E_ARG;
  vct::launch(advance_seed_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, seed);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

// dst[c][r] = src[r][c] for 2-byte elements: 64x64 tiles through LDS.  Loads are 16-byte vectors along the source rows (128-byte
// row segments), each thread then gathers eight elements DOWN a column of the tile (lanes = consecutive columns: conflict-free
// 2-byte LDS reads) and stores them as one 16-byte vector of the destination row.  VEC: both leading dimensions multiples of 8 and
// both bases 16-byte aligned; otherwise the element-wise form.
template <bool VEC>
__global__ __launch_bounds__(256) void transpose16_kernel(int rows, int cols, const uint16_t* __restrict__ src, int64_t lds_,
                                                          uint16_t* __restrict__ dst, int64_t ldd) {
  constexpr int STR = 72;                                    // LDS row stride in elements (144 B: 16-byte aligned rows)
  __shared__ __attribute__((aligned(16))) uint16_t tile[64 * STR];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, t = threadIdx.x;
  struct alignas(16) V8 { uint16_t e[8]; };
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int row = (t >> 3) + 32 * i, ch = t & 7;
      const int r = r0 + row, c = c0 + ch * 8;
      V8 v;
#pragma unroll
      for (int j = 0; j < 8; j++) v.e[j] = 0;
      if (r < rows) {
        if (c + 8 <= cols) v = *reinterpret_cast<const V8*>(src + (size_t)r * lds_ + c);
        else for (int j = 0; j < 8; j++) if (c + j < cols) v.e[j] = src[(size_t)r * lds_ + c + j];
      }
      *reinterpret_cast<V8*>(tile + row * STR + ch * 8) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int cc = t & 63, ch = (t >> 6) + 4 * i;          // output row c0 + cc, its elements r0 + ch*8 .. +8
      V8 v;
#pragma unroll
      for (int j = 0; j < 8; j++) v.e[j] = tile[(ch * 8 + j) * STR + cc];
      const int c = c0 + cc, r = r0 + ch * 8;
      if (c < cols && r < rows) {
        if (r + 8 <= rows) *reinterpret_cast<V8*>(dst + (size_t)c * ldd + r) = v;
        else for (int j = 0; j < 8; j++) if (r + j < rows) dst[(size_t)c * ldd + r + j] = v.e[j];
      }
    }
  } else {
    const int tx = t & 63, ty = t >> 6;
    for (int i = ty; i < 64; i += 4) {
      const int r = r0 + i, c = c0 + tx;
      tile[i * STR + tx] = (r < rows && c < cols) ? src[(size_t)r * lds_ + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
      const int c = c0 + i, r = r0 + tx;
      if (c < cols && r < rows) dst[(size_t)c * ldd + r] = tile[tx * STR + i];
    }
  }
}

extern "C" int vct_transpose(int dtype, int rows, int cols, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, void* stream) {
  if (dtype != VCT_BF16 || src == nullptr || dst == nullptr) return VCT_E_ARG;
  if (rows <= 0 || cols <= 0 || ld_src < cols || ld_dst < rows) return VCT_E_SHAPE;
  const bool vec = (ld_src % 8 == 0) && (ld_dst % 8 == 0) && (((uintptr_t)src & 15) == 0) && (((uintptr_t)dst & 15) == 0);
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (vec) vct::launch(transpose16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, rows, cols,
                       reinterpret_cast<const uint16_t*>(src), ld_src, reinterpret_cast<uint16_t*>(dst), ld_dst);
  else vct::launch(transpose16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, rows, cols,
                   reinterpret_cast<const uint16_t*>(src), ld_src, reinterpret_cast<uint16_t*>(dst), ld_dst);
  VCT_CHECK_LAUNCH();
  return VCT_OK;
}

extern "C" int vct_abi_version(void) { return VCT_ABI_VERSION; }
extern "C" int vct_build_info(char* buf, int buflen) {
  static const char info[] = "libvct_hip gfx950 (CDNA4) abi 2; bf16 mfma 16x16x32 + f32 mfma 16x16x4";
  int n = (int)sizeof(info) - 1;
  if (buf && buflen > 0) {
    int c = n < buflen - 1 ? n : buflen - 1;
    for (int i = 0; i < c; i++) buf[i] = info[i];
    buf[c] = 0;
  }
  return n;
}
