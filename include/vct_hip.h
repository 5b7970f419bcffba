/*
 * vct_hip.h -- C ABI of libvct_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * video-caption Transformer training / greedy-decode path.
 *
 * The reference (Kamino666/Video-Captioning-Transformer) has no FFI for this path: every FLOP is a
 * stock torch.nn module call.  Each entry point below therefore cites the reference call site
 * (file:line under /root/reference) and the torch operator whose arithmetic it replaces.  A
 * reference maintainer binds them with ctypes; nothing here knows about torch.
 *
 * Conventions
 *  - every function returns int: 0 ok, <0 argument/shape error (VCT_E_*), >0 a hipError_t;
 *    nothing throws, nothing allocates, nothing synchronises the host: all work is enqueued on the
 *    caller's `stream` (a hipStream_t passed as void*), so every call is hipGraph-capturable.
 *  - all pointers are DEVICE pointers owned by the caller; tensors are row-major; "ld*" are leading
 *    dimensions in ELEMENTS.  Activations are VCT_F32 or VCT_BF16 ("compute dtype"); statistics,
 *    biases, LayerNorm parameters, losses and every parameter gradient are fp32.
 *  - leading dimensions of VCT_BF16 matrices must be multiples of 8 elements and base pointers
 *    16-byte aligned (VCT_F32: multiples of 4).  A K extent that is not a multiple of 8 (4) along a
 *    contiguous dimension must be zero-padded up to it by the producer (the SCE-loss kernel does
 *    this for the vocabulary dimension).
 *  - dropout: mask(site, idx) = hash(seed[0], site, idx) >= p * 2^32, scaled by 1/(1-p); `seed` is
 *    a DEVICE pointer (so a captured graph sees a new seed each replay); p == 0 or seed == NULL
 *    disables it.  The backward kernels recompute the same mask from (site, idx).
 */
#ifndef VCT_HIP_H
#define VCT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCT_ABI_VERSION 15

enum { VCT_F32 = 0, VCT_BF16 = 1 };
enum { VCT_ACT_NONE = 0, VCT_ACT_GELU = 1, VCT_ACT_RELU = 2 };
enum {
  VCT_OK = 0,
  VCT_E_ARG = -1,      /* null pointer / bad enum */
  VCT_E_SHAPE = -2,    /* unsupported or inconsistent shape */
  VCT_E_ALIGN = -3,    /* leading dimension / pointer alignment */
  VCT_E_WORKSPACE = -4 /* workspace too small */
};

int vct_abi_version(void);
/* writes "gfx950 ..." build string; returns its length */
int vct_build_info(char* buf, int buflen);

/* ---------------------------------------------------------------------------------------------
 * GEMM family: C[M,N] = epilogue(op(A)[M,K] * op(B)[K,N])   -- MFMA (bf16 16x16x32 / f32 16x16x4)
 *   ta == 0: A stored [M,K] (K contiguous)      ta == 1: A stored [K,M] (M contiguous)
 *   tb == 1: B stored [N,K] (nn.Linear weight)  tb == 0: B stored [K,N] (N contiguous)
 * replaces: nn.Linear forward (MMEncoder.py:246, CapDecoder.py:55, the linear1/linear2/in_proj/out_proj
 * inside nn.Transformer{En,De}coderLayer -- torch nn/modules/transformer.py:951-982,1143-1199 and
 * nn/functional.py:5785,6637) and its autograd backward (dX = dY*W: ta=0,tb=0; dW = dY^T*X: ta=1,tb=0).
 * --------------------------------------------------------------------------------------------- */
/* Optional OPTIMIZER epilogue of the weight-gradient form (dtype bf16, out fp32, ta=1, tb=0; vct_gemm and vct_gemm_grouped): the
 * gradient element g the product would store to C[r, c] is consumed in the epilogue registers by torch.optim.Adam's update of the
 * parameter element it belongs to,
 *     m += (1-b1)(g-m);  v = b2 v + (1-b2) g^2;  p = p (1 - lr wd) - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * (exactly vct_adam_step's arithmetic; t = *step + 1, hyper = {lr, beta1, beta2, eps, weight_decay} in DEVICE memory), and the
 * refreshed bf16 shadow / stream-order packed copy of the weight are written from the same registers.
 * replaces: `optimizer.step()` of the reference (train.py:24-26,126: torch.optim.Adam over the weight) for THIS matrix, and the
 * 4 B / parameter gradient store + re-load between `loss.backward()` (train.py:125) and it -- the optimizer's HBM pass over the
 * matrix runs inside the MFMA-bound product that produced its gradient instead of as a serial tail of the step.
 *   param / exp_avg / exp_avg_sq: fp32 [M, ldc] -- element (r, c) at r * ldc + c, the layout of C (views of flat buffers at C's offset).
 *   shadow: bf16 [M, ld_shadow] or NULL.  pk_*: the matrix's stream-order packed copy as in vct_adam_pack_seg (K = row length of the
 *   WHOLE weight, pk_row0 = row of the whole weight that C's row 0 is, pk_stream NULL = none).
 *   store_grad != 0: C is written as well (tests, gradient hooks); 0: C is left untouched.
 * Single GPU only: a data-parallel exchange has to average g first (trainer.ShardedExchange keeps the separate pass). */
typedef struct vct_gemm_adam {
  float* param; float* exp_avg; float* exp_avg_sq;
  void* shadow; int64_t ld_shadow;
  void* pk_stream; int32_t pk_K, pk_mode; int32_t pk_chunk0[4]; int32_t pk_row0; int32_t store_grad;
  const float* hyper; const int32_t* step;
} vct_gemm_adam;

typedef struct vct_gemm_desc {
  int32_t dtype;      /* VCT_F32 | VCT_BF16 : element type of A and B */
  int32_t out_dtype;  /* element type of C, preact, addend */
  int32_t ta, tb;
  int32_t M, N, K;
  int32_t act;        /* VCT_ACT_*: applied after bias */
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  const float* bias;                        /* [N] or NULL */
  void* preact; int64_t ld_preact;          /* optional: acc+bias before act (saved for backward) */
  const void* addend; int64_t ld_addend;    /* optional: C += addend (residual-gradient accumulate) */
  const void* dact_src; int64_t ld_dact;    /* optional: C = acc * act'(dact_src) (type = out_dtype) */
  const uint32_t* seed; uint32_t site; float p_drop; /* dropout on C (after act / inside dact) */
  float* bias_grad;                         /* optional [M] fp32: sum over K of op(A) (db for ta=1) */
  void* workspace; int64_t workspace_bytes; /* split-K partials (fp32); NULL = never split */
  int32_t split_k;                          /* 0 auto, 1 none, >1 forced */
  int32_t reserved;                         /* 0; kernel-selection override for tests / A-B probes: 1-9 (+10 x buffers) = a tile of
                                               the general kernel, 100 = force / 99 = forbid the persistent-tile layer kernel */
  int32_t n_tile_counters;                  /* ints available at tile_counters */
  int32_t* tile_counters;                   /* optional, see below */
  const vct_gemm_adam* adam;                /* optional (NULL = none): optimizer epilogue of the dW form, see vct_gemm_adam */
} vct_gemm_desc;
/* tile_counters: device ints that are ZERO on entry (they are zero again when the GEMM has finished, so one
 * zero-filled allocation serves every later call on the same stream).  With them a split-K GEMM reduces inside
 * the producing kernel -- the last workgroup to finish a tile sums the partials in fixed split order
 * (bit-reproducible) -- instead of in a second launch.  Needs one int per output tile (<= M*N/4096 + M/64 + N/64
 * + 1 is always enough); without (NULL / too few) the two-pass reduce is used.  GEMMs that may run CONCURRENTLY
 * (different streams) must not share counters or workspace. */
int vct_gemm(const vct_gemm_desc* d, void* stream);
/* bytes of workspace vct_gemm may use for this descriptor (0 if it will not split) */
int64_t vct_gemm_workspace_bytes(const vct_gemm_desc* d);

/* n (<= VCT_GEMM_GROUP_MAX) independent weight-gradient GEMMs in ONE launch: every descriptor must be the dW form
 * (dtype bf16, out fp32, ta=1, tb=0, no epilogue other than bias_grad).  replaces: the per-Linear
 * `grad_weight = grad_out.t() @ input` / `grad_bias = grad_out.sum(0)` autograd nodes of one Transformer layer
 * (torch autograd of F.linear as used by nn.TransformerEncoderLayer / DecoderLayer, MMEncoder.py:236,
 * CapDecoder.py:18), which the reference runs as 8-14 separate kernels per layer.
 * Each descriptor carries its OWN workspace (vct_gemm_grouped_workspace_bytes(descs, n, i) bytes) and its OWN
 * tile_counters (required whenever that is non-zero); split_k: 0 auto, >=1 forced. */
#define VCT_GEMM_GROUP_MAX 8
int vct_gemm_grouped(const vct_gemm_desc* descs, int32_t n, void* stream);
int64_t vct_gemm_grouped_workspace_bytes(const vct_gemm_desc* descs, int32_t n, int32_t i);

/* ---------------------------------------------------------------------------------------------
 * Multi-head attention core: O = softmax(Q K^T / sqrt(hd) + mask) V per (batch, head), one wave each,
 * QK^T and PV as MFMA tiles, K/V staged in LDS, row softmax by wave shuffles, dropout on P.
 * replaces: F.scaled_dot_product_attention + mask merge inside nn.MultiheadAttention
 * (torch nn/functional.py:6553-6570,6629) as used at MMEncoder.py:274, CapDecoder.py:49-52,70-75.
 *   q: rows b*Lq+i, cols h*hd..; k,v: rows b*Lk+j.  ld in elements.  causal: key j > query i masked.
 *   key_pad: uint8 / bool [B, Lk - key_pad_shift], 1 = padded key (masked) or NULL; the first key_pad_shift keys are never
 *   padded (the encoder passes the raw frame mask [B,T] with shift 1: key 0 is the aggregation token, MMEncoder.py:252-257).
 *   key_ids: alternative to key_pad -- int64 token ids, key j of batch b is padded iff key_ids[b*key_ids_bs + j] == pad_id
 *   (the decoder passes its id matrix: tgt_padding_mask[:, :-1] of CapDecoder.py:45-47 without materialising it).
 *   Limits: Lq, Lk <= 64, hd <= 128.
 * --------------------------------------------------------------------------------------------- */
typedef struct vct_attn_desc {
  int32_t dtype;
  int32_t B, H, Lq, Lk, hd;
  int32_t causal;
  int32_t key_pad_shift;
  const void* q; int64_t ldq;
  const void* k; int64_t ldk;
  const void* v; int64_t ldv;
  void* o; int64_t ldo;
  const uint8_t* key_pad;
  const uint32_t* seed; uint32_t site; float p_drop;
  /* backward only */
  const void* d_o; int64_t ld_do;
  void* dq; int64_t ld_dq;
  void* dk; int64_t ld_dk;
  void* dv; int64_t ld_dv;
  /* optional batch strides in ELEMENTS (0 = dense: L * ld).  A KV cache [B, Lmax, ...] read with
   * Lk < Lmax sets k_bs = v_bs = Lmax * ld. */
  int64_t q_bs, k_bs, v_bs, o_bs;
  const int64_t* key_ids; int64_t key_ids_bs; int64_t pad_id;
} vct_attn_desc;
int vct_attn_fwd(const vct_attn_desc* d, void* stream);
int vct_attn_bwd(const vct_attn_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sample-stationary Transformer stack forward (bf16): ONE launch = up to 4 whole encoder / decoder layers (and, with last != 0
 * in the last descriptor, the stack-final LayerNorm); one 512-thread workgroup per SAMPLE keeps that sample's rows in LDS
 * from the first layer's input to the last layer's output and streams the weights from L2 straight into MFMA operand
 * registers (csrc/vct_layer_ss.hip).  layers[0 .. n_layers): one descriptor per layer, identical in shape, masks, memory, seed
 * and p_drop; layers[0].x is the stack input (the other x fields are ignored: a layer reads its predecessor's output from
 * LDS); the packed weight streams lie back to back (layers[l].wpk = layers[0].wpk + l * nchunks * 64 KiB).
 * replaces: nn.TransformerEncoderLayer.forward (torch nn/modules/transformer.py:951-982) / nn.TransformerDecoderLayer.forward
 * (:1143-1199) as built at MMEncoder.py:236-238 / CapDecoder.py:18-20 and run by MMEncoder.py:274 / CapDecoder.py:49-52
 * (post-norm, gelu / relu, eps 1e-5; nn.MultiheadAttention = torch nn/functional.py:6206-6640), plus the final
 * nn.LayerNorm of the stack (MMEncoder.py:238, CapDecoder.py:20) -- i.e. the vct_gemm / vct_attn_fwd / vct_add_ln_fwd /
 * vct_add_ln_ln_fwd launches of one layer on the unfused path.  It SAVES exactly what those save (qkv, o, a, y, mean, rstd,
 * cross q / kv / o / a, pre-activation, dropped activation, f) and draws the same dropout counter streams
 * (site_*: attention probabilities, residual dropouts, feed-forward dropout), so the unfused backward kernels run behind it.
 *   x bf16 [B*L, 512] layer input (an OUTPUT when layers[0].pro != 0: the rows the prologue builds); mem bf16 [B*Lm, 512] (decoder layers; NULL = encoder layer: no cross-attention block,
 *   n2 unused).  Self-attention masks as vct_attn_desc (causal; key_pad + key_pad_shift; key_ids / pad_id).
 *   wpk: the layer's weights packed in STREAM ORDER by vct_ss_pack -- 64-KiB chunks (one 512-row block x 64 K columns of a
 *   weight, as 8 waves x 8 MFMA fragments of 1 KiB: wave w, column tile t < 4, k-step s < 2 = rows 64w + 16t .. +15,
 *   columns 32s .. 32s + 31 of the chunk), blocks in consumption order:
 *     in_proj rows [0,512) [512,1024) [1024,1536) | out_proj | (decoder: cross in_proj rows [0,512) | [512,1024) [1024,1536) |
 *     cross out_proj) | linear1 rows [0,512) | for j < ff/512: linear1 rows [512(j+1), 512(j+2)) (while j+1 < ff/512) ,
 *     linear2 columns [512j, 512j+512)        (the feed-forward block is software-pipelined: the GELU / dropout of chunk j is
 *     issued between the K steps of linear1's block j+1)
 *   each block 8 chunks (K = 512); nchunks = vct_layer_ss_stream_chunks(ff, cross).
 *   The feed-forward activation is built on the pre-activation as stored (bf16), which is also what the backward's GELU' reads.
 * vct_layer_ss_supported: bf16, d = 512, 8 heads of 64, ff a multiple of 512 and <= 2048, L <= 32, Lm <= 16 (Lm = 0: encoder layer).
 * vct_ss_pack: segs[i] = rows 0..511 x columns 0..64*nchunks-1 of the bf16 matrix at `w` (leading dimension ldw; the caller
 *   offsets `w` to the block) -> chunks dst_chunk .. dst_chunk+nchunks-1 of `dst`.  One launch per 48 segments.
 * --------------------------------------------------------------------------------------------- */
typedef struct vct_ss_norm { const float* gamma; const float* beta; void* y; float* mean; float* rstd; } vct_ss_norm;
typedef struct vct_layer_ss_desc {
  int32_t dtype, B, L, Lm, d, H, ff, act;
  int32_t last, causal, key_pad_shift, reserved;
  const void* wpk; int64_t nchunks;
  const void* x; const void* mem;
  const float* b_qkv; const float* b_o;
  void* qkv; void* o; void* a;
  vct_ss_norm n1;
  const float* b_cq; const float* b_ckv; const float* b_co;
  void* cq; void* ckv; void* co; void* ca;
  vct_ss_norm n2;
  const float* b1; const float* b2;
  void* hpre; void* h; void* f;
  vct_ss_norm n3;
  vct_ss_norm nf;
  const uint8_t* key_pad; const int64_t* key_ids; int64_t key_ids_bs; int64_t pad_id;
  const uint32_t* seed; float p_drop;
  uint32_t site_sa, site_n1, site_ca, site_n2, site_ff, site_n3;
  uint32_t site_emb;
  /* stack prologue (layers[0] only): pro = 0: x is the input.  pro = 1 (encoder stacks): x is BUILT from the frame features --
   * u = feats W_u^T + b_unify (the unify weight's 512 x 512 block = the FIRST 8 chunks of the stream, in front of layer 0's),
   * row 0 = mean_t(u) + pe_rows[0], row t+1 = u_t + pe_rows[t+1] (MMEncoder.py:246-271) -- stored to x, and the bf16 copy of fp32
   * features to x_in (what the unify weight gradient reads; NULL when feats are bf16).  pro = 2 (decoder stacks): x = dropout(
   * emb_table[emb_ids[b, s]] + emb_pos[s]) (CapDecoder.py:48, Embedding.py:23-25; counter stream of vct_embed_fwd, site_emb) -- stored to x. */
  int32_t pro, feats_dtype;
  const void* feats; void* x_in; const float* b_unify; const float* pe_rows;
  const int64_t* emb_ids; int64_t emb_ids_bs; const float* emb_table; const float* emb_pos;
} vct_layer_ss_desc;
typedef struct vct_ss_pack_seg { const void* w; int64_t ldw; int32_t nchunks; int32_t transposed; int64_t dst_chunk; } vct_ss_pack_seg;
int vct_layer_ss_supported(int dtype, int d, int H, int ff, int L, int Lm);
int64_t vct_layer_ss_stream_chunks(int ff, int cross);
int vct_ss_pack(const vct_ss_pack_seg* segs, int nseg, void* dst, void* stream);
int vct_layer_ss_fwd(const vct_layer_ss_desc* layers, int n_layers, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sample-stationary BACKWARD of a self-attention + feed-forward layer stack (the activation-gradient chain; bf16, d = 512, 8 heads,
 * rows <= 32, <= 4 layers): ONE launch, one workgroup per sample, top layer first.
 * replaces: the autograd nodes of nn.TransformerEncoder's layers between the gradient of the stack output and the gradient of the
 * stack input (torch nn/modules/transformer.py:951-982 as built at MMEncoder.py:236-238): LayerNorm backward, linear2 / linear1 input
 * gradients with GELU' and dropout, out_proj input gradient, attention backward, in_proj input gradient.  The weight gradients are
 * NOT computed here: the kernel stores d f, d hpre, d a, d qkv (what vct_gemm_grouped multiplies with the saved activations) and one
 * partial (dgamma | dbeta) row per sample and LayerNorm ([B][2][512] fp32, the layout vct_ln_param_finalize_batched sums).
 * layers[0] is the TOP layer (last != 0: the stack-final norm is in front of it, y_last = that layer's output, dy = gradient of the
 * normed output); layers[n-1].dx receives the gradient of the stack input.  wpk: the transposed weight blocks in stream order,
 * vct_layer_ss_bwd_stream_chunks(ff) chunks per layer, layers back to back in processing order:
 *   [linear2.weight^T block j (vct_ss_pack transposed segment: w = W2 + 512 j, ld = ff) | linear1.weight^T K slice j (w = W1 + 512 j * 512,
 *    ld = 512)] for j < ff/512 | out_proj.weight^T (8 chunks) | in_proj_weight^T (24 chunks).
 * x / qkv / a / x1 / hpre / f, the statistics and the dropout sites are the forward's (vct_layer_ss_desc: x, qkv, a, n1, hpre, f, n3).
 * --------------------------------------------------------------------------------------------- */
typedef struct vct_ss_bwd_norm { const float* gamma; const float* mean; const float* rstd; float* ws; } vct_ss_bwd_norm;
typedef struct vct_layer_ss_bwd_desc {
  int32_t dtype, B, L, d, H, ff, act, last, causal, key_pad_shift;
  const void* wpk; int64_t nchunks;
  const void* dy; void* dx; const void* y_last;
  const void* x; const void* qkv; const void* a; const void* x1; const void* hpre; const void* f;
  vct_ss_bwd_norm n1, n3, nf;
  void* df; void* dhpre; void* da; void* dqkv;
  const uint8_t* key_pad; const int64_t* key_ids; int64_t key_ids_bs; int64_t pad_id;
  const uint32_t* seed; float p_drop;
  uint32_t site_sa, site_n1, site_ff, site_n3;
} vct_layer_ss_bwd_desc;
int64_t vct_layer_ss_bwd_stream_chunks(int ff);
int vct_layer_ss_bwd(const vct_layer_ss_bwd_desc* layers, int n_layers, void* stream);

/* ---------------------------------------------------------------------------------------------
 * y = LayerNorm(res + dropout(x)) (eps 1e-5, biased variance, affine); res may be NULL (plain LN).
 * replaces: dropoutN + residual add + nn.LayerNorm (torch nn/modules/transformer.py:951-957,
 * 1143-1153) and the stack-final norms (MMEncoder.py:238, CapDecoder.py:20).
 * mean/rstd: fp32 [M] saved for backward.
 * bwd: ds = dLN(dy) (gradient of the pre-norm sum), dxo = dropout-masked ds (may alias ds when p==0,
 * may be NULL when res == NULL and p == 0); dgamma/dbeta fp32 [d] (overwritten), param_ws fp32
 * [2 * vct_ln_ws_rows(M) * d] scratch.
 * --------------------------------------------------------------------------------------------- */
int vct_add_ln_fwd(int dtype, int M, int d, const void* x, const void* res, const float* gamma,
                   const float* beta, void* y, float* mean, float* rstd, const uint32_t* seed,
                   uint32_t site, float p_drop, void* stream);
/* the same with the stack-final norm behind it, one launch: y2 = LayerNorm2(y) (gamma2 / beta2), built on the rows of y as
 * stored -- the last layer's norm2 / norm3 + transformer_encoder.norm / decoder.norm (MMEncoder.py:238, CapDecoder.py:20) */
int vct_add_ln_ln_fwd(int dtype, int M, int d, const void* x, const void* res, const float* gamma, const float* beta,
                      void* y, float* mean, float* rstd, const float* gamma2, const float* beta2, void* y2, float* mean2,
                      float* rstd2, const uint32_t* seed, uint32_t site, float p_drop, void* stream);
int vct_add_ln_bwd(int dtype, int M, int d, const void* dy, const void* x, const void* res,
                   const float* gamma, const float* mean, const float* rstd, void* ds, void* dxo,
                   float* dgamma, float* dbeta, float* param_ws, const uint32_t* seed, uint32_t site,
                   float p_drop, void* stream);
/* the backward of vct_add_ln_ln_fwd in one launch: dy2 = gradient of y2 = LayerNorm2(y) (gamma2, mean2, rstd2; y as stored), then
 * vct_add_ln_bwd of y = LayerNorm(res + dropout(x)) on the gradient of y ROUNDED to the activation type -- bit-identical to the two
 * vct_add_ln_bwd launches it replaces (the gradient of y is not stored).  param_ws2 / param_ws: one partial-row set per norm
 * (dgamma / dbeta always deferred to vct_ln_param_finalize_batched).  replaces the autograd nodes of decoder.norm / encoder.norm
 * (CapDecoder.py:20, MMEncoder.py:238) and of the last layer's closing norm under `loss.backward()` (train.py:125). */
int vct_add_ln_ln_bwd(int dtype, int M, int d, const void* dy2, const void* y, const float* gamma2, const float* mean2,
                      const float* rstd2, float* param_ws2, const void* x, const void* res, const float* gamma,
                      const float* mean, const float* rstd, void* ds, void* dxo, float* param_ws, const uint32_t* seed,
                      uint32_t site, float p_drop, void* stream);
int vct_ln_ws_rows(int M);
/* dgamma == dbeta == NULL in vct_add_ln_bwd defers the column reduction of param_ws; this call then finalizes
 * n_entries LayerNorms in ONE launch.  table_dev: DEVICE int64 [n_entries][4] = {param_ws ptr, dgamma ptr,
 * dbeta ptr, vct_ln_ws_rows(M)} (every param_ws must stay untouched until then). */
int vct_ln_param_finalize_batched(const int64_t* table_dev, int n_entries, int d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder front end after the `unify` GEMM: z[b,0] = mean_t u[b,t] (all T rows, pads included),
 * z[b,t+1] = u[b,t] + pe_rows[t+1]  (pe_rows fp32 [T+1,d], row 0 = 0).
 * replaces: GlobalAggregation('avg') + cat + TemporalEncoding add (MMEncoder.py:196-197,248-250,
 * 89-104,271).  bwd: du[b,t] = dz[b,t+1] + dz[b,0]/T.
 * --------------------------------------------------------------------------------------------- */
int vct_enc_frontend_fwd(int dtype, int B, int T, int d, const void* u, const float* pe_rows, void* z,
                         void* stream);
int vct_enc_frontend_bwd(int dtype, int B, int T, int d, const void* dz, void* du, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Token embedding: x[n] = dropout(table[ids[n]] + pos[n % S])   (no sqrt(d) scaling)
 * replaces: nn.Embedding(padding_idx) + PositionalEmbedding (CapDecoder.py:26,48; Embedding.py:23-25).
 * ids: int64 [N] read with element stride id_stride from ids + b*id_batch_stride (so the token-shift
 * view tgt[:, :-1] needs no copy): token n = (b = n / S, s = n % S) -> ids[b*id_batch_stride + s].
 * bwd: dtable fp32 [V,d] = scatter-add of dx rows (deterministic order), row pad_id zero.
 *      id_ws: int32 [id_ws_ints >= 2*V + 4 + B*S] scratch: first-occurrence / count tables (built with integer atomics) and the
 *      batch's ids in position order, which the NEXT call reads.  incremental != 0: dtable is known to be zero outside the rows of the ids of the PREVIOUS call (same
 *      dtable, same id_ws, nothing else writing dtable in between), so only those rows are zeroed (10 MB instead
 *      of 62.5 MB at cfg-B); incremental == 0 zeroes all V rows.
 * --------------------------------------------------------------------------------------------- */
int vct_embed_fwd(int dtype, int B, int S, int d, const int64_t* ids, int64_t id_batch_stride,
                  const void* table, const float* pos, void* x, const uint32_t* seed, uint32_t site,
                  float p_drop, void* stream);
int vct_embed_bwd(int dtype, int B, int S, int d, int V, const int64_t* ids, int64_t id_batch_stride,
                  int64_t pad_id, const void* dx, float* dtable, int32_t* id_ws, int64_t id_ws_ints, int incremental,
                  const uint32_t* seed, uint32_t site, float p_drop, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Symmetric cross-entropy loss + gradient w.r.t. logits, one workgroup per row, row held in registers
 * (one HBM read + one exp + one HBM write per logit).  Limits: ldl, ld_dl multiples of 8 (bf16) / 4 (fp32),
 * >= V rounded up to that, and <= 65536 (bf16) / 32768 (fp32) columns.
 * replaces: SCELoss.forward (loss.py:78-92) / nn.CrossEntropyLoss(ignore_index) when alpha == 1
 * (CapDecoder.py:28-32,56-59) and their autograd backward.
 *   logits [N, ldl] (V valid columns); labels int64: row n = (b = n / S, s = n % S) ->
 *   labels[b*label_batch_stride + s]  (so tgt[:, 1:] needs no copy).
 *   loss_out fp32 [1]; dlogits (may alias logits; columns V..ldl-1 are written as 0) or NULL.
 *   row_ws fp32 [2*N + 2] scratch.
 * --------------------------------------------------------------------------------------------- */
int vct_sce_loss(int dtype, int N, int S, int V, const void* logits, int64_t ldl, const int64_t* labels,
                 int64_t label_batch_stride, int64_t pad_id, float alpha, float* loss_out, void* dlogits,
                 int64_t ld_dl, float* row_ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batch assembly from a DEVICE-RESIDENT feature store (a split's clips packed into store [rows, E] fp32, clip i =
 * rows offsets[i] .. offsets[i+1]): out[b, t, :] = store[offsets[idx[b]] + t] for t < len(clip idx[b]), else 0;
 * mask[b, t] = 1 where padded.  out: [B, Tmax, E] of out_dtype (fp32, or bf16 = the encoder's compute type, which
 * saves its input cast), mask: uint8/bool [B, Tmax].  E % 4 == 0 takes the 16-byte path.
 * replaces: per-sample np.load + torch.tensor (dataloader.py:378-386), zero-pad + mask build on the host
 * (_make_mask_video, dataloader.py:233-247) and the per-step H2D copies (train.py:120-121).
 * --------------------------------------------------------------------------------------------- */
int vct_gather_pad_rows(int out_dtype, int B, int Tmax, int E, const float* store, const int64_t* offsets,
                        const int64_t* idx, void* out, uint8_t* mask, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused Adam / AdamW step over the flat fp32 parameter buffer + bf16 shadow refresh, one launch.
 * replaces: torch.optim.Adam(...).step() / AdamW when weight_decay != 0 (train.py:24-31,126); same
 * arithmetic as torch's single-tensor Adam (no amsgrad).  step_dev: DEVICE int32 counter of steps
 * already taken (read for the bias corrections, then incremented by a 1-thread kernel).
 * shadow_bf16 may be NULL; elements [shadow_skip_begin, shadow_skip_end) get no shadow (the token
 * embedding is gathered from the fp32 master; offsets relative to `param`).  n must be a multiple of 4.
 * A step may be applied range by range (each range as soon as its gradients are final, on a side stream):
 * pass bump_step = 0 for all ranges and finish with one call n = 0, bump_step = 1.
 * hyper_dev: optional DEVICE fp32 [5] = {lr, beta1, beta2, eps, weight_decay}; when given it overrides the scalar
 * arguments, so a captured graph / recorded launch list follows an LR schedule (scalars are frozen at record time).
 * --------------------------------------------------------------------------------------------- */
int vct_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int32_t* step_dev, int64_t shadow_skip_begin, int64_t shadow_skip_end, int32_t bump_step,
                  const float* hyper_dev, void* stream);
/* The same step over ONE 2-D weight [rows, cols] (cols a multiple of 64; contiguous) that also writes its TRANSPOSED bf16 shadow
 * shadow_t [cols][ld_t >= rows] -- W_g^T for the NT form of the vocabulary dX -- in the same pass (no step-counter bump). */
/* vct_adam_step that ALSO writes the stream-order packed copies (vct_ss_pack layout) of the weight matrices listed in segs_dev
 * (device array, sorted by `begin`; flat element indices of the WHOLE parameter buffer, `base` = index of param[0] in it):
 * mode 0 = the matrix [N, K] is packed as 512-row blocks, block i at chunk chunk0[i]; mode 1 = a 512-row matrix packed as 512-column
 * K slices, slice i at chunk0[i] (chunk0 < 0: not packed).  No reference counterpart (a layout copy refreshed with the shadow). */
typedef struct vct_adam_pack_seg { int64_t begin, end; int32_t K, mode; int32_t chunk0[4]; void* stream; } vct_adam_pack_seg;
int vct_adam_step_pk(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                     int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                     int32_t* step_dev, int64_t shadow_skip_begin, int64_t shadow_skip_end, int32_t bump_step,
                     const float* hyper_dev, const vct_adam_pack_seg* segs_dev, int32_t nseg, int64_t base, void* stream);
/* The same step over a LIST of flat ranges in ONE launch -- what is left for a separate pass when the weight matrices are stepped by
 * the optimizer epilogue of their own weight-gradient GEMMs (vct_gemm_adam): biases, LayerNorm parameters, the token-embedding table.
 * param / grad / exp_avg / exp_avg_sq / shadow_bf16: the WHOLE flat buffers; ranges_dev: device array sorted by begin, [begin, end)
 * flat elements (multiples of 4), blk0 = number of 4096-element workgroups of all earlier ranges, shadow != 0: write the bf16 shadow
 * (and the packed copies of segs_dev, flat indices as in vct_adam_step_pk with base 0); total_blocks = workgroups of all ranges.
 * No step-counter bump.  replaces torch.optim.Adam.step (train.py:24-26,126) for those parameters. */
typedef struct vct_adam_range { int64_t begin, end; int32_t blk0, shadow; } vct_adam_range;
int vct_adam_step_ranges(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                         const vct_adam_range* ranges_dev, int32_t nranges, int32_t total_blocks, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int32_t* step_dev, const float* hyper_dev,
                         const vct_adam_pack_seg* segs_dev, int32_t nseg, void* stream);
int vct_adam_step_2d(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                     void* shadow_t_bf16, int32_t rows, int32_t cols, int64_t ld_t, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int32_t* step_dev, const float* hyper_dev, void* stream);

/* elementwise helpers ------------------------------------------------------------------------ */
/* dst[i] = (dst_dtype) src[i], n elements (fp32 <-> bf16 parameter / feature casts) */
int vct_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream);
/* read-only pass over `bytes` bytes at src (16-byte aligned): brings a buffer that the NEXT launch streams (the packed weight streams
 * of the sample-stationary stacks, whose first touch is otherwise an HBM miss per chunk) into the memory-side cache.  No reference
 * counterpart (cache placement). */
int vct_warm(const void* src, int64_t bytes, void* stream);
/* first-index argmax per row (torch.max(dim=1) tie-break, MMT4Caption.py:165); out[row * out_stride] int64
 * (out_stride lets the decode loop write straight into column t of the id matrix ys[B, max_len]) */
/* dst[c, r] = src[r, c] (bf16 only): the transposed shadow of the vocabulary projection's weight, so that the input gradient
 * dX = dlogits W_g runs in the K-contiguous NT form (replaces nothing in the reference: autograd's `grad_output.mm(weight)`,
 * torch F.linear backward, reads W_g row-major). */
int vct_transpose(int dtype, int rows, int cols, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, void* stream);
int vct_argmax_rows(int dtype, int rows, int cols, const void* x, int64_t ldx, int64_t* out, int64_t out_stride,
                    void* stream);
/* one greedy-decode selection step: vct_argmax_rows into column t of the id matrix PLUS the loop's end bookkeeping
 * (MMT4Caption.py:166-171) on the device: ended[row] (uint8, sticky) is set when the row emits end_id; ended_count
 * counts such rows; the row that completes the set stores t into all_ended_at[0] (atomic min).  Integer atomics
 * only.  The host reads all_ended_at every few steps instead of syncing per token (`.tolist()`, MMT4Caption.py:168).
 * Before a decode: ended = 0, ended_count = 0, all_ended_at = max_len. */
int vct_greedy_select(int dtype, int rows, int cols, const void* x, int64_t ldx, int64_t* out, int64_t out_stride,
                      int64_t end_id, uint8_t* ended, int32_t* ended_count, int64_t* all_ended_at, int32_t t,
                      void* stream);
/* ---------------------------------------------------------------------------------------------
 * One stage of the greedy-decode step at SMALL batch (B <= 4): out[b, n] = epilogue(W[n, :] . prologue(...)[b, :]), a weight-
 * streaming matrix-vector kernel whose prologue builds the input vector and whose epilogue finishes the stage, so that the
 * embedding / LayerNorm / attention launches of the per-token step vanish into their consumers (csrc/vct_decode.hip):
 *   pro = VCT_DEC_PRO_NONE        x = x_in[b, 0:K)                                  (fp32)
 *         VCT_DEC_PRO_EMBED       x = table[ids[b*id_stride], :] + pos_row          (nn.Embedding + PositionalEmbedding, CapDecoder.py:64-66)
 *         VCT_DEC_PRO_LN          x = LayerNorm(x_in; g1, b1)                       (the previous stage stored the PRE-norm sum)
 *         VCT_DEC_PRO_LN_LN       x = LayerNorm(LayerNorm(x_in; g1, b1); g2, b2)    (last layer's norm3, then decoder.norm)
 *         VCT_DEC_PRO_SELF_ATTN / _CROSS_ATTN   x = MHA core of ONE query per batch row: q [B] rows (stride q_bs), keys / values
 *                                 kc / vc + b*kv_bs + l*kv_ld for l < Lk (<= 64), H heads of K/H columns; no mask (CapDecoder.py:70-75:
 *                                 the newest token attends to every cached position)
 *   epilogue: + bias[n], act, + res[b*ld_res + n] (fp32), stored as fp32 or -- out_native -- in the weight dtype (q|k|v into the KV-cache
 *   slot, logits).  x_out (optional, fp32 [B, K]): the prologue's vector, written once (the next stage's residual).
 *   W: [N, K] row-major, wdtype VCT_BF16 (K % 512 == 0) or VCT_F32 (K % 256 == 0), K <= 2048; KV cache and q in the same dtype.
 * replaces: per token and decoder layer, self_attn / multihead_attn (in_proj, SDPA, out_proj), linear1 / linear2, norm1-3 of
 * nn.TransformerDecoderLayer (torch nn/modules/transformer.py:1143-1199) as called from CapDecoder.decode_word (CapDecoder.py:62-79),
 * plus decoder.norm and the generator: 6 launches per layer + 1 instead of 11 + 2.
 * --------------------------------------------------------------------------------------------- */
enum { VCT_DEC_PRO_NONE = 0, VCT_DEC_PRO_EMBED = 1, VCT_DEC_PRO_LN = 2, VCT_DEC_PRO_LN_LN = 3, VCT_DEC_PRO_SELF_ATTN = 4,
       VCT_DEC_PRO_CROSS_ATTN = 5 };
typedef struct vct_decode_gemv_desc {
  int32_t wdtype, B, N, K;
  const void* W; int64_t ldw;
  const float* bias;
  int32_t pro, act;
  const float* x_in; int64_t ld_x;
  const float* g1; const float* b1; const float* g2; const float* b2;
  const int64_t* ids; int64_t id_stride; const float* table; const float* pos_row;
  const void* q; int64_t q_bs; const void* kc; const void* vc; int64_t kv_ld, kv_bs; int32_t H, Lk;
  const float* res; int64_t ld_res;
  void* out; int64_t ld_out; int32_t out_native; int32_t rows_per_wave;   /* rows_per_wave: 0 = auto */
  float* x_out;
} vct_decode_gemv_desc;
int vct_decode_gemv(const vct_decode_gemv_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One nn.Linear of the greedy-decode step at LARGER batch (2 <= M = captions in flight <= 256), applied to the M rows of the
 * current position:  out[M, N] = act(in W^T + bias) + res,  W bf16 [N, K] row-major (csrc/vct_gemm_skinny.hip: MFMA fragments
 * straight from L2, K split over the eight waves of a workgroup, 16 output columns per workgroup).  Input, one of
 *   x      bf16 rows (ldx)                                                    -- or --
 *   x_pre  fp32 PRE-norm rows: in = LayerNorm(x_pre; ln_g, ln_b, eps 1e-5), normalised in registers (K <= 1024); with x_norm
 *          the normalised rows are also stored (fp32, once) for the residual of a later call.  With ids (int64, row r at
 *          ids[r * id_stride]) the rows are instead EMBEDDED: in = x_pre[ids[r]] + ln_b, x_pre = the fp32 table [V, K] and ln_b the
 *          positional row (nn.Embedding + PositionalEmbedding, CapDecoder.py:64-66); ln_g is not read.
 * res: rows of res_dtype (VCT_F32 or VCT_BF16) added after the activation.  out: bf16 or fp32 rows (ldo: e.g. a KV-cache slot).
 * K % 8 == 0; pointers 16-byte aligned, leading dimensions multiples of 8 (bf16) / 4 (fp32).
 * replaces: norm1 / norm2 / norm3 + the Linear that consumes them, and the `x + sublayer(x)` adds, of nn.TransformerDecoderLayer
 * (torch nn/modules/transformer.py:1143-1199) as called per token from CapDecoder.decode_word (CapDecoder.py:62-79): 8 launches per
 * layer instead of 11.
 * --------------------------------------------------------------------------------------------- */
typedef struct vct_decode_linear_desc {
  int32_t M, N, K;
  int32_t out_dtype, act, res_dtype;
  const void* x; int64_t ldx;
  const float* x_pre; int64_t ld_pre;
  const float* ln_g; const float* ln_b;
  float* x_norm; int64_t ld_norm;
  const void* W; int64_t ldw;
  const float* bias;
  const void* res; int64_t ld_res;
  void* out; int64_t ldo;
  const int64_t* ids; int64_t id_stride;
} vct_decode_linear_desc;
int vct_decode_linear(const vct_decode_linear_desc* d, void* stream);
/* y[M, K] (bf16) = LayerNorm(LayerNorm(x; g1, b1); g2, b2) of fp32 rows (g2 = b2 = NULL: one LayerNorm): the last layer's norm3
 * followed by decoder.norm (CapDecoder.py:20) in front of the generator.  K <= 1024, K % 4 == 0. */
int vct_decode_ln2(int M, int K, const float* x, int64_t ldx, const float* g1, const float* b1, const float* g2, const float* b2,
                   void* y, int64_t ldy, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batch-1 greedy-decode step, one launch per BLOCK of a decoder layer (csrc/vct_decode_block.hip; bf16 weights, fp32 vectors):
 *   kind 0  self-attention block   x -> q|k|v of the new token (-> cache slot), attention over Lk cached keys (the last one is
 *                                  the new token), PARTIAL out-projection per head: part_out[h][d]
 *   kind 1  cross-attention block  x1 -> q, attention over the memory's Lk cached K/V rows, partial out-projection per head
 *   kind 2  feed-forward block     x2 -> act(W1 x2 + b1) per 64 hidden units, partial linear2 per unit group: part_out[ff/64][d]
 *   kind 3  generator              y -> logits fp32 [V] (part_out) (+ the greedy selection of the token, see sel_ws)
 * The input vector of every launch is built by each workgroup: the embedded token (id / table / pos_row) or
 * res + res_bias + sum_c part[c] (the previous block's residual, the bias of its second product, its partial vectors),
 * followed by LayerNorm(g1, b1) and LayerNorm(g2, b2) when given; x_out receives it (the next block's residual).
 * replaces: CapDecoder.decode_word for one caption (model/CapDecoder.py:62-79; torch nn/modules/transformer.py:1143-1199) --
 * 3 launches per layer instead of the 6 of vct_decode_gemv.  w_a / b_a: the first product's weight rows [*, d] and bias
 * (in_proj | q rows | linear1 | generator); w_b: the second product's weight TRANSPOSED, [*, d] = [in, out] (out_proj^T |
 * linear2^T: a workgroup's 64 input columns are then 64 contiguous rows).
 * vct_decode_block_supported: bf16, d = 512 with head_dim 64 (8 heads), ff a multiple of 64 (<= 2048), Lk <= 64.
 * --------------------------------------------------------------------------------------------- */
typedef struct vct_decode_block_desc {
  int32_t kind, d, ff, V, Lk, act;
  const int64_t* id; const float* table; const float* pos_row;
  const float* res; const float* res_bias; const float* part; int32_t n_part, pad0;
  const float* g1; const float* b1; const float* g2; const float* b2;
  float* x_out;
  const void* w_a; int64_t ld_a; const float* b_a;
  void* slot;
  const void* kc; const void* vc; int64_t kv_ld;
  const void* w_b; int64_t ld_b;
  float* part_out;
  /* kind 3 only, optional (sel_ws != NULL): the greedy selection of vct_greedy_select for this ONE caption inside the generator
   * launch -- every workgroup leaves its (max, first index) pair in sel_ws (fp32 [2 * ceil(V / 128) + 1], zero-initialised once:
   * the last word is a ticket counter the launch leaves at zero), the last one to finish writes tok_out[0] and the end-of-sequence
   * bookkeeping (ended[0], ended_count, all_ended_at = min(., t)). */
  float* sel_ws; int64_t* tok_out; int64_t end_id; uint8_t* ended; int32_t* ended_count; int64_t* all_ended_at; int32_t t, pad1;
} vct_decode_block_desc;
int vct_decode_block_supported(int dtype, int d, int H, int ff, int Lk);
int vct_decode_block(const vct_decode_block_desc* d, void* stream);

/* seed[0] += 1 (one-thread kernel, keeps the dropout stream advancing inside a captured graph) */
int vct_advance_seed(uint32_t* seed, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Host runtime: recorded launch lists, cross-stream ordering, live kernel timing (csrc/vct_runtime.hip).
 * replaces: the Python interpreter + torch dispatcher that issue the reference's step op by op (train.py:119-131);
 * there is no counterpart object in the reference.
 *
 * Launch list: between vct_cmdlist_begin and vct_cmdlist_end every vct_* call made ON THIS HOST THREAD is validated
 * and planned as usual but its kernel launches (and memsets, sync points, taps) are RECORDED instead of executed --
 * kernel, geometry, by-value argument copies and the stream they were aimed at.  vct_cmdlist_replay re-issues them in
 * the recorded order with eager semantics: launches aimed at `main_stream` of _begin go to `main_stream` of _replay,
 * every other stream is used as recorded (the caller keeps those streams alive).  Device pointers are baked: record
 * against static buffers.  A list is not thread safe; one recording per thread at a time.
 * --------------------------------------------------------------------------------------------- */
int vct_cmdlist_create(void** out_list);
int vct_cmdlist_destroy(void* list);
int vct_cmdlist_begin(void* list, void* main_stream);
int vct_cmdlist_end(void* list);
int vct_cmdlist_replay(void* list, void* main_stream);
int vct_cmdlist_size(void* list);     /* recorded commands */
int vct_cmdlist_streams(void* list);  /* distinct streams seen while recording */
/* A replayed command cannot hand a status back to the call that recorded it: the FIRST non-zero status any command of a
 * replay produces (a failed RCCL collective: ncclResult_t + 10000; a failed event record / wait / memset: hipError_t) is
 * kept and returned by that vct_cmdlist_replay.  vct_cmdlist_inject_status is the fault-injection hook of that path:
 * eager it returns `status`; while recording it appends a command on `stream` that reports `status` at every replay. */
int vct_cmdlist_inject_status(int status, void* stream);
/* Host work in launch order: fn(arg) runs on the calling host thread after everything enqueued on `stream` so far has
 * FINISHED (the stream is synchronised first) -- now, or at that point of every replay while recording.  A non-zero return
 * of fn is reported like any other command status.  For collectives that are host calls (gloo / torch.distributed in the
 * one-GPU multi-rank tests); RCCL collectives (vct_comm_*) are stream work and never need it. */
int vct_cmdlist_host_call(int (*fn)(void*), void* arg, void* stream);
/* Cross-stream ordering, recordable: `waiter` will not run work enqueued after this call before everything enqueued on
 * `signal` so far has finished (event record + stream wait; replaces torch's Stream.wait_stream on the fast path). */
int vct_stream_wait(void* waiter_stream, void* signal_stream);
/* Named sync points 0..63: vct_sync_record marks "everything enqueued on `stream` so far"; a later vct_sync_wait on any
 * stream waits for the most recent mark with that id. */
int vct_sync_record(int id, void* stream);
int vct_sync_wait(int id, void* stream);
/* ---------------------------------------------------------------------------------------------
 * Gradient exchange over RCCL / xGMI (csrc/vct_comm.hip).  One communicator per process (one process per GPU).
 * replaces: DistributedDataParallel's bucketed NCCL all-reduce (train.py:217-219, utils.py:137-146).
 *   vct_comm_unique_id: rank 0 fills 128 bytes; the caller ships them to every rank (any side channel).
 *   vct_comm_init: collective over all ranks; the communicator owns a HIP stream for its collectives.
 *   Collectives are IN PLACE on device buffers and asynchronous: enqueued on the communicator's stream, ordered behind
 *   everything enqueued so far on after_stream when order_after != 0 (after_stream may be the NULL stream):
 *     vct_comm_allreduce_avg       buf[0:count)              <- mean over ranks
 *     vct_comm_reduce_scatter_avg  buf[r*n : (r+1)*n)        <- mean over ranks of that slice (n = count_per_rank, r = own rank)
 *     vct_comm_all_gather          buf[q*n : (q+1)*n)        <- rank q's slice, for every q
 *     vct_comm_broadcast           buf[0:count)              <- root's
 *   dtype VCT_F32 | VCT_BF16.  vct_comm_wait(comm, s): stream s waits for everything issued on the communicator so far.
 *   vct_comm_stream: the communicator's stream (to enqueue the optimizer of an owned shard between a reduce-scatter and
 *   the all-gather of its result).  All of these are recordable into launch lists.
 *   Return codes >= 10000 are ncclResult_t + 10000.  vct_comm_available() == 0: no RCCL could be bound at run time.
 * --------------------------------------------------------------------------------------------- */
int vct_comm_available(void);
int vct_comm_unique_id(uint8_t* out128);
int vct_comm_init(const uint8_t* id128, int rank, int world, void** out_comm);
int vct_comm_destroy(void* comm);
int vct_comm_rank(void* comm);
int vct_comm_world(void* comm);
int vct_comm_stream(void* comm, void** out_stream);
int vct_comm_allreduce_avg(void* comm, void* buf, int64_t count, int dtype, void* after_stream, int order_after);
int vct_comm_reduce_scatter_avg(void* comm, void* buf, int64_t count_per_rank, int dtype, void* after_stream, int order_after);
int vct_comm_all_gather(void* comm, void* buf, int64_t count_per_rank, int dtype, void* after_stream, int order_after);
int vct_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* after_stream, int order_after);
int vct_comm_wait(void* comm, void* stream);

/* A stream whose kernels may only run on the compute units set in cu_mask (bit i of word i/32 = CU i; `words` 32-bit
 * words; words == 0: an ordinary non-blocking stream).  MI355X has 256 CUs in 8 XCDs; a throughput-bound kernel (the
 * vocabulary weight gradient, Adam) confined to a subset of the CUs runs BESIDE a latency-bound chain of small kernels
 * confined to the complement instead of starving it (hipExtStreamCreateWithCUMask).  The caller owns the stream. */
int vct_stream_create_masked(const uint32_t* cu_mask, int words, void** out_stream);
int vct_stream_destroy(void* stream);
/* Live kernel timing with HIP events on the launch stream (bench.py's roofline block): vct_tap(tag, 0, s) / vct_tap(tag,
 * 1, s) bracket a region of stream s; every bracket executed (eagerly or by a replay) while taps are enabled adds one
 * (start, end) pair to tag's pool.  vct_tap_collect waits for the pairs, writes their elapsed milliseconds
 * (<= cap values) and resets the tag.  tag 0..15.  Disabled taps cost nothing and are not recorded. */
int vct_tap_enable(int on);
int vct_tap(int tag, int phase, void* stream);
int vct_tap_collect(int tag, float* ms_out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* VCT_HIP_H */
