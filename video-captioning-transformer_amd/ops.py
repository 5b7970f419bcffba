"""Tensor-level wrappers over the C ABI (include/vct_hip.h).  PyTorch supplies device memory and the
current HIP stream; every function here enqueues hand-written gfx950 kernels and nothing else."""
from typing import Optional, Tuple

import torch

from . import _lib as L

Drop = Optional[Tuple[torch.Tensor, int, float]]  # (seed tensor uint32[1] on device, site id, p)

# Live timing of tagged launches (bench.py roofline): see taps_enable / tap / tap_collect at the end of this file --
# HIP event pairs recorded by the C runtime on the launch stream, so they also fire inside replayed launch lists.


def _drop(d: Drop):
    if d is None or d[2] <= 0.0:
        return 0, 0, 0.0
    return d[0].data_ptr(), int(d[1]), float(d[2])


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D view required"
    return t.stride(0)


class GemmScratch:
    """Split-K scratch of ONE stream: fp32 partials + the zero-initialised tile counters that let the split-K
    reduction happen inside the producing kernel (include/vct_hip.h, tile_counters).  GEMMs that may run
    concurrently must use different GemmScratch objects."""
    COUNTERS_PER_SLOT = 16384

    def __init__(self, device, nbytes: int = 64 << 20):
        self.ws = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        self.counters = torch.zeros(L.GEMM_GROUP_MAX * self.COUNTERS_PER_SLOT, dtype=torch.int32, device=device)


def _gemm_desc(a, b, out, ta, tb, bias, act, preact, addend, dact_src, dropout, bias_grad, workspace, split_k,
               n_valid, k_valid, m_valid, adam=None):
    d = L.GemmDesc()
    if adam is not None:      # L.GemmAdam: optimizer epilogue of the weight-gradient form (the caller keeps it alive through the call)
        d.adam = L.C.pointer(adam)
    d.dtype, d.out_dtype = L.dtype_code(a.dtype), L.dtype_code(out.dtype)
    assert b.dtype == a.dtype
    d.ta, d.tb = int(ta), int(tb)
    M = a.shape[1] if ta else a.shape[0]
    K = a.shape[0] if ta else a.shape[1]
    N = b.shape[0] if tb else b.shape[1]
    Kb = b.shape[1] if tb else b.shape[0]
    if k_valid is not None:
        K = k_valid
    else:
        assert K == Kb, (a.shape, b.shape, ta, tb)
    if n_valid is not None:
        N = n_valid
    if m_valid is not None:
        M = m_valid
    d.M, d.N, d.K = M, N, K
    d.act = L.ACT[act]
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), out.data_ptr(), _ld(out)
    d.bias = L.ptr(bias)
    if preact is not None:
        d.preact, d.ld_preact = preact.data_ptr(), _ld(preact)
    if addend is not None:
        d.addend, d.ld_addend = addend.data_ptr(), _ld(addend)
    if dact_src is not None:
        d.dact_src, d.ld_dact = dact_src.data_ptr(), _ld(dact_src)
    d.seed, d.site, d.p_drop = _drop(dropout)
    d.bias_grad = L.ptr(bias_grad)
    if isinstance(workspace, GemmScratch):
        d.workspace, d.workspace_bytes = workspace.ws.data_ptr(), workspace.ws.numel() * 4
        d.tile_counters, d.n_tile_counters = workspace.counters.data_ptr(), GemmScratch.COUNTERS_PER_SLOT
    elif workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    d.split_k = split_k
    return d


def gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, ta: bool = False, tb: bool = True,
         bias: Optional[torch.Tensor] = None, act: Optional[str] = None, preact: Optional[torch.Tensor] = None,
         addend: Optional[torch.Tensor] = None, dact_src: Optional[torch.Tensor] = None, dropout: Drop = None,
         bias_grad: Optional[torch.Tensor] = None, workspace=None, split_k: int = 0,
         n_valid: Optional[int] = None, k_valid: Optional[int] = None, m_valid: Optional[int] = None,
         tag: Optional[str] = None, tile: int = 0, adam=None) -> torch.Tensor:
    """out[M,N] = epilogue(op(a) @ op(b)).  ta=False: a is [M,K]; ta=True: a is [K,M].
    tb=True: b is [N,K] (nn.Linear weight); tb=False: b is [K,N].  n_valid / k_valid override the
    logical N / K when a buffer is wider than its valid extent (zero-padded vocabulary columns).
    workspace: a fp32 tensor (two-pass split-K) or a GemmScratch (single-pass split-K).
    adam: L.GemmAdam (weight-gradient form only) -- torch.optim.Adam's step on the matrix runs in the epilogue (include/vct_hip.h)."""
    lib = L.load()
    d = _gemm_desc(a, b, out, ta, tb, bias, act, preact, addend, dact_src, dropout, bias_grad, workspace, split_k,
                   n_valid, k_valid, m_valid, adam)
    d.reserved = tile       # kernel selection override (include/vct_hip.h, vct_gemm_desc.reserved): tests and A/B probes
    timed = _taps_on and tag in TAPS and (_taps_only is None or tag in _taps_only)
    if timed:
        tap(tag, 0)
    L.check(lib.vct_gemm(d, L.stream_ptr()), "vct_gemm")
    if timed:
        tap(tag, 1)
    return out


def gemm_grouped(items, scratch: GemmScratch, split_k: int = 0, tile: int = 0):
    """One launch for up to 8 weight-gradient GEMMs: items = [(dy [rows,M], x [rows,N], dW [M,N] fp32, db [M] fp32 or None[, L.GemmAdam or None])]
    -> dW = dy^T x, db = column sums of dy (include/vct_hip.h, vct_gemm_grouped); with a GemmAdam the optimizer steps the matrix in
    the epilogue."""
    lib = L.load()
    n = len(items)
    if not 1 <= n <= L.GEMM_GROUP_MAX:
        raise ValueError(f"gemm_grouped: 1..{L.GEMM_GROUP_MAX} problems per launch, got {n}")
    arr = (L.GemmDesc * n)()
    for i, it in enumerate(items):
        dy, x, dw, db = it[:4]
        d = _gemm_desc(dy, x, dw, True, False, None, None, None, None, None, None, db, None, split_k, None, None, None,
                       it[4] if len(it) > 4 else None)
        C_ = L.C
        C_.memmove(C_.byref(arr[i]), C_.byref(d), C_.sizeof(L.GemmDesc))
    arr[0].reserved = tile
    off = 0
    for i in range(n):
        need = int(lib.vct_gemm_grouped_workspace_bytes(arr, n, i))
        need = (need + 255) // 256 * 256
        arr[i].workspace, arr[i].workspace_bytes = scratch.ws.data_ptr() + off, need
        arr[i].tile_counters = scratch.counters.data_ptr() + 4 * i * GemmScratch.COUNTERS_PER_SLOT
        arr[i].n_tile_counters = GemmScratch.COUNTERS_PER_SLOT
        off += need
    if off > scratch.ws.numel() * 4:
        raise ValueError(f"gemm_grouped: scratch too small ({off} > {scratch.ws.numel() * 4} bytes)")
    L.check(lib.vct_gemm_grouped(arr, n, L.stream_ptr()), "vct_gemm_grouped")


def gemm_workspace_bytes(M: int, N: int, K: int, dtype: torch.dtype) -> int:
    """Upper bound of the split-K workspace the weight-gradient GEMM [M,N] (reduction K) may use."""
    lib = L.load()
    d = L.GemmDesc()
    d.dtype, d.out_dtype, d.ta, d.tb, d.M, d.N, d.K = L.dtype_code(dtype), L.F32, 1, 0, M, N, K
    return int(lib.vct_gemm_workspace_bytes(d))


def _attn_desc(dtype, B, H, Lq, Lk, hd, causal, q, k, v, key_pad, dropout):
    d = L.AttnDesc()
    d.dtype, d.B, d.H, d.Lq, d.Lk, d.hd, d.causal = L.dtype_code(dtype), B, H, Lq, Lk, hd, int(causal)
    d.q, d.ldq, d.k, d.ldk, d.v, d.ldv = q.data_ptr(), _ld(q), k.data_ptr(), _ld(k), v.data_ptr(), _ld(v)
    # key_pad: uint8/bool [B, Lk] | (mask [B, Lk - shift], shift) = first `shift` keys never padded | ("ids", ids [B, >=Lk] int64, pad_id)
    if isinstance(key_pad, tuple) and key_pad[0] == "ids":
        _tag, ids, pad_id = key_pad
        assert ids.dtype == torch.int64 and ids.stride(1) == 1 and ids.shape[1] >= Lk
        d.key_ids, d.key_ids_bs, d.pad_id = ids.data_ptr(), ids.stride(0), int(pad_id)
    elif isinstance(key_pad, tuple):
        m, shift = key_pad
        assert m.is_contiguous() and m.element_size() == 1 and tuple(m.shape) == (B, Lk - shift)
        d.key_pad, d.key_pad_shift = m.data_ptr(), int(shift)
    else:
        d.key_pad = L.ptr(key_pad)
    d.seed, d.site, d.p_drop = _drop(dropout)
    return d


def attn_fwd(q, k, v, o, B, H, Lq, Lk, causal=False, key_pad=None, dropout: Drop = None, kv_batch_stride: int = 0):
    """q:[B*Lq, >=H*hd] views (any row stride), k,v:[B*Lk, ..]; o:[B*Lq, H*hd].  key_pad: uint8 [B,Lk].
    kv_batch_stride (elements): K/V live in a cache [B, Lmax, ...] and only the first Lk rows of each batch are read."""
    hd = o.shape[1] // H
    d = _attn_desc(q.dtype, B, H, Lq, Lk, hd, causal, q, k, v, key_pad, dropout)
    d.o, d.ldo = o.data_ptr(), _ld(o)
    d.k_bs = d.v_bs = kv_batch_stride
    L.check(L.load().vct_attn_fwd(d, L.stream_ptr()), "vct_attn_fwd")
    return o


# ---- sample-stationary layer forward (csrc/vct_layer_ss.hip) ---------------------------------------------------------------------
_ss_ok = {}
SS_CHUNK = 32768          # bf16 elements per 64-KiB chunk of a packed weight stream


def layer_ss_supported(dtype, d: int, H: int, ff: int, L_: int, Lm: int) -> bool:
    if dtype != torch.bfloat16:
        return False
    key = (d, H, ff, L_, Lm)
    r = _ss_ok.get(key)
    if r is None:
        r = _ss_ok[key] = bool(L.load().vct_layer_ss_supported(L.BF16, d, H, ff, L_, Lm))
    return r


def layer_ss_stream_chunks(ff: int, cross: bool) -> int:
    return int(L.load().vct_layer_ss_stream_chunks(int(ff), int(cross)))


def ss_pack(blocks, dst: torch.Tensor):
    """blocks: [(w2d bf16 view whose rows 0..511 x columns 0..64*nchunks-1 form the block, nchunks, dst_chunk)] -> the packed stream
    `dst` (bf16, stream order: include/vct_hip.h, vct_ss_pack).  One launch per 48 blocks."""
    n = len(blocks)
    segs = (L.SsPackSeg * n)()
    for i, blk in enumerate(blocks):
        w, nch, dc = blk[:3]
        tr = bool(blk[3]) if len(blk) > 3 else False        # transposed block: element (n, k) of the stream's matrix is w[k][n]
        if tr:
            assert w.dtype == torch.bfloat16 and w.stride(1) == 1 and w.shape[1] >= 512 and w.shape[0] >= 64 * nch
        else:
            assert w.dtype == torch.bfloat16 and w.stride(1) == 1 and w.shape[0] >= 512 and w.shape[1] >= 64 * nch
        segs[i].w, segs[i].ldw, segs[i].nchunks, segs[i].dst_chunk = w.data_ptr(), w.stride(0), int(nch), int(dc)
        segs[i].transposed = int(tr)
    L.check(L.load().vct_ss_pack(segs, n, dst.data_ptr(), L.stream_ptr()), "vct_ss_pack")
    return dst


def layer_ss_bwd_stream_chunks(ff: int) -> int:
    return int(L.load().vct_layer_ss_bwd_stream_chunks(int(ff)))


def layer_ss_bwd_desc(*, B, Lr, wpk, nchunks, ff, act, H, x, qkv, a, x1, hpre, f, n1, n3, outs, sites, nf=None, y_last=None, dy=None, dx=None,
                      causal=False, key_pad=None, seed=None, p_drop=0.0):
    """Descriptor of ONE layer of a sample-stationary backward launch (include/vct_hip.h, vct_layer_ss_bwd_desc).  nX = (gamma, mean,
    rstd, ws[B, 2, 512] fp32); outs = (df, dhpre, da, dqkv); sites = (self-attention probabilities, norm1, feed-forward, last norm)."""
    q = L.LayerSsBwdDesc()
    q.dtype, q.B, q.L, q.d, q.H, q.ff, q.act = L.BF16, int(B), int(Lr), x.shape[1], int(H), int(ff), L.ACT[act]
    q.last, q.causal = int(nf is not None), int(causal)
    q.wpk, q.nchunks = wpk.data_ptr(), int(nchunks)
    q.dy, q.dx, q.y_last = L.ptr(dy), L.ptr(dx), L.ptr(y_last)
    q.x, q.qkv, q.a, q.x1, q.hpre, q.f = (t.data_ptr() for t in (x, qkv, a, x1, hpre, f))

    def norm(dst, t):
        dst.gamma, dst.mean, dst.rstd, dst.ws = (z.data_ptr() for z in t)
    norm(q.n1, n1)
    norm(q.n3, n3)
    if nf is not None:
        norm(q.nf, nf)
    q.df, q.dhpre, q.da, q.dqkv = (t.data_ptr() for t in outs)
    q.key_pad_shift = 0
    if isinstance(key_pad, tuple) and key_pad[0] == "ids":
        _tag, ids, pad_id = key_pad
        q.key_ids, q.key_ids_bs, q.pad_id = ids.data_ptr(), ids.stride(0), int(pad_id)
    elif isinstance(key_pad, tuple):
        m, shift = key_pad
        q.key_pad, q.key_pad_shift = m.data_ptr(), int(shift)
    elif key_pad is not None:
        q.key_pad = key_pad.data_ptr()
    q.seed, q.p_drop = L.ptr(seed), float(p_drop)
    q.site_sa, q.site_n1, q.site_ff, q.site_n3 = (int(z) for z in sites)
    return q


def layer_ss_bwd(descs):
    """One launch: the activation-gradient chain of a self-attention + feed-forward stack, descs[0] = the TOP layer."""
    n = len(descs)
    arr = (L.LayerSsBwdDesc * n)(*descs)
    L.check(L.load().vct_layer_ss_bwd(arr, n, L.stream_ptr()), "vct_layer_ss_bwd")


def layer_ss_desc(*, B, Lr, x, wpk, nchunks, ff, act, H, bias, sa, n1, ffn, n3, cross=None, n2=None, nf=None, mem=None, Lm=0,
                  causal=False, key_pad=None, seed=None, p_drop=0.0, sites=(0, 0, 0, 0, 0, 0), frontend=None, embed=None):
    """Descriptor of ONE layer of a sample-stationary stack launch (include/vct_hip.h, vct_layer_ss_desc).  bias = dict(qkv, o,
    [cq, ckv, co], l1, l2) fp32 vectors; sa = (qkv, o, a); cross = (q, kv, o, a); ffn = (hpre, h, f); nX = (gamma, beta, y, mean, rstd);
    sites = dropout sites (self-attention probabilities, norm1, cross-attention probabilities, norm2, feed-forward, norm3)."""
    q = L.LayerSsDesc()
    q.dtype, q.B, q.L, q.Lm, q.d, q.H, q.ff, q.act = L.BF16, int(B), int(Lr), int(Lm), x.shape[1], int(H), int(ff), L.ACT[act]
    q.last, q.causal = int(nf is not None), int(causal)
    q.wpk, q.nchunks, q.x, q.mem = wpk.data_ptr(), int(nchunks), x.data_ptr(), L.ptr(mem)

    def norm(dst, t):
        dst.gamma, dst.beta, dst.y, dst.mean, dst.rstd = (z.data_ptr() for z in t)
    q.b_qkv, q.b_o = bias["qkv"].data_ptr(), bias["o"].data_ptr()
    q.qkv, q.o, q.a = (t.data_ptr() for t in sa)
    norm(q.n1, n1)
    if cross is not None:
        q.b_cq, q.b_ckv, q.b_co = bias["cq"].data_ptr(), bias["ckv"].data_ptr(), bias["co"].data_ptr()
        q.cq, q.ckv, q.co, q.ca = (t.data_ptr() for t in cross)
        norm(q.n2, n2)
    q.b1, q.b2 = bias["l1"].data_ptr(), bias["l2"].data_ptr()
    q.hpre, q.h, q.f = (t.data_ptr() for t in ffn)
    norm(q.n3, n3)
    if nf is not None:
        norm(q.nf, nf)
    if isinstance(key_pad, tuple) and key_pad[0] == "ids":
        _tag, ids, pad_id = key_pad
        assert ids.dtype == torch.int64 and ids.stride(1) == 1 and ids.shape[1] >= Lr
        q.key_ids, q.key_ids_bs, q.pad_id = ids.data_ptr(), ids.stride(0), int(pad_id)
    elif isinstance(key_pad, tuple):
        m, shift = key_pad
        assert m.is_contiguous() and m.element_size() == 1 and tuple(m.shape) == (B, Lr - shift)
        q.key_pad, q.key_pad_shift = m.data_ptr(), int(shift)
    elif key_pad is not None:
        q.key_pad = key_pad.data_ptr()
    if seed is not None and p_drop > 0.0:
        q.seed, q.p_drop = seed.data_ptr(), float(p_drop)
    q.site_sa, q.site_n1, q.site_ca, q.site_n2, q.site_ff, q.site_n3 = (int(s) for s in sites)
    if frontend is not None:       # (feats [B*T, 512] fp32 / bf16, x_in bf16 copy or None, unify bias, PE' rows [T+1, 512]): x is built and stored
        feats, x_in, b_u, pe_rows = frontend
        q.pro, q.feats_dtype, q.feats, q.x_in = 1, L.dtype_code(feats.dtype), feats.data_ptr(), L.ptr(x_in)
        q.b_unify, q.pe_rows = b_u.data_ptr(), pe_rows.data_ptr()
    if embed is not None:          # (ids [B, >= L] int64, fp32 table, fp32 positional rows, dropout site): x is built and stored
        ids, table, pos, site = embed
        assert ids.dtype == torch.int64 and ids.stride(1) == 1 and ids.shape[1] >= Lr
        q.pro, q.emb_ids, q.emb_ids_bs, q.emb_table, q.emb_pos, q.site_emb = 2, ids.data_ptr(), ids.stride(0), table.data_ptr(), pos.data_ptr(), int(site)
    return q


def layer_ss_fwd(descs):
    """A whole stack (list of layer_ss_desc results, first layer first) in ceil(n / 4) launches (vct_layer_ss_fwd)."""
    arr = (L.LayerSsDesc * len(descs))(*descs)
    L.check(L.load().vct_layer_ss_fwd(arr, len(descs), L.stream_ptr()), "vct_layer_ss_fwd")


def attn_bwd(q, k, v, d_o, dq, dk, dv, B, H, Lq, Lk, causal=False, key_pad=None, dropout: Drop = None):
    hd = d_o.shape[1] // H
    d = _attn_desc(q.dtype, B, H, Lq, Lk, hd, causal, q, k, v, key_pad, dropout)
    d.d_o, d.ld_do = d_o.data_ptr(), _ld(d_o)
    d.dq, d.ld_dq, d.dk, d.ld_dk, d.dv, d.ld_dv = dq.data_ptr(), _ld(dq), dk.data_ptr(), _ld(dk), dv.data_ptr(), _ld(dv)
    L.check(L.load().vct_attn_bwd(d, L.stream_ptr()), "vct_attn_bwd")


def add_ln_fwd(x, res, gamma, beta, y, mean, rstd, dropout: Drop = None):
    M, dm = x.shape
    s, site, p = _drop(dropout)
    L.check(L.load().vct_add_ln_fwd(L.dtype_code(x.dtype), M, dm, x.data_ptr(), L.ptr(res), gamma.data_ptr(),
                                    beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), s, site, p,
                                    L.stream_ptr()), "vct_add_ln_fwd")
    return y


def add_ln_ln_fwd(x, res, gamma, beta, y, mean, rstd, gamma2, beta2, y2, mean2, rstd2, dropout: Drop = None):
    """y = LayerNorm(res + dropout(x)); y2 = LayerNorm2(y) -- a layer's last norm and the stack-final norm in one launch."""
    M, dm = x.shape
    s, site, p = _drop(dropout)
    L.check(L.load().vct_add_ln_ln_fwd(L.dtype_code(x.dtype), M, dm, x.data_ptr(), L.ptr(res), gamma.data_ptr(), beta.data_ptr(),
                                       y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma2.data_ptr(), beta2.data_ptr(),
                                       y2.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), s, site, p, L.stream_ptr()),
            "vct_add_ln_ln_fwd")
    return y2


def ln_ws_rows(M: int) -> int:
    return int(L.load().vct_ln_ws_rows(M))


def add_ln_bwd(dy, x, res, gamma, mean, rstd, ds, dxo, dgamma, dbeta, param_ws, dropout: Drop = None):
    M, dm = x.shape
    s, site, p = _drop(dropout)
    L.check(L.load().vct_add_ln_bwd(L.dtype_code(x.dtype), M, dm, dy.data_ptr(), x.data_ptr(), L.ptr(res),
                                    gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ds.data_ptr(), L.ptr(dxo),
                                    L.ptr(dgamma), L.ptr(dbeta), param_ws.data_ptr(), s, site, p,
                                    L.stream_ptr()), "vct_add_ln_bwd")


def add_ln_ln_bwd(dy2, y, gamma2, mean2, rstd2, param_ws2, x, res, gamma, mean, rstd, ds, dxo, param_ws, dropout: Drop = None):
    """Backward of add_ln_ln_fwd in one launch (include/vct_hip.h, vct_add_ln_ln_bwd): the stack-final norm, then the last layer's norm."""
    M, dm = x.shape
    s, site, p = _drop(dropout)
    L.check(L.load().vct_add_ln_ln_bwd(L.dtype_code(x.dtype), M, dm, dy2.data_ptr(), y.data_ptr(), gamma2.data_ptr(), mean2.data_ptr(),
                                       rstd2.data_ptr(), param_ws2.data_ptr(), x.data_ptr(), L.ptr(res), gamma.data_ptr(), mean.data_ptr(),
                                       rstd.data_ptr(), ds.data_ptr(), L.ptr(dxo), param_ws.data_ptr(), s, site, p, L.stream_ptr()),
            "vct_add_ln_ln_bwd")


def ln_param_finalize_batched(table, n_entries, d):
    """table: int64 device tensor [n_entries, 4] = (param_ws ptr, dgamma ptr, dbeta ptr, ws rows)."""
    L.check(L.load().vct_ln_param_finalize_batched(table.data_ptr(), n_entries, d, L.stream_ptr()), "vct_ln_param_finalize_batched")


def enc_frontend_fwd(u, pe_rows, z, B, T):
    L.check(L.load().vct_enc_frontend_fwd(L.dtype_code(u.dtype), B, T, u.shape[1], u.data_ptr(), pe_rows.data_ptr(),
                                          z.data_ptr(), L.stream_ptr()), "vct_enc_frontend_fwd")
    return z


def enc_frontend_bwd(dz, du, B, T):
    L.check(L.load().vct_enc_frontend_bwd(L.dtype_code(dz.dtype), B, T, dz.shape[1], dz.data_ptr(), du.data_ptr(),
                                          L.stream_ptr()), "vct_enc_frontend_bwd")
    return du


def embed_fwd(ids, S, table, pos, x, dropout: Drop = None):
    """ids: int64 [B, S_total] (row stride = ids.stride(0)); uses the first S columns of each row."""
    B = ids.shape[0]
    s, site, p = _drop(dropout)
    L.check(L.load().vct_embed_fwd(L.dtype_code(x.dtype), B, S, x.shape[1], ids.data_ptr(), ids.stride(0),
                                   table.data_ptr(), pos.data_ptr(), x.data_ptr(), s, site, p, L.stream_ptr()),
            "vct_embed_fwd")
    return x


_embed_ws = {}
_embed_ws_retired = []


def embed_bwd(ids, S, pad_id, dx, dtable, dropout: Drop = None, id_ws=None, exclusive: bool = False, on_grow=None):
    """dtable fp32 [V, d] = deterministic scatter-add of the dx rows.  exclusive=True: the caller guarantees that nothing but
    this function writes `dtable` between calls (the single-GPU fast training path: gradients are overwritten every step, never
    accumulated or averaged in place) -- from the second exclusive call on, only the rows the previous call wrote are zeroed
    (10 MB at cfg-B) instead of all V (62.5 MB).  Any non-exclusive call falls back to the full zero-fill and breaks the chain."""
    B = ids.shape[0]
    s, site, p = _drop(dropout)
    V = dtable.shape[0]
    incremental = False
    if id_ws is None:
        # one workspace per gradient table: it remembers the ids of the LAST call into that table (what the next incremental call
        # zeroes) -- shared between two models it made a replayed step of one zero the rows of the other's batch
        key = (V, dx.device, dtable.data_ptr())
        ent = _embed_ws.get(key)
        need = 2 * V + 4 + max(B * S, 65536)
        if ent is None or ent[0].numel() < need:
            # [first_pos V | count V | header 4 (positions of the last call) | ids in position order N]; room
            # for 65536 positions up front (recordings bake its address); an outgrown one stays alive for the recordings that use it
            if ent is not None:
                _embed_ws_retired.append(ent[0])
                # recordings made against the old workspace carry incremental = 1 and ITS "previous ids": replayed after a step on
                # the new workspace they would zero the wrong rows -> the owner drops them (engine: ctx.generation += 1)
                if on_grow is not None:
                    on_grow()
            ent = _embed_ws[key] = [torch.zeros(need, dtype=torch.int32, device=dx.device), None]
        id_ws = ent[0]
        incremental = exclusive and ent[1] == dtable.data_ptr()
        ent[1] = dtable.data_ptr() if exclusive else None
    else:
        assert id_ws.numel() >= 2 * V + 4 + B * S
    L.check(L.load().vct_embed_bwd(L.dtype_code(dx.dtype), B, S, dx.shape[1], V, ids.data_ptr(),
                                   ids.stride(0), int(pad_id), dx.data_ptr(), dtable.data_ptr(), id_ws.data_ptr(), id_ws.numel(), int(incremental),
                                   s, site, p, L.stream_ptr()), "vct_embed_bwd")


def sce_loss(logits, V, labels, S, pad_id, alpha, loss_out, dlogits, row_ws):
    """logits [N, ld>=V]; labels: int64 2-D view [B, >=S] whose first S columns are the targets."""
    N = logits.shape[0]
    L.check(L.load().vct_sce_loss(L.dtype_code(logits.dtype), N, S, V, logits.data_ptr(), _ld(logits), labels.data_ptr(),
                                  labels.stride(0), int(pad_id), float(alpha), loss_out.data_ptr(), L.ptr(dlogits),
                                  _ld(dlogits) if dlogits is not None else 0, row_ws.data_ptr(), L.stream_ptr()),
            "vct_sce_loss")
    return loss_out


def warm(t: torch.Tensor):
    """Read-only pass over a tensor: pulls it into the memory-side cache for the launch that streams it next (include/vct_hip.h, vct_warm)."""
    L.check(L.load().vct_warm(t.data_ptr(), t.numel() * t.element_size(), L.stream_ptr()), "vct_warm")


def cast(src, dst):
    L.check(L.load().vct_cast(L.dtype_code(src.dtype), L.dtype_code(dst.dtype), src.data_ptr(), dst.data_ptr(),
                              src.numel(), L.stream_ptr()), "vct_cast")
    return dst


def argmax_rows(x, out, cols=None):
    """out: int64 1-D tensor (any stride): out[row] = first index of the row maximum."""
    L.check(L.load().vct_argmax_rows(L.dtype_code(x.dtype), x.shape[0], cols or x.shape[1], x.data_ptr(), _ld(x),
                                     out.data_ptr(), out.stride(0), L.stream_ptr()), "vct_argmax_rows")
    return out


def advance_seed(seed):
    L.check(L.load().vct_advance_seed(seed.data_ptr(), L.stream_ptr()), "vct_advance_seed")


def adam_step(param, grad, exp_avg, exp_avg_sq, shadow, lr, beta1, beta2, eps, weight_decay, step_dev, skip=(0, 0), bump=True,
              hyper=None, pack=None):
    """Fused Adam/AdamW over flat fp32 buffers (+ bf16 shadow refresh).  step_dev: int32[1] device counter.
    `param` may be a slice of the flat buffer (range-by-range stepping): pass the same slice of every buffer,
    `skip` relative to the slice start, bump=False, and finish with adam_bump(step_dev).  hyper: device fp32 [5]
    (lr, beta1, beta2, eps, weight_decay) read by the kernel instead of the scalars (graph / launch-list replays)."""
    if pack is not None:      # (device table of vct_adam_pack_seg, entries, flat index of param[0]): packed weight copies written in the same pass
        table, nseg, base = pack
        L.check(L.load().vct_adam_step_pk(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), L.ptr(shadow),
                                          param.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                          step_dev.data_ptr(), int(skip[0]), int(skip[1]), int(bool(bump)), L.ptr(hyper),
                                          table.data_ptr(), int(nseg), int(base), L.stream_ptr()), "vct_adam_step_pk")
        return
    L.check(L.load().vct_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), L.ptr(shadow),
                                   param.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                   step_dev.data_ptr(), int(skip[0]), int(skip[1]), int(bool(bump)), L.ptr(hyper), L.stream_ptr()),
            "vct_adam_step")


def adam_ranges_table(ranges, device):
    """Device table for adam_step_ranges: ranges = sorted [(begin, end, has_shadow)] of flat elements (multiples of 4).
    Returns (table tensor, entries, workgroups)."""
    EPB = 4096
    arr = (L.AdamRange * len(ranges))()
    blk = 0
    for i, (a, b, sh) in enumerate(ranges):
        assert a % 4 == 0 and b % 4 == 0 and b > a
        arr[i].begin, arr[i].end, arr[i].blk0, arr[i].shadow = int(a), int(b), blk, int(bool(sh))
        blk += (b - a + EPB - 1) // EPB
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    return table, len(ranges), blk


def adam_step_ranges(param, grad, exp_avg, exp_avg_sq, shadow, table, lr, beta1, beta2, eps, weight_decay, step_dev, hyper=None, pack=None):
    """vct_adam_step over a LIST of ranges of the WHOLE flat buffers in one launch (table from adam_ranges_table): what is left for a
    separate pass when the weight matrices are stepped inside their weight-gradient GEMMs.  pack = (device table of
    vct_adam_pack_seg with flat indices, entries) or None.  No step-counter bump."""
    tab, n, blocks = table
    pk, nseg = (pack[0].data_ptr(), int(pack[1])) if pack is not None and pack[1] else (0, 0)
    L.check(L.load().vct_adam_step_ranges(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), L.ptr(shadow),
                                          tab.data_ptr(), int(n), int(blocks), float(lr), float(beta1), float(beta2), float(eps),
                                          float(weight_decay), step_dev.data_ptr(), L.ptr(hyper), pk, nseg, L.stream_ptr()),
            "vct_adam_step_ranges")


def adam_step_2d(param, grad, exp_avg, exp_avg_sq, shadow, shadow_t, lr, beta1, beta2, eps, weight_decay, step_dev, hyper=None):
    """vct_adam_step on one 2-D weight (views [rows, cols] of the flat buffers) that also writes the transposed bf16 shadow
    shadow_t [cols, ld >= rows] in the same pass."""
    rows, cols = param.shape
    L.check(L.load().vct_adam_step_2d(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), L.ptr(shadow),
                                      shadow_t.data_ptr(), rows, cols, shadow_t.stride(0), float(lr), float(beta1), float(beta2),
                                      float(eps), float(weight_decay), step_dev.data_ptr(), L.ptr(hyper), L.stream_ptr()),
            "vct_adam_step_2d")


def adam_bump(step_dev):
    L.check(L.load().vct_adam_step(step_dev.data_ptr(), step_dev.data_ptr(), step_dev.data_ptr(), step_dev.data_ptr(), 0, 0,
                                   0.0, 0.0, 0.0, 0.0, 0.0, step_dev.data_ptr(), 0, 0, 1, 0, L.stream_ptr()), "vct_adam_step(bump)")


def greedy_select(x, out, end_id: int, ended, ended_count, all_ended_at, t: int, cols=None):
    """argmax_rows into `out` (column t of the id matrix) + the decode loop's end-of-sequence bookkeeping, one launch."""
    L.check(L.load().vct_greedy_select(L.dtype_code(x.dtype), x.shape[0], cols or x.shape[1], x.data_ptr(), _ld(x),
                                       out.data_ptr(), out.stride(0), int(end_id), ended.data_ptr(), ended_count.data_ptr(),
                                       all_ended_at.data_ptr(), int(t), L.stream_ptr()), "vct_greedy_select")
    return out


def decode_gemv(W, out, B, *, bias=None, pro="none", x_in=None, ln1=None, ln2=None, embed=None, attn=None, act=None, res=None,
                out_native=False, ld_out=None, x_out=None, n_valid=None):
    """One stage of the small-batch greedy-decode step (include/vct_hip.h, vct_decode_gemv).  W [N, K]; out: tensor whose row b starts
    at out.data_ptr() + b*ld_out elements.  embed = (ids view [B] (any stride), table fp32, pos_row fp32 [K]);
    attn = (q view [B, K], k view, v view, kv_ld, kv_bs, H, Lk) -- all in W's dtype; ln1 / ln2 = (gamma, beta)."""
    d = L.DecodeGemvDesc()
    d.wdtype, d.B, d.N, d.K = L.dtype_code(W.dtype), B, (n_valid or W.shape[0]), W.shape[1]
    d.W, d.ldw, d.bias = W.data_ptr(), _ld(W), L.ptr(bias)
    d.pro, d.act = L.DEC_PRO[pro], L.ACT[act]
    if x_in is not None:
        d.x_in, d.ld_x = x_in.data_ptr(), x_in.stride(0)
    if ln1 is not None:
        d.g1, d.b1 = ln1[0].data_ptr(), ln1[1].data_ptr()
    if ln2 is not None:
        d.g2, d.b2 = ln2[0].data_ptr(), ln2[1].data_ptr()
    if embed is not None:
        ids, table, pos_row = embed
        d.ids, d.id_stride, d.table, d.pos_row = ids.data_ptr(), ids.stride(0), table.data_ptr(), pos_row.data_ptr()
    if attn is not None:
        q, k, v, kv_ld, kv_bs, H, Lk = attn
        d.q, d.q_bs, d.kc, d.vc, d.kv_ld, d.kv_bs, d.H, d.Lk = q.data_ptr(), q.stride(0), k.data_ptr(), v.data_ptr(), kv_ld, kv_bs, H, Lk
    if res is not None:
        d.res, d.ld_res = res.data_ptr(), res.stride(0)
    d.out, d.ld_out, d.out_native = out.data_ptr(), (ld_out if ld_out is not None else out.stride(0)), int(out_native)
    d.x_out = L.ptr(x_out)
    L.check(L.load().vct_decode_gemv(d, L.stream_ptr()), "vct_decode_gemv")


def decode_block(kind: str, d: int, *, w_a, b_a, part_out, embed=None, res=None, res_bias=None, part=None, ln1=None, ln2=None, x_out=None,
                 slot=None, kc=None, vc=None, kv_ld=0, Lk=1, w_b=None, ff=0, act=None, V=0, select=None):
    """One block of the batch-1 greedy-decode step (include/vct_hip.h, vct_decode_block): kind in {'self', 'cross', 'ffn', 'gen'}.
    embed = (id view [1] int64, table fp32 [V, d], pos_row fp32 [d]) or the vector res + res_bias + sum(part) (fp32 [d] / [n, d])."""
    q = L.DecodeBlockDesc()
    q.kind, q.d, q.ff, q.V, q.Lk, q.act = {"self": 0, "cross": 1, "ffn": 2, "gen": 3}[kind], d, ff, V, Lk, L.ACT[act]
    if embed is not None:
        ids, table, pos_row = embed
        q.id, q.table, q.pos_row = ids.data_ptr(), table.data_ptr(), pos_row.data_ptr()
    q.res, q.res_bias = L.ptr(res), L.ptr(res_bias)
    if part is not None:
        q.part, q.n_part = part.data_ptr(), part.shape[0]
    if ln1 is not None:
        q.g1, q.b1 = ln1[0].data_ptr(), ln1[1].data_ptr()
    if ln2 is not None:
        q.g2, q.b2 = ln2[0].data_ptr(), ln2[1].data_ptr()
    q.x_out = L.ptr(x_out)
    q.w_a, q.ld_a, q.b_a = w_a.data_ptr(), _ld(w_a), b_a.data_ptr()
    q.slot, q.kc, q.vc, q.kv_ld = L.ptr(slot), L.ptr(kc), L.ptr(vc), kv_ld
    if w_b is not None:
        q.w_b, q.ld_b = w_b.data_ptr(), _ld(w_b)
    q.part_out = part_out.data_ptr()
    if select is not None:      # kind 'gen': (sel_ws fp32 [2 * ceil(V / 128) + 1] zeroed, out id view [1], end_id, ended, ended_count, all_ended_at, t)
        ws, out, end_id, ended, ended_count, all_ended_at, t = select
        assert ws.numel() >= 2 * ((V + 127) // 128) + 1
        q.sel_ws, q.tok_out, q.end_id, q.ended = ws.data_ptr(), out.data_ptr(), int(end_id), ended.data_ptr()
        q.ended_count, q.all_ended_at, q.t = ended_count.data_ptr(), all_ended_at.data_ptr(), int(t)
    L.check(L.load().vct_decode_block(q, L.stream_ptr()), "vct_decode_block")


_db_ok = {}


def decode_block_supported(dtype, d: int, H: int, ff: int, Lk: int) -> bool:
    if dtype != torch.bfloat16:
        return False
    key = (d, H, ff, Lk)
    r = _db_ok.get(key)
    if r is None:
        r = _db_ok[key] = bool(L.load().vct_decode_block_supported(L.BF16, d, H, ff, Lk))
    return r


def transpose(src: torch.Tensor, dst: torch.Tensor):
    """dst[c, r] = src[r, c] (bf16; row strides free)."""
    L.check(L.load().vct_transpose(L.dtype_code(src.dtype), src.shape[0], src.shape[1], src.data_ptr(), src.stride(0),
                                   dst.data_ptr(), dst.stride(0), L.stream_ptr()), "vct_transpose")
    return dst


def decode_linear(W, out, *, x=None, x_pre=None, ln=None, x_norm=None, bias=None, act=None, res=None, n_valid=None, embed=None):
    """One nn.Linear of the batched greedy-decode step on the M <= 256 rows of the current position (include/vct_hip.h,
    vct_decode_linear): out = act(in W^T + bias) + res with in = x (bf16 rows) or LayerNorm(x_pre; *ln) of fp32 pre-norm rows
    (x_norm: fp32 buffer that receives the normalised rows), or the embedded token rows: embed = (ids view [M] (any stride),
    table fp32 [V, K], pos_row fp32 [K]).  out / res: row views (any row stride)."""
    d = L.DecodeLinearDesc()
    d.M, d.N, d.K = out.shape[0], (n_valid or W.shape[0]), W.shape[1]
    d.out_dtype, d.act = L.dtype_code(out.dtype), L.ACT[act]
    if x is not None:
        d.x, d.ldx = x.data_ptr(), x.stride(0)
    if x_pre is not None:
        d.x_pre, d.ld_pre = x_pre.data_ptr(), x_pre.stride(0)
        d.ln_g, d.ln_b = ln[0].data_ptr(), ln[1].data_ptr()
    if embed is not None:
        ids, table, pos_row = embed
        d.x_pre, d.ld_pre, d.ln_b = table.data_ptr(), table.stride(0), pos_row.data_ptr()
        d.ids, d.id_stride = ids.data_ptr(), ids.stride(0)
    if x_norm is not None:
        d.x_norm, d.ld_norm = x_norm.data_ptr(), x_norm.stride(0)
    d.W, d.ldw, d.bias = W.data_ptr(), _ld(W), L.ptr(bias)
    if res is not None:
        d.res, d.ld_res, d.res_dtype = res.data_ptr(), res.stride(0), L.dtype_code(res.dtype)
    d.out, d.ldo = out.data_ptr(), out.stride(0)
    L.check(L.load().vct_decode_linear(d, L.stream_ptr()), "vct_decode_linear")
    return out


def decode_ln2(x, ln1, ln2, y):
    """y (bf16 [M, K]) = LayerNorm(LayerNorm(x; *ln1); *ln2) of fp32 rows; ln2 = None: one LayerNorm."""
    g2, b2 = (ln2[0].data_ptr(), ln2[1].data_ptr()) if ln2 is not None else (None, None)
    L.check(L.load().vct_decode_ln2(x.shape[0], x.shape[1], x.data_ptr(), x.stride(0), ln1[0].data_ptr(), ln1[1].data_ptr(), g2, b2,
                                    y.data_ptr(), y.stride(0), L.stream_ptr()), "vct_decode_ln2")
    return y


def gather_pad_rows(store: torch.Tensor, offsets: torch.Tensor, idx: torch.Tensor, tmax: int, out_dtype=torch.float32):
    """store fp32 [rows, E] (packed clips), offsets int64 [n+1], idx int64 [B] -> (feat [B, tmax, E], mask bool [B, tmax])."""
    B, E = idx.numel(), store.shape[1]
    out = torch.empty(B, tmax, E, dtype=out_dtype, device=store.device)
    mask = torch.empty(B, tmax, dtype=torch.bool, device=store.device)
    L.check(L.load().vct_gather_pad_rows(L.dtype_code(out_dtype), B, tmax, E, store.data_ptr(), offsets.data_ptr(), idx.data_ptr(),
                                         out.data_ptr(), mask.data_ptr(), L.stream_ptr()), "vct_gather_pad_rows")
    return out, mask


# ---- host runtime (csrc/vct_runtime.hip): cross-stream ordering, launch lists, live timing ----------------------------
def _sptr(stream) -> int:
    return stream.cuda_stream if stream is not None else L.stream_ptr()


def stream_wait(waiter, signal):
    """`waiter` (torch stream, None = current) waits for everything enqueued so far on `signal`.  Recordable."""
    L.check(L.load().vct_stream_wait(_sptr(waiter), _sptr(signal)), "vct_stream_wait")


def sync_record(ident: int, stream=None):
    L.check(L.load().vct_sync_record(int(ident), _sptr(stream)), "vct_sync_record")


def sync_wait(ident: int, stream=None):
    L.check(L.load().vct_sync_wait(int(ident), _sptr(stream)), "vct_sync_wait")


import threading

_rec = threading.local()     # .ll = the LaunchList recording on THIS thread (the C recorder is thread_local too): host-side collectives
                             # issued on another thread while this one records run immediately, as they must


def is_recording() -> bool:
    return getattr(_rec, "ll", None) is not None


_host_fns = []          # CFUNCTYPE objects of eager host calls made outside any recording (kept alive for the duration of the call only)


def host_call(fn, stream=None):
    """Run the Python callable `fn()` on the host once everything enqueued on `stream` (None = current) so far has finished --
    immediately, or at this point of every replay while a launch list records (include/vct_hip.h, vct_cmdlist_host_call).  For
    host-side collectives (torch.distributed / gloo) inside a recorded step; an exception in fn becomes status 1 of the replay."""
    def thunk(_arg):
        try:
            fn()
            return 0
        except Exception:                     # never unwind through the C frames of the replay loop
            import traceback
            traceback.print_exc()
            return 1
    cfn = L.HOST_FN(thunk)
    ll = getattr(_rec, "ll", None)
    if ll is not None:
        ll._thunks.append(cfn)            # the recording owns its thunks (and the tensors they capture): released with the list
    L.check(L.load().vct_cmdlist_host_call(L.C.cast(cfn, L.vp), None, _sptr(stream)), "vct_cmdlist_host_call")


def masked_stream(cu_bits, device=None):
    """torch stream restricted to the compute units whose indices are in `cu_bits` (iterable of ints < 256); an empty /
    None selection gives an ordinary stream.  Wraps vct_stream_create_masked; the stream lives as long as the process."""
    words = (L.u32 * 8)()
    n = 0
    for c in (cu_bits or ()):
        words[c // 32] |= (1 << (c % 32))
        n += 1
    h = L.vp()
    L.check(L.load().vct_stream_create_masked(words, 8 if n else 0, L.C.byref(h)), "vct_stream_create_masked")
    return torch.cuda.ExternalStream(h.value, device=device)


TAPS = {"gen_fwd": 0, "gen_dx": 1, "gen_dw": 2, "layers_fwd": 3, "loss": 4, "adam": 5, "step": 6,
        # data-parallel exchange (trainer.ShardedExchange): what the compute stream WAITS for the communicator at the end of a step,
        # and how long each gradient bucket (reduce-scatter -> Adam on the shard -> all-gather -> cast) occupies the communicator's stream
        "comm_wait": 7, **{f"comm_b{i}": 8 + i for i in range(8)},
        # the two sample-stationary stack launches of the forward (encoder incl. its front end / decoder incl. the token embedding)
        "ss_enc": 16, "ss_dec": 17}
_taps_on = False
_taps_only = None        # None = every tag, else the set of tags that are bracketed


def taps_enable(on: bool, only=None):
    """Switch the live timing brackets on / off; `only`: an iterable of tags to restrict them to.  Every bracket is two HIP event
    records in the step's stream -- seven brackets cost the cfg-B step ~55 us (tools/taps_cost.py) -- so a benchmark keeps only the
    one it needs inside its timed region.  Recorded launch lists / graphs contain the brackets that were active when they were
    made: drop them (CaptionTrainer.drop_recordings) after changing this."""
    global _taps_on, _taps_only
    _taps_on = bool(on)
    _taps_only = None if only is None else frozenset(only)
    L.check(L.load().vct_tap_enable(int(_taps_on)), "vct_tap_enable")


def tap(tag: str, phase: int, stream=None):
    """Bracket a region of `stream` with HIP timing events (phase 0 = start, 1 = end); no-op while taps are disabled."""
    if _taps_on and (_taps_only is None or tag in _taps_only):
        L.check(L.load().vct_tap(TAPS[tag], phase, _sptr(stream)), "vct_tap")


def tap_collect(tag: str, cap: int = 4096):
    """Elapsed milliseconds of every bracket of `tag` executed since the last collect (waits for them)."""
    buf = (L.f32 * cap)()
    n = L.load().vct_tap_collect(TAPS[tag], buf, cap)
    if n < 0:
        L.check(n, "vct_tap_collect")
    return [float(buf[i]) for i in range(n)]


class LaunchList:
    """One recorded step: `with ll.record(): <issue the step>` captures every vct_* launch made on this thread (nothing
    executes), `ll.replay()` re-issues them from C.  Launches aimed at the stream that was current at record time go to
    the stream current at replay time."""

    def __init__(self):
        h = L.vp()
        L.check(L.load().vct_cmdlist_create(L.C.byref(h)), "vct_cmdlist_create")
        self._h = h
        self._thunks = []                 # host-call thunks of this recording (ops.host_call)

    class _Rec:
        def __init__(self, ll):
            self.ll = ll

        def __enter__(self):
            L.check(L.load().vct_cmdlist_begin(self.ll._h, L.stream_ptr()), "vct_cmdlist_begin")
            self.ll._thunks = []          # a re-recording replaces the commands, and with them the thunks they call
            _rec.ll = self.ll
            return self.ll

        def __exit__(self, *exc):
            _rec.ll = None
            L.check(L.load().vct_cmdlist_end(self.ll._h), "vct_cmdlist_end")
            return False

    def record(self):
        return LaunchList._Rec(self)

    def replay(self):
        L.check(L.load().vct_cmdlist_replay(self._h, L.stream_ptr()), "vct_cmdlist_replay")

    def __len__(self):
        return int(L.load().vct_cmdlist_size(self._h))

    @property
    def n_streams(self):
        return int(L.load().vct_cmdlist_streams(self._h))

    def close(self):
        if self._h is not None and self._h.value:
            L.load().vct_cmdlist_destroy(self._h)
            self._h = None
            self._thunks = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
