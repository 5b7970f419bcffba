"""CPU oracle for the video-caption training / greedy-decode hot path.

TEST INFRASTRUCTURE ONLY.  This module is a numpy restatement of the reference's algorithm
(Kamino666/Video-Captioning-Transformer) for the one path BASELINE.json names.  It is imported
only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, and only as the checker
(or as the timed CPU baseline) -- never by the product package, which must fail loudly when the
HIP extension is missing.

Pinning: the reference holds no tests or golden vectors for this path, so
this oracle is pinned against outputs of the reference itself: oracle/make_golden.py imports
/root/reference in the build container, runs its unmodified modules on CPU and commits the
input/output vectors under tests/golden/; tests/test_oracle_golden.py checks every function here
against them.

The arithmetic of the reference lives in torch.nn (third-party, torch 2.10.0 CPU kernels); the
citations below give (a) the reference call site and (b) the torch source whose published
semantics are restated.

All tensors are numpy arrays; `dt` selects float32 (default, the reference's dtype) or float64.
Parameters are a dict keyed by the reference's state_dict() names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
from scipy.special import erf as _erf

P = Dict[str, np.ndarray]

NEG_LOG_1E4_F32 = float(-np.log(np.float32(1e-4)))  # loss.py:86-88: -log(clamp(onehot,1e-4,1)) off-target
ENC = "video_encoder."
DEC = "cap_decoder."


# --------------------------------------------------------------------------------------------
# tables and masks
# --------------------------------------------------------------------------------------------
def decoder_pos_table(maxlen: int, d: int) -> np.ndarray:
    """model/Embedding.py:13-17 -- sin/cos table, den = exp(-arange(0,d,2)*ln(10000)/d) in fp32."""
    den = np.exp(-np.arange(0, d, 2, dtype=np.float32) * np.float32(math.log(10000)) / np.float32(d)).astype(np.float32)
    pos = np.arange(0, maxlen, dtype=np.float32).reshape(maxlen, 1)
    tab = np.zeros((maxlen, d), np.float32)
    tab[:, 0::2] = np.sin(pos * den)
    tab[:, 1::2] = np.cos(pos * den)
    return tab


def encoder_pos_table(max_len: int, d: int) -> np.ndarray:
    """model/MMEncoder.py:71-81 -- same table, div_term = exp(arange(0,d,2) * -(ln(10000)/d))."""
    div = np.exp(np.arange(0, d, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / d))).astype(np.float32)
    pos = np.arange(0, max_len, dtype=np.float32).reshape(max_len, 1)
    tab = np.zeros((max_len, d), np.float32)
    tab[:, 0::2] = np.sin(pos * div)
    tab[:, 1::2] = np.cos(pos * div)
    return tab[None]  # [1, max_len, d] like the registered buffer `pe`


def generate_square_subsequent_mask(sz: int) -> np.ndarray:
    """utils.py:63-66 -- float [sz,sz]: 0 on/below the diagonal, -inf above."""
    m = np.zeros((sz, sz), np.float32)
    m[np.triu_indices(sz, k=1)] = -np.inf
    return m


def temporal_encoding_rows(pe: np.ndarray, T: int) -> np.ndarray:
    """model/MMEncoder.py:83-104 (separate=False, one modality): row 0 (aggregation token) = 0,
    row i+1 = pe[idx_i], idx = linspace(0, D-1, t).astype(int32) with D = t = T."""
    idx = np.linspace(0, T - 1, T).astype(np.int32)
    out = np.zeros((T + 1, pe.shape[-1]), pe.dtype)
    out[1:] = pe[0, idx, :]
    return out


# --------------------------------------------------------------------------------------------
# primitive ops (forward + backward)
# --------------------------------------------------------------------------------------------
def linear(x, w, b):
    """nn.Linear: y = x W^T + b, W:[out,in]."""
    return x @ w.T + b


def linear_bwd(dy, x, w):
    dx = dy @ w
    dy2 = dy.reshape(-1, dy.shape[-1])
    x2 = x.reshape(-1, x.shape[-1])
    return dx, dy2.T @ x2, dy2.sum(0)


def gelu(x):
    """F.gelu exact erf form (torch nn/modules/transformer.py:1207-1213 -> F.gelu)."""
    return (0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))).astype(x.dtype)


def gelu_bwd(dy, x):
    cdf = 0.5 * (1.0 + _erf(x / math.sqrt(2.0)))
    pdf = np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)
    return (dy * (cdf + x * pdf)).astype(x.dtype)


def relu(x):
    return np.maximum(x, 0)


def relu_bwd(dy, x):
    return dy * (x > 0)


def layer_norm(x, g, b, eps=1e-5):
    """nn.LayerNorm(d): biased variance, eps 1e-5, affine."""
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xh = xc * rstd
    return (xh * g + b).astype(x.dtype), (xh, rstd)


def layer_norm_bwd(dy, cache, g):
    xh, rstd = cache
    d = xh.shape[-1]
    dg = (dy * xh).reshape(-1, d).sum(0)
    db = dy.reshape(-1, d).sum(0)
    dxh = dy * g
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xh * (dxh * xh).mean(-1, keepdims=True))
    return dx.astype(xh.dtype), dg, db


def _softmax(s):
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    return e / e.sum(-1, keepdims=True)


def mha_forward(xq, xkv, w_in, b_in, w_o, b_o, nhead, add_mask):
    """nn.MultiheadAttention / F.multi_head_attention_forward (torch nn/functional.py:6206-6640):
    packed in-projection, [B,H,L,hd] heads, softmax(q k^T / sqrt(hd) + additive mask) v, out_proj.
    add_mask: None or float array broadcastable to [B,H,Lq,Lk] (bool masks already turned into
    0/-inf and merged, functional.py:6553-6566)."""
    B, Lq, d = xq.shape
    Lk = xkv.shape[1]
    hd = d // nhead
    q = linear(xq, w_in[:d], b_in[:d])
    k = linear(xkv, w_in[d:2 * d], b_in[d:2 * d])
    v = linear(xkv, w_in[2 * d:], b_in[2 * d:])
    qh = q.reshape(B, Lq, nhead, hd).transpose(0, 2, 1, 3)
    kh = k.reshape(B, Lk, nhead, hd).transpose(0, 2, 1, 3)
    vh = v.reshape(B, Lk, nhead, hd).transpose(0, 2, 1, 3)
    s = (qh @ kh.transpose(0, 1, 3, 2)) * xq.dtype.type(1.0 / math.sqrt(hd))
    if add_mask is not None:
        s = s + add_mask
    p = _softmax(s).astype(xq.dtype)
    oh = p @ vh
    o = oh.transpose(0, 2, 1, 3).reshape(B, Lq, d)
    y = linear(o, w_o, b_o)
    return y, (xq, xkv, qh, kh, vh, p, o)


def mha_backward(dy, cache, w_in, w_o, nhead):
    xq, xkv, qh, kh, vh, p, o = cache
    B, Lq, d = xq.shape
    Lk = xkv.shape[1]
    hd = d // nhead
    do, dw_o, db_o = linear_bwd(dy, o, w_o)
    doh = do.reshape(B, Lq, nhead, hd).transpose(0, 2, 1, 3)
    dp = doh @ vh.transpose(0, 1, 3, 2)
    dvh = p.transpose(0, 1, 3, 2) @ doh
    ds = p * (dp - (dp * p).sum(-1, keepdims=True))
    ds = ds * xq.dtype.type(1.0 / math.sqrt(hd))
    dqh = ds @ kh
    dkh = ds.transpose(0, 1, 3, 2) @ qh
    dq = dqh.transpose(0, 2, 1, 3).reshape(B, Lq, d)
    dk = dkh.transpose(0, 2, 1, 3).reshape(B, Lk, d)
    dv = dvh.transpose(0, 2, 1, 3).reshape(B, Lk, d)
    dxq, dwq, dbq = linear_bwd(dq, xq, w_in[:d])
    dxk, dwk, dbk = linear_bwd(dk, xkv, w_in[d:2 * d])
    dxv, dwv, dbv = linear_bwd(dv, xkv, w_in[2 * d:])
    dw_in = np.concatenate([dwq, dwk, dwv], 0)
    db_in = np.concatenate([dbq, dbk, dbv], 0)
    return dxq, dxk + dxv, dw_in, db_in, dw_o, db_o


def _act(name):
    if name == "gelu":
        return gelu, gelu_bwd
    if name == "relu":
        return relu, relu_bwd
    raise ValueError(name)


def _bool_to_add(mask_bool, dt):
    out = np.zeros(mask_bool.shape, dt)
    out[mask_bool] = -np.inf
    return out


# --------------------------------------------------------------------------------------------
# encoder  (model/MMEncoder.py:205-276 + torch nn/modules/transformer.py:951-982)
# --------------------------------------------------------------------------------------------
def encoder_frontend(p: P, feats: np.ndarray, dt=np.float32):
    """model/MMEncoder.py:244-273, one modality, aggregation 'avg', temporal 'encoding', do_norm False:
    u = unify(x); g = mean over ALL T rows (pads included, :196-197); z = cat([g,u]) + PE'."""
    x = feats.astype(dt)
    u = linear(x, p[ENC + "unify.0.weight"].astype(dt), p[ENC + "unify.0.bias"].astype(dt))
    g = u.mean(1, keepdims=True)
    z = np.concatenate([g, u], 1)
    pe_rows = temporal_encoding_rows(p[ENC + "temp_emb.pe"], feats.shape[1]).astype(dt)
    return z + pe_rows[None], (x,)


def encoder_frontend_bwd(dz, cache, p, dt=np.float32):
    (x,) = cache
    T = x.shape[1]
    du = dz[:, 1:] + dz[:, :1] / dt(T)
    _, dw, db = linear_bwd(du, x, p[ENC + "unify.0.weight"].astype(dt))
    return {ENC + "unify.0.weight": dw, ENC + "unify.0.bias": db}


def encoder_layer_fwd(p: P, pre: str, x, add_mask, nhead, act):
    """torch nn/modules/transformer.py:951-982 post-norm: x = LN1(x + SA(x)); x = LN2(x + FF(x))."""
    f, _ = _act(act)
    dt = x.dtype
    g = lambda k: p[pre + k].astype(dt)
    a, c_att = mha_forward(x, x, g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias"),
                           g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"), nhead, add_mask)
    x1, c_n1 = layer_norm(x + a, g("norm1.weight"), g("norm1.bias"))
    h_pre = linear(x1, g("linear1.weight"), g("linear1.bias"))
    h = f(h_pre)
    ff = linear(h, g("linear2.weight"), g("linear2.bias"))
    x2, c_n2 = layer_norm(x1 + ff, g("norm2.weight"), g("norm2.bias"))
    return x2, (c_att, c_n1, x1, h_pre, h, c_n2)


def encoder_layer_bwd(dy, cache, p: P, pre: str, nhead, act, grads: P):
    _, fb = _act(act)
    c_att, c_n1, x1, h_pre, h, c_n2 = cache
    dt = dy.dtype
    g = lambda k: p[pre + k].astype(dt)
    dr, grads[pre + "norm2.weight"], grads[pre + "norm2.bias"] = layer_norm_bwd(dy, c_n2, g("norm2.weight"))
    dh, grads[pre + "linear2.weight"], grads[pre + "linear2.bias"] = linear_bwd(dr, h, g("linear2.weight"))
    dh_pre = fb(dh, h_pre)
    dx1, grads[pre + "linear1.weight"], grads[pre + "linear1.bias"] = linear_bwd(dh_pre, x1, g("linear1.weight"))
    dx1 = dx1 + dr
    dr1, grads[pre + "norm1.weight"], grads[pre + "norm1.bias"] = layer_norm_bwd(dx1, c_n1, g("norm1.weight"))
    dxq, dxkv, dwi, dbi, dwo, dbo = mha_backward(dr1, c_att, g("self_attn.in_proj_weight"),
                                                 g("self_attn.out_proj.weight"), nhead)
    grads[pre + "self_attn.in_proj_weight"] = dwi
    grads[pre + "self_attn.in_proj_bias"] = dbi
    grads[pre + "self_attn.out_proj.weight"] = dwo
    grads[pre + "self_attn.out_proj.bias"] = dbo
    return dr1 + dxq + dxkv


def mm_encoder_forward(p: P, cfg: dict, feats: np.ndarray, mask: Optional[np.ndarray], dt=np.float32,
                       return_cache=False, return_layers=False):
    """MultiModalEncoder.forward (model/MMEncoder.py:244-276) in train-mode semantics with dropout 0:
    returns memory [B,T+1,d] (and the padded-key mask [B,T+1] or None)."""
    z, c_front = encoder_frontend(p, feats, dt)
    B, T1, _ = z.shape
    kpm = None
    add_mask = None
    if mask is not None:
        kpm = np.concatenate([np.zeros((B, 1), bool), mask.astype(bool)], 1)  # MMEncoder.py:252-257
        add_mask = _bool_to_add(kpm, dt)[:, None, None, :]
    x = z
    caches, layers_out = [], [z]
    for l in range(cfg["enc_layers"]):
        x, c = encoder_layer_fwd(p, f"{ENC}transformer_encoder.layers.{l}.", x, add_mask, cfg["enc_nhead"], cfg["activation"])
        caches.append(c)
        layers_out.append(x)
    mem, c_norm = layer_norm(x, p[ENC + "transformer_encoder.norm.weight"].astype(dt),
                             p[ENC + "transformer_encoder.norm.bias"].astype(dt))
    out = [mem, kpm]
    if return_cache:
        out.append((c_front, caches, c_norm))
    if return_layers:
        out.append(layers_out)
    return tuple(out)


def mm_encoder_backward(dmem, cache, p: P, cfg: dict, grads: P, dt=np.float32):
    c_front, caches, c_norm = cache
    dx, grads[ENC + "transformer_encoder.norm.weight"], grads[ENC + "transformer_encoder.norm.bias"] = \
        layer_norm_bwd(dmem, c_norm, p[ENC + "transformer_encoder.norm.weight"].astype(dt))
    for l in reversed(range(cfg["enc_layers"])):
        dx = encoder_layer_bwd(dx, caches[l], p, f"{ENC}transformer_encoder.layers.{l}.", cfg["enc_nhead"],
                               cfg["activation"], grads)
    grads.update(encoder_frontend_bwd(dx, c_front, p, dt))


# --------------------------------------------------------------------------------------------
# decoder  (model/CapDecoder.py:34-79 + torch nn/modules/transformer.py:1143-1199)
# --------------------------------------------------------------------------------------------
def embed_tokens(p: P, ids: np.ndarray, dt=np.float32):
    """model/CapDecoder.py:48 + model/Embedding.py:23-25: Emb[ids] + pos[:S] (no sqrt(d) scaling)."""
    S = ids.shape[1]
    return p[DEC + "tgt_to_emb.weight"].astype(dt)[ids] + p[DEC + "positional_encoding.pos_embedding"].astype(dt)[:S][None]


def decoder_layer_fwd(p: P, pre: str, x, mem, self_mask, nhead, act):
    """torch nn/modules/transformer.py:1143-1199 post-norm: self-attn, cross-attn (no memory mask:
    model/CapDecoder.py:49-52 passes none), feed-forward."""
    f, _ = _act(act)
    dt = x.dtype
    g = lambda k: p[pre + k].astype(dt)
    a, c_sa = mha_forward(x, x, g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias"),
                          g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"), nhead, self_mask)
    x1, c_n1 = layer_norm(x + a, g("norm1.weight"), g("norm1.bias"))
    c, c_ca = mha_forward(x1, mem, g("multihead_attn.in_proj_weight"), g("multihead_attn.in_proj_bias"),
                          g("multihead_attn.out_proj.weight"), g("multihead_attn.out_proj.bias"), nhead, None)
    x2, c_n2 = layer_norm(x1 + c, g("norm2.weight"), g("norm2.bias"))
    h_pre = linear(x2, g("linear1.weight"), g("linear1.bias"))
    h = f(h_pre)
    ff = linear(h, g("linear2.weight"), g("linear2.bias"))
    x3, c_n3 = layer_norm(x2 + ff, g("norm3.weight"), g("norm3.bias"))
    return x3, (c_sa, c_n1, c_ca, c_n2, x2, h_pre, h, c_n3)


def decoder_layer_bwd(dy, cache, p: P, pre: str, nhead, act, grads: P):
    _, fb = _act(act)
    c_sa, c_n1, c_ca, c_n2, x2, h_pre, h, c_n3 = cache
    dt = dy.dtype
    g = lambda k: p[pre + k].astype(dt)
    dr3, grads[pre + "norm3.weight"], grads[pre + "norm3.bias"] = layer_norm_bwd(dy, c_n3, g("norm3.weight"))
    dh, grads[pre + "linear2.weight"], grads[pre + "linear2.bias"] = linear_bwd(dr3, h, g("linear2.weight"))
    dh_pre = fb(dh, h_pre)
    dx2, grads[pre + "linear1.weight"], grads[pre + "linear1.bias"] = linear_bwd(dh_pre, x2, g("linear1.weight"))
    dx2 = dx2 + dr3
    dr2, grads[pre + "norm2.weight"], grads[pre + "norm2.bias"] = layer_norm_bwd(dx2, c_n2, g("norm2.weight"))
    dxq, dmem, dwi, dbi, dwo, dbo = mha_backward(dr2, c_ca, g("multihead_attn.in_proj_weight"),
                                                 g("multihead_attn.out_proj.weight"), nhead)
    grads[pre + "multihead_attn.in_proj_weight"] = dwi
    grads[pre + "multihead_attn.in_proj_bias"] = dbi
    grads[pre + "multihead_attn.out_proj.weight"] = dwo
    grads[pre + "multihead_attn.out_proj.bias"] = dbo
    dx1 = dr2 + dxq
    dr1, grads[pre + "norm1.weight"], grads[pre + "norm1.bias"] = layer_norm_bwd(dx1, c_n1, g("norm1.weight"))
    dxq, dxkv, dwi, dbi, dwo, dbo = mha_backward(dr1, c_sa, g("self_attn.in_proj_weight"),
                                                 g("self_attn.out_proj.weight"), nhead)
    grads[pre + "self_attn.in_proj_weight"] = dwi
    grads[pre + "self_attn.in_proj_bias"] = dbi
    grads[pre + "self_attn.out_proj.weight"] = dwo
    grads[pre + "self_attn.out_proj.bias"] = dbo
    return dr1 + dxq + dxkv, dmem


def sce_loss(logits: np.ndarray, labels: np.ndarray, alpha: float, pad_id: int = 0):
    """model/loss.py:69-92 (SCELoss) as constructed at model/CapDecoder.py:28-32:
    alpha==1.0 -> plain CrossEntropyLoss(ignore_index=pad); else alpha*CE + (1-alpha)*mean(RCE) where
    CE averages over non-pad rows and RCE over ALL rows; RCE_i = -sum_j clamp(p_ij,1e-7,1)*log(clamp(onehot,1e-4,1)).
    Returns (loss, dlogits)."""
    dt = logits.dtype
    N, V = logits.shape
    m = logits.max(-1, keepdims=True)
    e = np.exp(logits - m)
    se = e.sum(-1, keepdims=True)
    lse = (m + np.log(se))[:, 0]
    prob = e / se
    valid = labels != pad_id
    nvalid = max(int(valid.sum()), 0)
    rows = np.arange(N)
    ce_rows = lse - logits[rows, labels]
    ce = ce_rows[valid].sum() / dt.type(nvalid) if nvalid else dt.type(np.nan)
    onehot = np.zeros((N, V), bool)
    onehot[rows, labels] = True
    d_ce = (prob - onehot) * (valid[:, None] / dt.type(max(nvalid, 1)))
    if alpha == 1.0:
        return ce, d_ce.astype(dt)
    beta = 1.0 - alpha
    c = dt.type(NEG_LOG_1E4_F32)
    pc = np.clip(prob, 1e-7, 1.0)
    rce_rows = c * np.where(onehot, 0, pc).sum(-1)
    loss = dt.type(alpha) * ce + dt.type(beta) * rce_rows.mean()
    G = c * (~onehot) * (prob >= 1e-7)
    d_rce = prob * (G - (G * prob).sum(-1, keepdims=True)) / dt.type(N)
    return loss, (dt.type(alpha) * d_ce + dt.type(beta) * d_rce).astype(dt)


def cap_decoder_forward(p: P, cfg: dict, mem: np.ndarray, ids: np.ndarray, pad_mask: Optional[np.ndarray] = None,
                        dt=np.float32, return_cache=False, return_layers=False):
    """CapDecoder.forward (model/CapDecoder.py:34-60): token shift, float causal mask + bool key
    padding mask merged additively, decoder stack, final LN, generator, loss.  Returns (logits, loss)."""
    pad = cfg.get("pad_id", 0)
    if pad_mask is None:
        pad_mask = ids == pad
    tgt_in, tgt_out, kpm = ids[:, :-1], ids[:, 1:], pad_mask[:, :-1]
    Sd = tgt_in.shape[1]
    self_mask = generate_square_subsequent_mask(Sd).astype(dt)[None, None] + _bool_to_add(kpm, dt)[:, None, None, :]
    x = embed_tokens(p, tgt_in, dt)
    caches, layers_out = [], [x]
    for l in range(cfg["dec_layers"]):
        x, c = decoder_layer_fwd(p, f"{DEC}decoder.layers.{l}.", x, mem.astype(dt), self_mask, cfg["dec_nhead"], cfg["activation"])
        caches.append(c)
        layers_out.append(x)
    y, c_norm = layer_norm(x, p[DEC + "decoder.norm.weight"].astype(dt), p[DEC + "decoder.norm.bias"].astype(dt))
    logits = linear(y, p[DEC + "generator.weight"].astype(dt), p[DEC + "generator.bias"].astype(dt))
    V = logits.shape[-1]
    loss, dlogits = sce_loss(logits.reshape(-1, V), tgt_out.reshape(-1), cfg["sce_loss_alpha"], pad)
    out = [logits, loss]
    if return_cache:
        out.append((tgt_in, caches, c_norm, y, dlogits.reshape(logits.shape)))
    if return_layers:
        out.append(layers_out + [y])
    return tuple(out)


def cap_decoder_backward(cache, p: P, cfg: dict, grads: P, dt=np.float32):
    """Reverse of cap_decoder_forward for d(loss)=1; returns d(memory)."""
    tgt_in, caches, c_norm, y, dlogits = cache
    dy, grads[DEC + "generator.weight"], grads[DEC + "generator.bias"] = \
        linear_bwd(dlogits, y, p[DEC + "generator.weight"].astype(dt))
    dx, grads[DEC + "decoder.norm.weight"], grads[DEC + "decoder.norm.bias"] = \
        layer_norm_bwd(dy, c_norm, p[DEC + "decoder.norm.weight"].astype(dt))
    dmem = 0
    for l in reversed(range(cfg["dec_layers"])):
        dx, dm = decoder_layer_bwd(dx, caches[l], p, f"{DEC}decoder.layers.{l}.", cfg["dec_nhead"], cfg["activation"], grads)
        dmem = dmem + dm
    V, d = p[DEC + "tgt_to_emb.weight"].shape
    demb = np.zeros((V, d), dt)
    np.add.at(demb, tgt_in.reshape(-1), dx.reshape(-1, d))
    demb[cfg.get("pad_id", 0)] = 0  # nn.Embedding(padding_idx=pad): that row receives no gradient
    grads[DEC + "tgt_to_emb.weight"] = demb
    return dmem


# --------------------------------------------------------------------------------------------
# whole model: loss, gradients, one Adam step   (model/MMT4Caption.py:114-121, train.py:123-126)
# --------------------------------------------------------------------------------------------
def caption_loss_and_grads(p: P, cfg: dict, feats, mask, ids, dt=np.float32) -> Tuple[float, P, np.ndarray]:
    """loss = model(v_feats, v_masks, captions); loss.backward()  -> (loss, {name: grad}, logits)."""
    mem, _, c_enc = mm_encoder_forward(p, cfg, feats, mask, dt, return_cache=True)
    logits, loss, c_dec = cap_decoder_forward(p, cfg, mem, ids, None, dt, return_cache=True)
    grads: P = {}
    dmem = cap_decoder_backward(c_dec, p, cfg, grads, dt)
    mm_encoder_backward(dmem, c_enc, p, cfg, grads, dt)
    return loss, grads, logits


def adam_step(p: P, grads: P, state: dict, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam (train.py:24-26,126), weight_decay 0, no amsgrad -- torch/optim/adam.py
    single-tensor form: step_size = lr/bc1; denom = sqrt(v)/sqrt(bc2) + eps; p -= step_size*m/denom."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    bc1 = 1 - b1 ** t
    bc2s = math.sqrt(1 - b2 ** t)
    out = {}
    for k, g in grads.items():
        m = state.setdefault("m", {}).get(k, np.zeros_like(g))
        v = state.setdefault("v", {}).get(k, np.zeros_like(g))
        m = m + (1 - b1) * (g - m)  # torch: exp_avg.lerp_(grad, 1-beta1)
        v = b2 * v + (1 - b2) * g * g
        state["m"][k], state["v"][k] = m, v
        denom = np.sqrt(v) / bc2s + eps
        out[k] = (p[k] - (lr / bc1) * m / denom).astype(p[k].dtype)
    newp = dict(p)
    newp.update(out)
    return newp


# --------------------------------------------------------------------------------------------
# greedy decode   (model/CapDecoder.py:62-79, model/MMT4Caption.py:146-184)
# --------------------------------------------------------------------------------------------
def decode_word(p: P, cfg: dict, mem: np.ndarray, ys: np.ndarray, dt=np.float32) -> np.ndarray:
    """CapDecoder.decode_word: embed ALL tokens so far, bool causal mask, full decoder, generator on
    the last position -> logits [B,V].  (tgt_padding_mask is always None on the greedy path.)"""
    t = ys.shape[1]
    self_mask = generate_square_subsequent_mask(t).astype(dt)[None, None]
    x = embed_tokens(p, ys, dt)
    for l in range(cfg["dec_layers"]):
        x, _ = decoder_layer_fwd(p, f"{DEC}decoder.layers.{l}.", x, mem.astype(dt), self_mask, cfg["dec_nhead"], cfg["activation"])
    y, _ = layer_norm(x, p[DEC + "decoder.norm.weight"].astype(dt), p[DEC + "decoder.norm.bias"].astype(dt))
    return linear(y[:, -1], p[DEC + "generator.weight"].astype(dt), p[DEC + "generator.bias"].astype(dt))


def greedy_decode_ids(p: P, cfg: dict, feats, mask=None, max_len=30, start_id=101, end_id=102, dt=np.float32,
                      return_margins=False):
    """MMT4Caption.greedy_decode (model/MMT4Caption.py:146-172): encoder once, then <= max_len-1 steps of
    decode_word + first-index argmax (torch.max) + append; stop when EVERY row has emitted end_id at
    least once (sticky flags).  Returns the id matrix ys [B, <=max_len]."""
    mem = mm_encoder_forward(p, cfg, feats, mask, dt)[0]
    B = feats.shape[0]
    ys = np.full((B, 1), start_id, np.int64)
    end_flag = np.zeros(B, bool)
    margins = []
    for _ in range(max_len - 1):
        logits = decode_word(p, cfg, mem, ys, dt)
        nxt = logits.argmax(1)  # numpy argmax = first maximal index, like torch.max(dim=1)
        if return_margins:
            part = np.partition(logits, -2, axis=1)
            margins.append(part[:, -1] - part[:, -2])
        ys = np.concatenate([ys, nxt[:, None].astype(np.int64)], 1)
        end_flag |= nxt == end_id
        if end_flag.all():
            break
    if return_margins:
        return ys, np.stack(margins, 1)
    return ys


def ids_to_caption_ids(ys: np.ndarray, end_id=102) -> List[List[int]]:
    """model/MMT4Caption.py:174-181: cut at the first end_id; with no end_id the slice is [1:-1]
    (drops the last token -- reference quirk)."""
    out = []
    for row in ys.tolist():
        end_count = -1
        for i, t in enumerate(row):
            if t == end_id:
                end_count = i
                break
        out.append(row[1:end_count])
    return out


# --------------------------------------------------------------------------------------------
# config helper + synthetic inputs
# --------------------------------------------------------------------------------------------
def cfg_from_model_config(mc: dict, vocab_size=30522, pad_id=0) -> dict:
    """Flatten the reference's cfg['model'] block (configs/*.json:63-95) to what the oracle needs."""
    return dict(d=mc["embed_dim"], d_in=mc["modal_shape"][0], activation=mc["activation"],
                enc_layers=mc["video_encoder"]["layer"], enc_nhead=mc["video_encoder"]["nhead"],
                enc_ff=mc["video_encoder"]["feedforward"],
                dec_layers=mc["caption_decoder"]["layer"], dec_nhead=mc["caption_decoder"]["nhead"],
                dec_ff=mc["caption_decoder"]["feedforward"],
                sce_loss_alpha=mc["caption_decoder"]["sce_loss_alpha"], vocab=vocab_size, pad_id=pad_id)


def init_params(cfg: dict, seed=0, dt=np.float32) -> P:
    """Random parameters with the reference's shapes/keys (Appendix B).  Distribution is NOT torch's
    init stream (parity never depends on init RNG -- tests load identical weights on both sides);
    scales are chosen like torch's so activations are comparable."""
    rng = np.random.default_rng(seed)
    d, din, V = cfg["d"], cfg["d_in"], cfg["vocab"]
    p: P = {}

    def lin(name, o, i):
        b = 1.0 / math.sqrt(i)
        p[name + ".weight"] = rng.uniform(-b, b, (o, i)).astype(dt)
        p[name + ".bias"] = rng.uniform(-b, b, (o,)).astype(dt)

    def attn(pre):
        b = math.sqrt(6.0 / (4 * d))
        p[pre + ".in_proj_weight"] = rng.uniform(-b, b, (3 * d, d)).astype(dt)
        p[pre + ".in_proj_bias"] = (0.02 * rng.standard_normal(3 * d)).astype(dt)
        lin(pre + ".out_proj", d, d)

    def norm(pre):
        p[pre + ".weight"] = (1 + 0.05 * rng.standard_normal(d)).astype(dt)
        p[pre + ".bias"] = (0.05 * rng.standard_normal(d)).astype(dt)

    lin(ENC + "unify.0", d, din)
    p[ENC + "temp_emb.pe"] = encoder_pos_table(512, d)
    for l in range(cfg["enc_layers"]):
        pre = f"{ENC}transformer_encoder.layers.{l}"
        attn(pre + ".self_attn"); lin(pre + ".linear1", cfg["enc_ff"], d); lin(pre + ".linear2", d, cfg["enc_ff"])
        norm(pre + ".norm1"); norm(pre + ".norm2")
    norm(ENC + "transformer_encoder.norm")
    for l in range(cfg["dec_layers"]):
        pre = f"{DEC}decoder.layers.{l}"
        attn(pre + ".self_attn"); attn(pre + ".multihead_attn")
        lin(pre + ".linear1", cfg["dec_ff"], d); lin(pre + ".linear2", d, cfg["dec_ff"])
        norm(pre + ".norm1"); norm(pre + ".norm2"); norm(pre + ".norm3")
    norm(DEC + "decoder.norm")
    lin(DEC + "generator", V, d)
    emb = rng.standard_normal((V, d)).astype(dt)
    emb[cfg.get("pad_id", 0)] = 0
    p[DEC + "tgt_to_emb.weight"] = emb
    p[DEC + "positional_encoding.pos_embedding"] = decoder_pos_table(5000, d)
    return p


BUFFER_KEYS = (ENC + "temp_emb.pe", DEC + "positional_encoding.pos_embedding")


def synthetic_batch(B, T, d_in, S, vocab, seed=0, ragged=False):
    """randn features, ids uniform in [1000, min(30000,vocab)), ids[:,0]=CLS(101),
    last real token = SEP(102); ragged=True draws caption lengths in [6,S] and frame counts in [6,T]
    (right-padded with 0 / zero rows + mask True)."""
    rng = np.random.default_rng(seed)
    feats = rng.standard_normal((B, T, d_in)).astype(np.float32)
    lo, hi = (1000, min(30000, vocab)) if vocab > 2000 else (103, vocab)
    ids = rng.integers(lo, hi, (B, S)).astype(np.int64)
    ids[:, 0] = 101
    ids[:, -1] = 102
    mask = np.zeros((B, T), bool)
    if ragged:
        for b in range(B):
            L = int(rng.integers(min(6, S), S + 1))
            ids[b, L - 1] = 102
            ids[b, L:] = 0
            n = int(rng.integers(min(6, T), T + 1))
            feats[b, n:] = 0
            mask[b, n:] = True
        ids[0, -1] = 102  # keep at least one full-length row so S is the batch max
        if S > 1:
            full = ids[0] == 0
            ids[0, full] = lo
            ids[0, -1] = 102
    return feats, mask, ids


# ---- input path and early stopping --------------------------------
def load_clip(a: np.ndarray) -> np.ndarray:
    """dataloader.py:378-386: fp32; a file stored [E, T] (shape[0] > shape[1]) is transposed to [T, E]."""
    a = np.asarray(a, dtype=np.float32)
    return a.T.copy() if a.shape[0] > a.shape[1] else a


def make_mask_video(clips: Sequence[np.ndarray]):
    """dataloader.py:233-247: zero-pad clips [T_i, E] to [B, max T, E]; mask True = padded frame."""
    B, E, lens = len(clips), clips[0].shape[1], [c.shape[0] for c in clips]
    feat = np.zeros((B, max(lens), E), np.float32)
    mask = np.ones((B, max(lens)), bool)
    for i, c in enumerate(clips):
        feat[i, :lens[i]] = c
        mask[i, :lens[i]] = False
    return feat, mask


def early_stopping_trace(losses: Sequence[float], patience: int, delta: float = 0.0):
    """utils.py:36-59: the monitored value is negated once (`val_loss = -val_loss`) and everything after -- best
    score, the stored `val_loss_min` -- uses the negated value; a call saves iff it sets a new best."""
    counter, best, stop, vmin, saves, out = 0, None, False, float("inf"), 0, []
    for v in losses:
        s = -v
        if best is None:
            best, vmin, saves = s, s, saves + 1
        elif s < best + delta:
            counter += 1
            stop = stop or counter >= patience
        else:
            best, vmin, saves, counter = s, s, saves + 1, 0
        out.append({"counter": counter, "best_score": best, "early_stop": stop, "val_loss_min": vmin, "saves": saves})
    return out
