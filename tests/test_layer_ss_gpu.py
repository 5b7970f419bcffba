"""Sample-stationary layer forward (csrc/vct_layer_ss.hip: one launch per Transformer layer, one workgroup per sample).

  * vct_ss_pack writes the weight blocks in the documented stream order (checked element for element against a host restatement);
  * the fused forward against the ORACLE (numpy restatement of the reference, pinned by tests/golden): loss, logits, every gradient --
    the unfused backward kernels run behind the fused forward, so the gradients check every tensor the forward saves for them;
  * fused against the unfused kernel schedule on the same weights / batch: every saved activation and LayerNorm statistic, with
    dropout ON (the counter streams are the same, so the masks are, and the two paths differ by bf16 rounding only);
  * edge shapes: 32 decoder rows / 16 memory rows (the limits), a single row (decode_word's first step), ragged masks, pads;
  * the packed stream follows the weights (optimizer step inside a recorded launch list; load_state_dict).

Tolerances as tests/test_model_gpu.py (bf16: loss 1e-3 rel, logits 2e-2, gradients 5e-2 rel-Frobenius)."""
import numpy as np
import pytest
import torch

import vct_oracle as O
from helpers import build_model, load_golden, model_config_of, rel

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _mc(enc=1, dec=2, ff=1024, dropout=0.0, act="gelu"):
    mc = dict(model_config_of(load_golden("cfgA_slices.npz")))          # d = 512, 8 heads
    mc["dropout"], mc["activation"] = dropout, act
    mc["video_encoder"] = dict(mc["video_encoder"], layer=enc, feedforward=ff)
    mc["caption_decoder"] = dict(mc["caption_decoder"], layer=dec, feedforward=ff)
    return mc


def _fuse(on: bool):
    from vct_amd import engine
    old = engine._StackBase.fuse_layers
    engine._StackBase.fuse_layers = on
    return old


def test_pack_writes_the_documented_stream_order():
    from vct_amd import ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn(1536, 2048, generator=g).to(BF).to(DEV)
    # three blocks: rows 512..1023 x columns 0..511 (8 chunks) at chunk 2; rows 0..511 x columns 1024..1151 (2 chunks) at chunk 0; ...
    blocks = [(w[512:1024], 8, 2), (w[:, 1024:], 2, 0), (w[1024:, 512:], 4, 10)]
    dst = torch.zeros(14 * ops.SS_CHUNK, dtype=BF, device=DEV)
    ops.ss_pack(blocks, dst)
    got = dst.view(torch.int16).cpu().numpy().reshape(14, 8, 4, 2, 64, 8)          # [chunk][wave][tile][k-step][lane][8]
    wh = w.view(torch.int16).cpu().numpy()
    lane = np.arange(64)
    for src, nch, dc in ((wh[512:1024], 8, 2), (wh[:, 1024:], 2, 0), (wh[1024:, 512:], 4, 10)):
        for c in range(nch):
            for wv in range(8):
                for t in range(4):
                    for s in range(2):
                        rows = wv * 64 + t * 16 + (lane & 15)
                        cols = c * 64 + s * 32 + (lane >> 4) * 8
                        exp = np.stack([src[rows, cols + j] for j in range(8)], 1)
                        assert np.array_equal(got[dc + c, wv, t, s], exp), (dc, c, wv, t, s)


@pytest.mark.parametrize("shape", [dict(B=5, T=12, S=20), dict(B=3, T=15, S=33), dict(B=4, T=5, S=2), dict(B=2, T=12, S=18)])
@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_fused_forward_vs_oracle(shape, act):
    """Loss, logits and every parameter gradient against the oracle (dropout 0): rows 19 / 13 (cfg-B), the limits 32 / 16, one decoder
    row, and a 17-row case (second row tile barely used); ragged frame masks and padded captions in every case."""
    if act == "relu" and shape["S"] != 20:
        pytest.skip("one ReLU case is enough")
    B, T, S = shape["B"], shape["T"], shape["S"]
    V = 1000
    mc = _mc(enc=2, dec=2, ff=1024, act=act)
    cfg = O.cfg_from_model_config(mc, V)
    p = O.init_params(cfg, seed=17)
    f, mk, ids = O.synthetic_batch(B, T, 512, S, V, seed=S, ragged=True)
    ref_loss, ref_grads, ref_logits = O.caption_loss_and_grads(p, cfg, f, mk, ids)
    m = build_model(mc, V, DEV, BF, p)
    m.train()
    assert m.cap_decoder._engine()._ss_ok(S - 1, T + 1, B) and m.video_encoder._engine()._ss_ok(T + 1, 0, B)
    feats, mask, idt = (torch.from_numpy(a).to(DEV) for a in (f, mk, ids))
    loss, logits = m._forward_loss(feats, mask, idt, True, want_logits=True)
    assert abs(float(loss) - ref_loss) < 1e-3 * abs(ref_loss), (float(loss), ref_loss)
    valid = torch.from_numpy(ids[:, 1:] != 0).reshape(-1)
    lg = logits[:, :V].float().cpu()
    assert rel(lg[valid], ref_logits.reshape(-1, V)[valid.numpy()]) < 2e-2
    m._backward()
    for k, g in ref_grads.items():
        r = float(np.linalg.norm(g))
        if r < 1e-12:
            continue
        assert rel(m._ps.g[k], g) < 5e-2, k


@pytest.mark.parametrize("dropout", [0.0, 0.3])
def test_fused_equals_unfused_schedule(dropout):
    """Same weights, batch and dropout seed through the one-launch-per-layer forward and through the unfused kernels: every tensor
    the backward reads must agree (bf16 rounding apart), also with dropout on -- the masks are regenerated from the same counters."""
    V, B, T, S = 777, 6, 11, 14
    mc = _mc(enc=2, dec=2, ff=1024, dropout=dropout)
    cfg = O.cfg_from_model_config(mc, V)
    p = O.init_params(cfg, seed=5)
    f, mk, ids = O.synthetic_batch(B, T, 512, S, V, seed=2, ragged=True)
    feats, mask, idt = (torch.from_numpy(a).to(DEV) for a in (f, mk, ids))
    runs = {}
    old = _fuse(True)
    try:
        for fused in (True, False):
            _fuse(fused)
            m = build_model(mc, V, DEV, BF, p)
            m.train(); m._seed.fill_(4242)
            loss, logits = m._forward_loss(feats, mask, idt, True, want_logits=True)
            acts = {}
            for eng, pre in ((m.video_encoder._engine(), "enc."), (m.cap_decoder._engine(), "dec.")):
                for k, t in eng.cur.t.items():
                    if isinstance(k, str) and torch.is_tensor(t) and t.is_floating_point() and k not in ("logits", "dlogits", "row_ws", "loss"):
                        acts[pre + k] = t.float().clone()
            m._backward()
            runs[fused] = (float(loss), acts, m._ps.gflat.clone())
    finally:
        _fuse(old)
    (l1, a1, g1), (l0, a0, g0) = runs[True], runs[False]
    assert abs(l1 - l0) < 2e-3 * abs(l0)
    saved = [k for k in a0 if k in a1 and a0[k].shape == a1[k].shape]
    assert len(saved) >= 40, sorted(a0)
    for k in saved:
        # rows of padded FRAMES / tokens carry values nothing downstream reads identically in both schedules; statistics and
        # activations of all real rows must agree
        if k.endswith("mean"):          # row means of O(1) activations are ~1e-3: an absolute bound (bf16 rounding of 512 addends)
            assert float((a1[k] - a0[k]).abs().max()) < 5e-3, k
            continue
        tol = 3e-2 if k.endswith("rstd") else 2.5e-2
        assert rel(a1[k], a0[k]) < tol, (k, rel(a1[k], a0[k]))
    assert rel(g1, g0) < 3e-2


@pytest.mark.parametrize("dropout,T", [(0.0, 12), (0.3, 12), (0.3, 15), (0.3, 20)])
def test_fused_backward_equals_unfused_chain(dropout, T, monkeypatch):
    """The encoder stack's activation-gradient chain as ONE launch (csrc/vct_layer_ss_bwd.hip) against the unfused kernels behind the
    same forward: every gradient the weight-gradient GEMMs read (d f, d hpre, d a, d qkv), the gradient of the stack input and all
    parameter gradients (LayerNorm partial rows included), with dropout ON (regenerated masks) and at 13 / 16 / 21 rows per sample."""
    from vct_amd import engine, ops
    V, B, S = 600, 7, 12
    mc = _mc(enc=2, dec=1, ff=1024, dropout=dropout)
    cfg = O.cfg_from_model_config(mc, V)
    p = O.init_params(cfg, seed=11)
    f, mk, ids = O.synthetic_batch(B, T, 512, S, V, seed=T, ragged=True)
    feats, mask, idt = (torch.from_numpy(a).to(DEV) for a in (f, mk, ids))
    calls = []
    real = ops.layer_ss_bwd
    monkeypatch.setattr(ops, "layer_ss_bwd", lambda descs: (calls.append(len(descs)), real(descs))[1])
    runs = {}
    old = engine._StackBase.fuse_bwd
    try:
        for fused in (True, False):
            engine._StackBase.fuse_bwd = fused
            m = build_model(mc, V, DEV, BF, p)
            m.train(); m._seed.fill_(99)
            m._forward_loss(feats, mask, idt, True)
            m._backward()
            e = m.video_encoder._engine().cur.t
            def pick(k):     # (without dropout the unfused chain has no separate masked copy: it reads `ds`)
                return e[k] if (fused or dropout > 0.0 or not k.endswith("dxo")) else e[k[:-3] + "ds"]
            keep = {k: pick(k).float().clone() for k in ("L1.n2.dxo", "L1.ff.dhpre", "L1.n1.dxo", "L1.sa.dqkv", "L0.n2.dxo", "L0.ff.dhpre",
                                                         "L0.n1.dxo", "L0.sa.dqkv", "L0.sa.dx")}
            runs[fused] = (keep, m._ps.gflat.clone(), {k: m._ps.g[k].clone() for k in m._ps.g if k.startswith("video_encoder")})
    finally:
        engine._StackBase.fuse_bwd = old
    assert calls == [2]                                      # one launch, two layers, in the fused run only
    (k1, g1, e1), (k0, g0, e0) = runs[True], runs[False]
    for k in k0:
        assert rel(k1[k], k0[k]) < 2.5e-2, (k, rel(k1[k], k0[k]))
    for k in e0:
        if float(e0[k].float().norm()) > 1e-12:
            assert rel(e1[k], e0[k]) < 3e-2, (k, rel(e1[k], e0[k]))
    assert rel(g1, g0) < 2e-2


def test_packed_stream_follows_the_weights():
    """Training steps through a recorded launch list (the optimizer rewrites the shadow, the pack launches behind it are part of the
    recording) and a load_state_dict in between: the fused forward must always see the current weights -- equal to the unfused
    schedule run on the same sequence."""
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    V, B, T, S = 500, 4, 9, 11
    mc = _mc(enc=1, dec=1, ff=512, dropout=0.0)
    cfg = O.cfg_from_model_config(mc, V)
    p = O.init_params(cfg, seed=9)
    p2 = O.init_params(cfg, seed=10)
    batches = [tuple(torch.from_numpy(a).to(DEV) for a in O.synthetic_batch(B, T, 512, S, V, seed=20 + k)) for k in range(6)]
    out = {}
    old = _fuse(True)
    try:
        for fused in (True, False):
            _fuse(fused)
            m = build_model(mc, V, DEV, BF, p)
            m.train()
            tr = CaptionTrainer(m, FusedAdam(m, lr=3e-3), launch_list=True)
            losses = []
            for k, (fe, mk, ii) in enumerate(batches):
                if k == 4:
                    m.load_state_dict({n: torch.from_numpy(np.ascontiguousarray(v)) for n, v in p2.items()}, strict=False)
                losses.append(float(tr.step(fe, mk, ii)))
            torch.cuda.synchronize()
            out[fused] = (losses, m.flat_params.clone())
            if fused:      # the packed streams the optimizer's pass maintains == a fresh vct_ss_pack of the current shadow, bit for bit
                from vct_amd import ops
                assert m._ps.packed
                for key, (stream, _firsts, subs) in m._ps.packed.items():
                    fresh = ops.ss_pack([blk for sub in subs for blk in sub[2]], torch.zeros_like(stream))
                    torch.cuda.synchronize()
                    assert torch.equal(fresh.view(torch.int16), stream.view(torch.int16)), key
    finally:
        _fuse(old)
    for a, b in zip(out[True][0], out[False][0]):
        assert abs(a - b) < 3e-3 * abs(b), (out[True][0], out[False][0])
    assert out[True][0][1] < out[True][0][0] + 1.0          # it trains (lr 3e-3, not diverging)
    assert abs(out[True][0][4] - out[True][0][3]) > 1e-3    # the load_state_dict changed what the replay computes
    assert rel(out[True][1], out[False][1]) < 2e-2


def test_packed_stream_created_after_a_recording_is_kept_current():
    """ADVICE (round 4): a packed stream that comes into being AFTER another shape's launch list was recorded must not go stale.  A
    caption shape the sample-stationary decoder rejects (34 decoder rows > 32) is recorded first -- no decoder stream exists yet -- then an
    eligible shape creates the stream and records its own list; the shapes then alternate.  Every replay of either list must leave every
    packed stream equal to a fresh vct_ss_pack of the current shadow, bit for bit, and the eligible shape's losses must be those of the
    unfused schedule run on the same sequence."""
    from vct_amd import ops
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    V, B, T = 500, 4, 9
    mc = _mc(enc=1, dec=1, ff=512, dropout=0.0)
    cfg = O.cfg_from_model_config(mc, V)
    p = O.init_params(cfg, seed=12)
    long_b = tuple(torch.from_numpy(a).to(DEV) for a in O.synthetic_batch(B, T, 512, 35, V, seed=40))      # 34 decoder rows: unfused decoder
    short_b = [tuple(torch.from_numpy(a).to(DEV) for a in O.synthetic_batch(B, T, 512, 11, V, seed=41 + k)) for k in range(3)]
    seq = [long_b, short_b[0], long_b, short_b[1], long_b, short_b[2]]
    out = {}
    old = _fuse(True)
    try:
        for fused in (True, False):
            _fuse(fused)
            m = build_model(mc, V, DEV, BF, p)
            m.train()
            tr = CaptionTrainer(m, FusedAdam(m, lr=3e-3), launch_list=True)
            losses = []
            for k, (fe, mk, ii) in enumerate(seq):
                losses.append(float(tr.step(fe, mk, ii)))
                if fused:
                    torch.cuda.synchronize()
                    if k >= 1:
                        assert any("decoder" in key for key in m._ps.packed), list(m._ps.packed)
                    for key, (stream, _firsts, subs) in m._ps.packed.items():
                        fresh = ops.ss_pack([blk for sub in subs for blk in sub[2]], stream.clone())
                        torch.cuda.synchronize()
                        assert torch.equal(fresh.view(torch.int16), stream.view(torch.int16)), (k, key)
            out[fused] = losses
    finally:
        _fuse(old)
    for a, b in zip(out[True], out[False]):
        assert abs(a - b) < 3e-3 * abs(b), (out[True], out[False])


def test_fused_step_is_bitwise_deterministic_with_dropout():
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    V = 900
    mc = _mc(enc=1, dec=2, ff=1024, dropout=0.3)
    cfg = O.cfg_from_model_config(mc, V)
    p = O.init_params(cfg, seed=1)
    fe, mk, ii = (torch.from_numpy(a).to(DEV) for a in O.synthetic_batch(16, 12, 512, 20, V, seed=3))
    res = []
    for _ in range(2):
        m = build_model(mc, V, DEV, BF, p)
        m.train(); m._seed.fill_(7)
        tr = CaptionTrainer(m, FusedAdam(m, lr=1e-4), launch_list=True)
        losses = torch.cat([tr.step(fe, mk, ii).clone() for _ in range(4)])
        torch.cuda.synchronize()
        res.append((losses, m.flat_params.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert len(set(res[0][0].tolist())) == 4                # a fresh mask every step
