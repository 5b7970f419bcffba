"""Generate tests/golden/*.npz from the REAL reference (runs only in the build container).

    python oracle/make_golden.py            # needs /root/reference (read-only) and torch CPU

The reference's modules are imported unmodified from /root/reference; only the two host-side
objects that need network downloads are stubbed before construction:
CapPreprocessor (HF tokenizer download) -> accepts id lists, TextEncoder (CLIP download) -> .dim.
Nothing from the reference is copied: the fixtures are inputs and outputs only.
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("VCT_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

import model.MMT4Caption as RM  # noqa: E402  (the reference)
from model.loss import SCELoss  # noqa: E402
import vct_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
CPU = torch.device("cpu")


class _Tok:
    def __init__(self, v):
        self.vocab_size = v

    def convert_ids_to_tokens(self, ids):
        return [str(i) for i in ids]

    def convert_tokens_to_string(self, toks):
        return " ".join(toks)


def make_prep(vocab):
    class StubPrep:  # stands in for model/CapPreprocessor.py (tokenizer download)
        def __init__(self, *_a, **_k):
            self.tokenizer = _Tok(vocab)
            self.pad_id, self.start_id, self.end_id = 0, 101, 102

        def __call__(self, caps):
            ids = torch.as_tensor(np.asarray(caps), dtype=torch.long)
            return ids, ids == self.pad_id
    return StubPrep


class StubText:  # stands in for model/TextEncoder.py (CLIP download); .dim == embed_dim -> no v_proj
    def __init__(self, *_a, **_k):
        self.dim = None


def build_ref(mc, vocab):
    RM.CapPreprocessor = make_prep(vocab)
    StubText.dim = mc["embed_dim"]

    class T(StubText):
        def __init__(self, *a, **k):
            self.dim = mc["embed_dim"]
    RM.TextEncoder = T
    m = RM.MMT4Caption(mc, device=CPU)
    m.mode("caption")
    return m


def model_cfg(d, d_in, H, ff, Le, Ld, alpha=0.5, act="gelu", dropout=0.0):
    return {"modal": ["CLIP4Clip"], "modal_shape": [d_in], "tokenizer": "stub", "text_enc_type": "CLIP",
            "embed_dim": d, "dropout": dropout, "loss_beta": 0.5,
            "matching": {"enable_tem": False, "matching_loss": "CSL"}, "activation": act,
            "video_encoder": {"layer": Le, "nhead": H, "feedforward": ff,
                              "mme": {"temporal": "encoding", "modal_different": True, "do_norm": False,
                                      "aggregation": "avg"}},
            "caption_decoder": {"layer": Ld, "nhead": H, "feedforward": ff, "sce_loss_alpha": alpha},
            "pretrained_model": None}


def load_np_state(m, p):
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()}
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing
    assert all(k.startswith("matching") for k in missing.missing_keys), missing


def t2n(t):
    return t.detach().cpu().numpy()


def run_train_case(mc, vocab, p, feats, mask, ids, per_layer=True):
    """One reference forward/backward/Adam step in train() mode with dropout 0."""
    m = build_ref(mc, vocab)
    load_np_state(m, p)
    m.train()
    rec = {}
    if per_layer:
        enc = m.video_encoder.transformer_encoder
        dec = m.cap_decoder.decoder
        hooks = []
        hooks.append(enc.register_forward_pre_hook(lambda mod, a: rec.__setitem__("mm_src", t2n(a[0]))))
        for i, l in enumerate(enc.layers):
            hooks.append(l.register_forward_hook(lambda mod, a, o, i=i: rec.__setitem__(f"enc_layer{i}", t2n(o))))
        for i, l in enumerate(dec.layers):
            hooks.append(l.register_forward_hook(lambda mod, a, o, i=i: rec.__setitem__(f"dec_layer{i}", t2n(o))))
        hooks.append(dec.register_forward_pre_hook(lambda mod, a: rec.__setitem__("tgt_emb", t2n(a[0]))))
        def gen_hook(mod, a, o):
            rec["dec_out"], rec["logits"] = t2n(a[0]), t2n(o)
        hooks.append(m.cap_decoder.generator.register_forward_hook(gen_hook))
        hooks.append(m.video_encoder.register_forward_hook(lambda mod, a, o: rec.__setitem__("memory", t2n(o[0]))))
    opt = torch.optim.Adam(filter(lambda q: q.requires_grad, m.parameters()), lr=1e-4, betas=(0.9, 0.999))
    vf = [torch.from_numpy(feats)]
    vm = [torch.from_numpy(mask)] if mask is not None else None
    if vm is None:  # the reference's training path always passes masks (dataloader.py:507-510)
        vm = [torch.zeros(feats.shape[:2], dtype=torch.bool)]
    loss = m(vf, vm, ids.tolist())
    opt.zero_grad()
    loss.backward()
    grads = {k: t2n(q.grad) for k, q in m.named_parameters() if q.grad is not None}
    opt.step()
    after = {k: t2n(q) for k, q in m.named_parameters() if q.requires_grad}
    if per_layer:
        for h in hooks:
            h.remove()
    return float(loss), rec, grads, after, m


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    summary = {}

    # ---------------- A. tiny model, ragged batch: everything ----------------
    V = 131
    mc = model_cfg(d=64, d_in=48, H=4, ff=128, Le=2, Ld=2, alpha=0.5)
    cfg = O.cfg_from_model_config(mc, V)
    p = O.init_params(cfg, seed=11)
    feats, mask, ids = O.synthetic_batch(3, 5, 48, 7, V, seed=5, ragged=True)
    ids[1, 4:] = 0; ids[1, 3] = 102          # a short caption -> pad rows in the loss
    mask[2, 3:] = True; feats[2, 3:] = 0     # a padded video
    loss, rec, grads, after, m = run_train_case(mc, V, p, feats, mask, ids)
    # reference state_dict keys/shapes (compat surface) + torch buffers
    sd = m.state_dict()
    keys = {k: list(v.shape) for k, v in sd.items()}
    np.savez_compressed(
        os.path.join(OUT, "tiny_train.npz"),
        model_config=json.dumps(mc), vocab=V, state_keys=json.dumps(keys),
        feats=feats, mask=mask, ids=ids, loss=np.float64(loss),
        **{"param/" + k: v for k, v in p.items() if k not in O.BUFFER_KEYS},
        **{"act/" + k: v for k, v in rec.items()},
        **{"grad/" + k: v for k, v in grads.items()},
        **{"adam1/" + k: v for k, v in after.items()},
        pos_embedding_head=t2n(sd["cap_decoder.positional_encoding.pos_embedding"])[:64],
        pos_embedding_tail=t2n(sd["cap_decoder.positional_encoding.pos_embedding"])[4990:],
        temp_pe=t2n(sd["video_encoder.temp_emb.pe"]),
        layers_identical_at_init=np.bool_(True))
    summary["tiny_train.loss"] = loss

    # alpha == 1.0 (plain CE) and relu activation variant: loss + two grads
    mc1 = model_cfg(d=64, d_in=48, H=4, ff=128, Le=1, Ld=1, alpha=1.0, act="relu")
    cfg1 = O.cfg_from_model_config(mc1, V)
    p1 = O.init_params(cfg1, seed=12)
    loss1, rec1, grads1, _, _ = run_train_case(mc1, V, p1, feats, mask, ids)
    np.savez_compressed(
        os.path.join(OUT, "tiny_train_ce_relu.npz"), model_config=json.dumps(mc1), vocab=V,
        feats=feats, mask=mask, ids=ids, loss=np.float64(loss1), logits=rec1["logits"], memory=rec1["memory"],
        **{"grad/" + k: v for k, v in grads1.items()
           if k in ("cap_decoder.generator.weight", "video_encoder.unify.0.weight", "cap_decoder.tgt_to_emb.weight")})
    summary["tiny_ce_relu.loss"] = loss1

    # fresh reference model: deep-copied layers are identical at init (Appendix C.1), padding row zero
    fresh = build_ref(mc, V)
    fs = fresh.state_dict()
    assert torch.equal(fs["cap_decoder.decoder.layers.0.linear1.weight"], fs["cap_decoder.decoder.layers.1.linear1.weight"])
    assert torch.equal(fs["video_encoder.transformer_encoder.layers.0.self_attn.in_proj_weight"],
                       fs["video_encoder.transformer_encoder.layers.1.self_attn.in_proj_weight"])
    assert float(fs["cap_decoder.tgt_to_emb.weight"][0].abs().sum()) == 0.0

    # ---------------- B. greedy decode + decode_word (eval mode) ----------------
    dec = {}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}, strict=False)
    m.eval()
    for tag, B in (("b1", 1), ("b3", 3)):
        f, _, _ = O.synthetic_batch(B, 5, 48, 7, V, seed=21 + B)
        with torch.no_grad():
            vf = [torch.from_numpy(f)]
            caps_none = m.greedy_decode(vf, None, max_len=12)
            caps_false = m.greedy_decode(vf, [torch.zeros(B, 5, dtype=torch.bool)], max_len=12)
            assert caps_none == caps_false, (caps_none, caps_false)
            mem = m.video_encoder(vf, None)[0]
            # replay the loop to capture ys / logits per step (same code path: decode_word + torch.max)
            ys = torch.full((B, 1), 101, dtype=torch.long)
            logits_steps = []
            flags = [0] * B
            for _ in range(11):
                prob = m.cap_decoder.decode_word(mem, ys, None)
                logits_steps.append(t2n(prob))
                nxt = torch.max(prob, dim=1)[1]
                ys = torch.cat([ys, nxt[:, None]], 1)
                for k, fl in enumerate((nxt == 102).tolist()):
                    if fl:
                        flags[k] = 1
                if sum(flags) >= B:
                    break
        dec[f"{tag}/feats"] = f
        dec[f"{tag}/ys"] = t2n(ys)
        dec[f"{tag}/memory_eval"] = t2n(mem)
        dec[f"{tag}/logits_steps"] = np.stack(logits_steps, 1)
        dec[f"{tag}/captions"] = json.dumps(caps_none)
    # eval-mode encoder WITH a real padding mask (torch fast path / nested tensor quirk, Appendix C.6)
    with torch.no_grad():
        mem_eval_masked = m.video_encoder([torch.from_numpy(feats)], [torch.from_numpy(mask)])[0]
    dec["masked/feats"], dec["masked/mask"], dec["masked/memory_eval"] = feats, mask, t2n(mem_eval_masked)
    np.savez_compressed(os.path.join(OUT, "tiny_decode.npz"), model_config=json.dumps(mc), vocab=V, param_seed=11, **dec)

    # ---------------- C. SCE loss alone, wide logits (clamp gate exercised), pads ----------------
    rng = np.random.default_rng(3)
    lg = (rng.standard_normal((12, 257)) * 9.0).astype(np.float32)
    lb = rng.integers(1, 257, 12).astype(np.int64)
    lb[[2, 7]] = 0
    sce = {}
    for alpha in (0.5, 0.3):
        x = torch.from_numpy(lg).requires_grad_(True)
        fn = SCELoss(alpha, 1 - alpha, ignore_index=0, num_classes=257, device=CPU)
        l = fn(x, torch.from_numpy(lb))
        l.backward()
        sce[f"a{alpha}/loss"] = np.float64(float(l))
        sce[f"a{alpha}/dlogits"] = t2n(x.grad)
    x = torch.from_numpy(lg).requires_grad_(True)
    l = torch.nn.CrossEntropyLoss(ignore_index=0)(x, torch.from_numpy(lb))
    l.backward()
    sce["a1.0/loss"] = np.float64(float(l)); sce["a1.0/dlogits"] = t2n(x.grad)
    p_small = float((torch.softmax(torch.from_numpy(lg), 1) < 1e-7).float().mean())
    assert p_small > 0.2, p_small   # the clamp gate really is exercised
    np.savez_compressed(os.path.join(OUT, "sce_loss.npz"), logits=lg, labels=lb, **sce)

    # ---------------- D. nn.MultiheadAttention with merged causal(float)+pad(bool) mask ----------------
    from utils import generate_square_subsequent_mask as ref_mask  # reference utils.py:63-66
    mha = torch.nn.MultiheadAttention(32, 4, dropout=0.0, batch_first=True)
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.1); mha.out_proj.bias.normal_(0, 0.1)
    xq = torch.randn(2, 6, 32); xm = torch.randn(2, 4, 32)
    kpm = torch.tensor([[False] * 6, [False, False, False, True, True, True]])
    y_self = mha(xq, xq, xq, attn_mask=ref_mask(6), key_padding_mask=kpm, need_weights=False)[0]
    y_cross = mha(xq, xm, xm, need_weights=False)[0]
    np.savez_compressed(os.path.join(OUT, "mha.npz"), xq=t2n(xq), xm=t2n(xm), kpm=t2n(kpm), mask6=t2n(ref_mask(6)),
                        w_in=t2n(mha.in_proj_weight), b_in=t2n(mha.in_proj_bias), w_o=t2n(mha.out_proj.weight),
                        b_o=t2n(mha.out_proj.bias), y_self=t2n(y_self), y_cross=t2n(y_cross), mask19=t2n(ref_mask(19)))

    # ---------------- E. cfg-A full-size plumbing case (d=512 2/2, V=30522, B=8): slices only ----------------
    V2 = 30522
    mcA = model_cfg(d=512, d_in=512, H=8, ff=2048, Le=2, Ld=2, alpha=0.5)
    cfgA = O.cfg_from_model_config(mcA, V2)
    pA = O.init_params(cfgA, seed=666)
    fA, mA, iA = O.synthetic_batch(8, 12, 512, 20, V2, seed=0)
    lossA, recA, gradsA, afterA, mrefA = run_train_case(mcA, V2, pA, fA, mA, iA)
    lg = recA["logits"].astype(np.float64)
    lse = np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1)) + lg.max(-1)
    np.savez_compressed(
        os.path.join(OUT, "cfgA_slices.npz"), model_config=json.dumps(mcA), vocab=V2, param_seed=666, batch_seed=0,
        loss=np.float64(lossA), memory=recA["memory"], dec_out=recA["dec_out"],
        logits_head=recA["logits"][:, :, :96], logits_lse=lse.astype(np.float32),
        logits_argmax=recA["logits"].argmax(-1),
        grad_names=json.dumps(sorted(gradsA)),
        grad_norms=np.array([np.linalg.norm(gradsA[k].astype(np.float64)) for k in sorted(gradsA)]),
        grad_heads=np.stack([np.resize(gradsA[k].reshape(-1)[:32], 32) for k in sorted(gradsA)]),
        adam1_heads=np.stack([np.resize(afterA[k].reshape(-1)[:32], 32) for k in sorted(gradsA)]),
        n_params=np.int64(sum(v.numel() for v in mrefA.parameters())))
    summary["cfgA.loss"] = lossA
    summary["cfgA.n_params"] = int(sum(v.numel() for v in mrefA.parameters()))
    # full-size greedy decode ids on cfg-A weights (B=4, max_len 30), with per-step top-2 margins
    mrefA.load_state_dict({k: torch.from_numpy(v) for k, v in pA.items()}, strict=False)
    mrefA.eval()
    fD = O.synthetic_batch(4, 12, 512, 20, V2, seed=7)[0]
    with torch.no_grad():
        mem = mrefA.video_encoder([torch.from_numpy(fD)], None)[0]
        ys = torch.full((4, 1), 101, dtype=torch.long)
        margins = []
        for _ in range(29):
            prob = mrefA.cap_decoder.decode_word(mem, ys, None)
            top2 = torch.topk(prob, 2, dim=1)[0]
            margins.append(t2n(top2[:, 0] - top2[:, 1]))
            ys = torch.cat([ys, torch.max(prob, dim=1)[1][:, None]], 1)
    np.savez_compressed(os.path.join(OUT, "cfgA_decode.npz"), param_seed=666, feats_seed=7, ys=t2n(ys),
                        margins=np.stack(margins, 1), memory=t2n(mem))
    summary["cfgA.decode_min_margin"] = float(np.stack(margins, 1).min())

    with open(os.path.join(OUT, "SUMMARY.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
