"""Generate tests/golden/cfgD_slices.npz and cfgB_decode.npz from the REAL reference (build container only).

    python oracle/make_golden_cfgD.py       # needs /root/reference (read-only) and torch CPU; ~2 minutes

cfg-D = BASELINE.json configs[3]: 6 encoder + 6 decoder layers, d_model 1024, 8 heads (head_dim 128), ff 2048, 32 frames,
40-token captions, V = 30522, batch 8 (the CPU-sized batch SURVEY.md 8(d) names).  Only slices and norms are stored (the
logits alone would be 38 MB).  cfgB_decode = greedy decode of the d=512 2+2 model at batch 1 and 16 with per-step top-2
margins.  Helpers (reference import, stubs) come from make_golden.py; nothing of the reference is copied."""
import json
import os

import numpy as np
import torch

import make_golden as G
import vct_oracle as O

OUT = G.OUT


def main():
    V = 30522
    mc = G.model_cfg(d=1024, d_in=512, H=8, ff=2048, Le=6, Ld=6, alpha=0.5)
    cfg = O.cfg_from_model_config(mc, V)
    p = O.init_params(cfg, seed=4242)
    feats, mask, ids = O.synthetic_batch(8, 32, 512, 40, V, seed=3, ragged=True)
    loss, rec, grads, after, mref = G.run_train_case(mc, V, p, feats, mask, ids, per_layer=True)
    lg = rec["logits"].astype(np.float64)
    lse = np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1)) + lg.max(-1)
    names = sorted(grads)
    np.savez_compressed(
        os.path.join(OUT, "cfgD_slices.npz"), model_config=json.dumps(mc), vocab=V, param_seed=4242, batch_seed=3,
        loss=np.float64(loss), memory_head=rec["memory"][:, :, :64], dec_out_head=rec["dec_out"][:, :, :64],
        enc_layer5_head=rec["enc_layer5"][:, :, :32], dec_layer5_head=rec["dec_layer5"][:, :, :32],
        logits_head=rec["logits"][:, :, :64], logits_lse=lse.astype(np.float32), logits_argmax=rec["logits"].argmax(-1),
        grad_names=json.dumps(names),
        grad_norms=np.array([np.linalg.norm(grads[k].astype(np.float64)) for k in names]),
        grad_heads=np.stack([np.resize(grads[k].reshape(-1)[:32], 32) for k in names]),
        adam1_heads=np.stack([np.resize(after[k].reshape(-1)[:32], 32) for k in names]),
        param_order=json.dumps([k for k, _ in mref.named_parameters()]),
        n_params=np.int64(sum(v.numel() for v in mref.parameters())))
    print("cfgD loss", loss, "params", sum(v.numel() for v in mref.parameters()))

    # ---- greedy decode of the cfg-B model (d=512, 2+2), batch 1 and 16, 29 steps, top-2 margins per step ----
    mcB = G.model_cfg(d=512, d_in=512, H=8, ff=2048, Le=2, Ld=2, alpha=0.5)
    cfgB = O.cfg_from_model_config(mcB, V)
    pB = O.init_params(cfgB, seed=777)
    m = G.build_ref(mcB, V)
    G.load_np_state(m, pB)
    m.eval()
    out = {}
    for B, seed in ((1, 11), (16, 12)):
        f = O.synthetic_batch(B, 12, 512, 20, V, seed=seed)[0]
        with torch.no_grad():
            mem = m.video_encoder([torch.from_numpy(f)], None)[0]
            ys = torch.full((B, 1), 101, dtype=torch.long)
            margins = []
            for _ in range(29):
                prob = m.cap_decoder.decode_word(mem, ys, None)
                top2 = torch.topk(prob, 2, dim=1)[0]
                margins.append(G.t2n(top2[:, 0] - top2[:, 1]))
                ys = torch.cat([ys, torch.max(prob, dim=1)[1][:, None]], 1)
        out[f"ys_b{B}"] = G.t2n(ys)
        out[f"margins_b{B}"] = np.stack(margins, 1)
        out[f"feats_seed_b{B}"] = seed
        print("decode B", B, "min margin", float(np.stack(margins, 1).min()))
    np.savez_compressed(os.path.join(OUT, "cfgB_decode.npz"), model_config=json.dumps(mcB), vocab=V, param_seed=777, **out)
    for fn in ("cfgD_slices.npz", "cfgB_decode.npz"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
