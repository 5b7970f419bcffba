"""smoke(): one tiny training step + greedy decode on cuda:0, checked against the CPU oracle."""
import os
import sys

import numpy as np
import torch


def smoke():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import vct_oracle as O   # checker only
    from .model import MMT4Caption

    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    dev = torch.device("cuda", 0)
    mc = {"modal": ["CLIP4Clip"], "modal_shape": [64], "tokenizer": "ids", "vocab_size": 211, "text_enc_type": "CLIP",
          "embed_dim": 128, "dropout": 0.0, "loss_beta": 0.5, "matching": None, "activation": "gelu",
          "video_encoder": {"layer": 1, "nhead": 4, "feedforward": 256,
                            "mme": {"temporal": "encoding", "do_norm": False, "aggregation": "avg"}},
          "caption_decoder": {"layer": 2, "nhead": 4, "feedforward": 256, "sce_loss_alpha": 0.5},
          "pretrained_model": None}
    cfg = O.cfg_from_model_config(mc, 211)
    p = O.init_params(cfg, seed=3)
    feats, mask, ids = O.synthetic_batch(4, 6, 64, 9, 211, seed=1, ragged=True)
    ref_loss, ref_grads, ref_logits = O.caption_loss_and_grads(p, cfg, feats, mask, ids)
    for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 3e-2)):
        m = MMT4Caption(mc, device=dev, compute_dtype=dtype)
        m.mode("caption")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}, strict=False)
        m.train()
        f, mk, i = (torch.from_numpy(a).to(dev) for a in (feats, mask, ids))
        loss = m([f], [mk], i)
        loss.backward()
        assert abs(float(loss.detach()) - ref_loss) < tol * abs(ref_loss), (float(loss.detach()), ref_loss)
        for k, g in ref_grads.items():
            mine = m._ps.g[k].double().cpu().numpy()
            err = np.linalg.norm(mine - g) / max(np.linalg.norm(g), 1e-30)
            assert err < tol * 10, (k, err)
        if dtype == torch.float32:
            ys = m.greedy_decode_ids([f], None, max_len=8).cpu().numpy()
            ref = O.greedy_decode_ids(p, cfg, feats, None, max_len=8)
            assert np.array_equal(ys, ref), (ys, ref)
    print("smoke ok: HIP caption step + greedy decode match the CPU oracle (fp32 tight, bf16 loose)")
