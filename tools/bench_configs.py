#!/usr/bin/env python
"""Step time of the other BASELINE.json shapes (not the headline metric): cfg-D deep (d=1024, 6+6 layers, T=32, S=40,
head_dim 128) and the shipped MSR-VTT shape (d=768, 1 enc + 3 dec layers), batch 256, bf16, dropout 0.3, fwd+bwd+Adam."""
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import MODEL_CFG, TRAIN_CFG  # noqa: E402
from vct_amd.model import MMT4Caption  # noqa: E402
from vct_amd.trainer import CaptionTrainer, build_optimizer  # noqa: E402


def flops(B, T, S, d_in, d, ff, Le, Ld, V):
    Te, Sd = T + 1, S - 1
    unify = 2 * B * T * d_in * d
    enc = B * Te * (8 * d * d + 4 * d * ff) + 4 * B * Te * Te * d
    dec = B * Sd * (12 * d * d + 4 * d * ff) + 4 * B * Te * d * d + 4 * B * Sd * Sd * d + 4 * B * Sd * Te * d
    return 3 * (unify + Le * enc + Ld * dec + 2 * B * Sd * d * V)


def run(name, d, Le, Ld, T, S, B=256, steps=20):
    dev = torch.device("cuda", 0)
    mc = copy.deepcopy(MODEL_CFG)
    mc["embed_dim"] = d
    mc["video_encoder"]["layer"], mc["caption_decoder"]["layer"] = Le, Ld
    torch.manual_seed(666)
    m = MMT4Caption(mc, device=dev, compute_dtype=torch.bfloat16)
    m.mode("caption"); m.train()
    opt, _ = build_optimizer(TRAIN_CFG, m)
    tr = CaptionTrainer(m, opt)
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(B, T, 512, generator=g).to(dev)
    mask = torch.zeros(B, T, dtype=torch.bool, device=dev)
    ids = torch.randint(1000, 30000, (B, S), generator=g).to(dev); ids[:, 0] = 101; ids[:, -1] = 102
    for _ in range(5):
        tr.step(feats, mask, ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(feats, mask, ids)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    fl = flops(B, T, S, 512, d, 2048, Le, Ld, 30522)
    print(json.dumps({"config": name, "ms_per_step": round(ms, 3), "samples_per_s": round(B / ms * 1e3, 1),
                      "step_tflops": round(fl / ms / 1e9, 1), "params_M": round(sum(p.numel() for p in m.parameters()) / 1e6, 1),
                      "loss": float(loss)}), flush=True)
    del m, tr, opt
    torch.cuda.empty_cache()


if __name__ == "__main__":
    run("cfg-B d=512 2+2 T=12 S=20", 512, 2, 2, 12, 20)
    run("shipped d=768 1+3 T=12 S=20", 768, 1, 3, 12, 20)
    run("cfg-D d=1024 6+6 T=32 S=40", 1024, 6, 6, 32, 40)
