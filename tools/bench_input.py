#!/usr/bin/env python
"""Input-path measurement: how fast batches of the cfg-B shape can be PRODUCED, and the
end-to-end training rate when the step is fed by each loader.  Synthetic split written to a temp dir in the reference's
MSR-VTT format: N videos x 12 frames x 512 fp32, C captions per video.

  host   : reference-layout path -- Dataset.__getitem__ (np.load per sample) + collate_fn + .to(device) per batch
  device : DeviceLoader -- split resident in HBM, one gather-and-pad launch + one index_select per batch
Prints one JSON line per measurement.  Dev tool (not part of bench.py's contract)."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import data  # noqa: E402
from vct_amd.model import MMT4Caption  # noqa: E402
from vct_amd.trainer import CaptionTrainer, FusedAdam  # noqa: E402

MODEL_CFG = {"modal": ["CLIP4Clip"], "modal_shape": [512], "tokenizer": "ids", "text_enc_type": "CLIP", "embed_dim": 512,
             "dropout": 0.3, "loss_beta": 0.5, "matching": {"enable_tem": False, "matching_loss": "CSL"}, "activation": "gelu",
             "video_encoder": {"layer": 2, "nhead": 8, "feedforward": 2048,
                               "mme": {"temporal": "encoding", "modal_different": True, "do_norm": False, "aggregation": "avg"}, "aoa": False},
             "caption_decoder": {"layer": 2, "nhead": 8, "feedforward": 2048, "sce_loss_alpha": 0.5}, "pretrained_model": None}


class WordTok:
    """18 words + [CLS]/[SEP] per caption -> 20 ids (the cfg-B caption length)"""
    pad_id, start_id, end_id = 0, 101, 102

    def __call__(self, captions):
        rows = [[101] + [1000 + (hash(w) % 29000) for w in c.split()] + [102] for c in captions]
        S = max(map(len, rows))
        ids = torch.tensor([r + [0] * (S - len(r)) for r in rows], dtype=torch.long)
        return ids, ids == 0


def make_split(d, n_videos, caps_per_video, rng):
    os.makedirs(os.path.join(d, "feats"))
    vids = [f"video{i}" for i in range(n_videos)]
    for v in vids:
        np.save(os.path.join(d, "feats", v + ".npy"), rng.standard_normal((12, 512)).astype(np.float32))
    words = [f"w{i}" for i in range(500)]
    sents = [{"video_id": v, "caption": " ".join(words[j] for j in rng.integers(0, 500, 18))} for v in vids for _ in range(caps_per_video)]
    json.dump({"videos": [{"video_id": v, "split": "train"} for v in vids], "sentences": sents}, open(os.path.join(d, "ann.json"), "w"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=2048)
    ap.add_argument("--caps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--host-batches", type=int, default=12)
    args = ap.parse_args()
    dev = torch.device("cuda")
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        make_split(d, args.videos, args.caps, rng)
        ds = data.MSRVTT_Dataset([os.path.join(d, "feats")], os.path.join(d, "ann.json"), split_type="train", mode="by_caption")
        tok = WordTok()
        # ---- production rate of batches -------------------------------------------------------------
        host = torch.utils.data.DataLoader(ds, batch_size=args.batch, collate_fn=data.collate_fn, shuffle=True)
        t0, n = time.perf_counter(), 0
        for i, (f, m, caps, _v) in enumerate(host):
            f = [x.to(dev) for x in f]; m = [x.to(dev) for x in m]; ids = tok(caps)[0].to(dev)
            n += f[0].shape[0]
            if i + 1 == args.host_batches:
                break
        torch.cuda.synchronize()
        host_rate = n / (time.perf_counter() - t0)
        print(json.dumps({"what": "batch production, host path (np.load per sample + collate + tokenise + H2D)", "samples_per_s": round(host_rate, 1),
                          "batches": args.host_batches, "batch": args.batch}), flush=True)
        t0 = time.perf_counter()
        dl = data.DeviceLoader(ds, args.batch, tok, dev, shuffle=True, drop_last=True, feat_dtype=torch.bfloat16)
        torch.cuda.synchronize()
        setup = time.perf_counter() - t0
        for _ in dl:
            pass
        torch.cuda.synchronize()
        t0, n = time.perf_counter(), 0
        for e in range(3):
            dl.set_epoch(e)
            for f, m, caps, _v in dl:
                n += f[0].shape[0]
        torch.cuda.synchronize()
        dev_rate = n / (time.perf_counter() - t0)
        print(json.dumps({"what": "batch production, DeviceLoader (split resident in HBM, gather-pad kernel)", "samples_per_s": round(dev_rate, 1),
                          "one_time_upload_and_tokenise_s": round(setup, 2), "items": len(ds)}), flush=True)
        # ---- end to end: the cfg-B training step fed by DeviceLoader ----------------------------------
        torch.manual_seed(666)
        model = MMT4Caption(MODEL_CFG, device=dev, compute_dtype=torch.bfloat16)
        model.mode("caption"); model.train()
        tr = CaptionTrainer(model, FusedAdam(model, lr=1e-4))
        for f, m, caps, _v in dl:            # warm-up epoch (allocations)
            tr.step(f[0], m[0], caps)
        torch.cuda.synchronize()
        t0, n = time.perf_counter(), 0
        for e in range(2):
            dl.set_epoch(10 + e)
            for f, m, caps, _v in dl:
                loss = tr.step(f[0], m[0], caps)
                n += f[0].shape[0]
        torch.cuda.synchronize()
        e2e = n / (time.perf_counter() - t0)
        print(json.dumps({"what": "cfg-B training step fed by DeviceLoader (bf16 features, ids from the device-side cache)",
                          "samples_per_s": round(e2e, 1), "steps": n // args.batch, "loss": float(loss)}), flush=True)


if __name__ == "__main__":
    main()
