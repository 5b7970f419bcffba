#!/usr/bin/env python
"""How many cross-stream edges (event record + wait) one training step issues, and from where (dev tool, needs a GPU)."""
import os, sys, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from vct_amd import ops
from vct_amd.model import MMT4Caption
from vct_amd.trainer import CaptionTrainer, build_optimizer
dev = torch.device("cuda", 0)
m = MMT4Caption(B.MODEL_CFG, device=dev, compute_dtype=torch.bfloat16); m.mode("caption"); m.train()
opt, _ = build_optimizer(B.TRAIN_CFG, m)
tr = CaptionTrainer(m, opt, None)
feats, mask, ids = B.synthetic(256, 0, dev)
for _ in range(2):
    tr.step(feats, mask, ids)
cnt = collections.Counter()
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        st = traceback.extract_stack(limit=4)
        cnt[(name, " < ".join(f"{os.path.basename(fr.filename)}:{fr.lineno}:{fr.name}" for fr in reversed(st[:-1])))] += 1
        return orig(*a, **k)
    setattr(ops, name, f)
for n in ("stream_wait", "sync_record", "sync_wait"):
    wrap(n)
tr.step(feats, mask, ids)
torch.cuda.synchronize()
tot = collections.Counter()
for (n, where), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    tot[n] += c
    print(f"{c:3d}  {n:12s} {where}")
print(dict(tot))
