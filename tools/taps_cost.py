#!/usr/bin/env python
"""What the live timing taps (HIP event records inside the step) cost: step time with and without them (dev tool)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from vct_amd import ops
from vct_amd.model import MMT4Caption
from vct_amd.trainer import CaptionTrainer, build_optimizer
from vct_amd.utils import setup_seed
dev = torch.device("cuda", 0)
setup_seed(666)
for taps_on in (False, True, False, True):
    m = MMT4Caption(B.MODEL_CFG, device=dev, compute_dtype=torch.bfloat16); m.mode("caption"); m.train()
    opt, _ = build_optimizer(B.TRAIN_CFG, m)
    tr = CaptionTrainer(m, opt, None, launch_list=True)
    feats, mask, ids = B.synthetic(256, 0, dev)
    ops.taps_enable(taps_on)
    for _ in range(10):
        tr.step(feats, mask, ids)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        tr.step(feats, mask, ids)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    ops.taps_enable(False)
    print(f"taps {'on ' if taps_on else 'off'}: {dt * 1e3:.4f} ms/step  {256 / dt:.0f} samples/s", flush=True)
    del tr, opt, m
