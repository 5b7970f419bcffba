#!/usr/bin/env python
"""Is there idle time between two replays of the recorded step?  One launch list holding TWO steps vs two replays of a one-step
list (dev tool)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from vct_amd import ops
from vct_amd.model import MMT4Caption
from vct_amd.trainer import CaptionTrainer, build_optimizer
from vct_amd.utils import setup_seed
dev = torch.device("cuda", 0)
setup_seed(666)
m = MMT4Caption(B.MODEL_CFG, device=dev, compute_dtype=torch.bfloat16); m.mode("caption"); m.train()
opt, _ = build_optimizer(B.TRAIN_CFG, m)
tr = CaptionTrainer(m, opt, None, launch_list=True)
f, k, i = tr.adopt_inputs(*B.synthetic(256, 0, dev))
for _ in range(5):
    tr.step(f, k, i)
torch.cuda.synchronize()
for n in (1, 2, 4):
    ll = ops.LaunchList()
    with ll.record():
        for _ in range(n):
            tr._step_kernels(f, k, i)
    for _ in range(3):
        ll.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 40 // n
    for _ in range(reps):
        ll.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (reps * n)
    print(f"{n} step(s) per recorded list: {dt * 1e3:.4f} ms/step", flush=True)
