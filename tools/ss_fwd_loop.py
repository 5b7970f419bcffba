#!/usr/bin/env python
"""The forward of the cfg-B training step (two sample-stationary stack launches + vocabulary projection + loss) in a loop, nothing else
on the chip: kernel times for comparison with the same launches inside the whole step (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import vct_amd  # noqa: E402,F401
from vct_amd.model import MMT4Caption  # noqa: E402
from vct_amd.utils import setup_seed  # noqa: E402

dev = torch.device("cuda:0")
setup_seed(666)
m = MMT4Caption(bench.MODEL_CFG, device=dev, compute_dtype=torch.bfloat16)
m.mode("caption"); m.train()
feats, mask, ids = bench.synthetic(256, 0, dev)
only = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for it in range(30):
    loss, _ = m._forward_loss(feats, mask, ids, True)
    if only == "step":
        m._backward()
torch.cuda.synchronize()
print("loss", float(loss))
