#!/usr/bin/env python
"""Batch-128 greedy-decode steps, eager launches (for a rocprofv3 kernel trace of the step's kernel sequence)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import MODEL_CFG  # noqa: E402
from vct_amd.engine import DecodeState  # noqa: E402
from vct_amd.model import MMT4Caption  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(666)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = MMT4Caption(MODEL_CFG, device=dev, compute_dtype=torch.bfloat16); m.mode("caption"); m.eval()
m._ps.refresh_shadow()
enc, dec = m.video_encoder._engine(), m.cap_decoder._engine()
feats = torch.randn(B, 12, 512, device=dev)
mem = enc.forward(feats, None, False)
st = DecodeState(dec, B, 13, 30)
for rep in range(3):
    dec.decode_begin(st, mem, 101, 0)
    for t in range(1, 30):
        dec.decode_step(st, t, 102)
        torch.cuda.synchronize()
