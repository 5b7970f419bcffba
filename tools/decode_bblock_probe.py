#!/usr/bin/env python
"""Per-token cost of the batched greedy-decode step (graph replays): block-per-launch (vct_decode_bblock) vs skinny projections."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import MODEL_CFG  # noqa: E402
from vct_amd.engine import DecodeState, DecoderEngine  # noqa: E402
from vct_amd.model import MMT4Caption  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(666)
m = MMT4Caption(MODEL_CFG, device=dev, compute_dtype=torch.bfloat16); m.mode("caption"); m.eval()
m._ps.refresh_shadow()
enc, dec = m.video_encoder._engine(), m.cap_decoder._engine()
for B in (2, 16, 64, 128, 256):
    feats = torch.randn(B, 12, 512, device=dev)
    mem = enc.forward(feats, None, False)
    for bblock in (True, False):
        DecoderEngine.bblock_decode = bblock
        st = DecodeState(dec, B, 13, 30)
        dec.decode_begin(st, mem, 101, 0)
        for t in range(1, 30):
            dec.decode_step(st, t, 102)
        torch.cuda.synchronize()
        graphs = {}
        for t in range(1, 30):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                dec.decode_step(st, t, 102)
            graphs[t] = g
        for t in range(1, 30):
            graphs[t].replay()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(10):
            for t in range(1, 30):
                graphs[t].replay()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print(f"B {B:4d} bblock {int(bblock)}: {(t4 - t3) / 290 * 1e6:7.1f} us per token step (graph replay)", flush=True)
