#!/usr/bin/env python
"""Per-token cost of the greedy-decode step alone (no encoder, no host sync): eager launches vs per-position hipGraph replays."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import MODEL_CFG  # noqa: E402
from vct_amd.engine import DecodeState  # noqa: E402
from vct_amd.model import MMT4Caption  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(666)
for dtype in (torch.bfloat16,):
    m = MMT4Caption(MODEL_CFG, device=dev, compute_dtype=dtype); m.mode("caption"); m.eval()
    m._ps.refresh_shadow()
    enc, dec = m.video_encoder._engine(), m.cap_decoder._engine()
    for B in (1, 2, 4, 128):
        feats = torch.randn(B, 12, 512, device=dev)
        mem = enc.forward(feats, None, False)
        for small in (True, False):
            type(dec).small_batch_decode = small
            st = DecodeState(dec, B, 13, 30)
            dec.decode_begin(st, mem, 101, 0)
            for t in range(1, 30):
                dec.decode_step(st, t, 102)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(1, 30):
                dec.decode_step(st, t, 102)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
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
            for _ in range(5):
                for t in range(1, 30):
                    graphs[t].replay()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            print(f"B={B:3d} small={small!s:5s} eager: host {1e6*(t1-t0)/29:6.1f} us/tok, wall {1e6*(t2-t0)/29:6.1f} us/tok | graph replay {1e6*(t4-t3)/145:6.1f} us/tok", flush=True)
