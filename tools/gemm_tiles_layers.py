#!/usr/bin/env python
"""Every tile configuration of the general bf16 kernel on the LAYER GEMM shapes of cfg-B (decoder rows 4864, encoder rows 3328), isolated,
hip-event timed, random data: the landscape behind gemm_plan's choices.  Dev tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import run  # noqa: E402

TILES = [(5, "128x128w8"), (8, "128x64w8"), (4, "64x64w4"), (6, "256x128w8"), (7, "320x128w8"), (9, "64x128w8"), (0, "auto")]
SHAPES = []
for rows, tag in ((4864, "dec"), (3328, "enc")):
    SHAPES += [
        (f"{tag} qkv_fwd  NT {rows}x1536x512", 0, 1, rows, 1536, 512, torch.bfloat16),
        (f"{tag} ffn1_fwd NT {rows}x2048x512", 0, 1, rows, 2048, 512, torch.bfloat16),
        (f"{tag} ffn2_fwd NT {rows}x512x2048", 0, 1, rows, 512, 2048, torch.bfloat16),
        (f"{tag} out_fwd  NT {rows}x512x512", 0, 1, rows, 512, 512, torch.bfloat16),
        (f"{tag} ffn2_dx  NN {rows}x2048x512", 0, 0, rows, 2048, 512, torch.bfloat16),
        (f"{tag} ffn1_dx  NN {rows}x512x2048", 0, 0, rows, 512, 2048, torch.bfloat16),
        (f"{tag} qkv_dx   NN {rows}x512x1536", 0, 0, rows, 512, 1536, torch.bfloat16),
        (f"{tag} out_dx   NN {rows}x512x512", 0, 0, rows, 512, 512, torch.bfloat16),
    ]
SHAPES += [("enc kv_fwd   NT 3328x1024x512", 0, 1, 3328, 1024, 512, torch.bfloat16)]

if __name__ == "__main__":
    print(f"{'shape':34s} " + " ".join(f"{n:>14s}" for _, n in TILES))
    for s in SHAPES:
        row = []
        for t, _ in TILES:
            try:
                ms, tf = run(*s, (t + 20) if t else 0, workspace=False)
                row.append(f"{ms*1e3:6.1f}us {tf:4.0f}T")
            except Exception as e:
                row.append("      n/a     ")
        print(f"{s[0]:34s} " + " ".join(f"{r:>14s}" for r in row), flush=True)
