#!/usr/bin/env python
"""Tile sweep of the ENCODER's layer GEMMs (M = 3328 rows at cfg-B) beside the plan the library picks by itself (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gemm_bench as G
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3328
SH = [("qkv_fwd  NT", 0, 1, M, 1536, 512, torch.bfloat16), ("ffn1_fwd NT", 0, 1, M, 2048, 512, torch.bfloat16),
      ("ffn2_fwd NT", 0, 1, M, 512, 2048, torch.bfloat16), ("out_fwd  NT", 0, 1, M, 512, 512, torch.bfloat16),
      ("ffn2_dx  NN", 0, 0, M, 2048, 512, torch.bfloat16), ("ffn1_dx  NN", 0, 0, M, 512, 2048, torch.bfloat16),
      ("qkv_dx   NN", 0, 0, M, 512, 1536, torch.bfloat16), ("out_dx   NN", 0, 0, M, 512, 512, torch.bfloat16),
      ("kv_fwd   NT", 0, 1, M, 1024, 512, torch.bfloat16), ("unify    NT", 0, 1, 3072, 512, 512, torch.bfloat16)]
print(f"{'shape (M=%d)' % M:28s} {'auto':>14s} | " + " ".join(f"{t:>14s}" for t in ["128x128w8", "128x64w8", "64x64"]))
for s in SH:
    a_ms, a_tf = G.run(*s, 0)
    row = []
    for tile in (5, 8, 4):
        ms, tf = G.run(*s, tile + 20)
        row.append(f"{ms*1e3:6.1f}us {tf:4.0f}TF")
    print(f"{s[0]} {s[3]}x{s[4]}x{s[5]:<6d} {a_ms*1e3:6.1f}us {a_tf:4.0f}TF | " + " ".join(row), flush=True)
