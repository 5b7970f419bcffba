#!/usr/bin/env python
"""GEMM micro-benchmark over the caption path's real shapes (cfg-B, B=256): every tile
configuration of the bf16 kernel, hip-event timed, random data.  Dev tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import _lib as L  # noqa: E402

DEV = "cuda"
SHAPES = [  # name, ta, tb, M, N, K, out
    ("gen_fwd  NT 4864x30522x512", 0, 1, 4864, 30522, 512, torch.bfloat16),
    ("gen_dx   NN 4864x512x30522", 0, 0, 4864, 512, 30522, torch.bfloat16),
    ("gen_dw   TN 30522x512x4864", 1, 0, 30522, 512, 4864, torch.float32),
    ("qkv_fwd  NT 4864x1536x512", 0, 1, 4864, 1536, 512, torch.bfloat16),
    ("ffn1_fwd NT 4864x2048x512", 0, 1, 4864, 2048, 512, torch.bfloat16),
    ("ffn2_fwd NT 4864x512x2048", 0, 1, 4864, 512, 2048, torch.bfloat16),
    ("out_fwd  NT 4864x512x512", 0, 1, 4864, 512, 512, torch.bfloat16),
    ("ffn2_dx  NN 4864x2048x512", 0, 0, 4864, 2048, 512, torch.bfloat16),
    ("ffn1_dx  NN 4864x512x2048", 0, 0, 4864, 512, 2048, torch.bfloat16),
    ("qkv_dx   NN 4864x512x1536", 0, 0, 4864, 512, 1536, torch.bfloat16),
    ("ffn2_dw  TN 512x2048x4864", 1, 0, 512, 2048, 4864, torch.float32),
    ("ffn1_dw  TN 2048x512x4864", 1, 0, 2048, 512, 4864, torch.float32),
    ("qkv_dw   TN 1536x512x4864", 1, 0, 1536, 512, 4864, torch.float32),
    ("out_dw   TN 512x512x4864", 1, 0, 512, 512, 4864, torch.float32),
]


def run(name, ta, tb, M, N, K, odt, tile, iters=20):
    Kp, Np, Mp = (K + 31) // 32 * 32, (N + 31) // 32 * 32, (M + 31) // 32 * 32
    a = (torch.randn((Kp, Mp) if ta else (M, Kp), device=DEV)).to(torch.bfloat16)
    b = (torch.randn((N, Kp) if tb else (Kp, Np), device=DEV)).to(torch.bfloat16)
    if not ta and Kp != K:
        a[:, K:] = 0
    out = torch.empty(M, Np, dtype=odt, device=DEV)
    ws = torch.empty(64 * 1024 * 1024 // 4, device=DEV)
    lib = L.load()
    d = L.GemmDesc()
    d.dtype, d.out_dtype, d.ta, d.tb, d.M, d.N, d.K = L.BF16, L.dtype_code(odt), ta, tb, M, N, K
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.reserved = tile
    st = L.stream_ptr()
    for _ in range(3):
        L.check(lib.vct_gemm(d, st), name)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        lib.vct_gemm(d, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12


if __name__ == "__main__":
    sel = sys.argv[1:]
    print(f"{'shape':30s} nbuf " + " ".join(f"{t:>15s}" for t in ["128x128w8", "128x64w8", "64x64"]))
    for s in SHAPES:
        if sel and not any(x in s[0] for x in sel):
            continue
        for nbuf in (1, 2):
            row = []
            for tile in (5, 8, 4):
                ms, tf = run(*s, tile + 10 * nbuf)
                row.append(f"{ms*1e3:7.1f}us {tf:4.0f}TF")
            print(f"{s[0]:30s} {nbuf:4d} " + " ".join(f"{r:>15s}" for r in row), flush=True)
