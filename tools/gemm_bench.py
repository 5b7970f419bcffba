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
    ("square   NT 4096x4096x4096", 0, 1, 4096, 4096, 4096, torch.bfloat16),
    ("square   TN 4096x4096x4096", 1, 0, 4096, 4096, 4096, torch.float32),
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


def run(name, ta, tb, M, N, K, odt, tile, iters=20, workspace=True):
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
    if workspace:        # (the engine's layer GEMMs pass none: no split-K)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.reserved = tile
    d.split_k = int(os.environ.get("VCT_BENCH_SPLIT", "0"))     # 0: the library's own choice
    st = L.stream_ptr()
    for _ in range(int(os.environ.get("VCT_BENCH_WARM", "3"))):     # (a few hundred launches bring an idle part to its sustained clocks)
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


def vendor(name, ta, tb, M, N, K, odt, iters=20):
    """torch.matmul (hipBLASLt / rocBLAS) on the same shape: yardstick only"""
    a = torch.randn((K, M) if ta else (M, K), device=DEV).bfloat16()
    b = torch.randn((N, K) if tb else (K, N), device=DEV).bfloat16()
    f = lambda: (a.t() if ta else a) @ (b.t() if tb else b)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12


def grouped(rows, mem_rows, splits=(0, 1, 2, 3), iters=20):
    """a decoder / encoder layer's weight gradients as one grouped launch"""
    from vct_amd import ops
    dec = [(1536, 512, rows), (512, 512, rows), (512, 512, rows), (1024, 512, mem_rows), (512, 512, rows), (2048, 512, rows), (512, 2048, rows)]
    enc = [(1536, 512, mem_rows), (512, 512, mem_rows), (2048, 512, mem_rows), (512, 2048, mem_rows)]
    scratch = ops.GemmScratch(DEV, 256 << 20)
    for name, shp in (("dec layer", dec), ("enc layer", enc)):
        items = []
        fl = 0.0
        for M, N, r in shp:
            dy = torch.randn(r, M, device=DEV).bfloat16(); x = torch.randn(r, N, device=DEV).bfloat16()
            items.append((dy, x, torch.empty(M, N, device=DEV), torch.empty(M, device=DEV)))
            fl += 2.0 * M * N * r
        for tile in (0, 5, 8, 4):
          for sp in splits:
            for _ in range(3):
                ops.gemm_grouped(items, scratch, split_k=sp, tile=tile)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(iters):
                ops.gemm_grouped(items, scratch, split_k=sp, tile=tile)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            print(f"grouped {name} tile={tile} split={sp}: {ms*1e3:7.1f} us {fl/(ms*1e-3)/1e12:5.0f} TF", flush=True)


if __name__ == "__main__":
    sel = sys.argv[1:]
    if sel and sel[0] == "--grouped":
        grouped(4864, 3328)
        sys.exit(0)
    if sel and sel[0] == "--plan":   # the library's own plan for the named shapes (no vendor leg)
        for s in SHAPES:
            if any(x in s[0] for x in sel[1:]):
                ms, tf = run(*s, 0, iters=int(os.environ.get("VCT_BENCH_ITERS", "30")))
                print(f"{s[0]:30s} {ms*1e3:7.1f}us {tf:4.0f}TF", flush=True)
        sys.exit(0)
    if sel and sel[0] == "--auto":   # the plan the library picks by itself vs the vendor library
        print(f"{'shape':30s} {'auto':>15s} {'vendor':>15s}")
        for s in SHAPES:
            ms, tf = run(*s, 0)
            vm, vt = vendor(*s)
            print(f"{s[0]:30s} {ms*1e3:7.1f}us {tf:4.0f}TF {vm*1e3:7.1f}us {vt:4.0f}TF", flush=True)
        sys.exit(0)
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
