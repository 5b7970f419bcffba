#!/usr/bin/env python
"""Persistent-tile layer GEMM (VCT_GEMM_PT=1, csrc/vct_gemm256.hip gemm_pt_kernel) vs the one-tile-per-workgroup kernel on the layer
shapes of cfg-B: correctness against fp64 and recorded-replay timing.  Run once per setting of VCT_GEMM_PT / VCT_GEMM_PT_TILE.  Dev tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops  # noqa: E402

DEV, dt = "cuda", torch.bfloat16
SHAPES = [("dec qkv", 4864, 1536, 512), ("dec ffn1", 4864, 2048, 512), ("dec ffn2", 4864, 512, 2048), ("dec out", 4864, 512, 512),
          ("enc qkv", 3328, 1536, 512), ("enc ffn1", 3328, 2048, 512), ("enc ffn2", 3328, 512, 2048), ("enc kv", 3328, 1024, 512),
          ("ragged", 1000, 520, 200), ("big", 19456, 2048, 512)]


def timeit(fn, iters=50):
    for _ in range(3):
        fn()
    ll = ops.LaunchList()
    with ll.record():
        fn()
    ll.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        ll.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    tag = f"PT={os.environ.get('VCT_GEMM_PT', '0')} TILE={os.environ.get('VCT_GEMM_PT_TILE', '0')}"
    for name, M, N, K in SHAPES:
        Kp = (K + 7) // 8 * 8
        x = torch.zeros(M, Kp, dtype=dt, device=DEV); x[:, :K] = torch.randn(M, K, generator=g).to(dt).to(DEV)
        w = torch.zeros(N, Kp, dtype=dt, device=DEV); w[:, :K] = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        Np = (N + 7) // 8 * 8
        out = torch.full((M, Np), 7.0, dtype=dt, device=DEV)
        ops.gemm(x[:, :K] if Kp == K else x, w[:, :K] if Kp == K else w, out[:, :N] if Np != N else out, bias=b) if False else None
        xa, wa = (x, w) if Kp == K else (x, w)
        o = out if Np == N else out[:, :N]
        ops.gemm(xa, wa, o, bias=b)
        ref = x.double() @ w.double().t() + b.double()
        err = float((o.double() - ref).norm() / ref.norm())
        t = timeit(lambda: ops.gemm(xa, wa, o, bias=b))
        print(f"{tag} {name:9s} {M}x{N}x{K}: {t:6.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF  rel err {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
