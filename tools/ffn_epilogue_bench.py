#!/usr/bin/env python
"""What the fused epilogue of the FFN GEMMs costs (cfg-B shapes, bf16): plain GEMM vs + bias vs + GELU vs + saved
pre-activation vs + dropout, and the same for the dX GEMM with the activation derivative."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops  # noqa: E402

DEV, dt = "cuda", torch.bfloat16


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    M, d, ff = 4864, 512, 2048
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, d, generator=g).to(dt).to(DEV); w1 = (torch.randn(ff, d, generator=g) / 22).to(dt).to(DEV); b1 = torch.randn(ff, generator=g).to(DEV)
    w2 = (torch.randn(d, ff, generator=g) / 45).to(dt).to(DEV); b2 = torch.randn(d, generator=g).to(DEV)
    h = torch.empty(M, ff, dtype=dt, device=DEV); hpre = torch.empty_like(h); f = torch.empty(M, d, dtype=dt, device=DEV)
    df = torch.randn(M, d, generator=g).to(dt).to(DEV); dh = torch.empty_like(h); dx = torch.empty(M, d, dtype=dt, device=DEV)
    seed = torch.tensor([3], dtype=torch.int32, device=DEV)
    drop = (seed, 5, 0.3)
    rows = [("ffn1 plain", lambda: ops.gemm(x, w1, h)),
            ("ffn1 +bias", lambda: ops.gemm(x, w1, h, bias=b1)),
            ("ffn1 +bias+gelu", lambda: ops.gemm(x, w1, h, bias=b1, act="gelu")),
            ("ffn1 +bias+relu", lambda: ops.gemm(x, w1, h, bias=b1, act="relu")),
            ("ffn1 +bias+drop", lambda: ops.gemm(x, w1, h, bias=b1, dropout=drop)),
            ("ffn1 +bias+preact", lambda: ops.gemm(x, w1, h, bias=b1, preact=hpre)),
            ("ffn1 +bias+gelu+preact", lambda: ops.gemm(x, w1, h, bias=b1, act="gelu", preact=hpre)),
            ("ffn1 +bias+gelu+preact+drop", lambda: ops.gemm(x, w1, h, bias=b1, act="gelu", preact=hpre, dropout=drop)),
            ("ffn1 +bias+relu+preact+drop", lambda: ops.gemm(x, w1, h, bias=b1, act="relu", preact=hpre, dropout=drop)),
            ("ffn2 plain", lambda: ops.gemm(h, w2, f)),
            ("ffn2 +bias", lambda: ops.gemm(h, w2, f, bias=b2)),
            ("ffn2_dx plain", lambda: ops.gemm(df, w2, dh, ta=False, tb=False)),
            ("ffn2_dx +dgelu", lambda: ops.gemm(df, w2, dh, ta=False, tb=False, act="gelu", dact_src=hpre)),
            ("ffn2_dx +dgelu+drop", lambda: ops.gemm(df, w2, dh, ta=False, tb=False, act="gelu", dact_src=hpre, dropout=drop)),
            ("ffn1_dx plain", lambda: ops.gemm(dh, w1, dx, ta=False, tb=False)),
            ("ffn1_dx +addend", lambda: ops.gemm(dh, w1, dx, ta=False, tb=False, addend=df))]
    for name, fn in rows:
        print(f"{name:32s} {timeit(fn):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
