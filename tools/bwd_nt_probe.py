#!/usr/bin/env python
"""The layer backward's input-gradient products as the step issues them (NN against the [out, in] weight) and in NT form against a
transposed weight shadow, alone on the chip: d h = d f W2 (GELU' x dropout epilogue), d x = d hpre W1 (+ addend), d x = d qkv Win."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops  # noqa: E402

DEV, dt = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)


def rnd(*s, sc=1.0):
    return (torch.randn(*s, generator=g) * sc).to(dt).to(DEV)


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


seed = torch.tensor([1234], dtype=torch.int32, device=DEV)
for M in (4864, 3328):
    d, ff = 512, 2048
    df, hpre, dh = rnd(M, d), rnd(M, ff), torch.empty(M, ff, dtype=dt, device=DEV)
    w2 = rnd(d, ff, sc=0.03); w2t = w2.t().contiguous()
    w1 = rnd(ff, d, sc=0.03); w1t = w1.t().contiguous()
    win = rnd(3 * d, d, sc=0.03); wint = win.t().contiguous()
    dqkv, ds, dx = rnd(M, 3 * d), rnd(M, d), torch.empty(M, d, dtype=dt, device=DEV)
    drop = (seed, 7, 0.3)
    for tile, name in ((0, "auto"), (100, "forced persistent-tile")):
        def safe(fn):
            try:
                return "%.1f" % timeit(fn)
            except Exception as e:      # a forced kernel that is not eligible
                return "n/a"
        print(f"M={M} [{name}]")
        print("  d h  = d f W2    NN %s us   NT %s us" % (
            safe(lambda: ops.gemm(df, w2, dh, ta=False, tb=False, act="gelu", dact_src=hpre, dropout=drop, tile=tile)),
            safe(lambda: ops.gemm(df, w2t, dh, ta=False, tb=True, act="gelu", dact_src=hpre, dropout=drop, tile=tile))))
        print("  d x  = d hpre W1 NN %s us   NT %s us" % (
            safe(lambda: ops.gemm(dh, w1, dx, ta=False, tb=False, addend=ds, tile=tile)),
            safe(lambda: ops.gemm(dh, w1t, dx, ta=False, tb=True, addend=ds, tile=tile))))
        print("  d x  = d qkv Win NN %s us   NT %s us" % (
            safe(lambda: ops.gemm(dqkv, win, dx, ta=False, tb=False, addend=ds, tile=tile)),
            safe(lambda: ops.gemm(dqkv, wint, dx, ta=False, tb=True, addend=ds, tile=tile))))
