#!/usr/bin/env python
"""Micro-benchmark of the fused attention block (vct_attn_block_fwd) against the three kernels it replaces, at the cfg-B
shapes, with the kernel's experiment flags (1 = no attention phase, 2 = no projection phase, 4 = no chunk stagger)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    B, H, hd = 256, 8, 64
    d, dt = H * hd, torch.bfloat16
    for name, Lq, Lk, causal in (("dec self ", 19, 19, True), ("dec cross", 19, 13, False), ("enc self ", 13, 13, False)):
        g = torch.Generator().manual_seed(0)
        qkv = torch.randn(B * Lq, 3 * d, generator=g).to(dt).to(DEV); kvs = torch.randn(B * Lk, 3 * d, generator=g).to(dt).to(DEV)
        q, k, v = qkv[:, :d], kvs[:, d:2 * d], kvs[:, 2 * d:]
        wo = (torch.randn(d, d, generator=g) / math.sqrt(d)).to(dt).to(DEV); bo = torch.randn(d, generator=g).to(DEV)
        res = torch.randn(B * Lq, d, generator=g).to(dt).to(DEV); gamma = torch.ones(d, device=DEV); beta = torch.zeros(d, device=DEV)
        seed = torch.tensor([1], dtype=torch.int32, device=DEV)
        M = B * Lq
        o = torch.empty(M, d, dtype=dt, device=DEV); a = torch.empty_like(o); y = torch.empty_like(o)
        mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
        drop = (seed, 3, 0.3)

        def unfused():
            ops.attn_fwd(q, k, v, o, B, H, Lq, Lk, causal=causal, dropout=drop)
            ops.gemm(o, wo, a, bias=bo)
            ops.add_ln_fwd(a, res, gamma, beta, y, mean, rstd, dropout=(seed, 4, 0.3))
        t_un = timeit(unfused)
        t_parts = [timeit(lambda: ops.attn_fwd(q, k, v, o, B, H, Lq, Lk, causal=causal, dropout=drop)),
                   timeit(lambda: ops.gemm(o, wo, a, bias=bo)),
                   timeit(lambda: ops.add_ln_fwd(a, res, gamma, beta, y, mean, rstd, dropout=(seed, 4, 0.3)))]
        row = [f"{name} unfused {t_un:6.1f} us (attn {t_parts[0]:.1f} + proj {t_parts[1]:.1f} + ln {t_parts[2]:.1f})"]
        for flags in (0, 4, 1, 2, 3):
            t = timeit(lambda: ops.attn_block_fwd(q, k, v, o, B, H, Lq, Lk, wo, bo, res, gamma, beta, a, y, mean, rstd, causal=causal,
                                                  dropout=drop, site_res=4, flags=flags))
            row.append(f"fused[{flags}] {t:6.1f}")
        print("  ".join(row), flush=True)


if __name__ == "__main__":
    main()
