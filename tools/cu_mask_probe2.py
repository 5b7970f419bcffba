#!/usr/bin/env python
"""CU-masked streams, part 2: a LATENCY-bound chain (attention, out_proj-sized GEMMs, LayerNorm: the decoder backward's
kind of kernels) beside the vocabulary weight-gradient GEMM / Adam-like streaming kernel, on complementary CU sets."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops  # noqa: E402

DEV = "cuda"
dt = torch.bfloat16


def mk(*s):
    return torch.randn(*s, device=DEV).to(dt)


def main():
    M, V, d, B, H, L = 4864, 30528, 512, 256, 8, 19
    dl, y, dW, db = mk(M, V), mk(M, d), torch.empty(V, d, device=DEV), torch.empty(V, device=DEV)
    x, wo = mk(M, d), mk(d, d)
    qkv = mk(M, 3 * d); o = torch.empty(M, d, device=DEV, dtype=dt); a = torch.empty_like(o); yy = torch.empty_like(o)
    g = torch.ones(d, device=DEV); bt = torch.zeros(d, device=DEV); mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
    ws = ops.GemmScratch(DEV)
    big = torch.empty(46_000_000, device=DEV); big2 = torch.empty_like(big)

    def bulk():
        ops.gemm(dl, y, dW, ta=True, tb=False, bias_grad=db, m_valid=30522, workspace=ws)

    def stream_copy():          # Adam-like HBM streaming
        ops.cast(big, big2)

    def chain(n=10):
        for _ in range(n):
            ops.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, B, H, L, L, causal=True)
            ops.gemm(o, wo, a)
            ops.add_ln_fwd(a, x, g, bt, yy, mean, rstd)
            ops.gemm(yy, wo, a)

    def t_of(fn, stream, iters=5):
        with torch.cuda.stream(stream):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e6

    S = {"all": ops.masked_stream(None)}
    for n in (32, 64, 96, 128):
        S[f"first{n}"] = ops.masked_stream(range(n))
        S[f"last{256 - n}"] = ops.masked_stream(range(n, 256))
    print(f"{'mask':10s} {'bulk':>8s} {'copy':>8s} {'chain':>8s}")
    for k, s in S.items():
        print(f"{k:10s} {t_of(bulk, s):8.1f} {t_of(stream_copy, s):8.1f} {t_of(chain, s):8.1f}", flush=True)
    print("A(bulk) / B(chain) -> wall")
    for fa, name in ((bulk, "gemm"), (stream_copy, "copy")):
        for sa, sb in [("all", "all")] + [(f"first{n}", f"last{256 - n}") for n in (32, 64, 96, 128)] + [("first64", "all"), ("first96", "all")]:
            A, B_ = S[sa], S[sb]
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with torch.cuda.stream(A):
                    fa()
                with torch.cuda.stream(B_):
                    chain()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) * 1e6)
            print(f"{name} {sa:9s} / {sb:9s} -> {best:8.1f}", flush=True)


if __name__ == "__main__":
    main()
