#!/usr/bin/env python
"""Per-kernel times of a producer -> consumer CHAIN (the FFN block of a decoder layer: linear1+GELU+dropout -> linear2 -> add-LayerNorm)
against the same kernels run alone in a loop: how much of the in-step slowdown of a layer GEMM is its cold input.  Dev tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops  # noqa: E402

DEV, dt = "cuda", torch.bfloat16


def main():
    M, d, ff = 4864, 512, 2048
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g)
    x = mk(M, d).to(dt).to(DEV); w1 = (mk(ff, d) / 22).to(dt).to(DEV); b1 = mk(ff).to(DEV)
    w2 = (mk(d, ff) / 45).to(dt).to(DEV); b2 = mk(d).to(DEV)
    wq = (mk(3 * d, d) / 22).to(dt).to(DEV); bq = mk(3 * d).to(DEV)
    gam, bet = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    h = torch.empty(M, ff, dtype=dt, device=DEV); hpre = torch.empty_like(h); f = torch.empty(M, d, dtype=dt, device=DEV)
    y = torch.empty(M, d, dtype=dt, device=DEV); mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
    qkv = torch.empty(M, 3 * d, dtype=dt, device=DEV)
    seed = torch.tensor([3], dtype=torch.int32, device=DEV)
    ops_ = {
        "ffn1": lambda: ops.gemm(x, w1, h, bias=b1, act="gelu", preact=hpre, dropout=(seed, 5, 0.3)),
        "ffn2": lambda: ops.gemm(h, w2, f, bias=b2),
        "ln": lambda: ops.add_ln_fwd(f, x, gam, bet, y, mean, rstd, dropout=(seed, 6, 0.3)),
        "qkv": lambda: ops.gemm(y, wq, qkv, bias=bq),
    }

    def timed(seq, iters=30):
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in seq] for _ in range(iters)]
        for _ in range(3):
            for n in seq:
                ops_[n]()
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for it in range(iters):
            for k, n in enumerate(seq):
                ev[it][k][0].record(); ops_[n](); ev[it][k][1].record()
        t1.record()
        torch.cuda.synchronize()
        per = [sum(ev[it][k][0].elapsed_time(ev[it][k][1]) for it in range(iters)) / iters * 1e3 for k in range(len(seq))]
        return per, t0.elapsed_time(t1) / iters * 1e3

    for seq in (["ffn1"], ["ffn2"], ["ln"], ["qkv"], ["ffn1", "ffn2"], ["ffn1", "ffn2", "ln"], ["ffn1", "ffn2", "ln", "qkv"]):
        per, tot = timed(seq)
        print(" -> ".join(f"{n} {p:5.1f}" for n, p in zip(seq, per)) + f"   | wall per round {tot:6.1f} us", flush=True)
    # the same chain as ONE recorded launch list (no host gaps, no per-kernel events)
    ll = ops.LaunchList()
    with ll.record():
        for n in ("ffn1", "ffn2", "ln", "qkv"):
            ops_[n]()
    for _ in range(3):
        ll.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        ll.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"recorded chain ffn1 -> ffn2 -> ln -> qkv: {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us per round")


if __name__ == "__main__":
    main()
