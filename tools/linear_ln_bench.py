#!/usr/bin/env python
"""Row-complete linear + dropout + residual + LayerNorm (vct_linear_ln_fwd) against the two launches it replaces, cfg-B shapes.  Dev tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops  # noqa: E402

DEV, dt = "cuda", torch.bfloat16


def timeit(fn, iters=50):
    for _ in range(5):
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
    mk = lambda *s: torch.randn(*s, generator=g)
    d = 512
    seed = torch.tensor([3], dtype=torch.int32, device=DEV)
    for M in (4864, 3328):
        for K in (512, 2048):
            x = mk(M, K).to(dt).to(DEV); w = (mk(d, K) / K ** 0.5).to(dt).to(DEV); b = mk(d).to(DEV)
            res = mk(M, d).to(dt).to(DEV); gam = torch.ones(d, device=DEV); bet = torch.zeros(d, device=DEV)
            a = torch.empty(M, d, dtype=dt, device=DEV); y = torch.empty_like(a); z = torch.empty_like(a)
            mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV); m2 = torch.empty(M, device=DEV); r2 = torch.empty(M, device=DEV)
            for p in (0.0, 0.3):
                drop = (seed, 5, p) if p > 0 else None

                def unfused():
                    ops.gemm(x, w, a, bias=b)
                    ops.add_ln_fwd(a, res, gam, bet, y, mean, rstd, dropout=drop)
                t_g = timeit(lambda: ops.gemm(x, w, a, bias=b))
                t_u = timeit(unfused)
                row = f"M={M} K={K} p={p}: gemm {t_g:5.1f}  gemm+ln {t_u:5.1f}"
                for rows in (32, 16):
                    t_f = timeit(lambda: ops.linear_ln_fwd(x, w, b, res, gam, bet, a, y, mean, rstd, dropout=drop, rows_per_wg=rows))
                    row += f"  fused{rows} {t_f:5.1f}"
                t_f2 = timeit(lambda: ops.linear_ln_fwd(x, w, b, res, gam, bet, a, y, mean, rstd, dropout=drop, ln2=(gam, bet, z, m2, r2)))
                row += f"  fused+ln2 {t_f2:5.1f}"
                print(row, flush=True)


if __name__ == "__main__":
    main()
