#!/usr/bin/env python
"""What the optimizer epilogue costs a weight-gradient GEMM when the GEMM runs ALONE (vct_gemm_adam, include/vct_hip.h): the
vocabulary product (30522 x 512 x 4864) and a decoder layer's grouped launch, plain / plain + the separate optimizer pass / fused.
Recorded replays (launch list), HIP events around 20 replays.  Dev tool:  python tools/adam_epi_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vct_amd import _lib as L, ops  # noqa: E402

DEV = "cuda"


def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).to(dtype)


def timeit(fn, iters=20):
    ll = ops.LaunchList()
    fn()
    torch.cuda.synchronize()
    with ll.record():
        fn()
    for _ in range(3):
        ll.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ll.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def adam_desc(p, m, v, s, K, hyper, step):
    ad = L.GemmAdam()
    ad.param, ad.exp_avg, ad.exp_avg_sq, ad.shadow, ad.ld_shadow = p.data_ptr(), m.data_ptr(), v.data_ptr(), s.data_ptr(), K
    ad.hyper, ad.step, ad.store_grad = hyper.data_ptr(), step.data_ptr(), 0
    return ad


def main():
    hyper = torch.tensor([1e-4, 0.9, 0.999, 1e-8, 0.0, 0, 0, 0], dtype=torch.float32, device=DEV)
    step = torch.tensor([3], dtype=torch.int32, device=DEV)
    scratch = ops.GemmScratch(DEV)
    # vocabulary weight gradient
    V, d, rows = 30522, 512, 4864
    Vp = (V + 31) // 32 * 32
    dl, y = rnd(rows, Vp, scale=0.01), rnd(rows, d)
    dw, db = torch.empty(V, d, device=DEV), torch.empty(V, device=DEV)
    p, m, v = (torch.randn(V * d, device=DEV) * 0.02 for _ in range(3))
    v = v.abs() * 1e-4
    s = torch.empty(V * d, dtype=torch.bfloat16, device=DEV)
    st = torch.empty(d, Vp, dtype=torch.bfloat16, device=DEV)
    ad = adam_desc(p, m, v, s, d, hyper, step)
    t_plain = timeit(lambda: ops.gemm(dl, y, dw, ta=True, tb=False, bias_grad=db, m_valid=V, workspace=scratch))
    t_adam2d = timeit(lambda: ops.adam_step_2d(p.view(V, d), dw, m.view(V, d), v.view(V, d), s.view(V, d), st, 0, 0, 0, 0, 0, step, hyper=hyper))
    t_adam = timeit(lambda: ops.adam_step(p, dw.view(-1), m, v, s, 0, 0, 0, 0, 0, step, bump=False, hyper=hyper))
    t_fused = timeit(lambda: ops.gemm(dl, y, dw, ta=True, tb=False, bias_grad=db, m_valid=V, workspace=scratch, adam=ad))
    print(f"vocabulary dW {V}x{d}x{rows}: plain {t_plain:.1f} us | adam2d {t_adam2d:.1f} | flat adam {t_adam:.1f} | fused {t_fused:.1f} "
          f"(plain + flat adam = {t_plain + t_adam:.1f})")
    # a decoder layer's group (cfg-B)
    M = 4864
    shapes = [(1536, 512, M), (512, 512, M), (512, 512, M), (1024, 512, 3328), (512, 512, M), (2048, 512, M), (512, 2048, M)]
    items, fused_items, flat = [], [], []
    for (mo, ki, r) in shapes:
        dy, x = rnd(r, mo, scale=0.01), rnd(r, ki)
        g, b = torch.empty(mo, ki, device=DEV), torch.empty(mo, device=DEV)
        pp, mm, vv = (torch.randn(mo * ki, device=DEV) * 0.02 for _ in range(3))
        vv = vv.abs() * 1e-4
        ss = torch.empty(mo * ki, dtype=torch.bfloat16, device=DEV)
        items.append((dy, x, g, b))
        fused_items.append((dy, x, g, b, adam_desc(pp, mm, vv, ss, ki, hyper, step)))
        flat.append((pp, g, mm, vv, ss))
    t_g = timeit(lambda: ops.gemm_grouped(items, scratch))
    t_gf = timeit(lambda: ops.gemm_grouped(fused_items, scratch))
    n = sum(a * b for a, b, _ in shapes)
    pf, gf, mf, vf = (torch.randn(n, device=DEV) * 0.02 for _ in range(4))
    sf = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    t_a = timeit(lambda: ops.adam_step(pf, gf, mf, vf.abs(), sf, 0, 0, 0, 0, 0, step, bump=False, hyper=hyper))
    print(f"decoder-layer group ({n / 1e6:.1f} M parameters): plain {t_g:.1f} us | flat adam {t_a:.1f} | fused {t_gf:.1f} (plain + adam = {t_g + t_a:.1f})")


if __name__ == "__main__":
    main()
