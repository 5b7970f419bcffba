#!/usr/bin/env python
"""Row-panel Linear backward (vct_rp_linear, csrc/vct_rowpanel.hip) against the launches it replaces (vct_gemm NN form + vct_add_ln_bwd),
per product of the layers' dX chain at cfg-B: values (against the unfused kernels and an fp32 torch restatement) and time ALONE
(recorded replays, HIP events around 50 replays).  Dev tool / probe:  python tools/rp_bench.py [--rows 4864]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vct_amd import ops  # noqa: E402

DEV = "cuda"
bf16 = torch.bfloat16


def rnd(*shape, scale=1.0, dtype=bf16):
    return (torch.randn(*shape, device=DEV) * scale).to(dtype)


def timeit(fn, iters=50):
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


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4864)
    ap.add_argument("--p", type=float, default=0.3)
    a = ap.parse_args()
    M, d, ff = a.rows, 512, 2048
    seed = torch.tensor([1234], dtype=torch.int32, device=DEV)
    drop = lambda site: (seed, site, a.p)  # noqa: E731
    out = []

    # ---- epi 0: out_proj-style (K = 512) and in_proj-style (K = 1536), with addend -------------------------------------------------
    for K in (512, 1024, 1536):
        W = rnd(K, d, scale=0.05)
        dy, add = rnd(M, K), rnd(M, d)
        wpk = ops.rp_pack_transposed(W)
        o_ref, o_rp = torch.empty(M, d, dtype=bf16, device=DEV), torch.empty(M, d, dtype=bf16, device=DEV)
        ops.gemm(dy, W, o_ref, ta=False, tb=False, addend=add)
        ops.rp_linear(dy, wpk, d, out=o_rp, addend=add)
        torch.cuda.synchronize()
        exact = dy.float() @ W.float() + add.float()
        t_pack = timeit(lambda: ops.rp_pack_transposed(W, wpk))
        t_g = timeit(lambda: ops.gemm(dy, W, o_ref, ta=False, tb=False, addend=add))
        t_r = timeit(lambda: ops.rp_linear(dy, wpk, d, out=o_rp, addend=add))
        out.append(f"epi0 K={K:5d} N=512  gemm {t_g:6.1f} us  rp {t_r:6.1f} us  pack {t_pack:5.1f} us   err rp {relerr(o_rp, exact):.2e} gemm {relerr(o_ref, exact):.2e}")

    # ---- epi 1: linear2's input gradient (K = 512, N = ff) with GELU' and the feed-forward dropout mask -----------------------------
    W2 = rnd(d, ff, scale=0.05)
    df, hpre = rnd(M, d), rnd(M, ff)
    wpk2 = ops.rp_pack_transposed(W2)
    h_ref, h_rp = torch.empty(M, ff, dtype=bf16, device=DEV), torch.empty(M, ff, dtype=bf16, device=DEV)
    for act in ("gelu", "relu"):
        ops.gemm(df, W2, h_ref, ta=False, tb=False, act=act, dact_src=hpre, dropout=drop(7))
        ops.rp_linear(df, wpk2, ff, out=h_rp, hpre=hpre, act=act, site=7, seed=seed, p_drop=a.p)
        torch.cuda.synchronize()
        t_g = timeit(lambda: ops.gemm(df, W2, h_ref, ta=False, tb=False, act=act, dact_src=hpre, dropout=drop(7)))
        t_r = timeit(lambda: ops.rp_linear(df, wpk2, ff, out=h_rp, hpre=hpre, act=act, site=7, seed=seed, p_drop=a.p))
        nz = ((h_ref != 0) != (h_rp != 0)).float().mean().item()
        out.append(f"epi1 K=  512 N=2048 {act}  gemm {t_g:6.1f} us  rp {t_r:6.1f} us   rp vs gemm {relerr(h_rp, h_ref):.2e}  mask mismatch {nz:.2e}")

    # ---- epi 2: product + LayerNorm backward (K = 2048: linear1 dX -> norm2/norm1; K = 512: cross q dX -> norm1; K = 1536: in_proj dX -> norm3 below)
    for K in (512, 1536, 2048):
        W = rnd(K, d, scale=0.05)
        dy, add = rnd(M, K), rnd(M, d)
        xs, res = rnd(M, d), rnd(M, d)
        gamma = torch.rand(d, device=DEV) + 0.5
        wpk = ops.rp_pack_transposed(W)
        # forward statistics of z = res + drop(xs)
        y = torch.empty(M, d, dtype=bf16, device=DEV)
        mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        ops.add_ln_fwd(xs, res, gamma, torch.zeros(d, device=DEV), y, mean, rstd, dropout=drop(9))
        gy = torch.empty(M, d, dtype=bf16, device=DEV)
        ds_ref, dxo_ref = torch.empty_like(gy), torch.empty_like(gy)
        ds_rp, dxo_rp = torch.empty_like(gy), torch.empty_like(gy)
        nws = ops.ln_ws_rows(M)
        ws_ref = torch.empty(2 * nws * d, device=DEV)
        panels = (M + 31) // 32
        ws_rp = torch.empty(panels * 2 * d, device=DEV)
        dg_ref, db_ref = torch.empty(d, device=DEV), torch.empty(d, device=DEV)

        def unfused():
            ops.gemm(dy, W, gy, ta=False, tb=False, addend=add)
            ops.add_ln_bwd(gy, xs, res, gamma, mean, rstd, ds_ref, dxo_ref, None, None, ws_ref, dropout=drop(9))

        def fused():
            ops.rp_linear(dy, wpk, d, addend=add, seed=seed, p_drop=a.p,
                          norm=dict(gamma=gamma, mean=mean, rstd=rstd, ws=ws_rp, xs=xs, res=res, ds=ds_rp, dxo=dxo_rp, site=9))
        unfused(); fused()
        torch.cuda.synchronize()
        pr = ws_ref.view(nws, 2, d).sum(0)
        pp = ws_rp.view(panels, 2, d).sum(0)
        t_g, t_r = timeit(unfused), timeit(fused)
        t_gemm_only = timeit(lambda: ops.gemm(dy, W, gy, ta=False, tb=False, addend=add))
        out.append(f"epi2 K={K:5d} N=512  gemm+ln {t_g:6.1f} us (gemm {t_gemm_only:5.1f})  rp {t_r:6.1f} us   ds {relerr(ds_rp, ds_ref):.2e} dxo {relerr(dxo_rp, dxo_ref):.2e} "
                   f"dgamma {relerr(pp[0], pr[0]):.2e} dbeta {relerr(pp[1], pr[1]):.2e}")
    print("\n".join(out))


if __name__ == "__main__":
    main()
