#!/usr/bin/env python
"""Probe of CU-masked streams on MI355X: (1) how the vocabulary weight-gradient GEMM and a chain of small GEMMs scale with
the number of CUs they may use, for several mask patterns; (2) bulk kernel on X CUs beside the chain on the complement."""
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
    M, V, d, ff = 4864, 30528, 512, 2048
    dl, y, dW, db = mk(M, V), mk(M, d), torch.empty(V, d, device=DEV), torch.empty(V, device=DEV)
    x, w1, w2 = mk(M, d), mk(ff, d), mk(d, ff)
    h, f = torch.empty(M, ff, device=DEV, dtype=dt), torch.empty(M, d, device=DEV, dtype=dt)
    ws = ops.GemmScratch(DEV)

    def bulk():
        ops.gemm(dl, y, dW, ta=True, tb=False, bias_grad=db, m_valid=30522, workspace=ws)

    def chain(n=12):
        for _ in range(n):
            ops.gemm(x, w1, h)
            ops.gemm(h, w2, f)

    def t_of(fn, stream, iters=5):
        with torch.cuda.stream(stream):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e6

    patterns = {
        "all": None,
        "first128": range(128), "every2nd(128)": range(0, 256, 2), "first64": range(64), "every4th(64)": range(0, 256, 4),
        "first96": range(96), "3of8(96)": [c for c in range(256) if c % 8 < 3], "first32": range(32), "every8th(32)": range(0, 256, 8),
        "first192": range(192), "6of8(192)": [c for c in range(256) if c % 8 < 6],
    }
    print(f"{'mask':16s} {'bulk us':>9s} {'chain us':>9s}")
    streams = {}
    for name, bits in patterns.items():
        s = ops.masked_stream(bits)
        streams[name] = s
        print(f"{name:16s} {t_of(bulk, s):9.1f} {t_of(chain, s):9.1f}", flush=True)
    # concurrency: bulk on A beside chain on B
    print("bulk stream / chain stream -> wall us (both started together)")
    combos = [("all", "all")]
    for k in (2, 3, 4):
        a = [c for c in range(256) if c % 8 < k]; b = [c for c in range(256) if c % 8 >= k]
        streams[f"lo{k}"] = ops.masked_stream(a); streams[f"hi{k}"] = ops.masked_stream(b)
        combos += [(f"lo{k}", f"hi{k}"), (f"lo{k}", "all")]
    a = list(range(64)); b = list(range(64, 256))
    streams["blk64"] = ops.masked_stream(a); streams["blk192"] = ops.masked_stream(b)
    combos += [("blk64", "blk192")]
    for sa, sb in combos:
        A, B = streams[sa], streams[sb]
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(A):
                bulk()
            with torch.cuda.stream(B):
                chain()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e6
        print(f"{sa:8s} / {sb:8s} -> {wall:8.1f}", flush=True)


if __name__ == "__main__":
    main()
