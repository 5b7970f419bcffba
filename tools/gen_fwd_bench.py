#!/usr/bin/env python
"""The vocabulary projection exactly as the step issues it (bias, n_valid = 30522 inside a 30528-wide buffer, bf16 out)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops  # noqa: E402

DEV, dt = "cuda", torch.bfloat16
M, V, Vp, d = 4864, 30522, 30528, 512
g = torch.Generator().manual_seed(0)
y = torch.randn(M, d, generator=g).to(dt).to(DEV); w = (torch.randn(V, d, generator=g) / 22).to(dt).to(DEV)
bias = torch.randn(V, generator=g).to(DEV)
logits = torch.empty(M, Vp, dtype=dt, device=DEV)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("plain            %.1f us" % timeit(lambda: ops.gemm(y, w, logits)))
print("n_valid          %.1f us" % timeit(lambda: ops.gemm(y, w, logits, n_valid=V)))
print("bias             %.1f us" % timeit(lambda: ops.gemm(y, w, logits, bias=bias)))
print("bias + n_valid   %.1f us" % timeit(lambda: ops.gemm(y, w, logits, bias=bias, n_valid=V)))
