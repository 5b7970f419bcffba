#!/usr/bin/env python
"""linear1 forward (M x 2048 x 512, persistent-tile kernel) with its epilogue options one by one: bias | + GELU | + saved
pre-activation | + dropout.  Recorded replays.  Dev tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops  # noqa: E402
from gemm_pt_bench import timeit  # noqa: E402

DEV, dt = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)
seed = torch.tensor([3], dtype=torch.int32, device=DEV)
for M in (4864, 3328):
    N, K = 2048, 512
    x = torch.randn(M, K, generator=g).to(dt).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    h, pre = torch.empty(M, N, dtype=dt, device=DEV), torch.empty(M, N, dtype=dt, device=DEV)
    cases = [("bias", {}), ("+gelu", dict(act="gelu")), ("+gelu+preact", dict(act="gelu", preact=pre)),
             ("+gelu+preact+dropout", dict(act="gelu", preact=pre, dropout=(seed, 77, 0.3))),
             ("+relu+preact+dropout", dict(act="relu", preact=pre, dropout=(seed, 77, 0.3))),
             ("+dropout only", dict(dropout=(seed, 77, 0.3)))]
    for name, kw in cases:
        t = timeit(lambda: ops.gemm(x, w, h, bias=b, **kw))
        print(f"M={M} {name:24s} {t:6.1f} us", flush=True)
