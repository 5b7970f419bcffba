#!/usr/bin/env python
"""The batch-1024 linear1 product (19456 x 2048 x 512, persistent-tile kernel) in a loop, for PMC passes.  Dev tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops
M, N, K = 19456, 2048, 512
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda(); w = torch.randn(N, K, generator=g).to(torch.bfloat16).cuda()
b = torch.randn(N, generator=g).cuda(); o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(12):
    ops.gemm(x, w, o, bias=b)
torch.cuda.synchronize()
