#!/usr/bin/env python
"""Times vct_transpose on the generator weight shape (dev tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops
w = torch.randn(30522, 512, device="cuda").to(torch.bfloat16)
wt = torch.empty(512, 30528, dtype=torch.bfloat16, device="cuda")
for _ in range(3): ops.transpose(w, wt)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): ops.transpose(w, wt)
e1.record(); torch.cuda.synchronize()
print(f"transpose 30522x512 bf16: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", torch.equal(wt[:, :30522], w.t()))
