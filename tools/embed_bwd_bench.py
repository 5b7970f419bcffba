#!/usr/bin/env python
"""Token-embedding gradient (vct_embed_bwd) at the cfg-B shape, alone: recorded replays, full zero-fill vs dirty-row zeroing.  Dev tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops  # noqa: E402

DEV = "cuda"
B, S, d, V = 256, 19, 512, 30522
g = torch.Generator().manual_seed(0)
ids = torch.randint(1000, 30000, (B, S + 1), generator=g); ids[:, 0] = 101
ids = ids.to(DEV)
dx = torch.randn(B * S, d, generator=g).to(torch.bfloat16).to(DEV)
dt = torch.zeros(V, d, device=DEV)
seed = torch.tensor([3], dtype=torch.int32, device=DEV)
for excl in (False, True):
    for _ in range(3):
        ops.embed_bwd(ids, S, 0, dx, dt, dropout=(seed, 999, 0.3), exclusive=excl)
    ll = ops.LaunchList()
    with ll.record():
        ops.embed_bwd(ids, S, 0, dx, dt, dropout=(seed, 999, 0.3), exclusive=excl)
    ll.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        ll.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"exclusive={excl}: {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us per call ({len(ll)} launches)")
