#!/usr/bin/env python
"""vct_adam_step_2d against vct_adam_step + transpose on the vocabulary matrix, repeatedly: any run-to-run difference?  Dev tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops
rows, cols = 30522, 512
g = torch.Generator().manual_seed(1)
p0 = torch.randn(rows * cols, generator=g).cuda(); gr = (torch.randn(rows * cols, generator=g) * 1e-3).cuda()
m0 = (torch.randn(rows * cols, generator=g) * 1e-3).cuda(); v0 = (torch.rand(rows * cols, generator=g) * 1e-6).cuda()
step = torch.zeros(1, dtype=torch.int32, device="cuda")
hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 0, 0], dtype=torch.float32, device="cuda")
ld = (rows + 31) // 32 * 32
ref = None
bad = 0
for trial in range(int(os.environ.get("TRIALS", "30"))):
    p, m, v = p0.clone(), m0.clone(), v0.clone()
    s = torch.zeros(rows * cols, dtype=torch.bfloat16, device="cuda"); st = torch.zeros(cols, ld, dtype=torch.bfloat16, device="cuda")
    # something else running beside it on a second stream, like the encoder backward in the step
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        x = torch.randn(4096, 4096, device="cuda"); y = x @ x
    ops.adam_step_2d(p.view(rows, cols), gr.view(rows, cols), m.view(rows, cols), v.view(rows, cols), s.view(rows, cols), st, 1e-3, 0.9, 0.999,
                     1e-8, 0.0, step, hyper=hyper)
    torch.cuda.synchronize()
    out = (p, m, v, s, st)
    if ref is None:
        ref = [t.clone() for t in out]
        pa, ma, va = p0.clone(), m0.clone(), v0.clone(); sa = torch.zeros_like(s)
        ops.adam_step(pa, gr, ma, va, sa, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, bump=False, hyper=hyper)
        print("vs flat adam:", [bool(torch.equal(a, b)) for a, b in zip((p, m, v, s), (pa, ma, va, sa))],
              "transpose ok:", bool(torch.equal(st[:, :rows], sa.view(rows, cols).t())))
    else:
        eq = [bool(torch.equal(a, b)) for a, b in zip(out, ref)]
        if not all(eq):
            bad += 1
            d = (out[4] != ref[4]).nonzero()
            print("trial", trial, eq, "shadow_t diffs", d.shape[0], d[:4].tolist())
print("trials with a difference:", bad)
