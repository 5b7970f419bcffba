#!/usr/bin/env python
"""Per-launch floor of dependent kernels in one stream (recorded replays): a one-thread kernel, a LayerNorm over the decoder rows, and the
same LayerNorm when a second stream replays its own list beside it.  Dev tool."""
import os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vct_amd import ops  # noqa: E402

dev = "cuda"
seed = torch.zeros(1, dtype=torch.int32, device=dev)
M, d = 4864, 512
x = torch.randn(M, d, device=dev).to(torch.bfloat16); res = torch.randn(M, d, device=dev).to(torch.bfloat16)
g = torch.ones(d, device=dev); b = torch.zeros(d, device=dev)
y = torch.empty_like(x); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
x2, res2, y2 = x.clone(), res.clone(), torch.empty_like(x)
mean2, rstd2 = torch.empty(M, device=dev), torch.empty(M, device=dev)
N = 200


def rec(fn, stream):
    with torch.cuda.stream(stream):
        fn(); torch.cuda.synchronize()
        ll = ops.LaunchList()
        with ll.record():
            for _ in range(N):
                fn()
    return ll


def run(pairs, iters=10):
    def work(ll, st):
        with torch.cuda.stream(st):
            for _ in range(iters):
                ll.replay()
    for ll, st in pairs:
        with torch.cuda.stream(st):
            ll.replay()
    torch.cuda.synchronize()
    ths = [threading.Thread(target=work, args=a) for a in pairs]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters / N * 1e6


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
tiny = rec(lambda: ops.advance_seed(seed), s1)
ln = rec(lambda: ops.add_ln_fwd(x, res, g, b, y, mean, rstd), s1)
ln_b = rec(lambda: ops.add_ln_fwd(x2, res2, g, b, y2, mean2, rstd2), s2)
print(f"one-thread kernel, dependent launches in one stream: {run([(tiny, s1)]):5.2f} us per launch")
print(f"LayerNorm 4864 x 512 (15 MB of traffic), one stream:  {run([(ln, s1)]):5.2f} us per launch")
print(f"... two streams, each its own LayerNorm chain:        {run([(ln, s1), (ln_b, s2)]):5.2f} us per launch pair")
