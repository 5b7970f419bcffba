#!/usr/bin/env python
"""Would two half-batch forward chains side by side beat one full-batch chain?  Two models (own parameter sets, own streams) run the
layer-stack + generator forward of 128 samples each inside ONE captured graph (fork / join on the device, no host skew), against
one model on 256 samples.  Dropout off (the masks of a split batch would need row offsets in every kernel).  Dev tool."""
import os, sys, time
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
from bench import MODEL_CFG, synthetic  # noqa: E402
from vct_amd.model import MMT4Caption  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


def mk(seed):
    torch.manual_seed(seed)
    m = MMT4Caption(MODEL_CFG, device=dev, compute_dtype=torch.bfloat16)
    m.mode("caption"); m.eval()
    m._ps.refresh_shadow()
    return m


import threading
from vct_amd import ops  # noqa: E402


def record(fn, stream):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ll = ops.LaunchList()
        with ll.record():
            fn()
    return ll


def run_lists(lists_streams, iters=60):
    """every (list, stream) on its own host thread, `iters` replays each; wall time per replay round"""
    def work(ll, st):
        with torch.cuda.stream(st):
            for _ in range(iters):
                ll.replay()
    for ll, st in lists_streams:        # warm
        with torch.cuda.stream(st):
            ll.replay()
    torch.cuda.synchronize()
    ths = [threading.Thread(target=work, args=a) for a in lists_streams]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


mfull, ma, mb = mk(1), mk(2), mk(3)
f256 = synthetic(256, 0, dev); fa = synthetic(128, 1, dev); fb = synthetic(128, 2, dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def full():
    with torch.no_grad():
        mfull._forward_loss(*f256, False)


def one_half():
    with torch.no_grad():
        ma._forward_loss(*fa, False)


def other_half():
    with torch.no_grad():
        mb._forward_loss(*fb, False)


lf = record(full, s1)
la = record(one_half, s1)
lb = record(other_half, s2)
print(f"forward + loss, 256 samples, one chain:        {run_lists([(lf, s1)]):7.1f} us per round ({len(lf)} launches)")
print(f"forward + loss, 128 samples, one chain:        {run_lists([(la, s1)]):7.1f} us per round ({len(la)} launches)")
print(f"forward + loss, 2 x 128 samples, two chains:   {run_lists([(la, s1), (lb, s2)]):7.1f} us per round")
