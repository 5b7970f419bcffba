#!/usr/bin/env python
"""Idle time between dependent launches of one stream behind a LARGE kernel: records [vocabulary GEMM, tiny cast, vocabulary GEMM, tiny
cast] as a launch list and replays it; run under `rocprofv3 --kernel-trace --output-format csv` and feed the trace to this script's
--csv option to print start-to-end gaps.  Dev tool."""
import csv
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run():
    from vct_amd import ops
    dev = "cuda"
    M, V, d = 4864, 30522, 512
    Vp = (V + 31) // 32 * 32
    W = (torch.randn(V, d, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.randn(M, d, device=dev).to(torch.bfloat16)
    bias = torch.zeros(V, device=dev)
    out = torch.zeros(M, Vp, dtype=torch.bfloat16, device=dev)
    small, small2 = torch.zeros(4096, device=dev), torch.zeros(4096, dtype=torch.bfloat16, device=dev)
    mode = os.environ.get("GAP_MODE", "gemm")

    def body():
        for _ in range(2):
            if mode == "gemm":
                ops.gemm(y, W, out, bias=bias, n_valid=V)
            else:
                ops.warm(out)
            ops.cast(small, small2)
            ops.cast(small, small2)
    ll = ops.LaunchList()
    body(); torch.cuda.synchronize()
    with ll.record():
        body()
    for _ in range(20):
        ll.replay()
    torch.cuda.synchronize()


def gaps(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-24:]
    prev = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{r['Kernel_Name'][:50]:50s} dur {(e - s) / 1e3:8.1f} us  gap {((s - prev) / 1e3 if prev else 0):7.1f} us")
        prev = e


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--csv":
        gaps(sys.argv[2])
    else:
        run()
