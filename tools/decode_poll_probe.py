#!/usr/bin/env python
"""What the end-of-sequence poll costs after the device is already idle (dev tool)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda", 0)
flag = torch.full((1,), 30, dtype=torch.long, device=dev)
host = torch.empty(1, dtype=torch.long).pin_memory()
side = torch.cuda.Stream(device=dev)
ev = torch.cuda.Event()
x = torch.randn(1024, 1024, device=dev)
for name in ("item", "side_copy", "main_copy", "event_only"):
    ts = []
    for _ in range(20):
        y = x @ x
        ev.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name == "item":
            v = int(flag)
        elif name == "side_copy":
            with torch.cuda.stream(side):
                side.wait_event(ev); host.copy_(flag, non_blocking=True)
            side.synchronize(); v = int(host[0])
        elif name == "main_copy":
            host.copy_(flag, non_blocking=True); torch.cuda.current_stream().synchronize(); v = int(host[0])
        else:
            ev.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name:12s} median {ts[10]*1e6:7.1f} us  min {ts[0]*1e6:7.1f} max {ts[-1]*1e6:7.1f}")
