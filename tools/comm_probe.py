import sys, time, torch
sys.path.insert(0, "/root/repo")
from vct_amd.comm import RcclColl
from vct_amd import ops
dev = torch.device("cuda", 0)
c = RcclColl(device=dev)
x = torch.randn(16_000_000, device=dev)
xb = x.to(torch.bfloat16)
for name, fn in (("allreduce f32", lambda: c.allreduce_avg(x)), ("rs f32", lambda: c.reduce_scatter_avg(x, x.numel())),
                 ("ag f32", lambda: c.all_gather(x, x.numel())), ("rs bf16", lambda: c.reduce_scatter_avg(xb, xb.numel())),
                 ("edge only", lambda: ops.stream_wait(c.stream, None))):
    fn(); c.wait(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    t1 = time.perf_counter()
    c.wait(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:14s} host {1e6*(t1-t0)/20:8.1f} us/call   total {1e6*(t2-t0)/20:8.1f} us/call", flush=True)
