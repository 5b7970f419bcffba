import sys, time, os, torch
sys.path.insert(0, "/root/repo")
from bench import MODEL_CFG, TRAIN_CFG, synthetic
from vct_amd.model import MMT4Caption
from vct_amd.trainer import CaptionTrainer, build_optimizer, GradExchange
import torch.distributed as dist
dev = torch.device("cuda", 0)
force = "exchange" in sys.argv[1:]
executor = "eager" if "eager" in sys.argv[1:] else ("graph" if "graph" in sys.argv[1:] else "list")
if force:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(666)
m = MMT4Caption(MODEL_CFG, device=dev, compute_dtype=torch.bfloat16); m.mode("caption"); m.train()
opt, _ = build_optimizer(TRAIN_CFG, m)
ex = GradExchange(m, force=True) if force else None
tr = CaptionTrainer(m, opt, ex, use_graph=executor == "graph", launch_list=executor == "list")
b = synthetic(256, 0, dev)
for _ in range(8): tr.step(*b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): tr.step(*b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"executor={executor} exchange={force}: CPU enqueue {1e3*(t1-t0)/30:.3f} ms/step, wall {1e3*(t2-t0)/30:.3f} ms/step")
