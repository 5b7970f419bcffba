"""World-size-2 test of the data-parallel gradient exchange on CPU ranks (gloo): the same
GradExchange object bench.py / trainer.py use with RCCL.  Checks the constructor broadcast,
per-bucket async all-reduce in backward order, the 1/world scaling, and that train_epoch's loss
reduction is the mean of per-rank means."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_host_logic_cpu import SHIPPED_LIKE


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from helpers import build_model
    from vct_amd.trainer import GradExchange
    from vct_amd.utils import configure_hardware
    device, r, w = configure_hardware("gloo")
    assert (r, w) == (rank, world) and device.type == "cpu"
    torch.manual_seed(100 + rank)            # DIFFERENT init per rank: the broadcast must fix it
    mc = dict(SHIPPED_LIKE, embed_dim=64, modal_shape=[48])
    mc["video_encoder"] = dict(mc["video_encoder"], layer=1, nhead=4, feedforward=128)
    mc["caption_decoder"] = dict(mc["caption_decoder"], layer=2, nhead=4, feedforward=128)
    m = build_model(mc, 131, "cpu", torch.float32)
    before = m.flat_params.clone()
    ex = GradExchange(m)
    gathered = [torch.empty_like(m.flat_params) for _ in range(world)]
    dist.all_gather(gathered, m.flat_params)
    same_after = all(torch.equal(gathered[0], g) for g in gathered)
    changed = not torch.equal(before, m.flat_params)
    # synthetic per-rank gradients, exchanged bucket by bucket in backward order
    g = torch.Generator().manual_seed(7 + rank)
    local = torch.randn(m.flat_grads.numel(), generator=g)
    m.flat_grads.copy_(local)
    all_local = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(all_local, local)
    for i in range(len(m.grad_buckets())):
        ex.bucket_ready(i)
    ex.finish()
    expect = sum(all_local) / world
    ok_avg = torch.allclose(m.flat_grads, expect, rtol=1e-6, atol=1e-7)
    q.put((rank, same_after, changed, ok_avg))
    dist.destroy_process_group()


def test_grad_exchange_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_after, changed, ok_avg in res:
        assert same_after, "parameters differ across ranks after the constructor broadcast"
        assert ok_avg, "bucketed all-reduce did not produce the mean gradient"
    assert res[1][2], "rank 1 parameters were not overwritten by rank 0's"


class _SgdStandIn:
    """Stands in for FusedAdam on CPU ranks (the Adam kernel is HIP-only): same range protocol, p -= 0.1 g."""

    def __init__(self, model):
        self.model, self.ranges, self.finished = model, [], 0

    def step_range(self, a, b):
        self.ranges.append((a, b))
        self.model.flat_params[a:b] -= 0.1 * self.model.flat_grads[a:b]

    def finish_ranges(self):
        self.finished += 1


def _sharded_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from helpers import build_model
    from vct_amd.comm import C10dColl
    from vct_amd.trainer import ShardedExchange
    from vct_amd.utils import configure_hardware
    configure_hardware("gloo")
    torch.manual_seed(100 + rank)
    mc = dict(SHIPPED_LIKE, embed_dim=64, modal_shape=[48])
    mc["video_encoder"] = dict(mc["video_encoder"], layer=1, nhead=4, feedforward=128)
    mc["caption_decoder"] = dict(mc["caption_decoder"], layer=2, nhead=4, feedforward=128)
    m = build_model(mc, 131, "cpu", torch.float32)
    opt = _SgdStandIn(m)
    ex = ShardedExchange(m, opt, C10dColl())            # constructor broadcast: rank 0's parameters everywhere
    start = m.flat_params.clone()
    g = torch.Generator().manual_seed(7 + rank)
    local = torch.randn(m.flat_grads.numel(), generator=g)
    m.flat_grads.copy_(local)
    all_local = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(all_local, local)
    for i in range(len(m.grad_buckets())):
        ex.bucket_ready(i)
    ex.finish()
    expect = start - 0.1 * (sum(all_local) * (1.0 / world))
    gathered = [torch.empty_like(m.flat_params) for _ in range(world)]
    dist.all_gather(gathered, m.flat_params)
    same = all(torch.equal(gathered[0], x) for x in gathered)
    exact = torch.equal(m.flat_params, expect)
    # this rank stepped exactly its 1/world slice of every bucket
    mine_ok = all(b - a == (hi - lo) // world and a == lo + rank * ((hi - lo) // world)
                  for (a, b), (lo, hi) in zip(opt.ranges, [bk for bk in m.grad_buckets() if bk[1] > bk[0]]))
    q.put((rank, same, exact, mine_ok, opt.finished, len(opt.ranges)))
    dist.destroy_process_group()


def test_sharded_exchange_world2_gloo():
    """Reduce-scatter -> optimizer on the owned shard -> all-gather over two CPU ranks: every rank ends with the parameters
    of a full-range step on the mean gradient, bitwise, having stepped only its own half of every bucket."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, exact, mine_ok, finished, nr in res:
        assert same and exact and mine_ok and finished == 1 and nr >= 5, (rank, same, exact, mine_ok, finished, nr)
