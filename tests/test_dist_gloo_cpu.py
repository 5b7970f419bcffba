"""World-size-2 / 4 / 8 tests of the data-parallel gradient exchange on CPU ranks (gloo): the same
GradExchange / ShardedExchange objects bench.py / trainer.py use with RCCL.  Checks the constructor broadcast,
per-bucket async all-reduce in backward order, the 1/world scaling, the 1/W shard arithmetic of every bucket
(reduce-scatter -> optimizer on the owned slice -> all-gather) at the world sizes the driver's scaling run uses,
and sharded checkpoints."""
import os
import socket

import pytest
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


@pytest.mark.parametrize("world", [2, 4, 8])
def test_grad_exchange_gloo(world):
    port = _free_port()
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
    # two ranks: one addition per element, bitwise; more ranks: gloo's reduction order is not the host's left-to-right sum
    exact = torch.equal(m.flat_params, expect) if world == 2 else torch.allclose(m.flat_params, expect, rtol=1e-6, atol=1e-7)
    # this rank stepped exactly its 1/world slice of every bucket
    mine_ok = all(b - a == (hi - lo) // world and a == lo + rank * ((hi - lo) // world)
                  for (a, b), (lo, hi) in zip(opt.ranges, [bk for bk in m.grad_buckets() if bk[1] > bk[0]]))
    q.put((rank, same, exact, mine_ok, opt.finished, len(opt.ranges)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_exchange_gloo(world):
    """Reduce-scatter -> optimizer on the owned shard -> all-gather over 2 / 4 / 8 CPU ranks: every rank ends with the parameters
    of a full-range step on the mean gradient, having stepped only its own 1/W slice of every bucket (the slice offsets the
    8-GPU run uses: every bucket cut is a multiple of 64 elements, so it divides over 8 shards)."""
    port = _free_port()
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


def _cpu_adam_cls():
    """FusedAdam with the two HIP launches (vct_adam_step, bump) replaced by the same arithmetic in torch, so that the REAL
    state_dict / load_state_dict / pre_state_dict protocol and checkpoint.save/load_training_state run on CPU ranks."""
    from vct_amd.trainer import FusedAdam

    class CpuAdam(FusedAdam):
        @torch.no_grad()
        def step_range(self, a, b):
            b = min(b, self.end)
            if b <= a:
                return
            lr, b1, b2, eps, _wd = self._hyper_now()
            t = int(self.step_dev.item()) + 1
            ps = self.model._ps
            g, m, v = ps.gflat[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b]
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = v.sqrt() / (1 - b2 ** t) ** 0.5 + eps
            ps.flat[a:b].addcdiv_(m, denom, value=-lr / (1 - b1 ** t))

        @torch.no_grad()
        def finish_ranges(self):
            self.step_dev += 1
    return CpuAdam


def _resume_worker(rank, world, port, q, path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from helpers import build_model
    from vct_amd import checkpoint
    from vct_amd.comm import C10dColl
    from vct_amd.trainer import ShardedExchange
    from vct_amd.utils import configure_hardware
    configure_hardware("gloo")
    mc = dict(SHIPPED_LIKE, embed_dim=64, modal_shape=[48])
    mc["video_encoder"] = dict(mc["video_encoder"], layer=1, nhead=4, feedforward=128)
    mc["caption_decoder"] = dict(mc["caption_decoder"], layer=2, nhead=4, feedforward=128)
    CpuAdam = _cpu_adam_cls()

    def make(seed):
        torch.manual_seed(seed)
        m = build_model(mc, 131, "cpu", torch.float32)
        opt = CpuAdam(m, lr=1e-2)
        return m, opt, ShardedExchange(m, opt, C10dColl())

    def step(m, ex, k):
        g = torch.Generator().manual_seed(1000 * k + rank)
        m.flat_grads.copy_(torch.randn(m.flat_grads.numel(), generator=g))
        for i in range(len(m.grad_buckets())):
            ex.bucket_ready(i)
        ex.finish()

    m, opt, ex = make(100)
    for k in range(2):
        step(m, ex, k)
    # before the gather this rank's moments are only current on its own shards: remember a slice it does NOT own
    a, b = ex.buckets[0]
    n = (b - a) // world
    other = (rank + 1) % world
    stale_before = opt.exp_avg[a + other * n: a + (other + 1) * n].clone()
    try:                                  # a lone rank asking for the state must get an error, not a hidden (hanging) collective
        opt.state_dict()
        guarded = False
    except RuntimeError:
        guarded = True
    checkpoint.save_training_state(path, m, opt, epoch=3, write=(rank == 0))     # every rank calls: the gather is a collective
    gathered_now = opt.exp_avg[a + other * n: a + (other + 1) * n].clone()
    dist.barrier()
    step(m, ex, 2)
    want = {k: v.detach().clone() for k, v in m.state_dict().items()}     # (the flat buffer's alignment gaps are not model state)
    want_m = opt.exp_avg.clone()
    # resume into differently initialised objects on every rank, from the ONE file rank 0 wrote
    m2, opt2, ex2 = make(777 + rank)
    info = checkpoint.load_training_state(path, m2, opt2)
    step(m2, ex2, 2)
    own = slice(a + rank * n, a + (rank + 1) * n)
    got = m2.state_dict()
    q.put((rank, info["epoch"], all(torch.equal(got[k], want[k]) for k in want), torch.equal(opt2.exp_avg[own], want_m[own]),
           bool((stale_before - gathered_now).abs().max() > 0), int(opt2.step_dev.item()), guarded))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_optimizer_checkpoint_resume_gloo(tmp_path, world):
    """A checkpoint taken from a SHARDED data-parallel run holds every rank's Adam moments (all-gathered inside
    optimizer.state_dict()), so resuming all ranks from rank 0's file continues bit-identically.  Without the gather rank 1
    would resume from rank 0's stale moments for the shards rank 1 owns."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = str(tmp_path / "state.pt")
    procs = [ctx.Process(target=_resume_worker, args=(r, world, port, q, path)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, epoch, same_params, same_moments, was_stale, step_no, guarded in res:
        assert epoch == 4 and step_no == 3
        assert guarded, "state_dict() of a sharded optimizer without gather_state() must raise"
        assert was_stale, "the non-owned shard's moments were already current before the gather: the test checks nothing"
        assert same_params, f"rank {rank}: resumed run diverged from the uninterrupted one"
        assert same_moments, f"rank {rank}: resumed moments differ on the owned shard"
