"""Two data-parallel ranks through the REAL kernel schedule on one GPU (gloo moves the CUDA gradient buckets; the
driver's multi-GPU runs use RCCL with the same GradExchange / bucket hooks): every bucket must be complete --
weight gradients from the side stream, bias / LayerNorm gradients from the main stream -- at the moment its hook
fires, the averaged gradients must drive Adam per bucket, and the ranks must end up with identical parameters equal
to a single-process step on the mean gradient."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


MC = {"modal": ["clip"], "modal_shape": [48], "text_enc_type": "CLIP", "embed_dim": 64, "dropout": 0.0, "loss_beta": 0.5,
      "matching": {"enable_tem": False, "matching_loss": "CSL"}, "activation": "gelu",
      "video_encoder": {"layer": 2, "nhead": 4, "feedforward": 128,
                        "mme": {"temporal": "encoding", "modal_different": True, "do_norm": False, "aggregation": "avg"}, "aoa": False},
      "caption_decoder": {"layer": 2, "nhead": 4, "feedforward": 128, "sce_loss_alpha": 0.5}, "pretrained_model": None}
VOCAB = 301


def _batch(seed, dev):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(6, 7, 48, generator=g).to(dev)
    mask = torch.zeros(6, 7, dtype=torch.bool); mask[1, 5:] = True; mask[4, 3:] = True
    ids = torch.randint(3, VOCAB, (6, 9), generator=g); ids[:, 0] = 101; ids[2, 6:] = 0
    return feats, mask.to(dev), ids.to(dev)


def _worker(rank, world, port, dtype_name, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from helpers import build_model
    from vct_amd.trainer import CaptionTrainer, FusedAdam, GradExchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dtype = getattr(torch, dtype_name)
    torch.manual_seed(50 + rank)                       # different init per rank: the constructor broadcast fixes it
    m = build_model(MC, VOCAB, "cuda", dtype)
    m.train()
    ex = GradExchange(m)
    opt = FusedAdam(m, lr=1e-3)
    tr = CaptionTrainer(m, opt, ex)
    start = m.flat_params.clone()
    losses = [float(tr.step(*_batch(10 + rank + 2 * k, dev))) for k in range(2)]
    torch.cuda.synchronize()
    gathered = [torch.empty_like(m.flat_params) for _ in range(world)]
    dist.all_gather(gathered, m.flat_params)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    ok_ref = True
    if rank == 0:
        # single process: the two ranks' gradients computed one after the other, averaged, one Adam step -- twice
        r = build_model(MC, VOCAB, "cuda", dtype)
        r.train()
        r.flat_params.copy_(start)
        ropt = FusedAdam(r, lr=1e-3)
        for k in range(2):
            acc = torch.zeros_like(r.flat_grads)
            for rr in range(world):
                r._ps.refresh_shadow(force=True)
                r.train_step_kernels(*_batch(10 + rr + 2 * k, dev))
                acc += r.flat_grads
            r.flat_grads.copy_(acc / world)
            ropt.step()
        named = lambda mm: {k: v for k, v in mm.state_dict().items()}
        a, b = named(m), named(r)
        ok_ref = all(torch.equal(a[k], b[k]) for k in a)
    q.put((rank, same, ok_ref, losses))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_two_ranks_one_gpu_step_equals_mean_gradient_step(dtype_name):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dtype_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, same, ok_ref, losses in res:
        assert same, "ranks ended with different parameters"
        assert ok_ref, "exchanged step differs from the single-process step on the mean gradient"
        assert all(l == l and l > 0 for l in losses)
    assert res[0][3] != res[1][3]            # the ranks did see different batches
