"""dev: world-4 on one GPU -- is the exchanged gradient of bucket 2 wrong because of the exchange or of a rank's local gradient?"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import torch, torch.multiprocessing as mp


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import numpy as np
    import torch.distributed as dist
    import vct_oracle as O
    from helpers import build_model
    from test_dist_gpu import MC, VOCAB, _batch, _np_batch
    from vct_amd.trainer import CaptionTrainer, FusedAdam, GradExchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(50 + rank)
    m = build_model(MC, VOCAB, "cuda", torch.float32)
    m.train()
    opt = FusedAdam(m, lr=1e-3)
    ex = GradExchange(m)
    sd = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    cfg = O.cfg_from_model_config(MC, VOCAB)
    bk = m.grad_buckets()
    start = m.flat_params.clone()
    # exchanged step FIRST (the very first backward of this process)
    tr = CaptionTrainer(m, opt, ex)
    tr.step(*_batch(10 + rank, "cuda"))
    torch.cuda.synchronize()
    got = m.flat_grads.double().cpu().numpy()
    # then the local gradient on the same start parameters, no exchange
    m.flat_params.copy_(start)
    m._ps.refresh_shadow(force=True)
    m.train_step_kernels(*_batch(10 + rank, "cuda"))
    torch.cuda.synchronize()
    local = m.flat_grads.clone()
    ref = O.caption_loss_and_grads(sd, cfg, *_np_batch(10 + rank))[1]
    flat_ref = np.zeros(m._ps.total)
    for n, v in ref.items():
        flat_ref[m._ps.offsets[n]:m._ps.offsets[n] + v.size] = v.reshape(-1)
    loc = local.double().cpu().numpy()
    errs_local = [float(np.linalg.norm(loc[a:b] - flat_ref[a:b]) / max(np.linalg.norm(flat_ref[a:b]), 1e-30)) for a, b in bk]
    mean_hip = local.clone(); dist.all_reduce(mean_hip); mean_hip = (mean_hip / world).double().cpu().numpy()
    errs_ex = [float(np.linalg.norm(got[a:b] - mean_hip[a:b]) / max(np.linalg.norm(mean_hip[a:b]), 1e-30)) for a, b in bk]
    q.put((rank, [round(e, 6) for e in errs_local], [round(e, 6) for e in errs_ex]))
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    from test_dist_gpu import _free_port
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    for r in sorted(q.get(timeout=300) for _ in range(world)):
        print(r, flush=True)
    [p.join() for p in ps]
