"""Host-side helpers mirroring the reference's utils.py (mask builder, JSON config, seeding,
process-group setup)."""
import json
import os
import random

import numpy as np
import torch


def generate_square_subsequent_mask(sz: int) -> torch.Tensor:
    """Float [sz,sz] causal mask: 0 on/below the diagonal, -inf above (reference utils.py:63-66).
    The attention kernel bakes this predicate in (causal=1); the function exists for API parity."""
    return torch.triu(torch.full((sz, sz), float("-inf")), diagonal=1)


class Config:
    """JSON config loader (reference utils.py:82-89); `.data['model']` feeds MMT4Caption unchanged."""

    def __init__(self, path: str):
        with open(path) as f:
            self.data = json.load(f)

    def check(self):
        if self.data["model"]["video_encoder"].get("type", "mme") != "mme":
            raise ValueError("only the 'mme' video encoder is on the accelerated caption path")


def setup_seed(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def configure_hardware(backend: str = None):
    """One process per GPU (reference utils.py:126-149).  Reads RANK/LOCAL_RANK/WORLD_SIZE from the
    launcher env; backend 'nccl' is RCCL on ROCm, 'gloo' for CPU tests.  Returns (device, rank, world)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available() and backend != "gloo"
    backend = os.environ.get("VCT_DIST_BACKEND", backend)     # tests: several ranks on ONE GPU talk over gloo
    if use_cuda:
        n_dev = max(torch.cuda.device_count(), 1)
        if local >= n_dev:
            # more local ranks than visible GPUs: only the one-GPU multi-rank TESTS want that (they set VCT_DIST_BACKEND=gloo);
            # a real launch that over-subscribes a GPU would silently halve every rank's throughput -- refuse it
            if os.environ.get("VCT_DIST_BACKEND") is None:
                raise RuntimeError(f"LOCAL_RANK={local} but only {n_dev} GPU(s) are visible: one process per GPU "
                                   "(set VCT_DIST_BACKEND=gloo to share a GPU between ranks in tests)")
            local = local % n_dev
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {"device_id": device} if (use_cuda and backend in (None, "nccl")) else {}
        dist.init_process_group(backend=backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world, **kw)
    return device, rank, world


import contextlib
import gc


@contextlib.contextmanager
def capture_graph(graph: "torch.cuda.CUDAGraph"):
    """`with torch.cuda.graph(graph)` with the cyclic garbage collector held off for the duration of the capture.  A collection
    that happens to run INSIDE a capture destroys whatever unreachable objects it finds -- older CUDAGraphs, pinned host tensors,
    events of sessions that earlier code dropped -- and their destructors call HIP APIs that are illegal while a stream of this
    thread is capturing; the error surfaces inside a C++ destructor and aborts the process (seen once in ~4 runs of the GPU test
    suite, at a different place each time).  torch.cuda.graph collects BEFORE it starts capturing; this keeps it that way."""
    was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        if was_enabled:
            gc.enable()
