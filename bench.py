#!/usr/bin/env python
"""Benchmark of the caption TRAINING STEP on MI355X (BASELINE.json metric: train samples/sec, whole node).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward + backward (+ gradient all-reduce over RCCL when N > 1) + Adam + dropout-seed
advance on the configs[1] workload: 2 enc + 2 dec layers, d=512, ff=2048, 8 heads, V=30522, bf16
compute / fp32 masters, per-GPU batch 256 of synthetic (12 x 512) CLIP4Clip features -> 20-token
captions, dropout 0.3 active, SCE alpha 0.5.  Inputs are resident in HBM before the timed region.
Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL_CFG = {
    "modal": ["CLIP4Clip"], "modal_shape": [512], "tokenizer": "ids", "vocab_size": 30522, "text_enc_type": "CLIP",
    "embed_dim": 512, "dropout": 0.3, "loss_beta": 0.5, "matching": {"enable_tem": False, "matching_loss": "CSL"},
    "activation": "gelu",
    "video_encoder": {"layer": 2, "nhead": 8, "feedforward": 2048,
                      "mme": {"temporal": "encoding", "modal_different": True, "do_norm": False, "aggregation": "avg"}},
    "caption_decoder": {"layer": 2, "nhead": 8, "feedforward": 2048, "sce_loss_alpha": 0.5},
    "pretrained_model": None,
}
TRAIN_CFG = {"optimizer": {"name": "adam", "learning_rate": 1e-4, "beta": [0.9, 0.999], "weight_decay": 0,
                           "momentum": None, "lr_scheduler": None}}
T_FRAMES, S_TOK, D_IN, VOCAB = 12, 20, 512, 30522
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3


def algorithmic_flops(B, d=512, ff=2048, Le=2, Ld=2, T=T_FRAMES, S=S_TOK, V=VOCAB, d_in=D_IN):
    """SURVEY.md 8(d) FLOP model (2*m*n*k per GEMM, full QK^T / PV, elementwise ignored)."""
    Te, Sd = T + 1, S - 1
    unify = 2 * B * T * d_in * d
    enc = B * Te * (8 * d * d + 4 * d * ff) + 4 * B * Te * Te * d
    dec = B * Sd * (12 * d * d + 4 * d * ff) + 4 * B * Te * d * d + 4 * B * Sd * Sd * d + 4 * B * Sd * Te * d
    gen = 2 * B * Sd * d * V
    fwd = unify + Le * enc + Ld * dec + gen
    return {"fwd": fwd, "step": 3 * fwd, "gen": gen, "attn_ffn_fwd": Le * enc + Ld * dec}


def synthetic(B, rank, device):
    g = torch.Generator().manual_seed(0 + rank)
    feats = torch.randn(B, T_FRAMES, D_IN, generator=g)
    ids = torch.randint(1000, 30000, (B, S_TOK), generator=g)
    ids[:, 0], ids[:, -1] = 101, 102
    mask = torch.zeros(B, T_FRAMES, dtype=torch.bool)
    return feats.to(device), mask.to(device), ids.to(device)


def cpu_baseline(budget_s=20.0):
    """The numpy oracle (a port of the reference algorithm, oracle/vct_oracle.py) timed on the host:
    forward + backward + Adam at batch 32 of the SAME model/workload shape, fp32, no dropout."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vct_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    cfg = O.cfg_from_model_config(MODEL_CFG, VOCAB)
    p = O.init_params(cfg, seed=0)
    Bc = 32
    feats, mask, ids = O.synthetic_batch(Bc, T_FRAMES, D_IN, S_TOK, VOCAB, seed=0)
    state = {}
    t_total, n = 0.0, 0
    for it in range(1 + 50):
        t0 = time.perf_counter()
        loss, grads, _ = O.caption_loss_and_grads(p, cfg, feats, mask, ids)
        p = O.adam_step(p, grads, state)
        dt = time.perf_counter() - t0
        if it > 0:            # first iteration = warm-up
            t_total += dt; n += 1
        if t_total > budget_s or (it > 0 and t_total + dt > budget_s * 1.5):
            break
    return {"value": round(Bc * n / t_total, 2), "unit": "samples/s", "cores": int(threads), "kind": "port",
            "sample": f"{n} steps of fwd+bwd+Adam at batch {Bc} (same 4-layer d=512 model, T=12->S=20, fp32, dropout off) "
                      f"with the numpy oracle, {t_total:.1f} s", "loss": float(loss)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--payload", default="fp32", choices=["fp32", "bf16"], help="gradient all-reduce payload")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured hipGraph (N=1); the per-kernel roofline "
                    "timing then comes from an eager tail pass, so the default is eager launches")
    ap.add_argument("--overlap-adam", action="store_true", help="A/B: Adam per gradient bucket on the side stream during backward (measured slower)")
    ap.add_argument("--no-overlap-dw", action="store_true", help="A/B: weight-gradient GEMMs on the main stream")
    ap.add_argument("--no-overlap-kv", action="store_true", help="A/B: cross-attention K/V projections and d(memory) GEMMs on the main stream")
    ap.add_argument("--no-overlap-enc", action="store_true", help="A/B: encoder backward after (not beside) the decoder's tail")
    ap.add_argument("--no-group-dw", action="store_true", help="A/B: one launch per weight-gradient GEMM instead of one per layer")
    ap.add_argument("--force-exchange", action="store_true", help="run the RCCL gradient exchange even at world size 1 (plumbing test)")
    ap.add_argument("--torch-adam", action="store_true", help="A/B: torch.optim.Adam(fused=True) + cast kernels instead of vct_adam_step")
    args = ap.parse_args()

    import torch.distributed as dist
    import vct_amd  # noqa: F401
    from vct_amd import ops
    from vct_amd.model import MMT4Caption
    from vct_amd.trainer import CaptionTrainer, GradExchange, build_optimizer
    from vct_amd.utils import configure_hardware, setup_seed

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    device, rank, world = configure_hardware("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.no_overlap_dw:
        from vct_amd import engine as _eng
        _eng._StackBase.overlap_dw = False
    if args.no_overlap_kv:
        from vct_amd import engine as _eng
        _eng._StackBase.overlap_kv = False
    if args.no_overlap_enc:
        MMT4Caption.overlap_enc_bwd = False
        MMT4Caption.overlap_dec_prefix = False
    if os.environ.get("VCT_NO_DEC_PREFIX"):
        MMT4Caption.overlap_dec_prefix = False
    if args.no_group_dw:
        from vct_amd import engine as _eng
        _eng._StackBase.group_dw = False
    setup_seed(666)                       # reference train.py:308: same seed on every rank
    model = MMT4Caption(MODEL_CFG, device=device, compute_dtype={"bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype])
    model.mode("caption")
    model.train()
    if args.force_exchange and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    ex = (GradExchange(model, payload_dtype=torch.bfloat16 if args.payload == "bf16" else None, force=args.force_exchange)
          if (world > 1 or args.force_exchange) else None)
    if args.torch_adam:
        flat = torch.nn.Parameter(model.flat_params); flat.grad = model.flat_grads
        opt = torch.optim.Adam([flat], lr=1e-4, betas=(0.9, 0.999), fused=True)
    else:
        opt, _ = build_optimizer(TRAIN_CFG, model)
    trainer = CaptionTrainer(model, opt, ex, use_graph=args.graph)
    trainer.overlap_adam = args.overlap_adam
    feats, mask, ids = synthetic(args.batch, rank, device)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = trainer.step(feats, mask, ids)
    sync()
    for tag in ("gen_fwd", "gen_dx", "gen_dw"):
        ops.event_taps[tag] = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(feats, mask, ids)
    sync()
    elapsed = time.perf_counter() - t0
    if trainer.use_graph:      # replayed graphs bypass the Python taps: time the generator GEMMs in an eager tail pass
        trainer.use_graph = False
        for _ in range(5):
            trainer.step(feats, mask, ids)
        torch.cuda.synchronize()
        trainer.use_graph = True
    taps = {k: list(v) for k, v in ops.event_taps.items()}
    ops.event_taps.clear()
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    final_loss = float(loss)

    if rank == 0:
        fl = algorithmic_flops(args.batch)
        kern = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in taps.items() if v}   # ms per launch
        # roofline kernel = the generator forward GEMM: the largest single kernel and the only one of the three that
        # runs alone on the device (the dW GEMM shares the CUs with the dX chain on the side stream, so its event
        # bracket measures co-scheduled time, reported in all_ms for information only)
        dom = "gen_fwd"
        traffic = None        # HBM bytes per launch of the dominant kernel, from the committed PMC passes (profiles/)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_roofline_traffic.json")))
            if args.batch == 256 and args.dtype == "bf16":
                traffic = tj[dom]["hbm_bytes"]
        except Exception:
            traffic = None
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        achieved = fl["gen"] / (kern[dom] * 1e-3) / 1e12
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "video-caption train samples/sec (whole node)",
            "value": round(args.batch * world * args.steps / elapsed, 1), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: 2 enc + 2 dec layers d=512 ff=2048 H=8 V=30522, synthetic (256,12,512) "
                                   "features -> 20-token captions per GPU, fwd+bwd+Adam, dropout 0.3, SCE alpha 0.5",
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "seq_len": S_TOK, "frames": T_FRAMES,
                       "parallelism": f"dp{world}", "grad_allreduce_payload": args.payload if world > 1 else None,
                       "hipgraph_step": bool(trainer.use_graph)},
            "step_tflops": round(fl["step"] / (ms * 1e-3) / 1e12, 1),
            "step_frac_of_peak": round(fl["step"] / (ms * 1e-3) / 1e12 / peak, 4),
            "roofline": {"bound": "mfma", "kernel": {"gen_fwd": "generator GEMM fwd 4864x30522x512 (gemm_bf16_v2_kernel NT, 128x128 tile, 8 waves, LDS-DMA double buffer)",
                                                     "gen_dx": "generator dX GEMM (vct_gemm NN)",
                                                     "gen_dw": "generator dW GEMM (vct_gemm TN)"}[dom],
                         "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic, "traffic_source": "rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_roofline_traffic.json",
                         "flops_per_launch": fl["gen"], "avg_ms_per_launch": round(kern[dom], 4),
                         "all_ms": {k: round(v, 4) for k, v in kern.items()}},
            "loss": final_loss,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
