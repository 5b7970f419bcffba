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
PEAK_HBM_GBS = 8000.0       # HBM3E spec (MI355X_MICROARCH.md; ~6300 GB/s achievable)


def algorithmic_flops(B, d=512, ff=2048, Le=2, Ld=2, T=T_FRAMES, S=S_TOK, V=VOCAB, d_in=D_IN):
    """SURVEY.md 8(d) FLOP model (2*m*n*k per GEMM, full QK^T / PV, elementwise ignored)."""
    Te, Sd = T + 1, S - 1
    unify = 2 * B * T * d_in * d
    enc = B * Te * (8 * d * d + 4 * d * ff) + 4 * B * Te * Te * d
    dec = B * Sd * (12 * d * d + 4 * d * ff) + 4 * B * Te * d * d + 4 * B * Sd * Sd * d + 4 * B * Sd * Te * d
    gen = 2 * B * Sd * d * V
    fwd = unify + Le * enc + Ld * dec + gen
    return {"fwd": fwd, "step": 3 * fwd, "gen": gen, "attn_ffn_fwd": Le * enc + Ld * dec, "enc_stack": unify + Le * enc, "dec_stack": Ld * dec}


def synthetic(B, rank, device):
    g = torch.Generator().manual_seed(0 + rank)
    feats = torch.randn(B, T_FRAMES, D_IN, generator=g)
    ids = torch.randint(1000, 30000, (B, S_TOK), generator=g)
    ids[:, 0], ids[:, -1] = 101, 102
    mask = torch.zeros(B, T_FRAMES, dtype=torch.bool)
    return feats.to(device), mask.to(device), ids.to(device)


def cpu_baseline():
    """the reference's own module graph (stock torch.nn under autograd, restated in
    oracle/torch_ref.py and validated against the pinned numpy oracle in tests/test_oracle_golden.py) timed on THIS box's
    host cores: same model, same synthetic batches, fp32, dropout 0.3 ACTIVE, Adam step included,
    1 warm-up + 5 timed steps (median) at batch 256 (= the GPU workload; `value` is that one), 64 and 8 (configs[0])."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    threads = os.cpu_count() or 1
    import subprocess
    try:
        cpu = [l.split(":", 1)[1].strip() for l in subprocess.run(["lscpu"], capture_output=True, text=True).stdout.splitlines()
               if l.startswith("Model name")][0]
    except Exception:
        cpu = "unknown"
    # each batch size runs in its own process under a hard timeout: an over-subscribed torch CPU run (e.g. 256 threads
    # spinning on the tiny kernels of batch 8) can take minutes per step and must not stall the GPU benchmark
    phys = max(1, threads // 2) if threads >= 32 else threads        # SMT siblings do not help these GEMMs
    code = ("import sys, json; sys.path.insert(0, %r); import bench, torch_ref as TR, vct_oracle as O; "
            "cfg = O.cfg_from_model_config(bench.MODEL_CFG, bench.VOCAB); B, steps, th = map(int, sys.argv[1:4]); "
            "f, m, i = O.synthetic_batch(B, bench.T_FRAMES, bench.D_IN, bench.S_TOK, bench.VOCAB, seed=0); "
            "print(json.dumps(TR.time_training_steps(cfg, bench.MODEL_CFG['dropout'], f, m, i, steps, th)))")
    code = code % os.path.join(ROOT, "oracle")
    STEPS = 5
    per_batch, legs, spent, loss, used = {}, [], 0.0, None, None

    def leg(Bc, th, limit):
        env = dict(os.environ, OMP_NUM_THREADS=str(th), MKL_NUM_THREADS=str(th), PYTHONPATH=ROOT, HIP_VISIBLE_DEVICES="")
        r = subprocess.run([sys.executable, "-c", code, str(Bc), str(STEPS), str(th)], capture_output=True, text=True,
                           timeout=limit, env=env, cwd=ROOT)
        return json.loads(r.stdout.strip().splitlines()[-1])
    for th in (phys, min(phys, 32)):                     # the headline leg decides the thread count (fallback: 32 threads)
        t0 = time.perf_counter()
        try:
            rate, total, loss = leg(256, th, 120.0)
        except Exception:
            spent += time.perf_counter() - t0
            continue
        per_batch["256"], used = round(rate, 2), th
        spent += total
        legs.append(f"batch 256: 1 warm-up + {STEPS} timed steps")
        break
    if used is None:
        return {"value": None, "unit": "samples/s", "cores": int(phys), "kind": "port", "cpu": cpu,
                "sample": "torch-CPU baseline did not finish inside its time limit on this host"}
    for Bc, limit in ((64, 60.0), (8, 40.0)):
        # small batches do not scale to every core (the per-op work is tiny): 32 threads is what the survey container's numbers
        # were taken near, and it keeps a 256-thread host from spinning
        th = min(used, 32) if Bc == 8 else used
        try:
            rate, total, _ = leg(Bc, th, limit)
            per_batch[str(Bc)] = round(rate, 2)
            spent += total
            legs.append(f"batch {Bc}: 1 warm-up + {STEPS} timed steps" + (f" on {th} threads" if th != used else ""))
        except Exception:
            legs.append(f"batch {Bc}: did not finish in {limit:.0f} s (not reported)")
    return {"value": per_batch["256"], "unit": "samples/s", "cores": int(used), "kind": "port", "cpu": cpu,
            "by_batch": per_batch, "statistic": "batch / median step time",
            "sample": f"torch-CPU restatement of the reference modules (oracle/torch_ref.py), fp32, dropout 0.3 active, "
                      f"fwd+bwd+Adam, {used} threads (host: {threads} hardware threads), the same 4-layer d=512 model "
                      f"(T=12->S=20, V=30522); " + "; ".join(legs) + f"; {spent:.1f} s of timed CPU work", "loss": loss}


def exchange_path_n1(batch, dtype):
    """The schedule every N > 1 run takes -- reduce-scatter -> Adam on the owned shard -> all-gather per gradient bucket on the
    communicator's stream (`sharded`), or bucketed all-reduce + per-bucket Adam (`allreduce`) -- at world size 1, where the
    collectives move no bytes: what a rank pays for the exchange path BEFORE any wire time (no optimizer epilogue in the
    weight-gradient GEMMs there).  Untimed side line: each leg is this script in its own process (own RCCL communicator),
    --force-exchange, 4 warm-up + 10 timed steps."""
    import subprocess
    out = {"what": "bench.py --force-exchange at world size 1 (own process per leg, 4 warm-up + 10 timed steps)"}
    for kind in ("sharded", "allreduce"):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "4", "--batch", str(batch),
               "--dtype", dtype, "--force-exchange", "--exchange", kind, "--no-cpu-baseline", "--no-decode", "--no-b1024",
               "--no-other-configs", "--no-exchange-line"]
        env = dict(os.environ, MASTER_PORT=str(29531 + (kind == "allreduce")))
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
            line = json.loads(r.stdout.strip().splitlines()[-1])
            out[kind + "_ms"] = line["ms_per_step"]
            out[kind + "_comm"] = (line.get("comm") or {}).get("kind")
        except Exception as e:
            out[kind + "_ms"] = None
            out[kind + "_error"] = repr(e)[:200]
    return out


OTHER_CONFIGS = (("shipped d=768 1+3 T=12 S=20", 768, 1, 3, 12, 20), ("configs[3] d=1024 6+6 T=32 S=40", 1024, 6, 6, 32, 40))


def other_configs(device, dtype, peak):
    """Side lines (untimed w.r.t. the headline value): the same training step -- fwd + bwd + Adam, dropout 0.3, batch 256, recorded launch
    list -- on the other BASELINE.json shapes: the shipped MSR-VTT config (d=768, 1 enc + 3 dec layers) and configs[3] (d=1024, 6 + 6
    layers, 32 frames, 40 tokens, head_dim 128).  3 warm-up + 6 timed steps each, HIP events on the step's stream."""
    import copy
    from vct_amd.model import MMT4Caption
    from vct_amd.trainer import CaptionTrainer, build_optimizer
    out = {}
    for name, d, Le, Ld, T, S in OTHER_CONFIGS:
        try:
            mc = copy.deepcopy(MODEL_CFG)
            mc["embed_dim"] = d
            mc["video_encoder"]["layer"], mc["caption_decoder"]["layer"] = Le, Ld
            torch.manual_seed(666)
            m = MMT4Caption(mc, device=device, compute_dtype=dtype)
            m.mode("caption"); m.train()
            opt, _ = build_optimizer(TRAIN_CFG, m)
            tr = CaptionTrainer(m, opt, launch_list=True)
            g = torch.Generator().manual_seed(0)
            feats = torch.randn(256, T, D_IN, generator=g).to(device)
            mask = torch.zeros(256, T, dtype=torch.bool, device=device)
            ids = torch.randint(1000, 30000, (256, S), generator=g); ids[:, 0], ids[:, -1] = 101, 102
            feats, mask, ids = tr.adopt_inputs(feats, mask, ids.to(device))
            for _ in range(3):
                tr.step(feats, mask, ids)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(6):
                loss = tr.step(feats, mask, ids)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 6
            fl = algorithmic_flops(256, d=d, ff=2048, Le=Le, Ld=Ld, T=T, S=S)["step"]
            out[name] = {"ms_per_step": round(ms, 3), "samples_per_s": round(256 / ms * 1e3, 1), "step_tflops": round(fl / ms / 1e9, 1),
                         "step_frac_of_peak": round(fl / ms / 1e9 / peak, 4), "params_M": round(m.caption_param_end / 1e6, 1),
                         "loss": round(float(loss), 4)}
            del m, tr, opt
            torch.cuda.empty_cache()
        except Exception as e:          # a side line never takes the headline down with it
            out[name] = {"error": repr(e)[:200]}
    return out


def decode_line(device, dtype):
    """configs[4]: greedy decode (KV cache + captured per-token step) of the cfg-B model in eval mode, batch 1 and 128:
    whole-call time per token step (encoder forward, memory K/V projection and host checks included) and the step alone.
    A FRESH random-init model: it practically never emits [SEP], so every caption runs the full 29 steps (the trained-
    for-30-steps bench model stops after one).  Reported beside the training metric; not the headline value."""
    from vct_amd.model import MMT4Caption
    torch.manual_seed(666)
    model = MMT4Caption(MODEL_CFG, device=device, compute_dtype=dtype)
    model.mode("caption")
    model.eval()
    out = {}
    for B in (1, 128):
        feats = torch.randn(B, T_FRAMES, D_IN, generator=torch.Generator().manual_seed(0)).to(device)
        for _ in range(2):
            ys = model.greedy_decode_ids([feats], None, max_len=30)
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            ys = model.greedy_decode_ids([feats], None, max_len=30)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        steps = ys.shape[1] - 1
        out[f"batch{B}"] = {"us_per_token_step": round(dt / steps * 1e6, 1), "tokens_per_s": round(B * steps / dt, 1), "steps": steps}
        # the token step alone (its captured graphs replayed back to back, HIP events): without the encoder forward, the
        # cross-attention K/V projection of the memory and the host's end-of-sequence check every 4 tokens
        st = next((s for k, s in model.__dict__.get("_decode_sessions", {}).items() if k[0] == B), None)
        if st is not None and all(t in st.graphs for t in range(1, 30)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                for t in range(1, 30):
                    st.graphs[t].replay()
            e1.record()
            torch.cuda.synchronize()
            out[f"batch{B}"]["us_per_step_replay_only"] = round(e0.elapsed_time(e1) * 1e3 / (reps * 29), 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--payload", default="bf16", choices=["fp32", "bf16"], help="gradient reduce-scatter / all-reduce payload (N > 1)")
    ap.add_argument("--exchange", default="sharded", choices=["sharded", "allreduce", "c10d"],
                    help="N > 1: reduce-scatter + Adam on the owned 1/N + all-gather over the library's RCCL communicator (default), "
                         "all-reduce + replicated Adam over the same communicator, or torch.distributed's all-reduce (round-1 path)")
    ap.add_argument("--executor", default="list", choices=["list", "eager", "graph"],
                    help="N=1: how the ~100 launches of a step are issued -- a C-side recorded launch list (default), Python/ctypes "
                         "eager launches, or a captured hipGraph")
    ap.add_argument("--graph", action="store_true", help="same as --executor graph")
    ap.add_argument("--no-decode", action="store_true", help="skip the greedy-decode line (configs[4])")
    ap.add_argument("--no-b1024", action="store_true", help="skip the north_star_b1024 side line (same kernels, per-GPU batch 1024)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the side lines of the shipped d=768 shape and configs[3] (d=1024, 6+6)")
    ap.add_argument("--comm-cu-mask", type=int, default=0, help="N > 1: confine the communicator's stream (RCCL kernels) to the first N CUs")
    ap.add_argument("--overlap-adam", action="store_true", help="A/B: Adam per gradient bucket on the side stream during backward (measured slower)")
    ap.add_argument("--no-overlap-dw", action="store_true", help="A/B: weight-gradient GEMMs on the main stream")
    ap.add_argument("--no-overlap-kv", action="store_true", help="A/B: cross-attention K/V projections and d(memory) GEMMs on the main stream")
    ap.add_argument("--no-overlap-enc", action="store_true", help="A/B: encoder backward after (not beside) the decoder's tail")
    ap.add_argument("--no-group-dw", action="store_true", help="A/B: one launch per weight-gradient GEMM instead of one per layer")
    ap.add_argument("--force-exchange", action="store_true", help="run the RCCL gradient exchange even at world size 1 (plumbing test)")
    ap.add_argument("--no-exchange-line", action="store_true", help="skip the exchange_path_n1 side line (the N > 1 schedule at world size 1)")
    ap.add_argument("--torch-adam", action="store_true", help="A/B: torch.optim.Adam(fused=True) + cast kernels instead of vct_adam_step")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): RCCL prints a version banner to the C-level stdout when a communicator is
    # created, so everything else written to fd 1 by this process goes to stderr and the JSON goes to a private duplicate
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch.distributed as dist
    import vct_amd  # noqa: F401
    from vct_amd import ops
    from vct_amd.model import MMT4Caption
    from vct_amd.trainer import CaptionTrainer, GradExchange, build_optimizer
    from vct_amd.utils import configure_hardware, setup_seed

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.comm_cu_mask > 0:
        os.environ["VCT_COMM_CU_MASK"] = str(args.comm_cu_mask)      # read by vct_comm_init when it creates the communicator's stream
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
    if args.torch_adam:
        flat = torch.nn.Parameter(model.flat_params); flat.grad = model.flat_grads
        opt = torch.optim.Adam([flat], lr=1e-4, betas=(0.9, 0.999), fused=True)
    else:
        opt, _ = build_optimizer(TRAIN_CFG, model)
    ex, exchange_kind, rccl_ranks = None, None, None
    if world > 1 or args.force_exchange:
        payload = torch.bfloat16 if args.payload == "bf16" else None
        if args.exchange != "c10d" and not args.torch_adam:
            from vct_amd.comm import C10dColl, RcclColl
            from vct_amd.trainer import ShardedExchange
            coll, why = None, None
            if os.environ.get("VCT_DIST_BACKEND", "nccl") == "nccl":
                good = 0
                try:
                    coll = RcclColl(device=device)
                    good = 1 if coll.self_test() else 0
                    if not good:
                        why = "self-test of the collectives failed"
                except Exception as e:      # never silently: say why the library's communicator is not carrying the gradients
                    coll, why = None, repr(e)
                # ONE decision for the whole job: a rank that fell back alone would wait in torch.distributed collectives the
                # others never enter
                ok = torch.tensor([good], device=device)
                if dist.is_initialized():
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok) != 1:
                    coll, why = None, why or "vct_comm failed on another rank"
            else:
                why = "VCT_DIST_BACKEND is not nccl"
            if coll is None:
                print(f"[bench] WARNING rank {rank}: vct_comm (own RCCL communicator) unavailable: {why}; using torch.distributed collectives",
                      file=sys.stderr, flush=True)
                coll = C10dColl()
                payload = None
            rccl_ranks = getattr(coll, "lib_world", None)      # what the RCCL communicator itself reports (None: torch.distributed fallback)
            ex = ShardedExchange(model, opt, coll, sharded=args.exchange == "sharded", payload_dtype=payload)
            exchange_kind = f"{args.exchange}/{'vct_comm' if coll.owns_stream else 'c10d'}"
            if os.environ.get("VCT_COMM_IDLE") == "1":      # experiment: the communicator exists but carries nothing
                ex, exchange_kind = None, "idle communicator"
        else:
            ex = GradExchange(model, payload_dtype=payload, force=args.force_exchange)
            exchange_kind = "allreduce/torch.distributed"
    if args.graph:
        args.executor = "graph"
    # the recorded launch list is the default at every N: with the sharded exchange the collectives are part of the recording
    # (vct_comm_* stream work; tests/test_dist_gpu.py runs the recorded exchange with two ranks).  VCT_LIST_MULTI=0: eager at N > 1.
    use_list = args.executor == "list" and (world == 1 or os.environ.get("VCT_LIST_MULTI", "1") != "0")
    trainer = CaptionTrainer(model, opt, ex, use_graph=args.executor == "graph", launch_list=use_list)
    trainer.overlap_adam = args.overlap_adam
    feats, mask, ids = synthetic(args.batch, rank, device)
    # the batch is resident in HBM (contract): hand it over in the executor's own input buffers, so that no staging copy runs per step
    feats, mask, ids = trainer.adopt_inputs(feats, mask, ids)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # live kernel timing: HIP events recorded by the C runtime on the launch stream, inside the timed region (they are part
    # of the recorded launch list, so replays carry them too).  Every bracket is two event records in the stream and all of them
    # cost the step ~55 us (tools/taps_cost.py), so inside the timed region only ONE kernel is bracketed: the roofline kernel =
    # the LONGEST single launch of the step among the sample-stationary encoder / decoder stacks, the three generator GEMMs, the loss
    # and the optimizer's pass, chosen from the first half of the warm-up, which runs with those brackets on.  The other brackets
    # (north_star, the rest of the top-5, ...) come from a short second pass.
    GEN = ("gen_fwd", "gen_dx", "gen_dw")
    # candidates for the roofline kernel: every bracket that is ONE launch (or one launch + its fixed-order reduce) of the step
    CAND = GEN + ("ss_enc", "ss_dec", "loss", "adam")
    dom = "gen_dw"
    w_probe = args.warmup // 2 if args.warmup >= 4 else 0
    if w_probe:
        ops.taps_enable(True, only=CAND)
        for _ in range(w_probe):
            loss = trainer.step(feats, mask, ids)
        sync()
        probe = {tag: ops.tap_collect(tag) for tag in CAND}
        probe = {k: float(np.mean(v[1:] if len(v) > 1 else v)) for k, v in probe.items() if v}
        if probe:
            dom = max(probe, key=probe.get)
        trainer.drop_recordings()
    ops.taps_enable(True, only=(dom,))
    for _ in range(args.warmup - w_probe):
        loss = trainer.step(feats, mask, ids)
    sync()
    for tag in ops.TAPS:
        ops.tap_collect(tag)              # drop the warm-up brackets
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(feats, mask, ids)
    sync()
    elapsed = time.perf_counter() - t0
    final_loss = float(loss)          # of the last TIMED step (the loss buffer is static: the untimed passes below overwrite it)
    taps = {tag: ops.tap_collect(tag) for tag in ops.TAPS}
    # second pass, untimed: every bracket, fresh recordings (a recording contains the brackets that were active when it was made)
    ops.taps_enable(True)
    trainer.drop_recordings()
    extra_steps = max(4, min(10, args.steps))
    for _ in range(3):
        trainer.step(feats, mask, ids)
    sync()
    for tag in ops.TAPS:
        ops.tap_collect(tag)
    for _ in range(extra_steps):
        trainer.step(feats, mask, ids)
    sync()
    for tag in ops.TAPS:
        if tag != dom:
            taps[tag] = ops.tap_collect(tag)
    # side line: the SAME kernels at a per-GPU batch of 1024 (what the 288 GB leave room for): how much of the north_star gap is
    # the row count of configs[1] (M = 4864 / 3328 rows per GEMM) rather than the kernels
    b1024 = None
    if world == 1 and args.batch == 256 and not args.no_b1024:
        ops.taps_enable(True, only=("layers_fwd", "step"))
        trainer.drop_recordings()
        big = trainer.adopt_inputs(*synthetic(1024, rank, device))
        for _ in range(3):
            trainer.step(*big)
        sync()
        for tag in ("layers_fwd", "step"):
            ops.tap_collect(tag)
        for _ in range(6):
            trainer.step(*big)
        sync()
        b1024 = {tag: float(np.mean(ops.tap_collect(tag))) for tag in ("layers_fwd", "step")}
    ops.taps_enable(False)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)

    if rank == 0:
        fl = algorithmic_flops(args.batch)
        kern = {k: float(np.mean(v)) for k, v in taps.items() if v}   # ms per bracket, averaged over the timed steps
        # roofline kernel = the LONGEST single launch of the step (picked above among CAND, bracketed inside the timed region).  The
        # vocabulary weight gradient shares the chip with the encoder backward on the second stream, so its bracket is co-scheduled
        # time -- that is what the step pays for it, and what is reported.
        # HBM bytes per launch of the dominant kernel, from the committed PMC passes (profiles/): the entry must name the SAME
        # kernel symbol the tag runs today -- a stale file (the kernel behind a tag changed) yields null, not a wrong number
        fused_adam = bool(getattr(trainer, "fuse_adam", False))
        Md, Vp = args.batch * (S_TOK - 1), (VOCAB + 31) // 32 * 32
        esz = 2 if args.dtype == "bf16" else 4
        n_par = model.caption_param_end
        n_adam = getattr(opt, "range_elems", {}).get((0, model.encoder_param_begin), model.encoder_param_begin)
        # tag -> (bound, algorithmic work per launch [FLOP | bytes], kernel symbol in a rocprofv3 kernel trace, description)
        INFO = {
            "gen_fwd": ("mfma", fl["gen"], "gemm256_kernel<0, 1, bf16>",
                        "generator GEMM fwd 4864x30522x512 (gemm256_kernel NT: persistent 256x256 tiles, 8 waves, LDS-DMA double buffer)"),
            "gen_dx": ("mfma", fl["gen"], "gemm256_kernel<0, 1, float>" if getattr(model.cap_decoder._engine(), "_wgt", None) is not None
                       else "g32_kernel<0, 0, float",
                       "generator dX GEMM 4864x512x30522 (g32_kernel NN: persistent 256x256 tiles, software-pipelined K loop on 32x32x16, "
                       "split over K, + fixed-order reduce)"),
            "gen_dw": ("mfma", fl["gen"], "g32_kernel<1, 0, float",
                       "generator dW GEMM 30522x512x4864 (g32_kernel TN: persistent 256x256 tiles, software-pipelined K loop, transpose reads of "
                       "both operands, bias gradient balanced over the waves, fp32 out" + (", torch.optim.Adam's step on W_g in its epilogue" if fused_adam else "")
                       + "; runs beside the encoder backward)"),
            "ss_enc": ("mfma", fl["enc_stack"], "layer_ss_fwd_kernel<false, 1>",
                       "sample-stationary ENCODER stack forward, one launch: unify Linear + mean token + temporal encoding + 2 layers + final norm"),
            "ss_dec": ("mfma", fl["dec_stack"], "layer_ss_fwd_kernel<true, 2>",
                       "sample-stationary DECODER stack forward, one launch: token embedding + 2 layers (self-, cross-attention, feed-forward) + final norm"),
            "loss": ("hbm", 2 * Md * Vp * esz, "sce_loss_kernel",
                     "SCE loss + d/dlogits over the materialised logits: one read + one write"),
            "adam": ("hbm", 30 * n_adam, "adam_ranges_kernel" if fused_adam else "adam_kernel",
                     "optimizer pass over " + ("what the weight-gradient GEMMs' epilogues did not step (token embedding, biases, LayerNorm parameters)"
                                                if fused_adam else "everything but the encoder") + ": 16 B read + 14 B written per parameter"),
        }
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS

        def roof(tag, ms_):
            bound, work, _, _ = INFO[tag]
            if bound == "mfma":
                ach, pk, unit = work / (ms_ * 1e-3) / 1e12, peak, "TFLOP/s"
            else:
                ach, pk, unit = work / (ms_ * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
            return bound, ach, pk, unit
        traffic, traffic_src = None, None
        for name in ("r06_roofline_traffic.json", "r05_roofline_traffic.json", "r04_roofline_traffic.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", name)))
                ent = tj[dom]
                if args.batch == 256 and args.dtype == "bf16" and INFO[dom][2] in str(ent.get("kernel", "")):
                    traffic, traffic_src = ent["hbm_bytes"], f"rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE of {INFO[dom][2]}, profiles/{name}"
                    break
            except Exception:
                continue
        r_bound, achieved, r_peak, r_unit = roof(dom, kern[dom])
        ms = elapsed / args.steps * 1e3
        top5 = sorted(((k, kern[k]) for k in CAND if k in kern), key=lambda kv: -kv[1])[:5]
        hbm = {}
        if "loss" in kern:      # SCE loss + d/dlogits: one read + one write of the logits
            by = 2 * Md * Vp * esz
            hbm["sce_loss"] = {"bytes": by, "ms": round(kern["loss"], 4), "achieved": round(by / (kern["loss"] * 1e-3) / 1e9, 1),
                               "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(by / (kern["loss"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
        if "adam" in kern:      # Adam over everything but the encoder: 16 B read + 12 B written (+2 B shadow) per parameter
            by = 30 * n_adam
            hbm["adam"] = {"bytes": by, "ms": round(kern["adam"], 4), "achieved": round(by / (kern["adam"] * 1e-3) / 1e9, 1),
                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(by / (kern["adam"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                           "note": f"{n_adam} of {n_par} parameters in this launch" + (" (weight matrices are stepped inside their weight-gradient GEMMs)"
                                                                                         if fused_adam else " (the encoder's follow in a second launch)")}
        north = None
        if "layers_fwd" in kern:
            lf = kern["layers_fwd"]
            north = {"what": "attention + FFN forward of the 2 encoder + 2 decoder layers (north_star target: >= 40 % of bf16 MFMA peak); "
                             "bracket = main-stream HIP events from the input cast to the decoder's final LayerNorm, i.e. it also "
                             "contains the unify GEMM, the encoder front end and the token embedding (not counted as FLOPs)",
                     "flops": fl["attn_ffn_fwd"], "ms": round(lf, 4), "tflops": round(fl["attn_ffn_fwd"] / (lf * 1e-3) / 1e12, 1),
                     "frac_of_peak": round(fl["attn_ffn_fwd"] / (lf * 1e-3) / 1e12 / peak, 4), "target_frac": 0.40,
                     "two_stream_overlap": not args.no_overlap_enc}
        out = {
            "metric": "video-caption train samples/sec (whole node)",
            "value": round(args.batch * world * args.steps / elapsed, 1), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: 2 enc + 2 dec layers d=512 ff=2048 H=8 V=30522, synthetic (256,12,512) "
                                   "features -> 20-token captions per GPU, fwd+bwd+Adam, dropout 0.3, SCE alpha 0.5",
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "seq_len": S_TOK, "frames": T_FRAMES,
                       "parallelism": f"dp{world}", "grad_exchange": exchange_kind, "grad_payload": args.payload if ex is not None else None,
                       "executor": "eager" if not (trainer.use_list or trainer.use_graph) else ("list" if trainer.use_list else "graph")},
            "step_tflops": round(fl["step"] / (ms * 1e-3) / 1e12, 1),
            "step_frac_of_peak": round(fl["step"] / (ms * 1e-3) / 1e12 / peak, 4),
            "roofline": {"bound": r_bound, "kernel_tag": dom, "kernel": INFO[dom][3], "kernel_symbol": INFO[dom][2],
                         "selection": "longest single launch of the step among " + " / ".join(CAND) + " in the warm-up probe",
                         "achieved": round(achieved, 1), "peak": r_peak, "unit": r_unit, "frac": round(achieved / r_peak, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         ("flops_per_launch" if r_bound == "mfma" else "bytes_per_launch"): INFO[dom][1], "avg_ms_per_launch": round(kern[dom], 4),
                         "top5": [{"kernel_tag": k, "kernel": INFO[k][2], "bound": roof(k, v)[0], "ms": round(v, 4),
                                   "frac": round(roof(k, v)[1] / roof(k, v)[2], 4)} for k, v in top5],
                         "generator_gemms": {k: {"ms": round(kern[k], 4), "tflops": round(fl["gen"] / (kern[k] * 1e-3) / 1e12, 1),
                                                 "frac": round(fl["gen"] / (kern[k] * 1e-3) / 1e12 / peak, 4)} for k in GEN if k in kern},
                         "all_ms": {k: round(v, 4) for k, v in kern.items()},
                         "all_ms_source": f"{dom}: HIP events inside the timed region; the other brackets: a second, untimed pass "
                                          f"of {extra_steps} steps with every bracket on (all seven cost the step ~55 us)"},
            "north_star": north,
            "north_star_b1024": None if b1024 is None else {
                "what": "the same kernels and schedule at a per-GPU batch of 1024 (untimed side pass, 6 steps): attention + FFN forward",
                "flops": algorithmic_flops(1024)["attn_ffn_fwd"], "ms": round(b1024["layers_fwd"], 4),
                "tflops": round(algorithmic_flops(1024)["attn_ffn_fwd"] / (b1024["layers_fwd"] * 1e-3) / 1e12, 1),
                "frac_of_peak": round(algorithmic_flops(1024)["attn_ffn_fwd"] / (b1024["layers_fwd"] * 1e-3) / 1e12 / peak, 4),
                "step_ms": round(b1024["step"], 4), "samples_per_s": round(1024 / (b1024["step"] * 1e-3), 1)},
            "hbm_kernels": hbm,
            "loss": final_loss,
            # data-parallel exchange evidence (null at N = 1): ranks of the RCCL communicator, what the compute stream waits for it at
            # the end of a step, and how long each gradient bucket occupies the communicator's stream (second, untimed pass)
            "comm": None if ex is None else {
                "rccl_ranks": rccl_ranks, "kind": exchange_kind, "cu_mask": args.comm_cu_mask or None,
                "exposed_wait_ms": round(kern["comm_wait"], 4) if "comm_wait" in kern else None,
                "bucket_ms_on_comm_stream": [round(kern[f"comm_b{i}"], 4) if f"comm_b{i}" in kern else None
                                             for i in range(min(8, len(model.grad_buckets())))],
                "buckets": "generator | decoder layers top-down | token embedding | encoder layers top-down (+ unify)"},
        }
        if world == 1 and args.batch == 256 and not args.no_other_configs:
            out["other_configs"] = other_configs(device, model.compute_dtype, peak)
        if world == 1 and not args.no_decode:
            out["decode"] = decode_line(device, model.compute_dtype)
        if world == 1 and not args.no_exchange_line and not args.force_exchange and args.batch == 256:
            out["exchange_path_n1"] = exchange_path_n1(args.batch, args.dtype)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
