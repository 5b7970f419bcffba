"""Training step / epoch of the caption task and the data-parallel gradient exchange.

Restates reference train.py:113-148 (train_epoch), train.py:20-49 (optimizer / scheduler factory)
and replaces torch.nn.parallel.DistributedDataParallel (train.py:217-219) with an explicit bucketed
all-reduce of the flat gradient buffer over RCCL (torch.distributed 'nccl' on ROCm), launched per
bucket as soon as the backward schedule has enqueued the kernels that complete it, so the exchange
of the generator gradients (a third of the bytes, ready first) overlaps the rest of backward."""
from typing import List, Optional
import os

import torch
import torch.distributed as dist

from . import ops
from .utils import capture_graph


class GradExchange:
    """Gradient averaging across data-parallel ranks (one process per GPU).

    * init: rank 0's parameters (the flat fp32 buffer) are broadcast, like DDP's constructor.
    * per step: for each bucket of MMT4Caption.grad_buckets() an async all-reduce(SUM) is issued on
      the communication stream the backend owns, ordered after the kernels already enqueued on the
      compute stream; `finish()` makes the compute stream wait for them and applies 1/world.
    * payload: fp32 by default; `payload_dtype=torch.bfloat16` halves the xGMI bytes (cast kernels
      from libvct_hip.so on both sides)."""

    def __init__(self, model, group=None, payload_dtype: Optional[torch.dtype] = None, broadcast: bool = True,
                 force: bool = False):
        self.model, self.group = model, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())   # force: run the collectives even alone
        self.buckets = model.grad_buckets()
        self.payload_dtype = payload_dtype
        self._work: List = []
        self._stage = None
        self._avg = dist.is_initialized() and dist.get_backend(group) == "nccl"   # RCCL has ReduceOp.AVG; gloo does not
        if payload_dtype is not None and payload_dtype != torch.float32:
            self._stage = torch.empty(model.flat_grads.numel(), dtype=payload_dtype, device=model.flat_grads.device)
        if broadcast and self.active:
            dist.broadcast(model.flat_params, src=0, group=group)
            model._ps.refresh_shadow(force=True)

    def bucket_ready(self, i: int):
        if not self.active:
            return
        a, b = self.buckets[i]
        if b <= a:
            return
        g = self.model.flat_grads[a:b]
        if self._stage is not None and g.is_cuda:
            s = self._stage[a:b]
            ops.cast(g, s)
            g = s
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if g.is_cuda and not self._avg:
            # gloo carrying CUDA buffers (several ranks sharing one GPU in the tests): its staging copy runs on a pool stream
            # behind an event of the current stream; seen once in a while with 4 ranks time-slicing one GPU: a rank's bucket read
            # before its producers had finished.  The host waits for the stream here (test path only; RCCL is stream-ordered
            # through the library's own event edges and never takes this branch)
            torch.cuda.current_stream().synchronize()
        self._work.append((dist.all_reduce(g, op=op, group=self.group, async_op=True), i))

    def finish(self, on_bucket_done=None):
        """Wait for every bucket's reduction (in issue order).  on_bucket_done(a, b) runs right after bucket
        [a, b) holds the averaged gradient -- the trainer uses it to start Adam on that range while later
        buckets are still on the wire."""
        if not self.active:
            return
        for w, i in self._work:
            w.wait()
            a, b = self.buckets[i]
            if self._stage is not None and self.model.flat_grads.is_cuda:
                ops.cast(self._stage[a:b], self.model.flat_grads[a:b])
            if not self._avg:
                self.model.flat_grads[a:b].mul_(1.0 / self.world)
            if on_bucket_done is not None:
                on_bucket_done(a, b)
        self._work.clear()


class ShardedExchange:
    """Gradient exchange + optimizer of the data-parallel step with the optimizer SHARDED over the ranks (ZeRO-1 style):

        per gradient bucket [a, b), as soon as backward has enqueued its last kernel (n = (b - a) / W, r = own rank):
            reduce-scatter(AVG)  g[a + r n : a + (r+1) n)  <- mean over ranks          (xGMI: (b-a)/W per link)
            Adam                 on that owned shard only                                (1/W of the 1.39 GB optimizer pass)
            all-gather           p[a:b) fp32 masters <- every rank's updated shard       (xGMI: (b-a)/W per link)
            cast                 bf16 shadow of [a:b) from the gathered masters          (local HBM: cheaper than shipping it)
        all of it on the communicator's own stream (coll.owns_stream), behind an event edge to the compute stream, so the
        wire and the optimizer run beside the rest of backward; finish() joins and bumps the Adam step counter.

    Against a plain all-reduce + replicated Adam the wire carries the same bytes (RS + AG = all-reduce) and the
    optimizer's HBM traffic drops by (W-1)/W.  payload_dtype=torch.bfloat16 halves the reduce-scatter bytes (gradients
    are rounded to bf16 before the mean; the all-gather stays fp32 so every rank holds identical masters).
    sharded=False (or a bucket that does not divide by W): all-reduce + Adam on the whole bucket.

    replaces: DistributedDataParallel's reducer + the replicated torch.optim.Adam (reference train.py:24-26, 217-219)."""

    def __init__(self, model, opt: "FusedAdam", coll, sharded: bool = True, payload_dtype: Optional[torch.dtype] = None,
                 broadcast: bool = True):
        self.model, self.opt, self.coll = model, opt, coll
        self.world, self.rank = coll.world, coll.rank
        self.active = True
        self.sharded = sharded
        self.buckets = model.grad_buckets()
        self.payload_dtype = payload_dtype if payload_dtype not in (None, torch.float32) else None
        self._stage = (torch.empty(model.flat_grads.numel(), dtype=self.payload_dtype, device=model.flat_grads.device)
                       if self.payload_dtype is not None else None)
        if broadcast and self.world > 1:
            coll.broadcast(model.flat_params, 0)
            coll.wait()
            model._ps.refresh_shadow(force=True)
        # each rank steps exp_avg / exp_avg_sq on its own shards only: the optimizer's state_dict() (checkpoints) must see
        # the gathered moments, so it calls back here first -- on every rank, it is a collective
        if self.sharded and self.world > 1 and hasattr(opt, "pre_state_dict"):
            opt.pre_state_dict = self.gather_optimizer_state

    def _on_comm(self):
        return torch.cuda.stream(self.coll.stream) if self.coll.owns_stream else _NullCtx()

    def shard_of(self, i: int):
        a, b = self.buckets[i]
        W = self.world
        if not self.sharded or (b - a) % (8 * W) != 0:
            return None
        n = (b - a) // W
        return a + self.rank * n, a + (self.rank + 1) * n, n

    def bucket_ready(self, i: int):
        a, b = self.buckets[i]
        if b <= a:
            return
        m, coll = self.model, self.coll
        g, p = m.flat_grads, m.flat_params
        sh = self.shard_of(i)
        cur = torch.cuda.current_stream() if g.is_cuda else None
        if coll.owns_stream:
            ops.stream_wait(coll.stream, cur)          # the communicator's stream follows everything enqueued so far
        with self._on_comm():
            tag = f"comm_b{i}" if i < 8 else None          # live timing bracket of this bucket on the communicator's stream
            if tag:
                ops.tap(tag, 0)
            src = g
            if self._stage is not None:
                ops.cast(g[a:b], self._stage[a:b])
                src = self._stage
            if sh is None:
                coll.allreduce_avg(src[a:b], after=False)
                if src is not g:
                    ops.cast(src[a:b], g[a:b])
                self.opt.step_range(a, b)
                if tag:
                    ops.tap(tag, 1)
                return
            lo, hi, n = sh
            coll.reduce_scatter_avg(src[a:b], n, after=False)
            if src is not g:
                ops.cast(src[lo:hi], g[lo:hi])
            self.opt.step_range(lo, hi)
            coll.all_gather(p[a:b], n, after=False)
            m._ps.cast_range(a, b)
            if tag:
                ops.tap(tag, 1)

    def finish(self):
        ops.tap("comm_wait", 0)                  # compute stream: from "backward enqueued" to "the communicator's stream has drained"
        self.coll.wait()
        ops.tap("comm_wait", 1)
        self.opt.finish_ranges()

    def gather_optimizer_state(self):
        """Every rank's Adam moments are only current on its own shards: all-gather them (before a checkpoint)."""
        for i, (a, b) in enumerate(self.buckets):
            sh = self.shard_of(i)
            if sh is None or b <= a:
                continue
            for t in (self.opt.exp_avg, self.opt.exp_avg_sq):
                self.coll.all_gather(t[a:b], sh[2])
        self.coll.wait()


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam / AdamW semantics (reference train.py:24-31) as ONE kernel over the model's flat
    fp32 parameter buffer, which also rewrites the bf16 shadow the GEMMs read.  `param_groups[0]['lr']`
    is honoured every step, so torch LR schedulers work unchanged."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.model = model
        flat = torch.nn.Parameter(model.flat_params, requires_grad=True)
        flat.grad = model.flat_grads
        super().__init__([flat], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        ps = model._ps
        self.exp_avg = torch.zeros_like(ps.flat)
        self.exp_avg_sq = torch.zeros_like(ps.flat)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=ps.flat.device)
        emb = "cap_decoder.tgt_to_emb.weight"
        a = ps.offsets[emb]
        self.skip = (a, a + (ps.params[emb].numel() + ps.ALIGN - 1) // ps.ALIGN * ps.ALIGN)
        # the reference builds its optimizer over filter(requires_grad) (train.py:24): parameters outside the caption
        # path (matching.*: frozen by mode('caption'), never given a gradient here) are neither stepped nor decayed
        self.end = model.caption_param_end
        # lr / betas / eps / weight decay live in DEVICE memory (read by the kernel): a captured hipGraph or a recorded
        # launch list follows LR schedulers and load_state_dict instead of freezing the values of the recording step
        self.hyper = torch.zeros(8, dtype=torch.float32, device=ps.flat.device)
        self._hyper_host = None
        self.sync_hyper()
        self.pre_state_dict = None      # set by ShardedExchange: all-gather the moments before they are read (collective)
        # weight matrices stepped INSIDE their weight-gradient GEMMs (enable_dw_fusion): flat ranges registered by desc_for() while the
        # backward of the current step is being enqueued; step_range() then covers only what is left
        self.dw_fusion = False
        self._dw_ranges = []
        self.range_elems = {}
        self._dw_desc = {}
        self._range_tables = {}

    def _hyper_now(self):
        g = self.param_groups[0]
        return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))

    def sync_hyper(self):
        """Upload the hyper-parameters if a scheduler / user changed them (host check, rare H2D copy).  Called by
        CaptionTrainer.step before every step, outside any capture."""
        h = self._hyper_now()
        if h != self._hyper_host:
            self.hyper[:5].copy_(torch.tensor(h, dtype=torch.float32))
            self._hyper_host = h

    # ---- the optimizer inside the weight-gradient GEMMs (single GPU) ---------------------------------------------------------------
    # A/B switch.  The reference steps every parameter after backward (train.py:125-126); on one GPU nothing sits between a weight's
    # gradient and its update, so the update of every 2-D weight runs in the epilogue of the GEMM that produces the gradient
    # (include/vct_hip.h, vct_gemm_adam): the gradient never goes to HBM and the optimizer's 28 B per parameter move inside MFMA-bound
    # kernels instead of forming a 0.3 ms HBM-bound tail of the step.  What is left (biases, LayerNorm parameters, the embedding
    # table) takes ONE multi-range launch per step_range() call.
    fuse_dw_default = os.environ.get("VCT_FUSE_ADAM", "1") != "0"
    keep_grads = os.environ.get("VCT_FUSE_ADAM_KEEP_GRAD", "0") == "1"      # also store the weight gradients (hooks / inspection)

    def enable_dw_fusion(self, on: bool = True):
        """Called by CaptionTrainer when it owns the whole step (no gradient exchange).  bf16 compute mode on a GPU only."""
        ps = self.model._ps
        ok = bool(on) and ps.compute_dtype == torch.bfloat16 and ps.flat.is_cuda
        self.dw_fusion = ok
        ps.dw_adam = self if ok else None
        self._dw_ranges = []
        return ok

    def set_keep_grads(self, on: bool):
        """Store the weight gradients from the optimizer epilogues as well (CaptionTrainer(keep_weight_grads=...))."""
        if bool(on) != bool(self.keep_grads):
            self.keep_grads = bool(on)
            self._dw_desc.clear()            # the cached epilogue descriptors carry the flag

    def begin_step(self):
        """Forget the matrices registered by the previous enqueue of a step (the set is rebuilt as the backward is enqueued)."""
        self._dw_ranges = []

    def desc_for(self, dw: torch.Tensor):
        """ops.L.GemmAdam for the weight whose gradient view `dw` (fp32 [rows, K], rows of one parameter) a GEMM is about to
        produce, and note that this step's step_range() calls must leave its flat range alone.  None: not steppable there."""
        ps = self.model._ps
        off = (dw.data_ptr() - ps.gflat.data_ptr()) // 4
        if not self.dw_fusion or dw.dim() != 2 or dw.dtype != torch.float32 or not (0 <= off < self.end):
            return None
        name, base = ps.name_at(off)
        shape = ps.params[name].shape
        rows, K = dw.shape
        if len(shape) != 2 or K != shape[1] or dw.stride(0) != K or dw.stride(1) != 1 or (off - base) % K or name in ps.no_shadow:
            return None
        # the epilogue's own preconditions (vct_gemm's check_desc): 16-byte vectors of gradient / parameter / moments, 8-byte vectors of
        # the shadow.  A matrix that fails them keeps its separate optimizer pass (its range is NOT registered) instead of aborting.
        if K % 4 or off % 4 or (ps.flat.data_ptr() | self.exp_avg.data_ptr() | self.exp_avg_sq.data_ptr() | ps.gflat.data_ptr()) & 15 \
                or ps.cflat.data_ptr() & 7:
            return None
        key = (off, rows, tuple(sorted(ps.packed)))
        ad = self._dw_desc.get(key)
        if ad is None:
            ad = ops.L.GemmAdam()
            ad.param, ad.exp_avg, ad.exp_avg_sq = (t.data_ptr() + 4 * off for t in (ps.flat, self.exp_avg, self.exp_avg_sq))
            ad.shadow, ad.ld_shadow = ps.cflat.data_ptr() + 2 * off, K
            seg = ps.pack_seg(name)
            if seg is not None:
                ad.pk_K, ad.pk_mode, ad.pk_stream, ad.pk_row0 = seg[0], seg[1], seg[3], (off - base) // K
                for i in range(4):
                    ad.pk_chunk0[i] = seg[2][i]
            ad.hyper, ad.step = self.hyper.data_ptr(), self.step_dev.data_ptr()
            ad.store_grad = int(self.keep_grads)
            self._dw_desc[key] = ad
        self._dw_ranges.append((off, off + rows * K))
        return ad

    def _left_ranges(self, a: int, b: int):
        """[(begin, end, has_shadow)] of [a, b) minus the matrices registered for this step, split at the shadow-less tensors."""
        cuts = sorted(set((max(x, a), min(y, b)) for x, y in self._dw_ranges if y > a and x < b))
        out, cur = [], a
        for x, y in cuts:
            if x > cur:
                out.append((cur, x))
            cur = max(cur, y)
        if cur < b:
            out.append((cur, b))
        res = []
        s0, s1 = self.skip
        for x, y in out:                      # the embedding table has no bf16 shadow
            for lo, hi, sh in ((x, min(y, s0), True), (max(x, s0), min(y, s1), False), (max(x, s1), y, True)):
                if hi > lo:
                    res.append((lo, hi, sh))
        return res

    @torch.no_grad()
    def step(self, closure=None):
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        self.step_range(0, self.end)
        self.finish_ranges()

    # A/B switch: the optimizer's pass writes the stream-order packed weight copies itself (instead of vct_ss_pack launches behind it)
    pack_in_adam = os.environ.get("VCT_ADAM_PACK", "1") != "0"

    @torch.no_grad()
    def step_range(self, a: int, b: int):
        """Adam on flat elements [a, b) only, without advancing the step counter (range-by-range stepping as
        gradient buckets complete); call finish_ranges() after the last range of the step."""
        b = min(b, self.end)
        if b <= a:
            return
        lr, b1, b2, eps, wd = self._hyper_now()
        ps = self.model._ps
        bf = ps.compute_dtype != torch.float32
        if self.dw_fusion and self._dw_ranges:
            # the matrices of [a, b) were stepped by their weight-gradient GEMMs (which also wrote their shadows and packed copies): one
            # launch over what is left -- vectors, the embedding table, any matrix whose GEMM did not take the epilogue
            left = self._left_ranges(a, b)
            self.range_elems[(a, b)] = sum(r[1] - r[0] for r in left)        # what this call's launch touches (bench.py: bytes of the bracket)
            table, nseg, pk_parts = ps.adam_pack_table(a, b) if (self.pack_in_adam and ps.packed) else (None, 0, [])
            if left:
                key = (tuple(left), )
                tab = self._range_tables.get(key)
                if tab is None:
                    tab = self._range_tables[key] = ops.adam_ranges_table(left, ps.flat.device)
                ops.adam_step_ranges(ps.flat, ps.gflat, self.exp_avg, self.exp_avg_sq, ps.cflat, tab, lr, b1, b2, eps, wd, self.step_dev,
                                     hyper=self.hyper, pack=(table, nseg) if nseg else None)
            ps.refresh_transposed(a, b, packed_done=pk_parts)
            return
        # 2-D weights with an eager transposed shadow inside the range (W_g^T): their own pass writes the transposed copy too
        fused = sorted(ps.eager_transposed_in(a, b), key=lambda x: x[2]) if bf else []
        fused = [f for f in fused if ps.params[f[0]].shape[1] % 64 == 0 and not (f[2] < self.skip[1] and f[3] > self.skip[0])]
        if os.environ.get("VCT_ADAM2D", "1") == "0":      # A/B switch: transposed shadow by a transpose launch behind the flat pass
            fused = []

        # stream-order packed weight copies (the sample-stationary stack kernels' operand) inside the range: written by the same pass
        table, nseg, pk_parts = ps.adam_pack_table(a, b) if (bf and self.pack_in_adam and ps.packed) else (None, 0, [])

        def flat_range(lo, hi):
            if hi <= lo:
                return
            shadow = ps.cflat[lo:hi] if bf else None
            s0, s1 = max(self.skip[0], lo) - lo, min(self.skip[1], hi) - lo
            ops.adam_step(ps.flat[lo:hi], ps.gflat[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], shadow, lr, b1, b2, eps, wd,
                          self.step_dev, (s0, s1) if s1 > s0 else (0, 0), bump=False, hyper=self.hyper,
                          pack=(table, nseg, lo) if nseg else None)
        cur = a
        for name, t, x, y in fused:
            flat_range(cur, x)
            shape = ps.params[name].shape
            ops.adam_step_2d(ps.flat[x:y].view(shape), ps.gflat[x:y].view(shape), self.exp_avg[x:y].view(shape),
                             self.exp_avg_sq[x:y].view(shape), ps.cflat[x:y].view(shape), t, lr, b1, b2, eps, wd, self.step_dev,
                             hyper=self.hyper)
            cur = y
        flat_range(cur, b)
        if bf:
            ps.refresh_transposed(a, b, skip=[f[0] for f in fused], packed_done=pk_parts)   # other eager copies follow the shadow this pass rewrote

    @torch.no_grad()
    def finish_ranges(self):
        ops.adam_bump(self.step_dev)
        ps = self.model._ps
        ps.version += 1
        ps._stamp = sum(p._version for p in ps.params.values())

    def zero_grad(self, set_to_none: bool = True):
        pass   # the backward schedule OVERWRITES every gradient (no accumulation across backward calls on the fast path)

    # ---- checkpointing (checkpoint.save_training_state): moments and step live outside torch's per-param state ----
    def gather_state(self):
        """Data-parallel runs with a sharded optimizer: bring every rank's Adam moments up to date on this rank (a COLLECTIVE --
        every rank calls it; checkpoint.save_training_state does).  No-op otherwise."""
        if self.pre_state_dict is not None:
            self.pre_state_dict()
        self._gathered_at = int(self.step_dev.item())

    def state_dict(self):
        if self.pre_state_dict is not None and getattr(self, "_gathered_at", None) != int(self.step_dev.item()):
            # the moments of the shards other ranks own are stale here: gathering them is a collective and must not hide in a call
            # that a single rank may make (`if rank == 0: save(...)` would hang in it)
            raise RuntimeError("FusedAdam.state_dict() under a sharded exchange: call optimizer.gather_state() on EVERY rank first "
                               "(checkpoint.save_training_state does), then state_dict() on the rank(s) that write")
        sd = super().state_dict()
        ps = self.model._ps
        sd["vct_fused_adam"] = {"exp_avg": self.exp_avg.detach().clone(), "exp_avg_sq": self.exp_avg_sq.detach().clone(),
                                "step": int(self.step_dev.item()), "numel": ps.flat.numel(),
                                "layout": [(k, int(ps.offsets[k])) for k in ps.params]}
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        fused = sd.pop("vct_fused_adam", None)
        if fused is None:
            raise ValueError("not a FusedAdam state (no 'vct_fused_adam' entry)")
        ps = self.model._ps
        if fused["numel"] != ps.flat.numel() or [tuple(x) for x in fused["layout"]] != [(k, int(ps.offsets[k])) for k in ps.params]:
            raise ValueError("optimizer state was saved for a different parameter layout")
        super().load_state_dict(sd)
        self.exp_avg.copy_(fused["exp_avg"])
        self.exp_avg_sq.copy_(fused["exp_avg_sq"])
        self.step_dev.fill_(fused["step"])
        self._gathered_at = None      # (a restored step count says nothing about the other ranks' shards: gather_state() again before a save)
        self.sync_hyper()


def build_optimizer(train_cfg: dict, model):
    """Optimizer + scheduler factory with the reference's config surface (train.py:20-49).  The
    optimizer sees ONE parameter -- the flat fp32 buffer, whose .grad is the flat gradient buffer --
    so Adam is a single fused multi-tensor kernel instead of ~70 small ones."""
    oc = train_cfg["optimizer"]
    flat = torch.nn.Parameter(model.flat_params, requires_grad=True)
    flat.grad = model.flat_grads
    if oc["name"] == "adam":
        if flat.is_cuda:
            opt = FusedAdam(model, lr=oc["learning_rate"], betas=tuple(oc["beta"]), weight_decay=oc.get("weight_decay", 0) or 0.0)
        elif oc.get("weight_decay", 0) == 0:
            opt = torch.optim.Adam([flat], lr=oc["learning_rate"], betas=tuple(oc["beta"]))
        else:
            opt = torch.optim.AdamW([flat], lr=oc["learning_rate"], betas=tuple(oc["beta"]), weight_decay=oc["weight_decay"])
    elif oc["name"] == "sgd":
        opt = torch.optim.SGD([flat], lr=oc["learning_rate"], momentum=oc["momentum"])
    else:
        raise ValueError("Do not support optimizer: {}".format(oc["name"]))
    sched = None
    sc = oc.get("lr_scheduler")
    if sc:
        if sc["name"] == "CosineAnnealingLR":
            sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=sc["T_max"], eta_min=sc["eta_min"])
        elif sc["name"] == "ReduceLROnPlateau":
            sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=sc["patience"])
        else:
            raise ValueError("Do not support lr_scheduler: {}".format(sc["name"]))
    return opt, sched


class CaptionTrainer:
    """One object = the reference's `model(...) -> zero_grad -> backward -> step` loop body
    (train.py:123-126) on the kernel fast path, with the gradient exchange folded into backward.

    Executors of the ~100-launch step (single GPU, FusedAdam):
      * eager (default off the GPU fast path): Python issues every launch through ctypes;
      * launch_list=True: the step is recorded ONCE per input shape into a C-side launch list (ops.LaunchList:
        every launch with its stream, every cross-stream edge) and re-issued by one C call per step -- eager
        two-stream semantics without the Python/ctypes cost per launch;
      * use_graph=True: the step is captured into a hipGraph (bitwise equal, but replay serialises the two streams).
    Inputs are copied into static buffers; the dropout seed, the Adam step counter and the Adam hyper-parameters live
    in device memory, so every replay sees fresh values.  Recordings are dropped when an activation buffer of THIS model had
    to grow (engine.StepContext.generation), because they bake device pointers."""

    def __init__(self, model, optimizer, exchange: Optional[GradExchange] = None, use_graph: bool = False,
                 launch_list: Optional[bool] = None, keep_weight_grads: Optional[bool] = None):
        """keep_weight_grads: on the single-GPU bf16 FusedAdam path the 2-D weights are stepped inside their weight-gradient GEMMs and
        their gradients are NOT stored (model.grads_valid is False after a step; the reference leaves valid .grad after backward,
        train.py:125).  True stores them as well (for clipping, logging, hooks; costs the 4 B per parameter the fusion saved);
        None: the VCT_FUSE_ADAM_KEEP_GRAD environment switch (default off)."""
        self.model, self.opt, self.ex = model, optimizer, exchange
        if keep_weight_grads is not None and isinstance(optimizer, FusedAdam):
            optimizer.set_keep_grads(bool(keep_weight_grads))
        model._unit_loss_grad = True
        single = (exchange is None or not exchange.active) and isinstance(optimizer, FusedAdam)
        # a launch list can also carry the exchange when every collective is recordable: stream work of the library's own RCCL
        # communicator, or host commands of the list (comm.C10dColl under ops.host_call: the one-GPU multi-rank tests)
        listable = single or (isinstance(exchange, ShardedExchange) and getattr(exchange.coll, "recordable", False))
        self.use_graph = bool(use_graph) and single
        self.use_list = (bool(launch_list) if launch_list is not None else False) and listable and not self.use_graph
        self._graphs = {}
        self._lists = {}
        self._gen = None
        if single and model.flat_grads.is_cuda:
            # this trainer (zero_grad implicit, no exchange, no in-place averaging) is the only writer of the gradient buffer
            model.cap_decoder._engine().exclusive_grads = True
        # single GPU: per-bucket Adam on the side stream was measured SLOWER (3.28 vs 3.14 ms/step: the 6.5 TB/s
        # optimizer pass steals HBM bandwidth from the GEMMs it overlaps), so it is opt-in; with a gradient exchange
        # Adam always runs per bucket as each all-reduce lands (it overlaps the wire, not the GEMMs)
        self.overlap_adam = False
        # single GPU, FusedAdam, bf16: every weight matrix is stepped in the epilogue of its own weight-gradient GEMM (FusedAdam.
        # enable_dw_fusion); the hook is installed only WHILE this trainer enqueues a step (a plain loss.backward() outside it must
        # keep producing gradients and nothing else)
        self.fuse_adam = bool(single and isinstance(optimizer, FusedAdam) and optimizer.fuse_dw_default
                              and model._ps.compute_dtype == torch.bfloat16 and model.flat_grads.is_cuda)

    # A/B switch (single GPU): the whole Adam pass after the joined backward instead of 86 % of it beside the encoder backward
    adam_after_backward = os.environ.get("VCT_ADAM_TAIL", "0") == "1"
    warm_streams = os.environ.get("VCT_WARM_STREAMS", "1") != "0"

    def _step_kernels(self, feats, mask, ids):
        if not self.fuse_adam:
            return self._step_kernels_body(feats, mask, ids)
        self.opt.enable_dw_fusion(True)
        self.opt.begin_step()
        try:
            return self._step_kernels_body(feats, mask, ids)
        finally:
            self.opt.enable_dw_fusion(False)

    def _step_kernels_body(self, feats, mask, ids):
        m = self.model
        fused = isinstance(self.opt, FusedAdam)
        ops.tap("step", 0)
        if not fused:
            m._ps.refresh_shadow(force=True)      # a torch optimizer wrote the fp32 masters: re-cast the shadow
            m._ps._stamp = sum(p._version for p in m._ps.params.values())
        else:
            m._ps.refresh_shadow()                # FusedAdam keeps the shadow current (first step: cast once)
        exchanging = self.ex is not None and self.ex.active
        if exchanging and isinstance(self.ex, ShardedExchange):
            # reduce-scatter -> Adam on the owned shard -> all-gather, per bucket, on the communicator's stream
            loss = m.train_step_kernels(feats, mask, ids, bucket_ready=self.ex.bucket_ready)
            self.ex.finish()
        elif exchanging:
            loss = m.train_step_kernels(feats, mask, ids, bucket_ready=self.ex.bucket_ready)
            if fused:                             # Adam per bucket as its averaged gradient lands
                self.ex.finish(on_bucket_done=self.opt.step_range)
                self.opt.finish_ranges()
            else:
                self.ex.finish()
                self.opt.step()
        elif fused and self.overlap_adam and feats.is_cuda:
            # single GPU: Adam on each gradient bucket the moment backward completes it, on the side stream, so the
            # 1.4 GB optimizer pass hides under the rest of backward instead of trailing it
            buckets = m.grad_buckets()
            ctx = m._ps.ctx

            def hook(i):
                side = ctx.side
                if side is None:
                    self.opt.step_range(*buckets[i])
                    return
                ops.stream_wait(side, None)
                with torch.cuda.stream(side):
                    self.opt.step_range(*buckets[i])
            loss = m.train_step_kernels(feats, mask, ids, bucket_ready=hook)   # zero_grad is implicit: grads are overwritten
            if ctx.side is not None:
                ops.stream_wait(None, ctx.side)
            self.opt.finish_ranges()
        elif fused and feats.is_cuda and m.overlap_enc_bwd and not self.adam_after_backward:
            # the encoder backward is still running on the side stream when the decoder's tail is done: Adam on everything
            # but the encoder (86 % of the parameters at cfg-B) fills that gap on the main stream, the rest follows the join
            loss = m.train_step_kernels(feats, mask, ids, defer_join=True)
            a = m.encoder_param_begin
            # the decoder layers' weight gradients and the d(memory) GEMMs were issued on the SIDE stream: Adam reads
            # those gradients and rewrites the weights those kernels read, so the main stream joins the side stream first
            # (it is idle here: the encoder backward has not been enqueued yet)
            m.cap_decoder._engine().join_side()
            if m.encoder_backward_is_one_launch():
                # the sample-stationary backward takes whole compute units: alone on the main stream, ahead of the optimizer's pass; its
                # weight-gradient GEMMs (side stream) then run beside that pass
                m.launch_encoder_backward(main=True)
            ops.tap("adam", 0)
            self.opt.step_range(0, a)            # enqueued BEFORE the encoder backward: one launch vs ~35
            ops.tap("adam", 1)
            m.launch_encoder_backward()
            m.join_backward()
            self.opt.step_range(a, m._ps.total)
            self.opt.finish_ranges()
        else:
            loss = m.train_step_kernels(feats, mask, ids)
            self.opt.step()
        if self.warm_streams and feats.is_cuda:
            # the next step begins with the two sample-stationary stack launches, which stream these packed weights chunk by chunk with
            # two chunks of prefetch: behind the optimizer's passes (1.4 GB of traffic through the memory-side cache) every chunk is an
            # HBM miss
            for ent in m._ps.packed.values():
                ops.warm(ent[0])
        if m.training and m.video_encoder.cfg["dropout"] > 0:
            ops.advance_seed(m._seed)
        ops.tap("step", 1)
        return loss

    def _fresh_shadow(self):
        """Replays skip _step_kernels, which is where the bf16 shadow / transposed copies follow the fp32 masters: if the
        masters were written outside FusedAdam since the last step (load_state_dict, load_weights, restoring the best
        checkpoint), re-cast them eagerly on the current stream before the replay (host-only version-stamp check otherwise)."""
        self.model._ps.refresh_shadow()

    def drop_recordings(self):
        """Forget every recorded launch list / captured graph (they are re-made on the next step of each shape): needed after
        anything that changes WHAT a step launches, e.g. ops.taps_enable(...)."""
        self._lists.clear()
        self._graphs.clear()

    def _static_inputs(self, key, feats, mask, ids):
        st = getattr(self, "_static", None)
        if st is None:
            st = self._static = {}
        s = st.get(key)
        if s is None:
            s = st[key] = (feats.clone(), None if mask is None else mask.clone(), ids.clone())
        else:
            # a caller that already works in the static buffers (adopt_inputs) skips the staging copies: three small device copies
            # and the launch gaps around them are ~35 us at the head of a 2.4 ms step
            if s[0].data_ptr() != feats.data_ptr():
                s[0].copy_(feats, non_blocking=True)
            if mask is not None and s[1].data_ptr() != mask.data_ptr():
                s[1].copy_(mask, non_blocking=True)
            if s[2].data_ptr() != ids.data_ptr():
                s[2].copy_(ids, non_blocking=True)
        return s

    def _input_key(self, feats, mask, ids):
        return (tuple(feats.shape), feats.dtype, None if mask is None else tuple(mask.shape), tuple(ids.shape), self.model.training)

    def adopt_inputs(self, feats: torch.Tensor, mask: Optional[torch.Tensor], ids: torch.Tensor):
        """The trainer's own static input buffers for this shape, initialised with the given batch.  Recorded launch lists and
        captured graphs read their inputs from these buffers, so step() normally copies every batch into them; a producer that
        writes its batches INTO them (a device-side loader, a benchmark with resident data) and passes them to step() has no copy
        left.  Returns (feats, mask, ids) views of the buffers; without a recording executor the inputs are returned unchanged."""
        if not (self.use_graph or self.use_list):
            return feats, mask, ids
        return self._static_inputs(self._input_key(feats, mask, ids), feats, mask, ids)

    def step(self, feats: torch.Tensor, mask: Optional[torch.Tensor], ids: torch.Tensor) -> torch.Tensor:
        """Returns this rank's loss as a device tensor [1] (no host sync)."""
        if isinstance(self.opt, FusedAdam):
            self.opt.sync_hyper()
        if not (self.use_graph or self.use_list):
            return self._step_kernels(feats, mask, ids)
        ctx = self.model._ps.ctx                  # this model's buffers / side stream (engine.StepContext)
        if self._gen != ctx.generation:           # a buffer grew since the recordings were made: they bake stale pointers
            self._graphs.clear()
            self._lists.clear()
            self._gen = ctx.generation
        key = self._input_key(feats, mask, ids)
        static = self._static_inputs(key, feats, mask, ids)
        if self.use_list:
            ll = self._lists.get(key)
            if ll is None:
                # first step of this shape: run it eagerly on the static copies (allocates every buffer), then record the
                # same schedule (recording executes nothing); later calls replay the recording
                eager_loss = self._step_kernels(*static).clone()
                if self._gen != ctx.generation:       # the eager step allocated: older recordings are stale, this one is not made yet
                    self._graphs.clear()
                    self._lists.clear()
                    self._gen = ctx.generation
                ll = ops.LaunchList()
                with ll.record():
                    loss = self._step_kernels(*static)
                self._lists[key] = (ll, loss)
                return eager_loss
            self._fresh_shadow()
            ll[0].replay()
            self.model._ps.version += 1          # the replayed optimizer rewrote the shadow (lazy transposed copies key on this)
            return ll[1]
        g = self._graphs.get(key)
        if g is None:
            eager_loss = self._step_kernels(*static).clone()
            if self._gen != ctx.generation:
                self._graphs.clear()
                self._lists.clear()
                self._gen = ctx.generation
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                with capture_graph(graph):
                    loss = self._step_kernels(*static)
            except Exception:                       # capture is an optimisation, never a requirement
                self.use_graph = False
                return eager_loss
            self._graphs[key] = (graph, loss)
            return eager_loss
        graph, loss = g
        self._fresh_shadow()
        graph.replay()
        self.model._ps.version += 1
        return loss


def train_epoch(model, optimizer, dataloader, mode: str = "caption", exchange: Optional[GradExchange] = None,
                log_every: int = 0):
    """reference train.py:113-148 for mode != 'cross'.  `dataloader` yields (v_feats, v_masks, captions, vids)
    with the reference's layouts (lists of tensors; captions = id rows or strings).  Returns the epoch-mean of
    the all-rank mean loss -- one device->host sync per epoch instead of one per step."""
    if mode != "caption":
        raise NotImplementedError("only the caption task is on the accelerated path")
    model.train()
    model.mode(mode)
    trainer = CaptionTrainer(model, optimizer, exchange)
    dev = model.flat_params.device
    total = torch.zeros(1, device=dev)
    n = 0
    for v_feats, v_masks, captions, _vids in dataloader:
        feats = v_feats[0].to(dev, non_blocking=True)
        mask = v_masks[0].to(dev, non_blocking=True) if v_masks is not None else None
        ids, _ = model.cap_preprocessor(captions)
        total += trainer.step(feats, mask, ids)
        n += 1
    if exchange is not None and exchange.world > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=exchange.group)
        total /= exchange.world
    return float(total) / max(n, 1)
