"""Forward / backward schedules of the caption path over the gfx950 kernels.

The reference runs this path as ~60 stock torch.nn calls under autograd (model/MMEncoder.py:244-276,
model/CapDecoder.py:34-60, torch nn/modules/transformer.py:951-982,1143-1199).  Here the graph is
static and known, so forward and backward are explicit kernel schedules over pre-allocated HBM
buffers: no autograd tape, no temporaries, no host synchronisation -- the whole step is
hipGraph-capturable.  Every arithmetic step is a libvct_hip.so kernel (ops.py); torch is used for
memory, streams and a few boolean mask preparations only.

Data layout in HBM (row-major, tokens x features):
  encoder tokens   Me = B*(T+1) rows, decoder tokens Md = B*(S-1) rows, d columns
  packed projections qkv [M,3d], cross kv [Me,2d]; FFN hidden [M,ff]; logits [Md, Vp] with
  Vp = V rounded up to 32 and zero-padded columns; statistics (mean, rstd) fp32 [M].
  Activations are bf16 (throughput mode) or fp32 (parity mode); parameters stay fp32 masters with a
  bf16 shadow refreshed once per step; every parameter gradient is fp32.
"""
from typing import Dict, List, Optional

import os

import torch

from . import ops

ENC_SITE, DEC_SITE, EMB_SITE = 0, 1000, 999
DMEM_SYNC = 0      # named sync point (ops.sync_record / sync_wait): d(memory) is final on the main stream


_CUS = {}


def _cu_count(dev) -> int:
    n = _CUS.get(dev)
    if n is None:
        n = _CUS[dev] = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == "cuda" else 0
    return n


class StepContext:
    """Per-model execution state shared by the engines of ONE parameter set: the side HIP stream (+ its split-K scratch) that
    weight-gradient GEMMs / the encoder backward / the decoder prefix run on, and the buffer generation counter that
    invalidates baked pointer sets (launch lists, hipGraphs, pointer tables).  It used to be process-global: two models in one
    process (a training model and an evaluation copy) then shared a side stream and dropped each other's recordings."""

    def __init__(self):
        self.side = None
        self.side_ws = None
        self.generation = 0


class ParamSet:
    """fp32 master parameters (views of one flat buffer), their fp32 gradient views and the
    compute-dtype shadow used by the GEMMs.  Order = gradient-ready order of the backward pass so
    that contiguous slices of the flat gradient buffer are the all-reduce buckets."""

    ALIGN = 64  # elements; keeps every view 16-byte aligned in fp32 and bf16

    def __init__(self, named: List, device, compute_dtype: torch.dtype, no_shadow=()):
        self.names = [n for n, _ in named]
        self.params = {n: p for n, p in named}
        self.device, self.compute_dtype = device, compute_dtype
        self.ctx = StepContext()
        self.offsets, off = {}, 0
        for n, p in named:
            self.offsets[n] = off
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.gflat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.no_shadow = set(no_shadow)
        for n, p in named:
            o, k = self.offsets[n], p.numel()
            view = self.flat[o:o + k].view(p.shape)
            view.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = view
        self.g = {n: self.gflat[self.offsets[n]:self.offsets[n] + p.numel()].view(p.shape) for n, p in named}
        if compute_dtype == torch.float32:
            self.cflat = self.flat
            self.c = {n: p.data for n, p in named}
        else:
            self.cflat = torch.zeros(self.total, dtype=compute_dtype, device=device)
            self.c = {n: self.cflat[self.offsets[n]:self.offsets[n] + p.numel()].view(p.shape) for n, p in named}
        self._stamp = None
        self.version = 0        # bumped whenever the shadow is (or may have been) rewritten: lazy transposed copies key on it
        # transposed copies of individual weight matrices in the compute dtype (name -> (tensor [cols, ld], start, end)): kept in
        # step with the shadow by refresh_shadow / cast_range / FusedAdam.step_range (refresh_transposed)
        self.transposed = {}
        # STREAM-ORDER packed copies of whole stacks (key -> [tensor, chunks per layer, [[start, end, blocks, dirty] per layer]]): the operand of the
        # sample-stationary layer kernels (ops.layer_ss_fwd), kept in step with the shadow exactly like the eager transposed copies
        self.packed = {}
        self._adam_pack = {}
        # the optimizer that steps weight matrices INSIDE their weight-gradient GEMMs (trainer.FusedAdam.enable_dw_fusion; None: nobody
        # does): _StackBase.dw_gemm asks it for the epilogue descriptor of every gradient view it is about to produce
        self.dw_adam = None
        # False after a step whose weight-gradient GEMMs consumed their gradients in registers (optimizer epilogue, store_grad = 0): the
        # 2-D weights' regions of gflat (and their .grad views) then hold values of an EARLIER backward.  MMT4Caption.grads_valid.
        self.weight_grads_valid = True
        self._starts = None
        # contiguous [start, end) ranges to cast (everything except the no_shadow tensors)
        self.cast_ranges, start = [], 0
        for n in self.names:
            if n in self.no_shadow:
                if self.offsets[n] > start:
                    self.cast_ranges.append((start, self.offsets[n]))
                start = self.offsets[n] + (self.params[n].numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        if start < self.total:
            self.cast_ranges.append((start, self.total))

    def intact(self) -> bool:
        """True while every nn.Parameter still aliases the flat buffer (Module.to() would break it)."""
        base = self.flat.data_ptr()
        for n in (self.names[0], self.names[-1]):
            if self.params[n].data_ptr() != base + 4 * self.offsets[n]:
                return False
        return True

    def refresh_shadow(self, force=False):
        """bf16 shadow <- fp32 masters (one cast kernel per contiguous range)."""
        if self.compute_dtype == torch.float32:
            return
        stamp = None
        if not force:
            stamp = sum(p._version for p in self.params.values())
            if stamp == self._stamp:
                return
        for a, b in self.cast_ranges:
            ops.cast(self.flat[a:b], self.cflat[a:b])
        self.version += 1
        self.refresh_transposed(0, self.total)
        self._stamp = stamp if stamp is not None else sum(p._version for p in self.params.values())

    def cast_range(self, a: int, b: int):
        """bf16 shadow <- fp32 masters for flat elements [a, b) (minus the tensors that have no shadow)."""
        if self.compute_dtype == torch.float32:
            return
        for x, y in self.cast_ranges:
            lo, hi = max(a, x), min(b, y)
            if hi > lo:
                ops.cast(self.flat[lo:hi], self.cflat[lo:hi])
        self.version += 1
        self.refresh_transposed(a, b)

    def want_transposed(self, name: str, eager: bool = False) -> torch.Tensor:
        """A [cols, ld >= rows] transposed copy (compute dtype) of the 2-D weight `name`, created on first use.
        eager=True (a training-step operand: W_g^T of the NT-form vocabulary dX): rewritten whenever the shadow of that weight
        is -- by the optimizer's own pass when it is FusedAdam (vct_adam_step_2d), else by a transpose launch behind it.
        eager=False (decode-time operands): refreshed HERE, on demand, when the shadow has changed since the copy was made
        (`version` counts shadow rewrites): a training loop that validates between epochs pays nothing per step for them."""
        ent = self.transposed.get(name)
        if ent is None:
            w = self.c[name]
            rows, cols = w.shape
            t = torch.zeros(cols, (rows + 31) // 32 * 32, dtype=w.dtype, device=w.device)
            a = self.offsets[name]
            ent = self.transposed[name] = [t, a, a + w.numel(), bool(eager), -1]
        was_lazy = not ent[3]
        if eager:
            ent[3] = True
        if ent[4] != self.version:
            if ent[4] < 0 or was_lazy:                # new, or a (so far) lazy copy that is out of date
                ops.transpose(self.c[name], ent[0])
            ent[4] = self.version
        return ent[0]

    def want_packed(self, key: str, layers):
        """(stream, [first chunk of every part]): the weights of a whole Transformer STACK packed, part after part, in the order the
        sample-stationary kernel consumes them (include/vct_hip.h, vct_ss_pack).  layers = [(names, blocks_fn)] per part (a layer, or
        the unify weight in front of an encoder stack), blocks_fn() -> [(2-D shadow view, nchunks, first chunk inside the part)].
        Created on first use; every part is rewritten whenever the shadow of its weights is (refresh_transposed: behind the
        optimizer's pass, inside the recorded step)."""
        ent = self.packed.get(key)
        if ent is None:
            parts, at, firsts = [], 0, []
            for names, blocks_fn in layers:
                blocks = blocks_fn()                        # (w, nchunks, first chunk[, transposed])
                n = max(blk[2] + blk[1] for blk in blocks)
                parts.append((names, [(blk[0], blk[1], blk[2] + at) + tuple(blk[3:]) for blk in blocks]))
                firsts.append(at)
                at += n
            t = torch.empty(at * ops.SS_CHUNK, dtype=self.compute_dtype, device=self.device)
            subs = []
            for names, blocks in parts:
                a = min(self.offsets[n] for n in names)
                b = max(self.offsets[n] + self.params[n].numel() for n in names)
                subs.append([a, b, blocks, True])
            ent = self.packed[key] = [t, firsts, subs]
            # a NEW stream changes what a step launches (the optimizer's pass / the pack launch behind it now also writes this
            # stream): recordings made before it existed would replay without refreshing it -> drop them (ctx.generation is what
            # CaptionTrainer keys its launch lists / graphs on)
            self.ctx.generation += 1
        todo = []
        for sub in ent[2]:
            if sub[3]:
                todo += sub[2]
                sub[3] = False
        if todo:
            ops.ss_pack(todo, ent[0])
        return ent[0], ent[1]

    def refresh_lazy_transposed(self):
        """Bring every on-demand transposed copy up to date (decode entry points call this before replaying captured steps,
        which bake the copies' addresses but cannot notice that the weights moved on)."""
        for name, ent in self.transposed.items():
            if not ent[3] and ent[4] != self.version:
                ops.transpose(self.c[name], ent[0])
                ent[4] = self.version

    def name_at(self, off: int):
        """(parameter name, its first flat element) of the parameter that holds flat element `off`."""
        import bisect
        if self._starts is None:
            self._starts = sorted((o, n) for n, o in self.offsets.items())
        i = bisect.bisect_right(self._starts, (off, chr(0x10ffff))) - 1
        return self._starts[i][1], self._starts[i][0]

    def pack_seg(self, name: str):
        """(K, mode, [chunk0 x 4], stream pointer) of the stream-order packed copy the optimizer can maintain for the weight `name`
        (adam_pack_table's eligibility rules), or None."""
        _t, _n, _parts = self.adam_pack_table(0, self.total)
        sg = self._adam_pack[(0, self.total, tuple(sorted(self.packed)))][3].get(name)
        return None if sg is None else (sg[2], sg[3], list(sg[4]), sg[5])

    def adam_pack_table(self, a: int, b: int):
        """(device table of vct_adam_pack_seg, entries, [parts]) for the packed parts whose weights lie inside flat elements [a, b):
        the optimizer's pass over [a, b) writes their stream-order copies itself (ops.adam_step(pack=...)).  Parts with transposed
        blocks (the backward's stream) and matrices whose blocks are not whole 512-row blocks / 512-column slices stay with vct_ss_pack.
        The table is built once per (a, b) and set of streams (pointers are static)."""
        key = (a, b, tuple(sorted(self.packed)))
        hit = self._adam_pack.get(key)
        if hit is not None:
            return hit[:3]
        import bisect
        starts = sorted((off, n) for n, off in self.offsets.items())
        segs, parts = {}, []
        c0 = self.cflat.data_ptr()
        # ONE choice of stream per weight for the whole parameter set: the table of a sub-range is the whole-buffer selection
        # filtered to [a, b) -- the GEMM epilogues take their packed-stream targets from the (0, total) table (pack_seg) and step_range
        # marks parts as written from the (a, b) one; built independently the two could settle on different streams for a weight
        # that two eligible streams hold, and the one nobody writes would go stale
        full = None
        if (a, b) != (0, self.total):
            self.adam_pack_table(0, self.total)
            full = set(id(x) for x in self._adam_pack[(0, self.total, key[2])][2])
        for ent in self.packed.values():
            for sub in ent[2]:
                if not (a <= sub[0] and sub[1] <= b):
                    continue
                if full is not None and id(sub) not in full:
                    continue
                ok, mine = True, {}
                for blk in sub[2]:
                    w, nch, dc = blk[:3]
                    if len(blk) > 3 and blk[3]:
                        ok = False
                        break
                    eo = (w.data_ptr() - c0) // 2
                    i = bisect.bisect_right(starts, (eo, chr(0x10ffff))) - 1
                    mo, name = starts[i]
                    shape = self.params[name].shape
                    if len(shape) != 2 or w.stride(0) != shape[1]:
                        ok = False
                        break
                    N, K = shape
                    r0, cc0 = (eo - mo) // K, (eo - mo) % K
                    if cc0 == 0 and nch * 64 == K and r0 % 512 == 0 and r0 // 512 < 4:
                        mode, idx = 0, r0 // 512
                    elif r0 == 0 and N == 512 and nch == 8 and cc0 % 512 == 0 and cc0 // 512 < 4:
                        mode, idx = 1, cc0 // 512
                    else:
                        ok = False
                        break
                    if dc >= 0xffff:
                        ok = False            # the kernels carry the first chunk of a block as a 16-bit field (0xffff = "not packed"): a
                        break                 # stream of >= 4 GiB stays with vct_ss_pack instead of silently going stale
                    if name in segs and segs[name][5] != ent[0].data_ptr():
                        ok = False            # another stream already holds this weight: the kernel's table has ONE stream per weight,
                        break                 # so this part stays with vct_ss_pack (refresh_transposed) instead of going stale
                    sg = mine.setdefault(name, [mo, mo + N * K, K, mode, [-1, -1, -1, -1], ent[0].data_ptr()])
                    if sg[3] != mode:
                        ok = False
                        break
                    sg[4][idx] = dc
                if ok and mine:
                    segs.update(mine)
                    parts.append(sub)
        rows = sorted(segs.values())
        if rows:
            arr = (ops.L.AdamPackSeg * len(rows))()
            for i, (bg, en, K, mode, ch, ptr) in enumerate(rows):
                arr[i].begin, arr[i].end, arr[i].K, arr[i].mode, arr[i].stream = bg, en, K, mode, ptr
                for j in range(4):
                    arr[i].chunk0[j] = ch[j]
            raw = bytes(arr)
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        else:
            table = None
        hit = self._adam_pack[key] = (table, len(rows), parts, segs)
        return hit[:3]

    def refresh_transposed(self, a: int, b: int, skip=(), packed_done=()):
        """Shadow elements [a, b) were just rewritten: eager transposed copies inside follow (except `skip`: already written by
        the optimizer's fused pass); lazy ones are picked up by want_transposed through `version`.  packed_done: parts of packed
        streams the optimizer's pass wrote itself (adam_pack_table)."""
        for name, ent in self.transposed.items():
            if ent[3] and name not in skip and a <= ent[1] and ent[2] <= b:
                ops.transpose(self.c[name], ent[0])
        done = set(id(x) for x in packed_done)
        for ent in self.packed.values():
            todo = []
            for sub in ent[2]:
                if id(sub) in done:
                    sub[3] = False
                    continue
                if sub[1] > a and sub[0] < b:                 # the rewritten range touches this part
                    if a <= sub[0] and sub[1] <= b:
                        todo += sub[2]
                        sub[3] = False
                    else:                                     # partly rewritten (no schedule does this): re-pack at the next use
                        sub[3] = True
            if todo:
                ops.ss_pack(todo, ent[0])                     # every part of the stack this pass rewrote: one launch (<= 48 blocks)

    def eager_transposed_in(self, a: int, b: int):
        """[(name, tensor, start, end)] of the eager transposed copies whose weight lies inside flat elements [a, b)."""
        return [(n, e[0], e[1], e[2]) for n, e in self.transposed.items() if e[3] and a <= e[1] and e[2] <= b]

    def install_grads(self):
        for n, p in self.params.items():
            if p.requires_grad and p.grad is not self.g[n]:
                if p.grad is not None and p.grad.data_ptr() != self.g[n].data_ptr():
                    self.g[n].add_(p.grad)  # honour a pre-existing accumulated gradient
                p.grad = self.g[n]


class _Buf:
    """Named device buffers with STATIC addresses.  One instance serves every shape configuration of an engine:
    `get` hands out a leading view of a per-name allocation that only ever grows, so a ragged epoch (the loader trims S
    to each batch's longest caption) neither re-allocates per step nor frees memory that a captured hipGraph / recorded
    launch list still points to.  `ctx.generation` (StepContext of the owning parameter set) counts (re)allocations: whoever
    bakes pointers (trainer graphs, launch lists, pointer tables) keys its cache on it and re-records after a growth."""

    def __init__(self, device, ctx: "StepContext"):
        self.device, self.t, self._store, self.ctx = device, {}, {}, ctx

    def get(self, name, shape, dtype):
        n = 1
        for s in shape:
            n *= int(s)
        st = self._store.get(name)
        if st is None or st.dtype != dtype or st.numel() < n:
            st = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._store[name] = st
            self.ctx.generation += 1
            for k in [k for k in self.t if isinstance(k, tuple) and k and k[0] == "ln_table"]:
                del self.t[k]          # pointer tables baked the old addresses
        t = st[:n].view(shape)
        self.t[name] = t
        return t


class _StackBase:
    def __init__(self, ps: ParamSet, prefix: str, cfg: dict, seed: torch.Tensor):
        self.ps, self.pre, self.cfg, self.seed = ps, prefix, cfg, seed
        self.dev, self.dt = ps.device, ps.compute_dtype
        self.bufs: Dict[tuple, _Buf] = {}
        self.p_drop = 0.0
        self._ws = None
        self._ln_pending = []
        self._dw_pending = []
        self._kv_prefetched = None
        self._kv_inplace = set()
        self._prefix = None

    # parameter access: compute-dtype weight, fp32 vector, fp32 gradient
    def W(self, k): return self.ps.c[self.pre + k]
    def WT(self, k): return self.ps.want_transposed(self.pre + k)      # [in, out] copy, kept in step with the shadow
    def F(self, k): return self.ps.params[self.pre + k].data
    def G(self, k): return self.ps.g[self.pre + k]

    def drop(self, site):
        return (self.seed, site, self.p_drop) if self.p_drop > 0.0 else None

    def gemm_ws(self):
        """Split-K scratch of the main stream (partials + zeroed tile counters)."""
        if self._ws is None:
            self._ws = ops.GemmScratch(self.dev)
        return self._ws

    # ---- weight-gradient GEMMs: grouped per layer, on a side HIP stream ---------------------------
    # dW = dY^T X only depends on dY (produced by the dX chain) and X (saved), and nothing downstream in backward
    # reads it.  A layer's 4-7 weight gradients are small (16-128 output tiles each): they are queued while the
    # layer's dX chain runs and issued as ONE grouped launch (ops.gemm_grouped) on a second stream, where they
    # overlap the next layer's dX chain.  The two big ones (generator) go out immediately, alone.
    overlap_dw = True
    group_dw = True
    overlap_kv = True      # cross-attention K/V projections and d(memory) accumulation off the critical path
    defer_gen_dw = True    # single GPU: vocabulary weight gradient at the end of the main stream's tail (A/B switch)

    @property
    def side(self):
        """The model's side stream (None until something needed it)."""
        return self.ps.ctx.side

    def ensure_side(self):
        ctx = self.ps.ctx
        if ctx.side is None:
            # Normal priority.  (Round 5 measured a HIGH-priority side stream: without a gradient exchange the encoder backward's short
            # kernels win freed slots beside the vocabulary weight gradient -- step -0.8 % in three same-box pairs, 0 in a fourth, i.e.
            # inside the run-to-run spread -- but WITH an exchange, collectives in flight on the communicator's stream, every kernel of
            # the step is stretched: world size 1, sharded exchange, 2.88 -> 4.69 ms.  Not worth a stream whose effect depends on what
            # else is in flight: dropped.)
            ctx.side = torch.cuda.Stream(device=self.dev)
            ctx.side_ws = ops.GemmScratch(self.dev)
        return ctx.side

    def _on_side(self, fn):
        if not (self.overlap_dw and self.dev.type == "cuda"):
            return fn(self.gemm_ws())
        ctx = self.ps.ctx
        self.ensure_side()
        cur = torch.cuda.current_stream()
        if cur == ctx.side:               # already running on the side stream (encoder backward beside the decoder's tail)
            return fn(ctx.side_ws)
        ops.stream_wait(ctx.side, cur)
        with torch.cuda.stream(ctx.side):
            return fn(ctx.side_ws)

    def dw_gemm(self, dy, x, dw, *, bias_grad=None, m_valid=None, tag=None):
        """dw[M,N] (fp32 gradient view) = dy^T x, bias_grad[M] = column sums of dy."""
        ad = self.dw_adam_desc(dy, dw)
        if self.group_dw and m_valid is None and tag is None and dy.dtype == torch.bfloat16:
            self._dw_pending.append((dy, x, dw, bias_grad, ad))
            if len(self._dw_pending) == ops.L.GEMM_GROUP_MAX:
                self.flush_dw()
            return
        self._on_side(lambda ws: ops.gemm(dy, x, dw, ta=True, tb=False, bias_grad=bias_grad, m_valid=m_valid, tag=tag,
                                          workspace=ws, adam=ad))

    def dw_adam_desc(self, dy, dw):
        """Epilogue descriptor (ops.L.GemmAdam) when the optimizer steps this weight inside the GEMM that produces its gradient
        `dw` (single GPU, trainer.FusedAdam.enable_dw_fusion), else None."""
        opt = self.ps.dw_adam
        if opt is None or dy.dtype != torch.bfloat16 or self.dev.type != "cuda":
            return None
        return opt.desc_for(dw)

    def flush_dw(self, main: bool = False):
        """Issue the queued weight-gradient GEMMs as one grouped launch (side stream, or the current one if `main`)."""
        if self._dw_pending:
            items, self._dw_pending = self._dw_pending, []
            if main:
                ops.gemm_grouped(items, self.gemm_ws())
            else:
                self._on_side(lambda ws: ops.gemm_grouped(items, ws))

    def flush_dw_across(self, other):
        """Issue the queued weight-gradient group on the stream `other` (behind everything enqueued so far on the current one) instead
        of in line on the current stream: the encoder backward runs as ONE serial chain on the side stream, and a layer's group in
        the middle of that chain (72 us alone, 187 us beside the vocabulary dW) delays every kernel behind it, while the main
        stream idles at the end of the step (profiles/r05_step_timeline_list.txt).  The group only needs this layer's dY / X."""
        if not self._dw_pending:
            return
        cur = torch.cuda.current_stream()
        if other is None or other == cur or not (self.overlap_dw and self.dev.type == "cuda"):
            return self.flush_dw()
        items, self._dw_pending = self._dw_pending, []
        ops.stream_wait(other, cur)
        with torch.cuda.stream(other):
            ops.gemm_grouped(items, self.gemm_ws() if other != self.ps.ctx.side else self.ps.ctx.side_ws)

    def bucket_on_side(self, bucket_ready, *args):
        """Hand a finished gradient bucket to `bucket_ready` WITHOUT stalling the dX chain: the hook runs with the
        side stream current, after that stream has been ordered behind everything enqueued so far on the main
        stream -- an all-reduce issued from the hook waits for this bucket's weight gradients (side stream) and
        LayerNorm/bias gradients (main stream) while the main stream goes straight on to the next layer."""
        self.flush_dw()
        side = self.ps.ctx.side
        if side is None or not self.overlap_dw or torch.cuda.current_stream() == side:
            return bucket_ready(*args)
        ops.stream_wait(side, None)
        with torch.cuda.stream(side):
            return bucket_ready(*args)

    def join_side(self):
        """Main stream waits for every weight-gradient GEMM issued so far (before a gradient bucket is
        handed to the exchange / the optimizer)."""
        self.flush_dw()
        side = self.ps.ctx.side
        if side is not None and self.overlap_dw and torch.cuda.current_stream() != side:
            ops.stream_wait(None, side)

    def buf(self, key) -> _Buf:
        """The engine's buffer set (`key` = the shape configuration, kept for diagnostics only: every configuration
        shares one set of grow-only allocations, see _Buf)."""
        b = self.bufs.get("all")
        if b is None:
            b = self.bufs["all"] = _Buf(self.dev, self.ps.ctx)
        return b

    # ---- shared sub-blocks -------------------------------------------------------------------
    def _attn_block_fwd(self, b, tag, lp, x, kv_src, Bn, Lq, Lk, causal, key_pad, site, self_attn=True):
        """x:[Mq,d] queries source; kv_src:[Mk,d].  Returns (a = out_proj(attn) [Mq,d])."""
        d, H = self.cfg["d"], self.cfg["nhead"]
        Mq, Mk = x.shape[0], kv_src.shape[0]
        if self_attn:
            qkv = b.get(tag + "qkv", (Mq, 3 * d), self.dt)
            ops.gemm(x, self.W(lp + "in_proj_weight"), qkv, bias=self.F(lp + "in_proj_bias"))
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        else:
            q = b.get(tag + "q", (Mq, d), self.dt)
            ops.gemm(x, self.W(lp + "in_proj_weight")[:d], q, bias=self.F(lp + "in_proj_bias")[:d])
            kv = self._cross_kv(b, tag, lp, kv_src)
            k, v = kv[:, :d], kv[:, d:]
        o = b.get(tag + "o", (Mq, d), self.dt)
        ops.attn_fwd(q, k, v, o, Bn, H, Lq, Lk, causal=causal, key_pad=key_pad, dropout=self.drop(site))
        a = b.get(tag + "a", (Mq, d), self.dt)
        ops.gemm(o, self.W(lp + "out_proj.weight"), a, bias=self.F(lp + "out_proj.bias"))
        return a

    def _attn_ln_fwd(self, b, tag, ntag, lp, np_, x, kv_src, Bn, Lq, Lk, causal, key_pad, site, site_ln, self_attn=True):
        """y = LayerNorm(x + dropout(out_proj(MHA(x, kv_src)))) of one attention block; saves o, a, mean, rstd for the backward.
        (Two fused forms of this block -- attention core + out_proj + add-LayerNorm as one launch, and out_proj + add-LayerNorm as one
        row-complete launch -- were built in rounds 3/4, measured slower in the step and removed in round 5: git tag
        archive/r5-off-kernels)"""
        a = self._attn_block_fwd(b, tag, lp, x, kv_src, Bn, Lq, Lk, causal, key_pad, site, self_attn=self_attn)
        return self._ln_fwd(b, ntag, np_, a, x, site_ln)

    def _cross_kv(self, b, tag, lp, mem):
        """K/V projection of the encoder memory for one cross-attention block.  It depends on the memory only, so
        prefetch_cross_kv() issues it for every layer on the side stream at the start of the decoder stack, off the
        critical path; here it is either picked up (after joining that stream) or computed in place."""
        d = self.cfg["d"]
        kv = b.get(tag + "kv", (mem.shape[0], 2 * d), self.dt)
        if self._kv_prefetched and tag not in self._kv_inplace:
            if self._kv_prefetched == "pending":
                self.join_side()
                self._kv_prefetched = "joined"
        else:
            ops.gemm(mem, self.W(lp + "in_proj_weight")[d:], kv, bias=self.F(lp + "in_proj_bias")[d:])
        return kv

    def prefetch_cross_kv(self, b, mem, tags_lps):
        if not self.overlap_kv or not tags_lps:
            return
        d = self.cfg["d"]
        for tag, lp in tags_lps:
            kv = b.get(tag + "kv", (mem.shape[0], 2 * d), self.dt)
            self._on_side(lambda ws, kv=kv, lp=lp: ops.gemm(mem, self.W(lp + "in_proj_weight")[d:], kv, bias=self.F(lp + "in_proj_bias")[d:]))
        self._kv_prefetched = "pending"

    def _attn_block_bwd(self, b, tag, lp, da, x, kv_src, Bn, Lq, Lk, causal, key_pad, site, self_attn, ds_res,
                        dkv_out=None, dkv_accumulate=False):
        """da: grad of out_proj output (dropout-masked).  Returns dx = grad wrt x (+ ds_res)."""
        d, H = self.cfg["d"], self.cfg["nhead"]
        Mq, Mk = x.shape[0], kv_src.shape[0]
        o = b.t[tag + "o"]
        d_o = b.get(tag + "d_o", (Mq, d), self.dt)
        ops.gemm(da, self.W(lp + "out_proj.weight"), d_o, ta=False, tb=False)
        self.dw_gemm(da, o, self.G(lp + "out_proj.weight"), bias_grad=self.G(lp + "out_proj.bias"))
        dx = b.get(tag + "dx", (Mq, d), self.dt)
        if self_attn:
            qkv = b.t[tag + "qkv"]
            dqkv = b.get(tag + "dqkv", (Mq, 3 * d), self.dt)
            ops.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], d_o, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:],
                         Bn, H, Lq, Lk, causal=causal, key_pad=key_pad, dropout=self.drop(site))
            ops.gemm(dqkv, self.W(lp + "in_proj_weight"), dx, ta=False, tb=False, addend=ds_res)
            self.dw_gemm(dqkv, x, self.G(lp + "in_proj_weight"), bias_grad=self.G(lp + "in_proj_bias"))
        else:
            q, kv = b.t[tag + "q"], b.t[tag + "kv"]
            dq = b.get(tag + "dq", (Mq, d), self.dt)
            dkv = b.get(tag + "dkv", (Mk, 2 * d), self.dt)
            ops.attn_bwd(q, kv[:, :d], kv[:, d:], d_o, dq, dkv[:, :d], dkv[:, d:], Bn, H, Lq, Lk, causal=causal,
                         key_pad=key_pad, dropout=self.drop(site))
            ops.gemm(dq, self.W(lp + "in_proj_weight")[:d], dx, ta=False, tb=False, addend=ds_res)
            self.dw_gemm(dq, x, self.G(lp + "in_proj_weight")[:d], bias_grad=self.G(lp + "in_proj_bias")[:d])
            # d(memory) is only read by the encoder backward: accumulate it on the side stream (in layer order)
            run = self._on_side if self.overlap_kv else (lambda fn: fn(None))
            run(lambda ws: ops.gemm(dkv, self.W(lp + "in_proj_weight")[d:], dkv_out, ta=False, tb=False,
                                    addend=dkv_out if dkv_accumulate else None))
            self.dw_gemm(dkv, kv_src, self.G(lp + "in_proj_weight")[d:],
                         bias_grad=self.G(lp + "in_proj_bias")[d:])
        return dx

    def _ln_fwd(self, b, tag, np_, x, res, site):
        M, d = x.shape
        y = b.get(tag + "y", (M, d), self.dt)
        ops.add_ln_fwd(x, res, self.F(np_ + "weight"), self.F(np_ + "bias"), y, b.get(tag + "mean", (M,), torch.float32),
                       b.get(tag + "rstd", (M,), torch.float32), dropout=self.drop(site) if site is not None else None)
        return y

    def _ln_ln_fwd(self, b, tag, np_, x, res, site, tag2, np2):
        """The last layer's closing norm and the stack-final norm in one launch: returns (y, y2)."""
        M, d = x.shape
        y, y2 = b.get(tag + "y", (M, d), self.dt), b.get(tag2 + "y", (M, d), self.dt)
        ops.add_ln_ln_fwd(x, res, self.F(np_ + "weight"), self.F(np_ + "bias"), y, b.get(tag + "mean", (M,), torch.float32),
                          b.get(tag + "rstd", (M,), torch.float32), self.F(np2 + "weight"), self.F(np2 + "bias"), y2,
                          b.get(tag2 + "mean", (M,), torch.float32), b.get(tag2 + "rstd", (M,), torch.float32),
                          dropout=self.drop(site) if site is not None else None)
        return y, y2

    def _ln_bwd(self, b, tag, np_, dy, x, res, site):
        """Returns (ds, dxo): gradient of the pre-norm sum and its dropout-masked copy.  The column
        reduction of the dgamma/dbeta partials is deferred: flush_ln_grads() does all of them in one launch."""
        M, d = x.shape
        ds = b.get(tag + "ds", (M, d), self.dt)
        drop = self.drop(site) if site is not None else None
        dxo = b.get(tag + "dxo", (M, d), self.dt) if drop is not None else ds
        ws = b.get(tag + "ln_ws", (2 * ops.ln_ws_rows(M) * d,), torch.float32)     # one per LayerNorm (kept until the flush)
        ops.add_ln_bwd(dy, x, res, self.F(np_ + "weight"), b.t[tag + "mean"], b.t[tag + "rstd"], ds, dxo,
                       None, None, ws, dropout=drop)
        self._ln_pending.append((ws.data_ptr(), self.G(np_ + "weight").data_ptr(), self.G(np_ + "bias").data_ptr(),
                                 ops.ln_ws_rows(M)))
        return ds, dxo

    fuse_ln_ln_bwd = os.environ.get("VCT_LN_LN_BWD", "1") != "0"      # A/B switch

    def _ln_ln_bwd(self, b, tag2, np2, dy2, y, tag, np_, x, res, site):
        """The stack-final norm's backward and the top layer's closing norm's backward in ONE launch (ops.add_ln_ln_bwd; bit-identical
        to _ln_bwd(final) followed by _ln_bwd(layer)).  Returns (ds, dxo) of the layer norm."""
        M, d = x.shape
        ds = b.get(tag + "ds", (M, d), self.dt)
        drop = self.drop(site) if site is not None else None
        dxo = b.get(tag + "dxo", (M, d), self.dt) if drop is not None else ds
        rows = ops.ln_ws_rows(M)
        ws2 = b.get(tag2 + "ln_ws", (2 * rows * d,), torch.float32)
        ws = b.get(tag + "ln_ws", (2 * rows * d,), torch.float32)
        ops.add_ln_ln_bwd(dy2, y, self.F(np2 + "weight"), b.t[tag2 + "mean"], b.t[tag2 + "rstd"], ws2, x, res, self.F(np_ + "weight"),
                          b.t[tag + "mean"], b.t[tag + "rstd"], ds, dxo, ws, dropout=drop)
        for w_, n_ in ((ws2, np2), (ws, np_)):
            self._ln_pending.append((w_.data_ptr(), self.G(n_ + "weight").data_ptr(), self.G(n_ + "bias").data_ptr(), rows))
        return ds, dxo

    def flush_ln_grads(self, b):
        """One launch: dgamma/dbeta of every LayerNorm whose backward ran since the last flush."""
        if not self._ln_pending:
            return
        key = tuple(self._ln_pending)
        tab = b.t.get(("ln_table", key))
        if tab is None:       # pointers are static per shape configuration: the table is uploaded once
            tab = torch.tensor(self._ln_pending, dtype=torch.int64).to(self.dev)
            b.t[("ln_table", key)] = tab
        ops.ln_param_finalize_batched(tab, len(self._ln_pending), self.cfg["d"])
        self._ln_pending = []

    # ---- ONE launch per layer: the sample-stationary forward (csrc/vct_layer_ss.hip) ---------------------------------------
    # A/B switch.  bf16, d = 512 / 8 heads, rows per sample <= 32, memory rows <= 16: every layer of the stack is one launch in
    # which a workgroup keeps its sample in LDS and streams the layer's weights (a stream-order packed second shadow) from L2.
    # It saves the tensors and draws the dropout streams of the unfused kernels: the backward schedule is unchanged behind it.
    fuse_layers = os.environ.get("VCT_FUSE_LAYERS", "1") != "0"

    def _ss_ok(self, Lr: int, Lm: int, Bn: int) -> bool:
        """One workgroup per sample streams ALL of a layer's weights: it pays while the samples fit the CUs in one round (cfg-B:
        256 samples on 256 CUs, forward bracket 0.52 vs 0.55 ms unfused); at a per-GPU batch of 1024 the tiled GEMMs amortise the
        weights over 4864+ rows and win (1.39 vs 1.84 ms)."""
        c = self.cfg
        if not (self.fuse_layers and self.dev.type == "cuda" and c["activation"] in ("gelu", "relu")):
            return False
        if Bn > _cu_count(self.dev) * 5 // 4:
            return False
        return ops.layer_ss_supported(self.dt, c["d"], c["nhead"], c["ff"], Lr, Lm)

    def _ss_stream(self, lps, cross: bool, lead=None):
        """The packed weight stream of the stack's layers `lps` (blocks in the kernel's consumption order, layer after layer); lead =
        name of a 512 x 512 weight the kernel's prologue consumes first (the encoder's unify Linear)."""
        ff = self.cfg["ff"]

        def one(lp):
            P = self.pre + lp
            names = [P + "self_attn.in_proj_weight", P + "self_attn.out_proj.weight", P + "linear1.weight", P + "linear2.weight"]
            if cross:
                names += [P + "multihead_attn.in_proj_weight", P + "multihead_attn.out_proj.weight"]

            def blocks():
                c, out, at = self.ps.c, [], 0
                mats = [(c[P + "self_attn.in_proj_weight"], 3), (c[P + "self_attn.out_proj.weight"], 1)]
                if cross:
                    mats += [(c[P + "multihead_attn.in_proj_weight"], 3), (c[P + "multihead_attn.out_proj.weight"], 1)]
                for w, nb in mats:
                    for i in range(nb):
                        out.append((w[512 * i:512 * (i + 1)], 8, at)); at += 8
                # feed-forward, software-pipelined: linear1 block 0 | for j: linear1 block j+1 (its K steps carry chunk j's GELU), linear2 K slice j
                w1, w2 = c[P + "linear1.weight"], c[P + "linear2.weight"]
                nj = ff // 512
                out.append((w1[0:512], 8, at)); at += 8
                for j in range(nj):
                    if j + 1 < nj:
                        out.append((w1[512 * (j + 1):512 * (j + 2)], 8, at)); at += 8
                    out.append((w2[:, 512 * j:512 * (j + 1)], 8, at)); at += 8
                return out
            return names, blocks
        parts = [one(lp) for lp in lps]
        if lead is not None:
            parts.insert(0, ([self.pre + lead], lambda: [(self.ps.c[self.pre + lead][0:512], 8, 0)]))
        return self.ps.want_packed(self.pre + lps[0] + f"x{len(lps)}" + (lead or ""), parts)

    def _ss_stream_bwd(self, lps):
        """The packed TRANSPOSED weight stream of the self-attention + feed-forward layers `lps` (processing order of the backward:
        top layer first) for vct_layer_ss_bwd: [linear2^T block j | linear1^T K slice j] x ff/512 | out_proj^T | in_proj^T."""
        ff = self.cfg["ff"]

        def one(lp):
            P = self.pre + lp
            names = [P + "self_attn.in_proj_weight", P + "self_attn.out_proj.weight", P + "linear1.weight", P + "linear2.weight"]

            def blocks():
                c, out, at = self.ps.c, [], 0
                w1, w2 = c[P + "linear1.weight"], c[P + "linear2.weight"]
                for j in range(ff // 512):
                    out.append((w2[:, 512 * j:], 8, at, True)); at += 8
                    out.append((w1[512 * j:512 * (j + 1)], 8, at, True)); at += 8
                out.append((c[P + "self_attn.out_proj.weight"], 8, at, True)); at += 8
                out.append((c[P + "self_attn.in_proj_weight"], 24, at, True)); at += 24
                return out
            return names, blocks
        return self.ps.want_packed(self.pre + lps[0] + f"bwd{len(lps)}", [one(lp) for lp in lps])

    # The activation-gradient chain of a self-attention + feed-forward stack as ONE launch (csrc/vct_layer_ss_bwd.hip): OFF by default.
    # Measured at cfg-B (round 4, same box): the launch takes 236 us ALONE for the two encoder layers against 253 us for the 14 unfused
    # launches alone -- but it takes whole compute units (152 KB of LDS per workgroup), so nothing runs beside it, while the unfused
    # chain shares the chip with the vocabulary weight gradient and the optimizer's pass in the step's tail: step 2.40-2.42 ms with it
    # (beside the tail, or alone on the main stream ahead of the optimizer) against 2.24-2.26 ms without.  VCT_FUSE_BWD=1 enables it.
    fuse_bwd = os.environ.get("VCT_FUSE_BWD", "0") == "1"

    def _stack_ss_bwd(self, b, lps, tags, sites0, dy, dx, Bn, Lr, *, ln_tag, ln_name, final, kpm=None, causal=False):
        """Gradient of the stack input from the gradient `dy` of the (final-normed) stack output: ONE launch; queues the layers' weight-
        gradient GEMMs (grouped, one launch per layer) and the LayerNorm parameter partials behind it.  lps / tags / sites0 in FORWARD
        order (bottom layer first)."""
        d, ff, H = self.cfg["d"], self.cfg["ff"], self.cfg["nhead"]
        M = Bn * Lr
        f32 = torch.float32
        order = list(reversed(range(len(lps))))
        wpk, firsts = self._ss_stream_bwd([lps[l] for l in order])
        per = ops.layer_ss_bwd_stream_chunks(ff)
        descs, after = [], []
        for k, l in enumerate(order):
            lp, tag, site = lps[l], tags[l], sites0[l]

            def nb(t, name):
                ws = b.get(t + "ss_ws", (Bn * 2 * d,), f32)
                self._ln_pending.append((ws.data_ptr(), self.G(name + "weight").data_ptr(), self.G(name + "bias").data_ptr(), Bn))
                return (self.F(name + "weight"), b.t[t + "mean"], b.t[t + "rstd"], ws)
            nf = nb("nf.", final) if k == 0 else None
            n3 = nb(tag + ln_tag, lp + ln_name)
            n1 = nb(tag + "n1.", lp + "norm1.")
            df, dhpre = b.get(tag + ln_tag + "dxo", (M, d), self.dt), b.get(tag + "ff.dhpre", (M, ff), self.dt)
            da, dqkv = b.get(tag + "n1.dxo", (M, d), self.dt), b.get(tag + "sa.dqkv", (M, 3 * d), self.dt)
            x, x1 = b.t[tag + "x"], b.t[tag + "n1.y"]
            descs.append(ops.layer_ss_bwd_desc(
                B=Bn, Lr=Lr, wpk=wpk[firsts[k] * ops.SS_CHUNK:], nchunks=per, ff=ff, act=self.cfg["activation"], H=H,
                x=x, qkv=b.t[tag + "sa.qkv"], a=b.t[tag + "sa.a"], x1=x1, hpre=b.t[tag + "ff.hpre"], f=b.t[tag + "ff.f"],
                n1=n1, n3=n3, nf=nf, y_last=b.t["x_last"] if k == 0 else None, dy=dy if k == 0 else None,
                dx=dx if k == len(order) - 1 else None, outs=(df, dhpre, da, dqkv),
                sites=(site + 1, site + 2, site + 3, site + 4), causal=causal, key_pad=kpm,
                seed=self.seed if self.p_drop > 0.0 else None, p_drop=self.p_drop))
            sa = lp + "self_attn."
            after.append([(df, b.t[tag + "ff.h"], lp + "linear2."), (dhpre, x1, lp + "linear1."), (da, b.t[tag + "sa.o"], sa + "out_proj."),
                          (dqkv, x, sa + "in_proj_")])
        ops.layer_ss_bwd(descs)
        return order, after

    def _stack_ss(self, b, lps, tags, x, Bn, Lr, sites0, *, ln_tag, ln_name, final, mem=None, Lm=0, causal=False, kpm=None,
                  frontend=None, embed=None):
        """The layers `lps` (buffer tags `tags`, dropout site bases `sites0`) on input x [Bn*Lr, d] in ONE launch per four layers.
        ln_tag / ln_name: buffer tag and parameter name of a layer's closing norm ('n2.' / 'norm2.' encoder, 'n3.' / 'norm3.' decoder);
        final = parameter prefix of the stack-final norm.  Returns (last layer's output, final-norm output)."""
        d, ff, H = self.cfg["d"], self.cfg["ff"], self.cfg["nhead"]
        M, cross = Bn * Lr, mem is not None
        f32 = torch.float32
        wpk, firsts = self._ss_stream(lps, cross, lead="unify.0.weight" if frontend is not None else None)
        if frontend is not None:
            firsts = firsts[1:]                  # (the kernel's prologue reads the unify block from the head of the stream itself)
        per = ops.layer_ss_stream_chunks(ff, cross)
        descs, y, y2 = [], x, None
        for l, (lp, tag, site) in enumerate(zip(lps, tags, sites0)):
            b.t[tag + "x"] = y

            def norm(t, name):
                return (self.F(name + "weight"), self.F(name + "bias"), b.get(t + "y", (M, d), self.dt), b.get(t + "mean", (M,), f32),
                        b.get(t + "rstd", (M,), f32))
            sa, st = lp + "self_attn.", tag + "sa."
            bias = {"qkv": self.F(sa + "in_proj_bias"), "o": self.F(sa + "out_proj.bias"), "l1": self.F(lp + "linear1.bias"),
                    "l2": self.F(lp + "linear2.bias")}
            kw = {}
            if cross:
                ca, ct = lp + "multihead_attn.", tag + "ca."
                bias.update(cq=self.F(ca + "in_proj_bias")[:d], ckv=self.F(ca + "in_proj_bias")[d:], co=self.F(ca + "out_proj.bias"))
                kw = dict(cross=(b.get(ct + "q", (M, d), self.dt), b.get(ct + "kv", (Bn * Lm, 2 * d), self.dt), b.get(ct + "o", (M, d), self.dt),
                                 b.get(ct + "a", (M, d), self.dt)),
                          n2=norm(tag + "n2.", lp + "norm2."), mem=mem, Lm=Lm)
                sites = (site + 1, site + 2, site + 3, site + 4, site + 5, site + 6)
            else:
                sites = (site + 1, site + 2, 0, 0, site + 3, site + 4)
            nl = norm(tag + ln_tag, lp + ln_name)
            nf = norm("nf.", final) if l == len(lps) - 1 else None
            descs.append(ops.layer_ss_desc(
                B=Bn, Lr=Lr, x=y, wpk=wpk[(firsts[l] if l else 0) * ops.SS_CHUNK:], nchunks=per, ff=ff, act=self.cfg["activation"], H=H, bias=bias,
                sa=(b.get(st + "qkv", (M, 3 * d), self.dt), b.get(st + "o", (M, d), self.dt), b.get(st + "a", (M, d), self.dt)),
                n1=norm(tag + "n1.", lp + "norm1."),
                ffn=(b.get(tag + "ff.hpre", (M, ff), self.dt), b.get(tag + "ff.h", (M, ff), self.dt), b.get(tag + "ff.f", (M, d), self.dt)),
                n3=nl, nf=nf, causal=causal, key_pad=kpm, seed=self.seed, p_drop=self.p_drop, sites=sites,
                frontend=frontend if l == 0 else None, embed=embed if l == 0 else None, **kw))
            y, y2 = nl[2], (nf[2] if nf is not None else None)
        tag_ = "ss_dec" if cross else "ss_enc"
        ops.tap(tag_, 0)
        ops.layer_ss_fwd(descs)
        ops.tap(tag_, 1)
        return y, y2

    def _ffn_fwd(self, b, tag, lp, x, site):
        M, d = x.shape
        ff = self.cfg["ff"]
        hpre = b.get(tag + "hpre", (M, ff), self.dt)
        h = b.get(tag + "h", (M, ff), self.dt)
        ops.gemm(x, self.W(lp + "linear1.weight"), h, bias=self.F(lp + "linear1.bias"), act=self.cfg["activation"],
                 preact=hpre, dropout=self.drop(site))
        f = b.get(tag + "f", (M, d), self.dt)
        ops.gemm(h, self.W(lp + "linear2.weight"), f, bias=self.F(lp + "linear2.bias"))
        return f

    def _ffn_bwd(self, b, tag, lp, df, x, site, ds_res):
        """df: grad wrt f (dropout-masked).  Returns grad wrt x (+ ds_res)."""
        M, d = x.shape
        ff = self.cfg["ff"]
        dhpre = b.get(tag + "dhpre", (M, ff), self.dt)
        ops.gemm(df, self.W(lp + "linear2.weight"), dhpre, ta=False, tb=False, act=self.cfg["activation"],
                 dact_src=b.t[tag + "hpre"], dropout=self.drop(site))
        self.dw_gemm(df, b.t[tag + "h"], self.G(lp + "linear2.weight"), bias_grad=self.G(lp + "linear2.bias"))
        dx = b.get(tag + "dxf", (M, d), self.dt)
        ops.gemm(dhpre, self.W(lp + "linear1.weight"), dx, ta=False, tb=False, addend=ds_res)
        self.dw_gemm(dhpre, x, self.G(lp + "linear1.weight"), bias_grad=self.G(lp + "linear1.bias"))
        return dx


class EncoderEngine(_StackBase):
    """MultiModalEncoder (one modality, 'avg' aggregation token, sinusoidal temporal encoding):
    model/MMEncoder.py:244-276."""

    def __init__(self, ps, prefix, cfg, seed, pe_buffer: torch.Tensor):
        super().__init__(ps, prefix, cfg, seed)
        self.pe = pe_buffer  # [1, 512, d] fp32 buffer `temp_emb.pe`
        self._pe_rows = {}

    def pe_rows(self, T):
        r = self._pe_rows.get(T)
        if r is None:
            import numpy as np
            idx = torch.from_numpy(np.linspace(0, T - 1, T).astype(np.int64)).to(self.dev)  # MMEncoder.py:98
            r = torch.zeros(T + 1, self.cfg["d"], dtype=torch.float32, device=self.dev)
            r[1:] = self.pe[0, idx, :]
            self._pe_rows[T] = r
        return r

    def forward(self, feats: torch.Tensor, mask: Optional[torch.Tensor], training: bool) -> torch.Tensor:
        """feats [B,T,Ein] fp32, mask [B,T] bool (True = padded) or None -> memory [B*(T+1), d]."""
        B, T, Ein = feats.shape
        d, L = self.cfg["d"], self.cfg["layers"]
        self.p_drop = self.cfg["dropout"] if training else 0.0
        b = self.buf((B, T))
        self.cur, self.shape = b, (B, T)
        Te, M = T + 1, B * (T + 1)
        x_in = feats.reshape(B * T, Ein).contiguous()
        # key-padding of the encoder's self-attention: the raw frame mask with a shift of one (key 0 = the aggregation
        # token, never padded) -- read by the attention kernel directly, no [B, T+1] mask is built
        kpm = None
        if mask is not None:
            mk = mask if mask.is_contiguous() else mask.contiguous()
            kpm = (mk.view(torch.uint8) if mk.dtype == torch.bool else mk, 1)
        b.t["kpm_used"] = kpm
        if self._ss_ok(Te, 0, B) and Ein == d and x_in.dtype in (torch.float32, self.dt):
            # the whole stack in one launch per four layers: front end (unify Linear, mean token, temporal encoding) in the kernel's
            # prologue, the stack-final norm in its last epilogue
            xc = None
            if x_in.dtype != self.dt:
                xc = b.get("feats_c", (B * T, Ein), self.dt)            # bf16 copy of the features: the unify weight gradient's operand
            b.t["x_in"] = xc if xc is not None else x_in
            x0 = b.get("x0", (M, d), self.dt)
            x, mem = self._stack_ss(b, [f"transformer_encoder.layers.{l}." for l in range(L)], [f"L{l}." for l in range(L)], x0, B, Te,
                                    [ENC_SITE + 16 * l for l in range(L)], ln_tag="n2.", ln_name="norm2.",
                                    final="transformer_encoder.norm.", kpm=kpm,
                                    frontend=(x_in, xc, self.F("unify.0.bias"), self.pe_rows(T)))
            b.t["x_last"] = x
            return mem
        if x_in.dtype != self.dt:        # fp32 features (reference contract); a DeviceLoader batch may already be bf16
            x_in = ops.cast(x_in, b.get("feats_c", (B * T, Ein), self.dt))
        b.t["x_in"] = x_in
        u = b.get("u", (B * T, d), self.dt)
        ops.gemm(x_in, self.W("unify.0.weight"), u, bias=self.F("unify.0.bias"))
        x = ops.enc_frontend_fwd(u, self.pe_rows(T), b.get("x0", (M, d), self.dt), B, T)
        if self._ss_ok(Te, 0, B):       # (features of another width: the front end stays on its own kernels)
            x, mem = self._stack_ss(b, [f"transformer_encoder.layers.{l}." for l in range(L)], [f"L{l}." for l in range(L)], x, B, Te,
                                    [ENC_SITE + 16 * l for l in range(L)], ln_tag="n2.", ln_name="norm2.",
                                    final="transformer_encoder.norm.", kpm=kpm)
            b.t["x_last"] = x
            return mem
        for l in range(L):
            lp, tag, site = f"transformer_encoder.layers.{l}.", f"L{l}.", ENC_SITE + 16 * l
            b.t[tag + "x"] = x
            x1 = self._attn_ln_fwd(b, tag + "sa.", tag + "n1.", lp + "self_attn.", lp + "norm1.", x, x, B, Te, Te, False, kpm,
                                   site + 1, site + 2)
            f = self._ffn_fwd(b, tag + "ff.", lp, x1, site + 3)
            if l == L - 1:       # norm2 of the last layer + the stack-final norm: one launch
                x, mem = self._ln_ln_fwd(b, tag + "n2.", lp + "norm2.", f, x1, site + 4, "nf.", "transformer_encoder.norm.")
                b.t["x_last"] = x
                return mem
            x = self._ln_fwd(b, tag + "n2.", lp + "norm2.", f, x1, site + 4)
        b.t["x_last"] = x
        return self._ln_fwd(b, "nf.", "transformer_encoder.norm.", x, None, None)

    def ss_bwd_ok(self) -> bool:
        """The current shape's activation-gradient chain runs as one sample-stationary launch (csrc/vct_layer_ss_bwd.hip)."""
        B, T = self.shape
        return bool(self.fuse_bwd and self._ss_ok(T + 1, 0, B) and self.cfg["layers"] <= 4)

    def backward(self, dmem: torch.Tensor, bucket_ready=None, join: bool = True):
        """join = False (one-launch backward on the main stream, trainer): the weight-gradient GEMMs stay un-joined on the side
        stream; the caller joins before it touches the encoder's gradients."""
        b = self.cur
        B, T = self.shape
        Te, L = T + 1, self.cfg["layers"]
        kpm = b.t["kpm_used"]
        if self.ss_bwd_ok() and dmem.dtype == self.dt:
            # the whole dX chain of the stack in one launch; behind it one grouped weight-gradient launch per layer
            dx = b.get("L0.sa.dx", (B * Te, self.cfg["d"]), self.dt)
            order, after = self._stack_ss_bwd(b, [f"transformer_encoder.layers.{l}." for l in range(L)], [f"L{l}." for l in range(L)],
                                              [ENC_SITE + 16 * l for l in range(L)], dmem, dx, B, Te, ln_tag="n2.", ln_name="norm2.",
                                              final="transformer_encoder.norm.", kpm=kpm)
            for l, items in zip(order, after):
                for dyv, xv, name in items:
                    self.dw_gemm(dyv, xv, self.G(name + "weight"), bias_grad=self.G(name + "bias"))
                if l > 0:
                    self.flush_dw()
                if bucket_ready is not None and l > 0:
                    self.flush_ln_grads(b)
                    self.bucket_on_side(bucket_ready, "enc_layer", l)
            du = ops.enc_frontend_bwd(dx, b.get("du", (B * T, self.cfg["d"]), self.dt), B, T)
            self.dw_gemm(du, b.t["x_in"], self.G("unify.0.weight"), bias_grad=self.G("unify.0.bias"))
            self.flush_ln_grads(b)
            if join or bucket_ready is not None:
                self.join_side()
            else:
                self.flush_dw()
            if bucket_ready is not None:
                bucket_ready("enc_layer", 0)
            return
        if not self.fuse_ln_ln_bwd:
            dx, _ = self._ln_bwd(b, "nf.", "transformer_encoder.norm.", dmem, b.t["x_last"], None, None)
        for l in reversed(range(L)):
            lp, tag, site = f"transformer_encoder.layers.{l}.", f"L{l}.", ENC_SITE + 16 * l
            x, x1 = b.t[tag + "x"], b.t[tag + "n1.y"]
            if l == L - 1 and self.fuse_ln_ln_bwd:      # stack-final norm + this layer's norm2: one launch
                ds2, df = self._ln_ln_bwd(b, "nf.", "transformer_encoder.norm.", dmem, b.t["x_last"], tag + "n2.", lp + "norm2.",
                                          b.t[tag + "ff.f"], x1, site + 4)
            else:
                ds2, df = self._ln_bwd(b, tag + "n2.", lp + "norm2.", dx, b.t[tag + "ff.f"], x1, site + 4)
            dx1 = self._ffn_bwd(b, tag + "ff.", lp, df, x1, site + 3, ds2)
            ds1, da = self._ln_bwd(b, tag + "n1.", lp + "norm1.", dx1, b.t[tag + "sa.a"], x, site + 2)
            dx = self._attn_block_bwd(b, tag + "sa.", lp + "self_attn.", da, x, x, B, Te, Te, False, kpm, site + 1, True, ds1)
            if l > 0:
                if bucket_ready is None and self.enc_dw_main and getattr(self, "main_stream", None) is not None:
                    self.flush_dw_across(self.main_stream)      # behind the main stream's tail (vocabulary dW, optimizer pass)
                else:
                    self.flush_dw()       # this layer's weight gradients: one grouped launch beside the next layer
            if bucket_ready is not None and l > 0:
                self.flush_ln_grads(b)
                self.bucket_on_side(bucket_ready, "enc_layer", l)
        du = ops.enc_frontend_bwd(dx, b.get("du", (B * T, self.cfg["d"]), self.dt), B, T)
        self.dw_gemm(du, b.t["x_in"], self.G("unify.0.weight"), bias_grad=self.G("unify.0.bias"))
        self.flush_ln_grads(b)
        if bucket_ready is None and self.enc_dw_main >= 2 and getattr(self, "main_stream", None) is not None:
            self.flush_dw_across(self.main_stream)       # (A/B: the bottom layer's group too)
        self.join_side()
        if bucket_ready is not None:
            bucket_ready("enc_layer", 0)


class DecoderEngine(_StackBase):
    """CapDecoder: embedding + positional table, decoder stack, generator, SCE loss
    (model/CapDecoder.py:34-60)."""

    def __init__(self, ps, prefix, cfg, seed, pos_buffer: torch.Tensor):
        super().__init__(ps, prefix, cfg, seed)
        self.pos = pos_buffer  # [5000, d] fp32 buffer
        self.V = cfg["vocab"]
        self.Vp = (self.V + 31) // 32 * 32
        # set per ENGINE by trainer.CaptionTrainer (single GPU, fused optimizer): nothing but this engine's backward schedule writes
        # the flat gradient buffer, so the token-embedding gradient only re-zeroes the rows it wrote in the previous step
        self.exclusive_grads = False

    def _ws_grew(self):
        self.ps.ctx.generation += 1      # recordings that baked the outgrown id workspace are dropped by their owners

    def _embed(self, b, ids, Sd, M):
        return ops.embed_fwd(ids, Sd, self.F("tgt_to_emb.weight"), self.pos, b.get("x0", (M, self.cfg["d"]), self.dt),
                             dropout=self.drop(EMB_SITE))

    def _self_block(self, b, l, x, Bn, Sd, kpm):
        """x1 = LN1(x + drop(SelfMHA(x))) of decoder layer l."""
        lp, tag, site = f"decoder.layers.{l}.", f"L{l}.", DEC_SITE + 16 * l
        b.t[tag + "x"] = x
        return self._attn_ln_fwd(b, tag + "sa.", tag + "n1.", lp + "self_attn.", lp + "norm1.", x, x, Bn, Sd, Sd, True, kpm,
                                 site + 1, site + 2)

    def forward_prefix(self, Bn: int, Te: int, ids: torch.Tensor, training: bool):
        """The part of the decoder forward that does not depend on the encoder memory -- token embedding and the bottom
        layer's self-attention block -- issued on the side stream so that it runs beside the encoder forward
        (MMT4Caption._forward_loss calls this BEFORE the encoder is enqueued; forward() picks the result up)."""
        pad, S = self.cfg["pad_id"], ids.shape[1]
        Sd, M = S - 1, Bn * (S - 1)
        self.p_drop = self.cfg["dropout"] if training else 0.0
        if self._ss_ok(Sd, Te, Bn):     # one workgroup per sample fills every CU: nothing to run beside the encoder
            self._prefix = None
            return
        b = self.buf((Bn, Te, S))
        kpm = ("ids", ids, pad)

        def run(_ws):
            x = self._embed(b, ids, Sd, M)
            return x, self._self_block(b, 0, x, Bn, Sd, kpm)
        x, x1 = self._on_side(run)
        self._prefix = (b, x, x1)

    def _run_stack(self, b, mem, Bn, Te, ids, Sd, kpm):
        """Embedding + decoder layers + final LayerNorm over the first Sd tokens of each ids row."""
        d, L = self.cfg["d"], self.cfg["layers"]
        M = Bn * Sd
        prefix, self._prefix = self._prefix, None
        if self._ss_ok(Sd, Te, Bn) and prefix is None:
            self._kv_prefetched, self._kv_inplace = None, set()
            x0 = b.get("x0", (M, d), self.dt)                              # built by the kernel's prologue (token embedding + positions + dropout)
            x, y = self._stack_ss(b, [f"decoder.layers.{l}." for l in range(L)], [f"L{l}." for l in range(L)], x0, Bn, Sd,
                                  [DEC_SITE + 16 * l for l in range(L)], ln_tag="n3.", ln_name="norm3.", final="decoder.norm.",
                                  mem=mem, Lm=Te, causal=True, kpm=kpm,
                                  embed=(ids, self.F("tgt_to_emb.weight"), self.pos, EMB_SITE))
            b.t["x_last"] = x
            return y
        if prefix is not None and prefix[0] is b:        # embedding + bottom self-attention already ran beside the encoder
            x, x1_0 = prefix[1], prefix[2]
            # the main stream has nothing else to do until the bottom cross-attention: its K/V projection runs here, in
            # place (no cross-stream hand-over on the critical path); the upper layers' go to the side stream
            self.join_side()
            self.prefetch_cross_kv(b, mem, [(f"L{l}.ca.", f"decoder.layers.{l}.multihead_attn.") for l in range(1, L)])
            self._kv_inplace = {"L0.ca."}
        else:
            self.prefetch_cross_kv(b, mem, [(f"L{l}.ca.", f"decoder.layers.{l}.multihead_attn.") for l in range(L)])
            self._kv_inplace = set()
            x, x1_0 = self._embed(b, ids, Sd, M), None
        for l in range(L):
            lp, tag, site = f"decoder.layers.{l}.", f"L{l}.", DEC_SITE + 16 * l
            if l == 0 and x1_0 is not None:
                x1 = x1_0
            else:
                x1 = self._self_block(b, l, x, Bn, Sd, kpm)
            x2 = self._attn_ln_fwd(b, tag + "ca.", tag + "n2.", lp + "multihead_attn.", lp + "norm2.", x1, mem, Bn, Sd, Te, False, None,
                                   site + 3, site + 4, self_attn=False)
            f = self._ffn_fwd(b, tag + "ff.", lp, x2, site + 5)
            if l == L - 1:       # norm3 of the last layer + the stack-final norm: one launch
                x, y = self._ln_ln_fwd(b, tag + "n3.", lp + "norm3.", f, x2, site + 6, "nf.", "decoder.norm.")
                self._kv_prefetched = None
                b.t["x_last"] = x
                return y
            x = self._ln_fwd(b, tag + "n3.", lp + "norm3.", f, x2, site + 6)
        self._kv_prefetched = None
        b.t["x_last"] = x
        return self._ln_fwd(b, "nf.", "decoder.norm.", x, None, None)

    def forward(self, mem: torch.Tensor, Bn: int, Te: int, ids: torch.Tensor, training: bool, want_logits=False):
        """mem [B*Te, d] compute dtype; ids int64 [B,S] (pads = pad_id).  Returns (loss[1] fp32, logits or None).
        The logits gradient is produced in the same pass (in place when logits are not requested)."""
        pad = self.cfg["pad_id"]
        S = ids.shape[1]
        Sd, M = S - 1, Bn * (S - 1)
        self.p_drop = self.cfg["dropout"] if training else 0.0
        b = self.buf((Bn, Te, S))
        self.cur, self.shape = b, (Bn, Te, S)
        b.t["ids"], b.t["mem"] = ids, mem
        kpm = ("ids", ids, pad)          # tgt_padding_mask[:, :-1] == (ids[:, :Sd] == pad), evaluated inside the attention kernel
        b.t["kpm"] = kpm
        self._wgt = None
        if training and self.gen_dx_nt and self.dt == torch.bfloat16 and self.dev.type == "cuda" and self.ps.dw_adam is None:
            # (Not with the optimizer inside the vocabulary weight-gradient GEMM: an epilogue that also emitted W_g^T -- per-wave LDS
            # transposition, 16-byte pieces of the transposed rows -- took that product from 328 to 413 us for a dX that is 40 us
            # faster in the NT form; round 5, gpurun_out/r5f.  dX then runs in its NN form.)
            # dX = dlogits W_g in the K-contiguous NT form (persistent 256x256 kernel, split over K): needs W_g^T, which the
            # parameter set keeps beside the bf16 shadow -- rewritten right after the optimizer has touched W_g (62 MB of traffic
            # in the main stream's slack at the end of the step), not here in front of the latency-bound layer stack
            self._wgt = self.ps.want_transposed(self.pre + "generator.weight", eager=True)
        y = self._run_stack(b, mem, Bn, Te, ids, Sd, kpm)
        ops.tap("layers_fwd", 1)
        logits = b.get("logits", (M, self.Vp), self.dt)
        ops.gemm(y, self.W("generator.weight"), logits, bias=self.F("generator.bias"), n_valid=self.V, tag="gen_fwd")
        loss = b.get("loss", (1,), torch.float32)
        dlogits = b.get("dlogits", (M, self.Vp), self.dt) if want_logits else logits
        ops.tap("loss", 0)
        ops.sce_loss(logits, self.V, ids[:, 1:], Sd, pad, self.cfg["sce_loss_alpha"], loss, dlogits,
                     b.get("row_ws", (2 * M + 2,), torch.float32))
        ops.tap("loss", 1)
        b.t["dlogits_used"] = dlogits
        return loss, (logits if want_logits else None)

    def decode_word(self, mem: torch.Tensor, Bn: int, Te: int, ys: torch.Tensor) -> torch.Tensor:
        """Reference algorithm of CapDecoder.decode_word (CapDecoder.py:62-79): re-run the decoder over
        ALL t tokens so far (causal mask, no padding mask), generator on the last position -> [B, V]."""
        d, t = self.cfg["d"], ys.shape[1]
        self.p_drop = 0.0
        b = self.buf(("decode", Bn, Te, t))
        y = self._run_stack(b, mem, Bn, Te, ys, t, None)
        last = y.view(Bn, t, d)[:, t - 1, :]           # strided [B, d] view (lda = t*d): no gather copy
        logits = b.get("logits1", (Bn, self.Vp), self.dt)
        ops.gemm(last, self.W("generator.weight"), logits, bias=self.F("generator.bias"), n_valid=self.V)
        return logits[:, :self.V]

    def backward(self, bucket_ready=None, on_dmem_ready=None, join: bool = True) -> torch.Tensor:
        """d(loss) = 1.  Returns d(memory) [B*Te, d].  bucket_ready(kind, layer) is called when a gradient bucket
        of MMT4Caption.grad_buckets() is complete ('generator', 'dec_layer' l, 'embedding').  on_dmem_ready(dmem) is
        called as soon as the last cross-attention backward has been enqueued -- d(memory) is final there, while the
        bottom layer's self-attention backward and the embedding gradient are still to come."""
        b = self.cur
        Bn, Te, S = self.shape
        d, L, pad = self.cfg["d"], self.cfg["layers"], self.cfg["pad_id"]
        Sd, M = S - 1, Bn * (S - 1)
        mem, ids, kpm = b.t["mem"], b.t["ids"], b.t["kpm"]
        dl, y = b.t["dlogits_used"], b.t["nf.y"]
        dy = b.get("dy", (M, d), self.dt)
        if getattr(self, "_wgt", None) is not None:
            # fp32 partials of the split over the vocabulary-long K: room for the 6-way split of cfg-B / the 5-way one at batch 1024
            ws = b.get("gen_dx_ws", (6 * dy.shape[0] * dy.shape[1],), torch.float32)
            ops.gemm(dl, self._wgt, dy, ta=False, tb=True, k_valid=self.V, workspace=ws, tag="gen_dx")
        else:
            ops.gemm(dl, self.W("generator.weight"), dy, ta=False, tb=False, k_valid=self.V, workspace=self.gemm_ws(), tag="gen_dx")
        # the vocabulary weight gradient: with a gradient exchange it goes out first (its bucket is a third of the bytes and
        # can be on the wire during the whole backward); without one and with the encoder backward on the side stream it
        # is DEFERRED to the end of the main stream's tail, where that stream would otherwise idle -- beside the decoder's
        # dX chain it slowed the critical path (a 34 us GEMM took 123 us next to it)
        defer_gen_dw = self.defer_gen_dw and bucket_ready is None and on_dmem_ready is not None
        early_gen_dw = self.early_gen_dw and bucket_ready is None and on_dmem_ready is not None
        if early_gen_dw:      # A/B: right behind the dX GEMM on the main stream, alone on the chip (nothing runs on the side stream yet)
            defer_gen_dw = False
            ops.gemm(dl, y, self.G("generator.weight"), ta=True, tb=False, bias_grad=self.G("generator.bias"), m_valid=self.V,
                     tag="gen_dw", workspace=self.gemm_ws(), adam=self.dw_adam_desc(dl, self.G("generator.weight")))

        def gen_dw():
            if early_gen_dw:
                return
            if defer_gen_dw:
                # (Measured and dropped, round 5: this product at ONE workgroup per CU -- 40 KB of idle dynamic LDS -- so that the encoder
                # backward's short kernels on the side stream find free registers beside it: the product went 320 -> 426 us and the
                # step 2.254 -> 2.289 ms.)
                ops.gemm(dl, y, self.G("generator.weight"), ta=True, tb=False, bias_grad=self.G("generator.bias"), m_valid=self.V,
                         tag="gen_dw", workspace=self.gemm_ws(), adam=self.dw_adam_desc(dl, self.G("generator.weight")))
            else:
                self.dw_gemm(dl, y, self.G("generator.weight"), bias_grad=self.G("generator.bias"), m_valid=self.V, tag="gen_dw")
        if not defer_gen_dw:
            gen_dw()
        if bucket_ready is not None:
            self.bucket_on_side(bucket_ready, "generator")
        if not self.fuse_ln_ln_bwd:
            dx, _ = self._ln_bwd(b, "nf.", "decoder.norm.", dy, b.t["x_last"], None, None)
        dmem = b.get("dmem", (Bn * Te, d), self.dt)
        for l in reversed(range(L)):
            lp, tag, site = f"decoder.layers.{l}.", f"L{l}.", DEC_SITE + 16 * l
            x, x1, x2 = b.t[tag + "x"], b.t[tag + "n1.y"], b.t[tag + "n2.y"]
            if l == L - 1 and self.fuse_ln_ln_bwd:      # stack-final norm + this layer's norm3: one launch
                ds3, df = self._ln_ln_bwd(b, "nf.", "decoder.norm.", dy, b.t["x_last"], tag + "n3.", lp + "norm3.", b.t[tag + "ff.f"], x2,
                                          site + 6)
            else:
                ds3, df = self._ln_bwd(b, tag + "n3.", lp + "norm3.", dx, b.t[tag + "ff.f"], x2, site + 6)
            dx2 = self._ffn_bwd(b, tag + "ff.", lp, df, x2, site + 5, ds3)
            ds2, dc = self._ln_bwd(b, tag + "n2.", lp + "norm2.", dx2, b.t[tag + "ca.a"], x1, site + 4)
            dx1 = self._attn_block_bwd(b, tag + "ca.", lp + "multihead_attn.", dc, x1, mem, Bn, Sd, Te, False, None, site + 3,
                                       False, ds2, dkv_out=dmem, dkv_accumulate=(l != L - 1))
            dmem_point = None
            if l == 0 and on_dmem_ready is not None:
                # d(memory) is final once everything enqueued so far has run: remember that point; the encoder backward is
                # ENQUEUED after this layer's short tail (the host needs ~0.3 ms to launch its ~35 kernels, during which the
                # main stream would starve) but only WAITS for this point
                dmem_point = DMEM_SYNC
                ops.sync_record(dmem_point)
                # the bottom layer's weight gradients run on the MAIN stream (whose tail is not the critical path any
                # more): the side stream is free for the encoder backward the moment d(memory) is final
                # (With the optimizer inside the weight-gradient GEMMs this group REWRITES the layer's weights, W_kv among them, which
                # the layer's d(memory) product -- queued on the SIDE stream -- reads: the group then waits for the end of the layer's
                # chain, where the main stream first joins the side stream; tests/test_executor_gpu.py delays the side stream to show it.)
                if self.ps.dw_adam is None:
                    self.flush_dw(main=self.l0_dw_main & 1 != 0)
            ds1, da = self._ln_bwd(b, tag + "n1.", lp + "norm1.", dx1, b.t[tag + "sa.a"], x, site + 2)
            dx = self._attn_block_bwd(b, tag + "sa.", lp + "self_attn.", da, x, x, Bn, Sd, Sd, True, kpm, site + 1, True, ds1)
            if l == 0 and on_dmem_ready is not None:
                if self.ps.dw_adam is not None and self.side is not None and self.overlap_dw and self.l0_dw_main:
                    ops.stream_wait(None, self.side)          # the d(memory) products are behind us: one group of seven on the main stream
                    self.flush_dw(main=True)
                else:
                    self.flush_dw(main=self.l0_dw_main & 2 != 0)
                if bucket_ready is not None:
                    self.flush_ln_grads(b)
                    # through the side stream like every other bucket: the hook's optimizer step REWRITES this layer's weights (and
                    # their bf16 shadow), which the layer's d(memory) GEMM -- still queued on the side stream -- reads.  Handing the
                    # bucket over from the main stream alone let Adam overtake that GEMM (found by the four-rank one-GPU test: wrong
                    # encoder gradients / parameter updates on boxes where the side stream lagged)
                    self.bucket_on_side(bucket_ready, "dec_layer", l)
                self.flush_ln_grads(b)
                ops.embed_bwd(ids, Sd, pad, dx, self.G("tgt_to_emb.weight"), dropout=self.drop(EMB_SITE),
                              exclusive=self.exclusive_grads and bucket_ready is None, on_grow=self._ws_grew)
                if bucket_ready is not None:
                    bucket_ready("embedding")
                if defer_gen_dw:
                    gen_dw()
                on_dmem_ready(dmem, dmem_point)           # the encoder backward goes to the side stream now
                if join:
                    self.join_side()
                return dmem
            self.flush_dw()               # this layer's weight gradients: one grouped launch beside the next layer
            if bucket_ready is not None:      # this layer's (and, for the top layer, the final norm's) gradients are complete
                self.flush_ln_grads(b)
                self.bucket_on_side(bucket_ready, "dec_layer", l)
        self.flush_ln_grads(b)
        ops.embed_bwd(ids, Sd, pad, dx, self.G("tgt_to_emb.weight"), dropout=self.drop(EMB_SITE),
                      exclusive=self.exclusive_grads and bucket_ready is None, on_grow=self._ws_grew)
        if bucket_ready is not None:
            bucket_ready("embedding")
        if join:
            self.join_side()
        return dmem


class DecodeState:
    """Static buffers of one greedy-decode session (B, Te, Lmax): id matrix, per-layer self-attention
    cache [B, Lmax, 3d] (packed q | k | v of every consumed token), per-layer cross-attention K/V of the memory (computed once), step temporaries
    and the hipGraphs of the per-token step (one per position; every kernel argument is baked)."""

    def __init__(self, eng: "DecoderEngine", Bn: int, Te: int, Lmax: int):
        d, L, dt, dev = eng.cfg["d"], eng.cfg["layers"], eng.dt, eng.dev
        self.B, self.Te, self.Lmax = Bn, Te, Lmax
        self.ys = torch.zeros(Bn, Lmax, dtype=torch.long, device=dev)
        self.ended = torch.zeros(Bn, dtype=torch.bool, device=dev)
        self.ended_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.all_ended_at = torch.full((1,), Lmax, dtype=torch.long, device=dev)   # first t at which every row had ended
        self.kv_self = [torch.zeros(Bn * Lmax, 3 * d, dtype=dt, device=dev) for _ in range(L)]    # [q | k | v] per slot
        self.kv_cross = [torch.empty(Bn * Te, 2 * d, dtype=dt, device=dev) for _ in range(L)]
        self.b = _Buf(dev, eng.ps.ctx)
        self.graphs = {}


def _decoder_decode_begin(self, st: DecodeState, mem: torch.Tensor, start_id: int, pad_id: int):
    """Encoder memory -> cross-attention K/V of every layer (once per decode); reset ids / cache."""
    d = self.cfg["d"]
    st.ys.fill_(pad_id)
    st.ys[:, 0] = start_id
    st.ended.zero_()
    st.ended_count.zero_()
    st.all_ended_at.fill_(st.Lmax)
    for l in range(self.cfg["layers"]):
        lp = f"decoder.layers.{l}.multihead_attn."
        ops.gemm(mem, self.W(lp + "in_proj_weight")[d:], st.kv_cross[l], bias=self.F(lp + "in_proj_bias")[d:])


def _decoder_decode_step(self, st: DecodeState, t: int, end_id: int):
    """One greedy step with the KV cache: consumes token ys[:, t-1], writes ys[:, t].  Equivalent to
    CapDecoder.decode_word on ys[:, :t] + torch.max (CapDecoder.py:62-79, MMT4Caption.py:164-172): the
    keys/values of positions < t-1 are the cached projections of the same inputs."""
    d, H, L, Bn, Te, Lmax = self.cfg["d"], self.cfg["nhead"], self.cfg["layers"], st.B, st.Te, st.Lmax
    b = st.b
    ws = self.gemm_ws()      # split-K over the reduction for these M = batch GEMMs (a few output tiles, long K), reduced in-kernel
    self.p_drop = 0.0
    pos_row = self.pos[t - 1:t]                       # positional row of the consumed token
    x = ops.embed_fwd(st.ys[:, t - 1:t], 1, self.F("tgt_to_emb.weight"), pos_row, b.get("x0", (Bn, d), self.dt))
    for l in range(L):
        lp, tag = f"decoder.layers.{l}.", f"L{l}."
        sa = lp + "self_attn."
        # ONE packed projection per token: q, k, v land in slot t-1 of the cache [B, Lmax, 3d] (k, v stay there for the
        # later steps; q is read once, through the same row stride)
        cache = st.kv_self[l]
        qkv_new = cache.view(Bn, Lmax, 3 * d)[:, t - 1, :]                  # [B, 3d] view, row stride Lmax*3d
        ops.gemm(x, self.W(sa + "in_proj_weight"), qkv_new, bias=self.F(sa + "in_proj_bias"), workspace=ws)
        o = b.get(tag + "o", (Bn, d), self.dt)
        ops.attn_fwd(qkv_new[:, :d], cache[:, d:2 * d], cache[:, 2 * d:], o, Bn, H, 1, t, kv_batch_stride=Lmax * 3 * d)
        a = b.get(tag + "a", (Bn, d), self.dt)
        ops.gemm(o, self.W(sa + "out_proj.weight"), a, bias=self.F(sa + "out_proj.bias"), workspace=ws)
        x1 = self._ln_fwd(b, tag + "n1.", lp + "norm1.", a, x, None)
        ca = lp + "multihead_attn."
        qc = b.get(tag + "qc", (Bn, d), self.dt)
        ops.gemm(x1, self.W(ca + "in_proj_weight")[:d], qc, bias=self.F(ca + "in_proj_bias")[:d], workspace=ws)
        oc = b.get(tag + "oc", (Bn, d), self.dt)
        ops.attn_fwd(qc, st.kv_cross[l][:, :d], st.kv_cross[l][:, d:], oc, Bn, H, 1, Te)
        c = b.get(tag + "c", (Bn, d), self.dt)
        ops.gemm(oc, self.W(ca + "out_proj.weight"), c, bias=self.F(ca + "out_proj.bias"), workspace=ws)
        x2 = self._ln_fwd(b, tag + "n2.", lp + "norm2.", c, x1, None)
        h = b.get(tag + "h", (Bn, self.cfg["ff"]), self.dt)
        ops.gemm(x2, self.W(lp + "linear1.weight"), h, bias=self.F(lp + "linear1.bias"), act=self.cfg["activation"], workspace=ws)
        f = b.get(tag + "f", (Bn, d), self.dt)
        ops.gemm(h, self.W(lp + "linear2.weight"), f, bias=self.F(lp + "linear2.bias"), workspace=ws)
        x = self._ln_fwd(b, tag + "n3.", lp + "norm3.", f, x2, None)
    y = self._ln_fwd(b, "nf.", "decoder.norm.", x, None, None)
    logits = b.get("logits", (Bn, self.Vp), self.dt)
    ops.gemm(y, self.W("generator.weight"), logits, bias=self.F("generator.bias"), n_valid=self.V, workspace=ws)
    # arg-max into column t + sticky end flags + the first step at which every row has ended: one launch, no host sync
    st.last_logits = logits          # [B, Vp] of this step (decode.teacher_forced_next_ids reads it)
    ops.greedy_select(logits, st.ys[:, t], end_id, st.ended, st.ended_count, st.all_ended_at, t, cols=self.V)


def _decoder_small_decode_ok(self, st: DecodeState) -> bool:
    """The weight-streaming GEMV step (ops.decode_gemv): model widths that are whole 16-byte lane rows (d, ff multiples of
    512 for bf16 / 256 for fp32, <= 2048), <= 64 cached positions, and batch 1 -- measured per token at cfg-B
    (tools/decode_probe.py, graph replay): batch 1 97 us vs 114 us on the batched MFMA kernels, batch 2 135 vs 116 (every
    workgroup recomputes the attention of all (batch, head) pairs and reduces 8 x B dot products per trip), so every larger
    batch stays on the batched kernels."""
    ki = 512 if self.dt == torch.bfloat16 else 256
    d, ff, H = self.cfg["d"], self.cfg["ff"], self.cfg["nhead"]
    # the C side's limits (csrc/vct_decode.hip, vct_decode_gemv): K / ki chunks per lane in {1, 2, 4} (8: fp32 only), the
    # LayerNorm prologues hold a whole row of K = d <= 1024 in registers, head_dim a whole number of 16-byte vectors.  Any
    # other width (the shipped d = 768 in fp32: 3 chunks) takes the batched step instead of raising VCT_E_SHAPE.
    def chunks_ok(k):
        return k % ki == 0 and (k // ki in (1, 2, 4) or (k // ki == 8 and self.dt == torch.float32))
    vec = 8 if self.dt == torch.bfloat16 else 4
    return (self.small_batch_decode and st.B <= 1 and chunks_ok(d) and chunks_ok(ff) and d <= 1024 and (d // H) % vec == 0
            and st.Lmax <= 64 and st.Te <= 64 and self.dev.type == "cuda")


def _decoder_decode_step_small(self, st: DecodeState, t: int, end_id: int):
    """The same step as _decoder_decode_step in 6 launches per layer + 2: embedding, LayerNorms and both attention cores run in the
    prologues of the matrix-vector kernels that consume them, residual adds in the epilogues of the producers; activations
    between stages are fp32 vectors (pre-norm sums s, normalised x kept for the next residual)."""
    d, H, L, B, Te, Lmax, ff = self.cfg["d"], self.cfg["nhead"], self.cfg["layers"], st.B, st.Te, st.Lmax, self.cfg["ff"]
    b = st.b
    f32 = torch.float32
    x, s, x1, x2 = (b.get(n, (B, d), f32) for n in ("sx", "ss", "sx1", "sx2"))
    s2, s3 = b.get("ss2", (B, d), f32), b.get("ss3", (B, d), f32)
    qc = b.get("sqc", (B, d), self.dt)
    h = b.get("sh", (B, ff), f32)
    prev_norm = None
    for l in range(L):
        lp = f"decoder.layers.{l}."
        sa, ca = lp + "self_attn.", lp + "multihead_attn."
        cache = st.kv_self[l]                                              # [B * Lmax, 3d]: q | k | v of every consumed token
        slot = cache.view(B, Lmax, 3 * d)[:, t - 1, :]
        if l == 0:      # x = Emb[ys[:, t-1]] + pos[t-1]  ->  q | k | v into slot t-1
            ops.decode_gemv(self.W(sa + "in_proj_weight"), slot, B, bias=self.F(sa + "in_proj_bias"), pro="embed",
                            embed=(st.ys[:, t - 1], self.F("tgt_to_emb.weight"), self.pos[t - 1]), out_native=True, x_out=x)
        else:           # x = norm3 of the layer below
            ops.decode_gemv(self.W(sa + "in_proj_weight"), slot, B, bias=self.F(sa + "in_proj_bias"), pro="ln", x_in=s3, ln1=prev_norm,
                            out_native=True, x_out=x)
        # s = x + out_proj(self-attention over the t cached positions)
        ops.decode_gemv(self.W(sa + "out_proj.weight"), s, B, bias=self.F(sa + "out_proj.bias"), pro="self_attn",
                        attn=(slot[:, :d], cache[:, d:2 * d], cache[:, 2 * d:], 3 * d, Lmax * 3 * d, H, t), res=x)
        # x1 = norm1(s); cross-attention query
        ops.decode_gemv(self.W(ca + "in_proj_weight")[:d], qc, B, bias=self.F(ca + "in_proj_bias")[:d], pro="ln", x_in=s,
                        ln1=(self.F(lp + "norm1.weight"), self.F(lp + "norm1.bias")), out_native=True, x_out=x1)
        kvc = st.kv_cross[l]                                               # [B * Te, 2d]
        ops.decode_gemv(self.W(ca + "out_proj.weight"), s2, B, bias=self.F(ca + "out_proj.bias"), pro="cross_attn",
                        attn=(qc, kvc[:, :d], kvc[:, d:], 2 * d, Te * 2 * d, H, Te), res=x1)
        # x2 = norm2(s2); feed-forward
        ops.decode_gemv(self.W(lp + "linear1.weight"), h, B, bias=self.F(lp + "linear1.bias"), pro="ln", x_in=s2,
                        ln1=(self.F(lp + "norm2.weight"), self.F(lp + "norm2.bias")), act=self.cfg["activation"], x_out=x2)
        ops.decode_gemv(self.W(lp + "linear2.weight"), s3, B, bias=self.F(lp + "linear2.bias"), pro="none", x_in=h, res=x2)
        prev_norm = (self.F(lp + "norm3.weight"), self.F(lp + "norm3.bias"))
    logits = b.get("slogits", (B, self.Vp), f32)
    ops.decode_gemv(self.W("generator.weight"), logits, B, bias=self.F("generator.bias"), pro="ln_ln", x_in=s3, ln1=prev_norm,
                    ln2=(self.F("decoder.norm.weight"), self.F("decoder.norm.bias")), n_valid=self.V)
    st.last_logits = logits          # [B, Vp] of this step (decode.teacher_forced_next_ids reads it)
    ops.greedy_select(logits, st.ys[:, t], end_id, st.ended, st.ended_count, st.all_ended_at, t, cols=self.V)


def _decoder_block_decode_ok(self, st: DecodeState) -> bool:
    """The batch-1 step with one launch per layer BLOCK (ops.decode_block): bf16, d = 512 with head_dim 64, ff <= 2048, <= 64 positions
    (vct_decode_block_supported is the authority: anything else falls through to the gemv / skinny / batched steps)."""
    d, ff, H = self.cfg["d"], self.cfg["ff"], self.cfg["nhead"]
    return (self.block_decode and st.B == 1 and self.dev.type == "cuda" and st.Lmax <= 64 and st.Te <= 64
            and ops.decode_block_supported(self.dt, d, H, ff, min(st.Lmax, 64)))


def _decoder_decode_step_block(self, st: DecodeState, t: int, end_id: int):
    """The step of _decoder_decode_step for ONE caption in 3 launches per layer + 1: self-attention block, cross-attention block,
    feed-forward block (each: first product + attention / activation + the second product split over the workgroups that own the
    first, as partial vectors), generator (its last workgroup also selects the token).  The partial vectors, the residual, the second product's bias and the
    LayerNorm(s) are folded by the prologue of the next launch (csrc/vct_decode_block.hip)."""
    d, H, L, Te, Lmax, ff = self.cfg["d"], self.cfg["nhead"], self.cfg["layers"], st.Te, st.Lmax, self.cfg["ff"]
    b = st.b
    f32 = torch.float32
    xa, x1, x2 = b.get("kx", (d,), f32), b.get("kx1", (d,), f32), b.get("kx2", (d,), f32)
    a_part, c_part, f_part = b.get("ka", (H, d), f32), b.get("kc", (H, d), f32), b.get("kf", (ff // 64, d), f32)
    prev = None                                   # (bias of linear2, norm3) of the layer below
    for l in range(L):
        lp = f"decoder.layers.{l}."
        sa, ca = lp + "self_attn.", lp + "multihead_attn."
        cache = st.kv_self[l]                                              # [Lmax, 3d]: q | k | v of every consumed token
        slot = cache[t - 1]
        src = (dict(embed=(st.ys[0, t - 1:t], self.F("tgt_to_emb.weight"), self.pos[t - 1])) if prev is None else
               dict(res=x2, res_bias=prev[0], part=f_part, ln1=prev[1]))
        ops.decode_block("self", d, w_a=self.W(sa + "in_proj_weight"), b_a=self.F(sa + "in_proj_bias"), slot=slot,
                         kc=cache[:, d:2 * d], vc=cache[:, 2 * d:], kv_ld=3 * d, Lk=t, w_b=self.WT(sa + "out_proj.weight"),
                         part_out=a_part, x_out=xa, **src)
        kvc = st.kv_cross[l]                                               # [Te, 2d]
        ops.decode_block("cross", d, res=xa, res_bias=self.F(sa + "out_proj.bias"), part=a_part,
                         ln1=(self.F(lp + "norm1.weight"), self.F(lp + "norm1.bias")), x_out=x1,
                         w_a=self.W(ca + "in_proj_weight")[:d], b_a=self.F(ca + "in_proj_bias")[:d], kc=kvc[:, :d], vc=kvc[:, d:],
                         kv_ld=2 * d, Lk=Te, w_b=self.WT(ca + "out_proj.weight"), part_out=c_part)
        ops.decode_block("ffn", d, res=x1, res_bias=self.F(ca + "out_proj.bias"), part=c_part,
                         ln1=(self.F(lp + "norm2.weight"), self.F(lp + "norm2.bias")), x_out=x2,
                         w_a=self.W(lp + "linear1.weight"), b_a=self.F(lp + "linear1.bias"), w_b=self.WT(lp + "linear2.weight"),
                         ff=ff, act=self.cfg["activation"], part_out=f_part)
        prev = (self.F(lp + "linear2.bias"), (self.F(lp + "norm3.weight"), self.F(lp + "norm3.bias")))
    logits = b.get("klogits", (1, self.Vp), f32)
    sel = b.t.get("ksel")
    if sel is None or sel.numel() < 2 * ((self.V + 127) // 128) + 1:
        sel = b.get("ksel", (2 * ((self.V + 127) // 128) + 1,), f32)
        sel.zero_()                               # the ticket counter: every launch leaves it at zero again
    ops.decode_block("gen", d, res=x2, res_bias=prev[0], part=f_part, ln1=prev[1],
                     ln2=(self.F("decoder.norm.weight"), self.F("decoder.norm.bias")), w_a=self.W("generator.weight"),
                     b_a=self.F("generator.bias"), V=self.V, part_out=logits,
                     select=(sel, st.ys[0, t:t + 1], end_id, st.ended, st.ended_count, st.all_ended_at, t))
    st.last_logits = logits


def _decoder_fused_decode_ok(self, st: DecodeState) -> bool:
    """The batched step with LayerNorms folded into the consuming projections (ops.decode_linear): bf16, up to 256 captions in
    flight, model width <= 1024 (a row's statistics come out of one pass over the MFMA fragments)."""
    d, ff = self.cfg["d"], self.cfg["ff"]
    return (self.fused_decode and self.dt == torch.bfloat16 and 2 <= st.B <= 256 and d % 32 == 0 and d <= 1024 and ff % 32 == 0
            and self.dev.type == "cuda")


def _decoder_decode_step_fused(self, st: DecodeState, t: int, end_id: int):
    """The same step as _decoder_decode_step in 8 launches per layer + 3 instead of 11 + 4: every projection is one skinny MFMA
    kernel (M = batch rows, K split over the waves); norm1 / norm2 / norm3 run as the prologue of the projection that consumes
    them (which also stores the normalised rows once, for the residual two launches later), the residual adds in the epilogues;
    the sums that feed a LayerNorm stay fp32."""
    d, H, L, Bn, Te, Lmax, ff = self.cfg["d"], self.cfg["nhead"], self.cfg["layers"], st.B, st.Te, st.Lmax, self.cfg["ff"]
    b = st.b
    f32 = torch.float32
    act = self.cfg["activation"]
    xin, xres, prev_norm = None, None, None      # layer input: the embedded tokens (layer 0) or (pre-norm sum, norm3 of the layer below)
    for l in range(L):
        lp, tag = f"decoder.layers.{l}.", f"F{l}."
        sa, ca = lp + "self_attn.", lp + "multihead_attn."
        cache = st.kv_self[l]
        slot = cache.view(Bn, Lmax, 3 * d)[:, t - 1, :]                     # q | k | v of the consumed token
        xres = b.get(tag + "xn", (Bn, d), f32)
        if prev_norm is None:                    # x = Emb[ys[:, t-1]] + pos[t-1], built in the projection's prologue
            ops.decode_linear(self.W(sa + "in_proj_weight"), slot, embed=(st.ys[:, t - 1], self.F("tgt_to_emb.weight"), self.pos[t - 1]),
                              x_norm=xres, bias=self.F(sa + "in_proj_bias"))
        else:
            ops.decode_linear(self.W(sa + "in_proj_weight"), slot, x_pre=xin, ln=prev_norm, x_norm=xres, bias=self.F(sa + "in_proj_bias"))
        o = b.get(tag + "o", (Bn, d), self.dt)
        ops.attn_fwd(slot[:, :d], cache[:, d:2 * d], cache[:, 2 * d:], o, Bn, H, 1, t, kv_batch_stride=Lmax * 3 * d)
        s1 = b.get(tag + "s1", (Bn, d), f32)                                # x + self-attention block
        ops.decode_linear(self.W(sa + "out_proj.weight"), s1, x=o, bias=self.F(sa + "out_proj.bias"), res=xres)
        x1 = b.get(tag + "x1", (Bn, d), f32)
        qc = b.get(tag + "qc", (Bn, d), self.dt)
        ops.decode_linear(self.W(ca + "in_proj_weight")[:d], qc, x_pre=s1, ln=(self.F(lp + "norm1.weight"), self.F(lp + "norm1.bias")),
                          x_norm=x1, bias=self.F(ca + "in_proj_bias")[:d])
        oc = b.get(tag + "oc", (Bn, d), self.dt)
        ops.attn_fwd(qc, st.kv_cross[l][:, :d], st.kv_cross[l][:, d:], oc, Bn, H, 1, Te)
        s2 = b.get(tag + "s2", (Bn, d), f32)
        ops.decode_linear(self.W(ca + "out_proj.weight"), s2, x=oc, bias=self.F(ca + "out_proj.bias"), res=x1)
        x2 = b.get(tag + "x2", (Bn, d), f32)
        h = b.get(tag + "h", (Bn, ff), self.dt)
        ops.decode_linear(self.W(lp + "linear1.weight"), h, x_pre=s2, ln=(self.F(lp + "norm2.weight"), self.F(lp + "norm2.bias")),
                          x_norm=x2, bias=self.F(lp + "linear1.bias"), act=act)
        s3 = b.get(tag + "s3", (Bn, d), f32)
        ops.decode_linear(self.W(lp + "linear2.weight"), s3, x=h, bias=self.F(lp + "linear2.bias"), res=x2)
        xin, prev_norm = s3, (self.F(lp + "norm3.weight"), self.F(lp + "norm3.bias"))
    y = b.get("fy", (Bn, d), self.dt)
    ops.decode_ln2(xin, prev_norm, (self.F("decoder.norm.weight"), self.F("decoder.norm.bias")), y)
    logits = b.get("logits", (Bn, self.Vp), self.dt)
    ops.gemm(y, self.W("generator.weight"), logits, bias=self.F("generator.bias"), n_valid=self.V, workspace=self.gemm_ws())
    st.last_logits = logits          # [B, Vp] of this step (decode.teacher_forced_next_ids reads it)
    ops.greedy_select(logits, st.ys[:, t], end_id, st.ended, st.ended_count, st.all_ended_at, t, cols=self.V)


def _decoder_decode_step_any(self, st: DecodeState, t: int, end_id: int):
    if _decoder_block_decode_ok(self, st):
        return _decoder_decode_step_block(self, st, t, end_id)
    if _decoder_small_decode_ok(self, st):
        return _decoder_decode_step_small(self, st, t, end_id)
    if _decoder_fused_decode_ok(self, st):
        return _decoder_decode_step_fused(self, st, t, end_id)
    return _decoder_decode_step(self, st, t, end_id)


# A/B switch: vocabulary dX through a transposed weight shadow (NT form on the persistent 256x256 kernel, split over K).  The shadow is
# maintained by the parameter set (ParamSet.want_transposed): one 35 us transpose behind the optimizer's pass over W_g, in the main
# stream's slack at the end of the step.  (Rebuilt in front of the layer stack at every step it cost the forward more than the dX
# gained.)  Measured in the step (same box): the dX bracket drops 0.218 -> 0.181 ms and the Adam bracket grows by the 40 us of the
# transpose; step 2.44-2.45 ms either way (the chip is work-bound: the side stream fills whatever the main stream leaves) -> off.
# (round 6: the NN form on the pipelined 256x256 kernel runs at the NT form's speed -- 0.183 ms both -- so the transposed shadow W_g^T
# (a 39 us transpose + the 2-D optimizer pass per step behind a gradient exchange) is no longer kept by default: exchange path -0.8 %)
DecoderEngine.gen_dx_nt = os.environ.get("VCT_GEN_DX_NT", "0") != "0"
EncoderEngine.enc_dw_main = int(os.environ.get("VCT_ENC_DW_MAIN", "1"))
DecoderEngine.early_gen_dw = os.environ.get("VCT_GEN_DW_EARLY", "0") == "1"
# bit 0 / bit 1: the bottom decoder layer's cross-attention + feed-forward / self-attention weight gradients on the MAIN stream (A/B)
DecoderEngine.l0_dw_main = int(os.environ.get("VCT_L0_DW_MAIN", "3"))
DecoderEngine.fused_decode = True             # A/B switch: LayerNorms folded into the skinny projections (2 <= batch <= 256, bf16)


DecoderEngine.block_decode = os.environ.get("VCT_BLOCK_DECODE", "1") != "0"   # A/B switch: 3 launches per layer at batch 1 (bf16)
DecoderEngine.small_batch_decode = True       # A/B switch: weight-streaming GEMV step for batch <= 4
DecoderEngine.decode_begin = _decoder_decode_begin
DecoderEngine.decode_step = _decoder_decode_step_any
