"""MMT4Caption -- drop-in for the reference's model/MMT4Caption.py:15-211 on the caption task.

Same constructor (`MMT4Caption(cfg['model'], device)`), `forward(video_feats, video_masks, captions)`,
`caption_forward`, `greedy_decode`, `mode`, attributes (`cap_preprocessor`, `cap_decoder`,
`video_encoder`, `matching`, `device`, `f_type`) and state_dict keys.  All parameters live in ONE
flat fp32 buffer laid out in gradient-ready order (with a matching flat gradient buffer and a bf16
shadow), so the data-parallel gradient exchange and the optimizer work on contiguous slices."""
import os
from typing import List, Optional

import torch
import torch.nn as nn

from .. import ops
from ..engine import ParamSet
from .CapDecoder import CapDecoder, grad_ready_order_decoder
from .CapPreprocessor import CapPreprocessor
from .Matching import Matching, TextEncoder
from .MMEncoder import MultiModalEncoder, grad_ready_order_encoder

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}


class _CaptionFn(torch.autograd.Function):
    """loss = caption_forward(...) as ONE autograd node: forward = encoder + decoder + loss kernels,
    backward = the explicit reverse schedule; gradients land in the flat gradient buffer."""

    @staticmethod
    def forward(ctx, model, feats, mask, ids, *params):
        ctx.model = model
        return model._forward_loss(feats, mask, ids, model.training)[0]

    @staticmethod
    def backward(ctx, gloss):
        m = ctx.model
        m._backward()
        m._ps.install_grads()
        if not m._unit_loss_grad:
            m._ps.gflat.mul_(gloss)
        return (None, None, None, None) + (None,) * len(m._ps.names)


class MMT4Caption(nn.Module):
    overlap_enc_bwd = True      # encoder backward beside the decoder's tail (A/B switch)
    overlap_dec_prefix = True   # decoder embedding + bottom self-attention beside the encoder forward (A/B switch)

    def __init__(self, model_config: dict, device=torch.device("cuda"), compute_dtype=None):
        super().__init__()
        self.device = device
        self.model_config = model_config
        self.loss_beta = model_config["loss_beta"]
        self.f_type = None
        cd = compute_dtype or model_config.get("compute_dtype") or os.environ.get("VCT_COMPUTE_DTYPE", "bf16")
        self.compute_dtype = _DTYPES[cd] if isinstance(cd, str) else cd

        self.cap_preprocessor = CapPreprocessor(model_config["tokenizer"], device=device,
                                                vocab_size=model_config.get("vocab_size"))
        self.text_encoder = TextEncoder(model_config["text_enc_type"], device=device)
        dec_cfg, enc_cfg = model_config["caption_decoder"], model_config["video_encoder"]
        self.cap_decoder = CapDecoder(
            num_layers=dec_cfg["layer"], embed_dim=model_config["embed_dim"], nhead=dec_cfg["nhead"],
            dim_feedforward=dec_cfg["feedforward"], dropout=model_config["dropout"],
            vocab_size=self.cap_preprocessor.tokenizer.vocab_size, pad_id=self.cap_preprocessor.pad_id,
            sce_loss_alpha=dec_cfg["sce_loss_alpha"], custom_decoder_type=dec_cfg.get("layer_type", None),
            activation=model_config["activation"], device=device, compute_dtype=self.compute_dtype)
        if enc_cfg.get("type", "mme") != "mme":
            raise NotImplementedError("video_encoder.type 'simple'/'hmme' are outside the accelerated caption path")
        mme = enc_cfg["mme"]
        self.video_encoder = MultiModalEncoder(
            d_feats=model_config["modal_shape"], d_model=model_config["embed_dim"], nhead=enc_cfg["nhead"],
            dim_feedforward=enc_cfg["feedforward"], num_encoder_layers=enc_cfg["layer"], dropout=model_config["dropout"],
            activation=model_config["activation"], global_type=mme["aggregation"],
            modal_different=mme.get("modal_different", True), temporal_type=mme.get("temporal", "encoding"),
            do_norm=mme.get("do_norm", False), device=device, compute_dtype=self.compute_dtype)
        if model_config.get("matching", None) is not None:
            self.matching = Matching((model_config["embed_dim"], self.text_encoder.dim),
                                     enable_tem=model_config["matching"]["enable_tem"],
                                     loss=model_config["matching"]["matching_loss"],
                                     loss_tem=model_config["matching"].get("temperature", None), device=device)
        self._ps: Optional[ParamSet] = None
        self._unit_loss_grad = False
        self._seed = None
        self._build_flat()

    # ---- flat parameter storage --------------------------------------------------------------------
    def _build_flat(self):
        named = dict(self.named_parameters())
        order = (grad_ready_order_decoder("cap_decoder.", self.cap_decoder.cfg["layers"]) +
                 grad_ready_order_encoder("video_encoder.", self.video_encoder.cfg["layers"]))
        order += [n for n in named if n not in set(order)]   # matching.* (not on the caption path)
        dev = named[order[0]].device
        self._ps = ParamSet([(n, named[n]) for n in order], dev, self.compute_dtype, no_shadow=("cap_decoder.tgt_to_emb.weight",))
        if self._seed is None or self._seed.device != dev:
            self._seed = torch.tensor([torch.initial_seed() & 0x7FFFFFFF], dtype=torch.int32, device=dev)
        self.cap_decoder._bind(self._ps, "cap_decoder.", self._seed, self._build_flat)
        self.video_encoder._bind(self._ps, "video_encoder.", self._seed, self._build_flat)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if self._ps is not None and not self._ps.intact():
            self._build_flat()
        return out

    @property
    def flat_params(self) -> torch.Tensor:
        return self._ps.flat

    @property
    def flat_grads(self) -> torch.Tensor:
        return self._ps.gflat

    @property
    def grads_valid(self) -> bool:
        """True when every `.grad` / `flat_grads` element is the gradient of the LAST backward, as after the reference's
        `loss.backward()` (train.py:125).  False after a CaptionTrainer step that stepped the weight matrices inside their
        weight-gradient GEMMs without storing the gradients (single GPU, bf16, FusedAdam -- the default there): bias, LayerNorm and
        embedding gradients are current, the 2-D weights' are stale.  `CaptionTrainer(..., keep_weight_grads=True)` keeps all valid."""
        return self._ps.weight_grads_valid

    def check_grads_valid(self):
        """Raise if the weight-matrix gradients are stale (call before gradient clipping / logging / hooks that read .grad)."""
        if not self._ps.weight_grads_valid:
            raise RuntimeError("the last CaptionTrainer.step() stepped the weight matrices inside their weight-gradient GEMMs and did not "
                               "store those gradients: .grad / flat_grads of 2-D weights are stale; construct the trainer with "
                               "keep_weight_grads=True (or set VCT_FUSE_ADAM_KEEP_GRAD=1) to keep them")

    def grad_buckets(self):
        """Contiguous [start, end) element ranges of the flat gradient buffer in the order the backward
        pass completes them: generator | decoder norm + top layer | ... | decoder layer 0 | token embedding |
        encoder norm + top layer | ... | encoder layer 0 + unify (+ parameters outside the caption path)."""
        ps, o = self._ps, self._ps.offsets
        Ld, Le = self.cap_decoder.cfg["layers"], self.video_encoder.cfg["layers"]
        cuts = [0, o["cap_decoder.decoder.norm.weight"]]
        cuts += [o[f"cap_decoder.decoder.layers.{l}.norm3.weight"] for l in reversed(range(Ld - 1))]
        cuts += [o["cap_decoder.tgt_to_emb.weight"], o["video_encoder.transformer_encoder.norm.weight"]]
        cuts += [o[f"video_encoder.transformer_encoder.layers.{l}.norm2.weight"] for l in reversed(range(Le - 1))]
        cuts.append(ps.total)
        return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]

    def bucket_index(self, kind: str, layer: int = 0) -> int:
        """Index into grad_buckets(): kind in {'generator', 'dec_layer', 'embedding', 'enc_layer'}; `layer` is the
        layer whose backward just finished."""
        Ld, Le = self.cap_decoder.cfg["layers"], self.video_encoder.cfg["layers"]
        if kind == "generator":
            return 0
        if kind == "dec_layer":
            return 1 + (Ld - 1 - layer)
        if kind == "embedding":
            return 1 + Ld
        if kind == "enc_layer":
            return 2 + Ld + (Le - 1 - layer)
        raise ValueError(kind)

    # ---- engine-level forward/backward (no autograd) ------------------------------------------------
    def _forward_loss(self, feats, mask, ids, training, want_logits=False):
        if not self._ps.intact():
            self._build_flat()
        self._ps.refresh_shadow()
        enc, dec = self.video_encoder._engine(), self.cap_decoder._engine()
        ops.tap("layers_fwd", 0)      # bench.py north_star bracket: input cast .. decoder final LayerNorm (main stream)
        if self.overlap_dec_prefix and dec.dev.type == "cuda" and dec.overlap_dw:
            # token embedding + the decoder's bottom self-attention block do not need the encoder: side stream, beside it
            dec.forward_prefix(feats.shape[0], feats.shape[1] + 1, ids, training)
        mem = enc.forward(feats, mask, training)
        loss, logits = dec.forward(mem, feats.shape[0], feats.shape[1] + 1, ids, training, want_logits=want_logits)
        return loss, logits

    @property
    def encoder_param_begin(self) -> int:
        """Flat offset where the encoder's parameters (and whatever follows them) start: everything before it -- generator,
        decoder stack, token embedding -- has its final gradient before the encoder backward has finished."""
        return self.grad_buckets()[2 + self.cap_decoder.cfg["layers"]][0]

    @property
    def caption_param_end(self) -> int:
        """Flat offset where the parameters OUTSIDE the caption path (matching.*) start -- the end of what the caption
        task's optimizer owns (reference train.py:24: filter(requires_grad) after mode('caption'))."""
        ps = self._ps
        for n in ps.names:
            if not (n.startswith("cap_decoder.") or n.startswith("video_encoder.")):
                return ps.offsets[n]
        return ps.total

    def join_backward(self):
        """train_step_kernels(defer_join=True) leaves the encoder backward un-enqueued: enqueue it (side stream) if that
        has not happened yet, then make the main stream wait for it."""
        self.launch_encoder_backward()
        self.cap_decoder._engine().join_side()

    def launch_encoder_backward(self, main: bool = False):
        """main: enqueue it on the CURRENT stream (the caller has joined the side stream: d(memory) is final) -- for the one-launch
        sample-stationary backward, which takes whole compute units and gains nothing from running beside other kernels."""
        fn, self._pending_enc_bwd = getattr(self, "_pending_enc_bwd", None), None
        if fn is not None:
            fn(main)

    def encoder_backward_is_one_launch(self) -> bool:
        enc = self.video_encoder._engine()
        return enc.ss_bwd_ok()

    def _backward(self, bucket_ready=None, join: bool = True):
        hook = None
        if bucket_ready is not None:
            def hook(kind, layer=0):
                bucket_ready(self.bucket_index(kind, layer))
        dec, enc = self.cap_decoder._engine(), self.video_encoder._engine()
        if self.overlap_enc_bwd and dec.dev.type == "cuda" and dec.overlap_dw:
            # the encoder's backward only needs d(memory): it runs on the side stream beside the decoder's bottom
            # self-attention backward and the embedding gradient (two chains of small kernels share the chip)
            def launch(dmem, dmem_point, main=False):
                if main:
                    enc.backward(dmem, hook, join=False)
                    return
                side = dec.ensure_side()
                ops.sync_wait(dmem_point, side)                   # d(memory) final (its last accumulate is on `side` itself)
                enc.main_stream = torch.cuda.current_stream()     # (EncoderEngine.enc_dw_main: upper layers' weight-gradient groups go there)
                with torch.cuda.stream(side):
                    enc.backward(dmem, hook)
                enc.main_stream = None

            def on_dmem(dmem, dmem_point):
                if join:
                    launch(dmem, dmem_point)
                else:      # the caller enqueues its own main-stream work first (the host launches ~35 kernels here)
                    self._pending_enc_bwd = lambda main=False: launch(dmem, dmem_point, main)
            dec.backward(hook, on_dmem_ready=on_dmem, join=join)  # join: ends with the main stream joining the side stream
        else:
            dmem = dec.backward(hook)
            enc.backward(dmem, hook)

    def train_step_kernels(self, feats: torch.Tensor, mask: Optional[torch.Tensor], ids: torch.Tensor,
                           bucket_ready=None, defer_join: bool = False) -> torch.Tensor:
        """Fast path used by the trainer and bench: forward + backward as one static kernel schedule
        (hipGraph-capturable, no autograd tape).  Gradients are WRITTEN (not accumulated) into the flat
        gradient buffer, whose views are installed as `.grad`.  Returns the loss tensor [1]."""
        loss, _ = self._forward_loss(feats, mask, ids, self.training)
        self._backward(bucket_ready, join=not defer_join)     # defer_join: the caller calls join_backward() itself
        opt = self._ps.dw_adam                                # the optimizer epilogue consumed the weight gradients unless told to store them
        self._ps.weight_grads_valid = opt is None or bool(opt.keep_grads)
        return loss

    # ---- reference API -----------------------------------------------------------------------------
    def forward(self, video_feats: List[torch.Tensor], video_masks: List[torch.Tensor], captions):
        if self.f_type == "caption":
            return self.caption_forward(video_feats, video_masks, captions)
        if self.f_type in ("match", "cross"):
            raise NotImplementedError("the video-text matching task is outside the MI355X caption path")
        raise ValueError

    def caption_forward(self, video_feats, video_masks, captions):
        text_ts, _text_mask_ts = self.cap_preprocessor(captions)
        mask = video_masks[0] if video_masks is not None else None
        feats = video_feats[0]
        if not torch.is_grad_enabled():
            return self._forward_loss(feats, mask, text_ts, self.training)[0][0].clone()
        return _CaptionFn.apply(self, feats, mask, text_ts, *[self._ps.params[n] for n in self._ps.names]).clone().reshape(())

    @torch.no_grad()
    def greedy_decode(self, video_feat: List[torch.Tensor], video_masks: Optional[List[torch.Tensor]] = None,
                      max_len: int = 30) -> List[str]:
        ys = self.greedy_decode_ids(video_feat, video_masks, max_len)
        end_id = self.cap_preprocessor.end_id
        result = []
        for idx_cap in ys.tolist():
            end_count = -1
            for i, idx in enumerate(idx_cap):
                if idx == end_id:
                    end_count = i
                    break
            idx_cap = idx_cap[1:end_count]   # reference quirk kept: without [SEP] the last token is dropped
            toks = self.cap_preprocessor.tokenizer.convert_ids_to_tokens(idx_cap)
            result.append(self.cap_preprocessor.tokenizer.convert_tokens_to_string(toks))
        return result

    @torch.no_grad()
    def greedy_decode_ids(self, video_feat, video_masks=None, max_len: int = 30, kv_cache: bool = True,
                          use_graphs: bool = True) -> torch.Tensor:
        """The id matrix ys [B, <=max_len] of MMT4Caption.greedy_decode (MMT4Caption.py:159-172).
        kv_cache=False runs the reference's O(L^2) algorithm (full decoder re-run per token)."""
        was_training = self.training
        self.eval()
        try:
            from .. import decode
            mask = video_masks[0] if video_masks is not None else None
            if kv_cache:
                return decode.greedy_decode_ids(self, video_feat[0], mask, max_len, use_graphs=use_graphs)
            return decode.greedy_decode_ids_reference_algorithm(self, video_feat[0], mask, max_len)
        finally:
            self.train(was_training)

    def beam_decode(self):
        pass

    def mode(self, forward_type="caption") -> None:
        self.f_type = forward_type
        matching = getattr(self, "matching", None)
        mparams = list(matching.parameters()) if matching is not None else []
        if forward_type == "caption":
            flags = (True, False)
        elif forward_type == "match":
            flags = (False, True)
        elif forward_type == "cross":
            flags = (True, True)
        else:
            raise ValueError
        for p in self.cap_decoder.parameters():
            p.requires_grad = flags[0]
        for p in mparams:
            p.requires_grad = flags[1]
