"""MultiModalEncoder -- drop-in for the reference's model/MMEncoder.py:205-276 (the shipped path:
one modality, `temporal: "encoding"`, `aggregation: "avg"`, do_norm False), executed by
hand-written gfx950 kernels (engine.EncoderEngine).  Same constructor signature, same forward
signature and return tuple, same state_dict keys."""
from typing import List, Optional

import torch
import torch.nn as nn

from ..engine import EncoderEngine, ParamSet
from ._params import LinearParams, StackParams, sinusoid_table


class TemporalEncoding(nn.Module):
    """Holds the `pe` buffer [1, max_len, d] (model/MMEncoder.py:63-81)."""

    def __init__(self, d_model=512, max_len=512, device=None):
        super().__init__()
        self.register_buffer("pe", sinusoid_table(max_len, d_model, "encoder", device).unsqueeze(0))


def grad_ready_order_encoder(prefix, n_layers):
    names = [prefix + "transformer_encoder.norm.weight", prefix + "transformer_encoder.norm.bias"]
    for l in reversed(range(n_layers)):
        lp = f"{prefix}transformer_encoder.layers.{l}."
        names += [lp + k for k in ("norm2.weight", "norm2.bias", "linear2.weight", "linear2.bias", "linear1.weight",
                                   "linear1.bias", "norm1.weight", "norm1.bias", "self_attn.out_proj.weight",
                                   "self_attn.out_proj.bias", "self_attn.in_proj_weight", "self_attn.in_proj_bias")]
    return names + [prefix + "unify.0.weight", prefix + "unify.0.bias"]


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, feats, mask, *params):
        eng = mod._engine()
        mem = eng.forward(feats, mask, mod.training)
        ctx.mod = mod
        B, T = feats.shape[0], feats.shape[1]
        return mem.view(B, T + 1, -1)

    @staticmethod
    def backward(ctx, dmem):
        mod = ctx.mod
        eng = mod._engine()
        eng.backward(dmem.reshape(-1, dmem.shape[-1]).contiguous())
        mod._ps.install_grads()
        return (None, None, None) + (None,) * len(mod._ps.names)


class MultiModalEncoder(nn.Module):
    def __init__(self, d_feats: List[int], d_model: int, nhead: int, dim_feedforward: int = 2048,
                 num_encoder_layers: int = 4, dropout: float = 0.1, activation: str = "gelu", global_type: str = "avg",
                 modal_different: bool = True, temporal_type: str = "embedding", do_norm: bool = False,
                 device=torch.device("cuda"), compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        if len(d_feats) != 1:
            raise NotImplementedError("multi-modal input (ModalEmbedding) is outside the accelerated caption path")
        if global_type != "avg" or temporal_type != "encoding" or do_norm:
            raise NotImplementedError("accelerated path = aggregation 'avg', temporal 'encoding', do_norm False "
                                      "(the shipped configs); other encoder variants are out of scope")
        self.device, self.num_modal, self.do_norm = device, 1, do_norm
        self.cfg = dict(d=d_model, nhead=nhead, ff=dim_feedforward, layers=num_encoder_layers, dropout=float(dropout),
                        activation=activation)
        self.compute_dtype = compute_dtype
        self.unify = nn.ModuleList([LinearParams(d_feats[0], d_model, device)])
        self.temp_emb = TemporalEncoding(d_model, device=device)
        self.transformer_encoder = StackParams(d_model, dim_feedforward, num_encoder_layers, False, device)
        self._ps: Optional[ParamSet] = None   # set by the owner (MMT4Caption) or lazily for standalone use
        self._prefix = ""
        self._eng: Optional[EncoderEngine] = None
        self._seed = None

    # ---- engine plumbing -------------------------------------------------------------------------
    def _bind(self, ps: ParamSet, prefix: str, seed: torch.Tensor, rebuild):
        self._ps, self._prefix, self._seed, self._eng, self._rebuild = ps, prefix, seed, None, rebuild

    def _engine(self) -> EncoderEngine:
        if getattr(self, "_rebuild", None) is not None:
            if not self._ps.intact():
                self._rebuild()
        elif self._ps is None or not self._ps.intact():
            named = dict(self.named_parameters())
            order = grad_ready_order_encoder("", self.cfg["layers"])
            dev = next(self.parameters()).device
            self._ps = ParamSet([(n, named[n]) for n in order], dev, self.compute_dtype)
            self._prefix, self._eng = "", None
            self._seed = torch.tensor([torch.initial_seed() & 0x7FFFFFFF], dtype=torch.int32, device=dev)
        if self._eng is None:
            self._eng = EncoderEngine(self._ps, self._prefix, self.cfg, self._seed, self.temp_emb.pe)
        return self._eng

    # ---- reference API ---------------------------------------------------------------------------
    def forward(self, srcs: List[torch.Tensor], src_padding_masks: Optional[List[torch.Tensor]]):
        """srcs: [Tensor[B,T,E]] fp32; src_padding_masks: [Tensor[B,T] bool] (True = padded) or None.
        Returns (memory[B,T+1,d], global_masks[B,T+1] or None, memory[:,0]) like MMEncoder.py:276."""
        feats = srcs[0]
        mask = src_padding_masks[0] if src_padding_masks is not None else None
        eng = self._engine()
        eng.ps.refresh_shadow()
        mem = _EncoderFn.apply(self, feats, mask, *[self._ps.params[n] for n in self._ps.names])
        mem = mem.float() if mem.dtype != torch.float32 else mem.clone()
        gmask = None
        if mask is not None:
            gmask = torch.cat([torch.zeros(mask.shape[0], 1, dtype=torch.bool, device=mask.device), mask], 1)
        return mem, gmask, mem[:, 0]
