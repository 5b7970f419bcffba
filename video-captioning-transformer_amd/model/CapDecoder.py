"""CapDecoder -- drop-in for the reference's model/CapDecoder.py:11-79, executed by hand-written
gfx950 kernels (engine.DecoderEngine).  Same constructor / forward / decode_word signatures, same
state_dict keys (decoder.layers.N.*, decoder.norm, generator, tgt_to_emb, positional_encoding)."""
from typing import Optional

import torch
import torch.nn as nn

from ..engine import DecoderEngine, ParamSet
from ._params import LinearParams, StackParams
from .Embedding import PositionalEmbedding
from .loss import SCELoss


class EmbeddingParams(nn.Module):
    """nn.Embedding(V, d, padding_idx) parameters: N(0,1) init, padding row zero and gradient-free."""

    def __init__(self, V, d, padding_idx, device=None):
        super().__init__()
        self.padding_idx = padding_idx
        self.weight = nn.Parameter(torch.randn(V, d, device=device))
        with torch.no_grad():
            self.weight[padding_idx].zero_()


def grad_ready_order_decoder(prefix, n_layers):
    names = [prefix + "generator.weight", prefix + "generator.bias", prefix + "decoder.norm.weight", prefix + "decoder.norm.bias"]
    for l in reversed(range(n_layers)):
        lp = f"{prefix}decoder.layers.{l}."
        names += [lp + k for k in (
            "norm3.weight", "norm3.bias", "linear2.weight", "linear2.bias", "linear1.weight", "linear1.bias",
            "norm2.weight", "norm2.bias", "multihead_attn.out_proj.weight", "multihead_attn.out_proj.bias",
            "multihead_attn.in_proj_weight", "multihead_attn.in_proj_bias", "norm1.weight", "norm1.bias",
            "self_attn.out_proj.weight", "self_attn.out_proj.bias", "self_attn.in_proj_weight", "self_attn.in_proj_bias")]
    return names + [prefix + "tgt_to_emb.weight"]


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, mem, ids, want_logits, *params):
        eng = mod._engine()
        B, Te, d = mem.shape
        loss, logits = eng.forward(mem.reshape(B * Te, d), B, Te, ids, mod.training, want_logits=want_logits)
        ctx.mod, ctx.shape = mod, (B, Te, d)
        out_logits = logits[:, :eng.V].reshape(B, ids.shape[1] - 1, eng.V) if want_logits else loss.new_zeros(())
        ctx.mark_non_differentiable(out_logits)
        return loss[0], out_logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        mod = ctx.mod
        eng = mod._engine()
        dmem = eng.backward()
        mod._ps.install_grads()
        mod._scale_grads_if_needed(gloss)
        B, Te, d = ctx.shape
        dmem = dmem.view(B, Te, d) * gloss.to(dmem.dtype)
        return (None, dmem, None, None) + (None,) * len(mod._ps.names)


class CapDecoder(nn.Module):
    def __init__(self, num_layers, embed_dim, nhead, dim_feedforward, dropout, vocab_size, pad_id, sce_loss_alpha: float,
                 custom_decoder_type: Optional[str] = None, activation="gelu", device=torch.device("cuda"),
                 compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        if custom_decoder_type is not None:
            raise NotImplementedError("Vis* decoder layers (attention-map visualisation) are outside the accelerated path")
        self.device, self.compute_dtype = device, compute_dtype
        self.cfg = dict(d=embed_dim, nhead=nhead, ff=dim_feedforward, layers=num_layers, dropout=float(dropout),
                        activation=activation, vocab=vocab_size, pad_id=pad_id, sce_loss_alpha=float(sce_loss_alpha))
        self.decoder = StackParams(embed_dim, dim_feedforward, num_layers, True, device)
        self.generator = LinearParams(embed_dim, vocab_size, device)
        self.tgt_to_emb = EmbeddingParams(vocab_size, embed_dim, pad_id, device)
        self.positional_encoding = PositionalEmbedding(embed_dim, dropout=dropout, maxlen=5000, device=device)
        # kept for API parity (CapDecoder.py:28-32); the fused kernel implements both branches
        self.loss_fn = (nn.CrossEntropyLoss(ignore_index=pad_id) if sce_loss_alpha == 1.0
                        else SCELoss(sce_loss_alpha, 1 - sce_loss_alpha, ignore_index=pad_id, num_classes=vocab_size))
        self._ps: Optional[ParamSet] = None
        self._prefix, self._eng, self._seed, self._rebuild = "", None, None, None
        self._unit_loss_grad = False

    def _bind(self, ps, prefix, seed, rebuild):
        self._ps, self._prefix, self._seed, self._eng, self._rebuild = ps, prefix, seed, None, rebuild

    def _engine(self) -> DecoderEngine:
        if self._rebuild is not None:
            if not self._ps.intact():
                self._rebuild()
        elif self._ps is None or not self._ps.intact():
            named = dict(self.named_parameters())
            dev = next(self.parameters()).device
            order = grad_ready_order_decoder("", self.cfg["layers"])
            self._ps = ParamSet([(n, named[n]) for n in order], dev, self.compute_dtype, no_shadow=("tgt_to_emb.weight",))
            self._prefix, self._eng = "", None
            self._seed = torch.tensor([torch.initial_seed() & 0x7FFFFFFF], dtype=torch.int32, device=dev)
        if self._eng is None:
            self._eng = DecoderEngine(self._ps, self._prefix, self.cfg, self._seed, self.positional_encoding.pos_embedding)
        return self._eng

    def _scale_grads_if_needed(self, gloss):
        """The kernels produce gradients for d(loss) = 1; apply the incoming scalar to this module's
        gradient slices (skipped by the fused training step, which always seeds 1)."""
        if self._unit_loss_grad:
            return
        pre = self._prefix
        for n in self._ps.names:
            if n.startswith(pre + "decoder.") or n.startswith(pre + "generator.") or n.startswith(pre + "tgt_to_emb."):
                self._ps.g[n].mul_(gloss)

    # ---- reference API ---------------------------------------------------------------------------
    def forward(self, memories: torch.Tensor, tgt: torch.Tensor, tgt_padding_mask: torch.Tensor, return_logits: bool = True):
        """memories [B,T,E]; tgt int64 [B,S]; tgt_padding_mask bool [B,S] (True = pad; must equal tgt == pad_id,
        which is how the reference builds it, CapPreprocessor.py:35).  Returns (logits[B,S-1,V] fp32, loss)."""
        eng = self._engine()
        eng.ps.refresh_shadow()
        mem = memories.to(eng.dt)
        loss, logits = _DecoderFn.apply(self, mem, tgt.contiguous(), return_logits, *[self._ps.params[n] for n in self._ps.names])
        return (logits.float() if return_logits else None), loss

    @torch.no_grad()
    def decode_word(self, memories: torch.Tensor, tgt: torch.Tensor, tgt_padding_mask: Optional[torch.Tensor] = None):
        """Next-token logits [B,V] given all tokens so far (CapDecoder.py:62-79)."""
        if tgt_padding_mask is not None:
            raise NotImplementedError("decode_word is only used with tgt_padding_mask=None (MMT4Caption.py:164)")
        eng = self._engine()
        eng.ps.refresh_shadow()
        B, Te, d = memories.shape
        return eng.decode_word(memories.to(eng.dt).reshape(B * Te, d), B, Te, tgt.contiguous()).float()
