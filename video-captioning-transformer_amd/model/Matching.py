"""Matching head parameter holder (reference: model/Matching.py:14-30).  The contrastive
video-text matching task is outside the accelerated caption path; the module exists because
MMT4Caption.mode() iterates `self.matching.parameters()` and released checkpoints carry
`matching.v_proj.*` when embed_dim != text-encoder dim."""
import torch.nn as nn

from ._params import LinearParams


class Matching(nn.Module):
    def __init__(self, vt_shape, enable_tem=False, loss="CSL", loss_tem=None, device=None):
        super().__init__()
        self.vt_shape, self.loss = vt_shape, loss
        self.v_proj = LinearParams(vt_shape[0], vt_shape[1], device) if vt_shape[0] != vt_shape[1] else None

    def forward(self, *_a, **_k):
        raise NotImplementedError("video-text matching task is not part of the MI355X caption path")


class TextEncoder:
    """Dimension-only stand-in for model/TextEncoder.py (frozen CLIP/BERT sentence encoder, matching
    task).  The reference constructs it unconditionally (MMT4Caption.py:30) and downloads CLIP; the
    caption path only needs `.dim` to size Matching.v_proj."""

    def __init__(self, enc_type, device=None):
        self.enc_type = enc_type
        self.dim = 512 if str(enc_type).upper() == "CLIP" else 768
