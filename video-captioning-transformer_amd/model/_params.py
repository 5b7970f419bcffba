"""Parameter-holder modules: they reproduce the reference's state_dict() key names and shapes
 and torch's default initialisation, but carry no arithmetic -- the math
runs in the engine's HIP kernels."""
import math

import torch
import torch.nn as nn


class LinearParams(nn.Module):
    """weight [out,in], bias [out]; nn.Linear's default init (kaiming_uniform(a=sqrt 5), U(+-1/sqrt(in)))."""

    def __init__(self, d_in, d_out, device=None, zero_bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d_out, d_in, device=device))
        self.bias = nn.Parameter(torch.empty(d_out, device=device))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if zero_bias:
            nn.init.zeros_(self.bias)
        else:
            bound = 1.0 / math.sqrt(d_in)
            nn.init.uniform_(self.bias, -bound, bound)


class AttnParams(nn.Module):
    """nn.MultiheadAttention's parameters: packed in_proj (xavier_uniform / zeros), out_proj (bias 0)."""

    def __init__(self, d, device=None):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d, device=device))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d, device=device))
        self.out_proj = LinearParams(d, d, device, zero_bias=True)
        nn.init.xavier_uniform_(self.in_proj_weight)


class NormParams(nn.Module):
    def __init__(self, d, device=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d, device=device))
        self.bias = nn.Parameter(torch.zeros(d, device=device))


class LayerParams(nn.Module):
    def __init__(self, d, ff, cross, device=None):
        super().__init__()
        self.self_attn = AttnParams(d, device)
        if cross:
            self.multihead_attn = AttnParams(d, device)
        self.linear1 = LinearParams(d, ff, device)
        self.linear2 = LinearParams(ff, d, device)
        self.norm1 = NormParams(d, device)
        self.norm2 = NormParams(d, device)
        if cross:
            self.norm3 = NormParams(d, device)


class StackParams(nn.Module):
    """`layers` + final `norm`, like nn.TransformerEncoder/Decoder.  torch deep-copies ONE layer, so
    every layer starts with identical weights -- reproduced here."""

    def __init__(self, d, ff, n_layers, cross, device=None):
        super().__init__()
        self.layers = nn.ModuleList([LayerParams(d, ff, cross, device) for _ in range(n_layers)])
        for l in range(1, n_layers):
            self.layers[l].load_state_dict(self.layers[0].state_dict())
        self.norm = NormParams(d, device)


def sinusoid_table(n_pos, d, variant, device=None):
    """Fixed sin/cos table.  variant 'decoder' follows model/Embedding.py:13-17, 'encoder'
    model/MMEncoder.py:71-81 (same values up to fp32 rounding order)."""
    pos = torch.arange(0, n_pos, dtype=torch.float32).unsqueeze(1)
    if variant == "decoder":
        den = torch.exp(-torch.arange(0, d, 2) * math.log(10000) / d)
    else:
        den = (torch.arange(0, d, 2).float() * -(math.log(10000.0) / d)).exp()
    tab = torch.zeros(n_pos, d)
    tab[:, 0::2] = torch.sin(pos * den)
    tab[:, 1::2] = torch.cos(pos * den)
    return tab.to(device) if device is not None else tab
