"""PositionalEmbedding buffer holder (reference: model/Embedding.py:7-25).  The add + dropout runs
inside the embedding-gather kernel (vct_embed_fwd)."""
import torch.nn as nn

from ._params import sinusoid_table


class PositionalEmbedding(nn.Module):
    def __init__(self, emb_size: int, dropout: float, maxlen: int = 5000, device=None):
        super().__init__()
        self.p = dropout
        self.register_buffer("pos_embedding", sinusoid_table(maxlen, emb_size, "decoder", device))
