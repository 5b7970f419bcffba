"""torch-CPU restatement of the reference's caption training step -- the CPU BASELINE of bench.py.

TEST INFRASTRUCTURE ONLY (like vct_oracle.py): imported by tests/ and by bench.py's cpu_baseline leg, never by the
product package.  The reference runs this path as stock torch.nn modules under autograd on whatever device it is given
(CPU mode: utils.py:128-131); this file builds the same module graph from the same cfg['model'] block so that it can be
timed on the GPU box's host cores, where /root/reference does not exist:

    unify Linear                      model/MMEncoder.py:241,246
    mean token + cat + temporal PE    model/MMEncoder.py:248-257,89-104,271
    nn.TransformerEncoder (post-norm, gelu, batch_first) + final LayerNorm      model/MMEncoder.py:236-238,274
    nn.Embedding(padding_idx) + positional table + dropout                      model/CapDecoder.py:26,48; Embedding.py:23-25
    nn.TransformerDecoder + final LayerNorm, causal float mask + padding mask   model/CapDecoder.py:18-20,49-52; utils.py:63-66
    generator Linear, SCE loss (alpha CE + beta RCE)                            model/CapDecoder.py:25,55-59; loss.py:78-92
    Adam(lr 1e-4, betas (.9,.999))                                              train.py:24-26

Parameter names equal the reference's state_dict keys, so weights move between this module, the numpy oracle and the
HIP model by name.  tests/test_oracle_golden.py checks loss and every gradient of this module against the numpy oracle
(which is itself pinned to outputs of the real reference)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Unify(nn.Sequential):
    pass


class _TempEmb(nn.Module):
    def __init__(self, d, max_len=512):
        super().__init__()
        pos = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
        pe = torch.zeros(max_len, d)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe.unsqueeze(0))


class _PosEmb(nn.Module):
    def __init__(self, d, maxlen=5000):
        super().__init__()
        den = torch.exp(-torch.arange(0, d, 2, dtype=torch.float32) * math.log(10000) / d)
        pos = torch.arange(0, maxlen, dtype=torch.float32).reshape(maxlen, 1)
        tab = torch.zeros(maxlen, d)
        tab[:, 0::2] = torch.sin(pos * den)
        tab[:, 1::2] = torch.cos(pos * den)
        self.register_buffer("pos_embedding", tab)


class _Encoder(nn.Module):
    def __init__(self, d_in, d, nhead, ff, layers, dropout, activation):
        super().__init__()
        self.unify = _Unify(nn.Linear(d_in, d))
        self.temp_emb = _TempEmb(d)
        layer = nn.TransformerEncoderLayer(d, nhead, ff, dropout, activation=activation, batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(layer, layers, nn.LayerNorm(d), enable_nested_tensor=False)

    def forward(self, feats, mask):
        B, T, _ = feats.shape
        u = self.unify(feats)
        g = u.mean(dim=1, keepdim=True)                                   # over ALL frames, padded ones included
        z = torch.cat([g, u], dim=1)
        pe = torch.cat([torch.zeros_like(self.temp_emb.pe[:, :1]), self.temp_emb.pe[:, :T]], dim=1)
        z = z + pe
        kpm = None
        if mask is not None:
            kpm = torch.cat([torch.zeros(B, 1, dtype=torch.bool), mask], dim=1)
        return self.transformer_encoder(z, src_key_padding_mask=kpm)


class _Decoder(nn.Module):
    def __init__(self, d, nhead, ff, layers, dropout, activation, vocab, pad_id, alpha):
        super().__init__()
        layer = nn.TransformerDecoderLayer(d, nhead, ff, dropout, activation=activation, batch_first=True)
        self.decoder = nn.TransformerDecoder(layer, layers, nn.LayerNorm(d))
        self.generator = nn.Linear(d, vocab)
        self.tgt_to_emb = nn.Embedding(vocab, d, padding_idx=pad_id)
        self.positional_encoding = _PosEmb(d)
        self.drop = nn.Dropout(dropout)
        self.vocab, self.pad_id, self.alpha = vocab, pad_id, alpha

    def loss_fn(self, logits, labels):
        N = logits.shape[0]
        if self.alpha == 1.0:
            return F.cross_entropy(logits, labels, ignore_index=self.pad_id)
        ce = F.cross_entropy(logits, labels, ignore_index=self.pad_id)
        p = F.softmax(logits, dim=1).clamp(min=1e-7, max=1.0)
        onehot = F.one_hot(labels, self.vocab).float().clamp(min=1e-4, max=1.0)
        rce = -(p * onehot.log()).sum(dim=1)
        return self.alpha * ce + (1.0 - self.alpha) * rce.mean()

    def forward(self, mem, ids):
        tgt_in, tgt_out = ids[:, :-1], ids[:, 1:]
        S = tgt_in.shape[1]
        x = self.drop(self.tgt_to_emb(tgt_in) + self.positional_encoding.pos_embedding[:S])
        causal = torch.full((S, S), float("-inf")).triu(1)
        y = self.decoder(x, mem, tgt_mask=causal, tgt_key_padding_mask=(tgt_in == self.pad_id))
        logits = self.generator(y)
        return logits, self.loss_fn(logits.reshape(-1, logits.shape[-1]), tgt_out.reshape(-1))


class RefCaptionModel(nn.Module):
    """cfg = vct_oracle.cfg_from_model_config(cfg['model'], vocab); dropout from cfg['model']['dropout']."""

    def __init__(self, cfg: dict, dropout: float):
        super().__init__()
        self.video_encoder = _Encoder(cfg["d_in"], cfg["d"], cfg["enc_nhead"], cfg["enc_ff"], cfg["enc_layers"], dropout,
                                      cfg["activation"])
        self.cap_decoder = _Decoder(cfg["d"], cfg["dec_nhead"], cfg["dec_ff"], cfg["dec_layers"], dropout, cfg["activation"],
                                    cfg["vocab"], cfg["pad_id"], cfg["sce_loss_alpha"])

    def forward(self, feats, mask, ids):
        return self.cap_decoder(self.video_encoder(feats, mask), ids)

    def load_oracle_params(self, p: dict):
        sd = {k: torch.from_numpy(v.copy()) for k, v in p.items()}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all("drop" in k for k in missing), missing


def time_training_steps(cfg, dropout, feats, mask, ids, steps, threads, lr=1e-4):
    """samples/s at the MEDIAN of `steps` timed steps (after one warm-up) of forward + zero_grad + backward + Adam
    (train.py:119-131; BASELINE.md section 3: >= 5 timed steps, median).  Returns (rate, seconds spent in timed steps, loss)."""
    import time
    torch.set_num_threads(threads)
    torch.manual_seed(666)
    m = RefCaptionModel(cfg, dropout)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr, betas=(0.9, 0.999))
    f, mk, i = torch.from_numpy(feats), torch.from_numpy(mask), torch.from_numpy(ids)
    times, loss = [], None
    for it in range(steps + 1):
        t0 = time.perf_counter()
        _, loss = m(f, mk, i)
        opt.zero_grad()
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    return feats.shape[0] / med, sum(times), float(loss)
