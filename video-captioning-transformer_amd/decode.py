"""Greedy decoding (reference: model/MMT4Caption.py:146-184, model/CapDecoder.py:62-79)."""
import torch

from . import ops


@torch.no_grad()
def greedy_decode_ids(model, feats: torch.Tensor, mask, max_len: int = 30) -> torch.Tensor:
    """Encoder once; then <= max_len-1 steps of decode_word + first-index arg-max + append; stop when
    EVERY row has emitted end_id at least once (sticky flags).  Returns ys int64 [B, <=max_len]."""
    pre = model.cap_preprocessor
    model._ps.refresh_shadow()
    enc, dec = model.video_encoder._engine(), model.cap_decoder._engine()
    B, T = feats.shape[0], feats.shape[1]
    mem = enc.forward(feats, mask, False)
    ys = torch.full((B, max_len), pre.pad_id, dtype=torch.long, device=feats.device)
    ys[:, 0] = pre.start_id
    nxt = torch.empty(B, dtype=torch.long, device=feats.device)
    ended = torch.zeros(B, dtype=torch.bool, device=feats.device)
    t = 1
    for _ in range(max_len - 1):
        logits = dec.decode_word(mem, B, T + 1, ys[:, :t])
        ops.argmax_rows(logits, nxt, cols=dec.V)
        ys[:, t] = nxt
        t += 1
        ended |= nxt == pre.end_id
        if bool(ended.all()):   # host sync per token, like the reference's .tolist() (MMT4Caption.py:168)
            break
    return ys[:, :t].clone()
