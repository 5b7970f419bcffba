"""Greedy decoding (reference: model/MMT4Caption.py:146-184, model/CapDecoder.py:62-79).

Two implementations with identical results (tests/test_model_gpu.py):
  * `greedy_decode_ids_reference_algorithm`: the reference's O(L^2) loop -- every step re-runs the whole
    decoder over all tokens so far -- on the HIP kernels; one host sync per token like the reference's
    `.tolist()` (MMT4Caption.py:168).
  * `greedy_decode_ids` (default): KV cache (self-attention K/V per layer [B, Lmax, 2d], cross-attention
    K/V of the memory computed once), ONE new token per step, the per-token kernel sequence captured
    in a hipGraph per position and replayed; end-of-sequence bookkeeping stays on the device and the
    host checks it every `sync_every` tokens.  The reference's stop rule (stop when EVERY row has
    emitted [SEP] at least once) is honoured by truncating the id matrix at that step."""
import torch

from . import ops
from .engine import DecodeState


@torch.no_grad()
def greedy_decode_ids_reference_algorithm(model, feats: torch.Tensor, mask, max_len: int = 30) -> torch.Tensor:
    pre = model.cap_preprocessor
    model._ps.refresh_shadow()
    enc, dec = model.video_encoder._engine(), model.cap_decoder._engine()
    B, T = feats.shape[0], feats.shape[1]
    mem = enc.forward(feats, mask, False)
    ys = torch.full((B, max_len), pre.pad_id, dtype=torch.long, device=feats.device)
    ys[:, 0] = pre.start_id
    ended = torch.zeros(B, dtype=torch.bool, device=feats.device)
    t = 1
    for _ in range(max_len - 1):
        logits = dec.decode_word(mem, B, T + 1, ys[:, :t])
        ops.argmax_rows(logits, ys[:, t], cols=dec.V)
        ended |= ys[:, t] == pre.end_id
        t += 1
        if bool(ended.all()):
            break
    return ys[:, :t].clone()


def _session(model, dec, B, Te, max_len) -> DecodeState:
    cache = model.__dict__.setdefault("_decode_sessions", {})
    key = (B, Te, max_len, dec.dt)
    st = cache.get(key)
    if st is None:
        if len(cache) > 3:
            cache.clear()
        st = cache[key] = DecodeState(dec, B, Te, max_len)
    return st


@torch.no_grad()
def greedy_decode_ids(model, feats: torch.Tensor, mask, max_len: int = 30, use_graphs: bool = True,
                      sync_every: int = 4) -> torch.Tensor:
    """Returns ys int64 [B, <= max_len], identical to the reference loop's id matrix."""
    pre = model.cap_preprocessor
    model._ps.refresh_shadow()
    enc, dec = model.video_encoder._engine(), model.cap_decoder._engine()
    B, T = feats.shape[0], feats.shape[1]
    st = _session(model, dec, B, T + 1, max_len)
    stamp = model._ps._stamp
    if st.__dict__.get("weights_stamp") != stamp:      # graphs bake weight pointers only, but keep it simple and safe
        st.weights_stamp = stamp
    mem = enc.forward(feats, mask, False)
    dec.decode_begin(st, mem, pre.start_id, pre.pad_id)
    stop = max_len
    for t in range(1, max_len):
        if use_graphs:
            g = st.graphs.get(t)
            if g is None:
                dec.decode_step(st, t, pre.end_id)         # warm-up run (allocates the step's temporaries)
                # the warm-up already wrote ys[:, t]; capture replays the same work
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    dec.decode_step(st, t, pre.end_id)
                st.graphs[t] = g
            else:
                g.replay()
        else:
            dec.decode_step(st, t, pre.end_id)
        if t % sync_every == 0 or t == max_len - 1:
            s = int(st.all_ended_at)                        # host sync (every sync_every tokens)
            if s < max_len:
                stop = s
                break
    return st.ys[:, :min(stop, max_len - 1) + 1].clone()
