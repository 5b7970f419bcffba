"""Greedy decoding (reference: model/MMT4Caption.py:146-184, model/CapDecoder.py:62-79).

Two implementations with identical results (tests/test_model_gpu.py):
  * `greedy_decode_ids_reference_algorithm`: the reference's O(L^2) loop -- every step re-runs the whole
    decoder over all tokens so far -- on the HIP kernels; one host sync per token like the reference's
    `.tolist()` (MMT4Caption.py:168).
  * `greedy_decode_ids` (default): KV cache (self-attention K/V per layer [B, Lmax, 2d], cross-attention
    K/V of the memory computed once), ONE new token per step, the per-token kernel sequence captured
    in a hipGraph per position and replayed; end-of-sequence bookkeeping stays on the device and the
    host reads it `lookahead` steps behind the launch front, so the device never idles on the check.
    The reference's stop rule (stop when EVERY row has emitted [SEP] at least once) is honoured by
    truncating the id matrix at that step."""
import torch

from . import ops
from .engine import DecodeState
from .utils import capture_graph


@torch.no_grad()
def greedy_decode_ids_reference_algorithm(model, feats: torch.Tensor, mask, max_len: int = 30) -> torch.Tensor:
    pre = model.cap_preprocessor
    model._ps.refresh_shadow()
    model._ps.refresh_lazy_transposed()
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
                      sync_every: int = 4, lookahead: int = 3) -> torch.Tensor:
    """Returns ys int64 [B, <= max_len], identical to the reference loop's id matrix.

    The host never waits for the step it has just launched: it keeps `lookahead` token steps queued and, every `sync_every`
    steps, reads the device-side "every caption has ended at step s" word as of the step that left the queue (a copy on a second
    stream issued once the host has seen THAT step's event complete) -- the reference syncs on every token (`.tolist()`, MMT4Caption.py:168).  A
    caption batch that ends at step s therefore costs at most s + lookahead steps; the id matrix is truncated at s."""
    pre = model.cap_preprocessor
    model._ps.refresh_shadow()
    model._ps.refresh_lazy_transposed()
    enc, dec = model.video_encoder._engine(), model.cap_decoder._engine()
    B, T = feats.shape[0], feats.shape[1]
    st = _session(model, dec, B, T + 1, max_len)
    stamp = model._ps._stamp
    if st.__dict__.get("weights_stamp") != stamp:      # graphs bake weight pointers only, but keep it simple and safe
        st.weights_stamp = stamp
    on_gpu = feats.device.type == "cuda"
    if on_gpu and st.__dict__.get("poll_stream") is None:
        st.poll_stream = torch.cuda.Stream(device=feats.device)
        st.poll_host = torch.empty(1, dtype=torch.long).pin_memory()
        st.events = [torch.cuda.Event() for _ in range(max_len)]
    if use_graphs and on_gpu:
        # the prologue (encoder forward over the batch, cross-attention K/V of the memory for every layer, cache / flag reset:
        # ~40 launches, host-bound at 0.33 ms when issued one by one) is ONE captured graph too; the inputs go through static
        # copies.  It bakes pointers into the engines' shared grow-only buffers: re-captured whenever any of them grew.
        key = (tuple(feats.shape), feats.dtype, mask is not None)
        bg = st.__dict__.get("begin")
        if bg is None or bg["key"] != key or bg["gen"] != model._ps.ctx.generation:
            fin = torch.empty_like(feats, memory_format=torch.contiguous_format)
            min_ = torch.empty_like(mask, memory_format=torch.contiguous_format) if mask is not None else None
            fin.copy_(feats)
            if mask is not None:
                min_.copy_(mask)
            dec.decode_begin(st, enc.forward(fin, min_, False), pre.start_id, pre.pad_id)      # warm-up: allocates
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):
                dec.decode_begin(st, enc.forward(fin, min_, False), pre.start_id, pre.pad_id)
            st.begin = {"key": key, "gen": model._ps.ctx.generation, "g": g, "fin": fin, "min": min_}
        else:
            bg["fin"].copy_(feats)
            if mask is not None:
                bg["min"].copy_(mask)
            bg["g"].replay()
    else:
        dec.decode_begin(st, enc.forward(feats, mask, False), pre.start_id, pre.pad_id)
    stop = max_len

    def ended_as_of(step: int) -> int:
        """all_ended_at once `step` has run (later steps may still be in flight: the word only ever decreases to the FIRST step
        at which every row had ended, so a value read early is either max_len or final)."""
        st.events[step].synchronize()       # the HOST waits for that step (later steps stay queued on the device) ...
        with torch.cuda.stream(st.poll_stream):
            # ... so the copy needs no device-side dependency: a cross-stream wait on an event that is still pending costs
            # ~0.8 ms of latency on this runtime (tools/decode_loop_probe2.py), an independent 8-byte copy 20 us
            st.poll_host.copy_(st.all_ended_at, non_blocking=True)
        st.poll_stream.synchronize()
        return int(st.poll_host[0])

    for t in range(1, max_len):
        if use_graphs:
            g = st.graphs.get(t)
            if g is None:
                dec.decode_step(st, t, pre.end_id)         # warm-up run (allocates the step's temporaries)
                # the warm-up already wrote ys[:, t]; capture replays the same work
                g = torch.cuda.CUDAGraph()
                with capture_graph(g):
                    dec.decode_step(st, t, pre.end_id)
                st.graphs[t] = g
            else:
                g.replay()
        else:
            dec.decode_step(st, t, pre.end_id)
        if not on_gpu:
            continue
        if t % sync_every == 0 or t == max_len - 1:
            st.events[t].record()                          # only the steps that will be polled: an event between two graph
        done = t - lookahead                               # launches costs the device ~15 us

        if done >= 1 and (done % sync_every == 0):
            s = ended_as_of(done)
            if s < max_len:
                stop = s
                break
    else:
        if on_gpu:
            stop = min(stop, ended_as_of(max_len - 1))
    return st.ys[:, :min(stop, max_len - 1) + 1].clone()


@torch.no_grad()
def teacher_forced_next_ids(model, feats: torch.Tensor, mask, prefix_ids: torch.Tensor, steps: int, return_logits: bool = False):
    """The KV-cache token step of greedy_decode_ids (same kernels, same cache) with the CONSUMED token of every step forced to
    `prefix_ids[:, t - 1]` instead of the step's own previous prediction: returns the predicted next ids [B, steps]
    (column t - 1 = arg-max after consuming prefix_ids[:, :t]).  This is CapDecoder.decode_word + torch.max of the reference
    (CapDecoder.py:62-79, MMT4Caption.py:164-165) evaluated along a given caption -- what a low-precision path can be held
    to where free-running ids would diverge after the first unresolvable logit gap.  return_logits: also the fp32 logits
    [B, steps, V] of every step."""
    pre = model.cap_preprocessor
    model._ps.refresh_shadow()
    model._ps.refresh_lazy_transposed()
    enc, dec = model.video_encoder._engine(), model.cap_decoder._engine()
    B, T = feats.shape[0], feats.shape[1]
    st = DecodeState(dec, B, T + 1, steps + 1)
    dec.decode_begin(st, enc.forward(feats, mask, False), pre.start_id, pre.pad_id)
    out = torch.empty(B, steps, dtype=torch.long, device=feats.device)
    logits = torch.empty(B, steps, dec.V, dtype=torch.float32, device=feats.device) if return_logits else None
    for t in range(1, steps + 1):
        st.ys[:, t - 1] = prefix_ids[:, t - 1]
        dec.decode_step(st, t, pre.end_id)
        out[:, t - 1] = st.ys[:, t]
        if return_logits:
            logits[:, t - 1] = st.last_logits[:, :dec.V].float()
    return (out, logits) if return_logits else out
