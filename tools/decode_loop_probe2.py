#!/usr/bin/env python
"""Bisect the whole-call overhead of greedy_decode_ids: the function's body re-typed with switches (dev tool)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import MODEL_CFG, T_FRAMES, D_IN  # noqa: E402
from vct_amd.model import MMT4Caption  # noqa: E402
from vct_amd import decode  # noqa: E402
dev = torch.device("cuda", 0)
torch.manual_seed(666)
m = MMT4Caption(MODEL_CFG, device=dev, compute_dtype=torch.bfloat16); m.mode("caption"); m.eval()
B = 128
feats = torch.randn(B, T_FRAMES, D_IN, device=dev)
for _ in range(2):
    decode.greedy_decode_ids(m, feats, None, 30)
enc, dec = m.video_encoder._engine(), m.cap_decoder._engine()
st = decode._session(m, dec, B, T_FRAMES + 1, 30)
pre = m.cap_preprocessor


def body(refresh, record, poll, no_grad):
    ctx = torch.no_grad() if no_grad else torch.enable_grad()
    with ctx:
        if refresh:
            m._ps.refresh_shadow()
        mem = enc.forward(feats, None, False)
        dec.decode_begin(st, mem, pre.start_id, pre.pad_id)
        for t in range(1, 30):
            st.graphs[t].replay()
        if record:
            st.events[29].record()
        if poll:
            st.events[29].synchronize()
            with torch.cuda.stream(st.poll_stream):
                st.poll_host.copy_(st.all_ended_at, non_blocking=True)
            st.poll_stream.synchronize()
        return st.ys[:, :30].clone()


for cfg in ((0, 0, 0, 0), (1, 0, 0, 0), (1, 1, 0, 0), (1, 1, 1, 0), (1, 1, 1, 1), (0, 0, 0, 1)):
    for _ in range(2):
        body(*cfg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        body(*cfg)
    torch.cuda.synchronize()
    print(cfg, f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    decode.greedy_decode_ids(m, feats, None, 30)
torch.cuda.synchronize()
print("function", f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call")
