#!/usr/bin/env python
"""Whole-call greedy decode time per token step for different host polling settings (dev tool)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import MODEL_CFG, T_FRAMES, D_IN  # noqa: E402
from vct_amd.model import MMT4Caption  # noqa: E402
from vct_amd import decode  # noqa: E402
dev = torch.device("cuda", 0)
torch.manual_seed(666)
m = MMT4Caption(MODEL_CFG, device=dev, compute_dtype=torch.bfloat16); m.mode("caption"); m.eval()
for B in (1, 128):
    feats = torch.randn(B, T_FRAMES, D_IN, device=dev)
    for se, la in ((4, 3), (4, 0), (100, 3)):
        for _ in range(2):
            ys = decode.greedy_decode_ids(m, feats, None, 30, True, se, la)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            ys = decode.greedy_decode_ids(m, feats, None, 30, True, se, la)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"B={B:3d} sync_every={se:3d} lookahead={la}: {dt / 29 * 1e6:6.1f} us/step (whole call {dt*1e3:.2f} ms)", flush=True)
    # the bare loop: graph replays only, host-timed and event-timed
    st = decode._session(m, m.cap_decoder._engine(), B, T_FRAMES + 1, 30)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
    for t in range(1, 30):
        st.graphs[t].replay()
    t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B:3d} bare replays: host enqueue {(t1 - t0) / 29 * 1e6:.1f} us/step, events {e0.elapsed_time(e1) / 29 * 1e3:.1f} us/step, wall {(t2 - t0) / 29 * 1e6:.1f} us/step")
    t0 = time.perf_counter()
    for _ in range(20):
        m._ps.refresh_shadow()
    print(f"refresh_shadow: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us host")
    # fixed part: encoder forward + decode_begin only
    enc, dec = m.video_encoder._engine(), m.cap_decoder._engine()
    st = decode._session(m, dec, B, T_FRAMES + 1, 30)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        mem = enc.forward(feats, None, False); dec.decode_begin(st, mem, 101, 0)
    torch.cuda.synchronize(); print(f"B={B:3d} encoder forward + decode_begin: {(time.perf_counter() - t0) / 20 * 1e6:.0f} us per call")

# where the whole call spends its time: eager prologue -> graph replays -> eager epilogue, event- and host-timed
for B in (128,):
    feats = torch.randn(B, T_FRAMES, D_IN, device=dev)
    enc, dec = m.video_encoder._engine(), m.cap_decoder._engine()
    st = decode._session(m, dec, B, T_FRAMES + 1, 30)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for rep in range(3):
        torch.cuda.synchronize(); h0 = time.perf_counter(); ev[0].record()
        mem = enc.forward(feats, None, False); dec.decode_begin(st, mem, 101, 0)
        h1 = time.perf_counter(); ev[1].record()
        for t in range(1, 30):
            st.graphs[t].replay()
        h2 = time.perf_counter(); ev[2].record()
        out = st.ys[:, :30].clone()
        h3 = time.perf_counter(); ev[3].record(); torch.cuda.synchronize(); h4 = time.perf_counter()
        print(f"B={B} host: begin {1e6*(h1-h0):.0f} replays {1e6*(h2-h1):.0f} clone {1e6*(h3-h2):.0f} drain {1e6*(h4-h3):.0f} us | "
              f"device: begin {1e3*ev[0].elapsed_time(ev[1]):.0f} replays {1e3*ev[1].elapsed_time(ev[2]):.0f} clone {1e3*ev[2].elapsed_time(ev[3]):.0f} us")

# the real function, instrumented from outside: host time of the call vs wall until the device is idle
for B in (128,):
    feats = torch.randn(B, T_FRAMES, D_IN, device=dev)
    for se, la in ((100, 3), (4, 3)):
        for rep in range(3):
            torch.cuda.synchronize(); h0 = time.perf_counter()
            ys = decode.greedy_decode_ids(m, feats, None, 30, True, se, la)
            h1 = time.perf_counter(); torch.cuda.synchronize(); h2 = time.perf_counter()
            print(f"B={B} se={se}: call returns after {1e6*(h1-h0):.0f} us, device idle after {1e6*(h2-h0):.0f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    ys = decode.greedy_decode_ids(m, feats, None, 30, True, 100, 3)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
