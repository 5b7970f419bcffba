#!/usr/bin/env python
"""Greedy-decode benchmark (BASELINE.json configs[4]): cfg-B model in eval mode, synthetic (B,12,512)
features, max_len 30, B in {1, 128}; KV-cache + hipGraph step vs KV-cache eager vs the reference's
O(L^2) algorithm (all on the HIP kernels).  Prints one JSON line per configuration."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import MODEL_CFG  # noqa: E402


def main():
    from vct_amd.model import MMT4Caption
    dev = torch.device("cuda", 0)
    torch.manual_seed(666)
    for dtype in (torch.bfloat16, torch.float32):
        m = MMT4Caption(MODEL_CFG, device=dev, compute_dtype=dtype)
        m.mode("caption"); m.eval()
        # random-init weights rarely emit [SEP]: every run decodes the full 29 steps (worst case)
        for B in (1, 128):
            feats = torch.randn(B, 12, 512, generator=torch.Generator().manual_seed(0)).to(dev)
            for name, kw in (("kv_cache+hipgraph", dict(kv_cache=True, use_graphs=True)),
                             ("kv_cache eager", dict(kv_cache=True, use_graphs=False)),
                             ("full re-run (reference algorithm)", dict(kv_cache=False))):
                for _ in range(2):
                    ys = m.greedy_decode_ids([feats], None, max_len=30, **kw)
                torch.cuda.synchronize()
                n = 5
                t0 = time.perf_counter()
                for _ in range(n):
                    ys = m.greedy_decode_ids([feats], None, max_len=30, **kw)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                steps = ys.shape[1] - 1
                print(json.dumps({"decode": name, "dtype": str(dtype).split(".")[-1], "batch": B, "steps": steps,
                                  "ms_per_caption_batch": round(dt * 1e3, 3), "us_per_token_step": round(dt / steps * 1e6, 1),
                                  "tokens_per_s": round(B * steps / dt, 1)}), flush=True)


if __name__ == "__main__":
    main()
