#!/usr/bin/env python
"""Yardstick only (NOT used by the product): what torch.matmul (hipBLASLt/rocBLAS) reaches on the
caption path's GEMM shapes, bf16, same box -- to know how far the hand-written kernel is from a
tuned vendor kernel."""
import torch
SH = [("gen_fwd NT", 4864, 30522, 512, "nt"), ("gen_dx NN", 4864, 512, 30522, "nn"), ("gen_dw TN", 30522, 512, 4864, "tn"),
      ("ffn1_fwd NT", 4864, 2048, 512, "nt"), ("ffn2_fwd NT", 4864, 512, 2048, "nt"), ("qkv_fwd NT", 4864, 1536, 512, "nt"),
      ("out_fwd NT", 4864, 512, 512, "nt"), ("ffn2_dx NN", 4864, 2048, 512, "nn"), ("ffn1_dw TN", 2048, 512, 4864, "tn"),
      ("out_dw TN", 512, 512, 4864, "tn")]
for name, M, N, K, lay in SH:
    if lay == "nt":
        a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16(); f = lambda: a @ b.t()
    elif lay == "nn":
        a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(K, N, device="cuda").bfloat16(); f = lambda: a @ b
    else:
        a = torch.randn(K, M, device="cuda").bfloat16(); b = torch.randn(K, N, device="cuda").bfloat16(); f = lambda: a.t() @ b
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:14s} {M}x{N}x{K}: {ms*1e3:8.1f} us  {2.0*M*N*K/(ms*1e-3)/1e12:6.0f} TF")
