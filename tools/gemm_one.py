#!/usr/bin/env python
"""Run ONE GEMM configuration a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py <shape-substr> <tile> <nbuf> [iters]"""
import sys
from gemm_bench import SHAPES, run
name, tile, nbuf = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
s = [x for x in SHAPES if name in x[0]][0]
ms, tf = run(*s, tile + 10 * nbuf, iters=iters)
print(s[0], tile, nbuf, f"{ms*1e3:.1f}us {tf:.0f}TF")
