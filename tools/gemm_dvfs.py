#!/usr/bin/env python3
"""Is the GEMM power/clock limited?  Same launch on random, sign-constant and zero operands."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops
from tools.bench_kernels import timeit
DEV = "cuda:0"
M = 100352
for N, K in ((4096, 1024), (1024, 4096), (8192, 8192)):
    MM = M if N != 8192 else 8192
    for kind in ("random", "abs", "zero"):
        x = torch.randn(MM, K, device=DEV)
        w = torch.randn(N, K, device=DEV) * 0.03
        if kind == "abs":
            x, w = x.abs(), w.abs()
        if kind == "zero":
            x, w = x * 0, w * 0
        x, w = x.to(torch.bfloat16), w.to(torch.bfloat16)
        out = torch.empty(MM, N, device=DEV, dtype=torch.bfloat16)
        us = timeit(lambda: ops.linear(x, w, None, out=out), iters=20)
        print(f"M={MM} N={N} K={K} {kind:7s}: {us:9.1f} us {2.0*MM*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
