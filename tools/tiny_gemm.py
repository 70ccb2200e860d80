#!/usr/bin/env python3
"""Fixed vs per-K-step cost of the small Q-Former GEMMs (rows = B*S = 256, N = 768): device time per launch from a
back-to-back stream of 200 launches (so launch gaps count as the dependent chain sees them)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops
from tools.bench_kernels import timeit
DEV = "cuda:0"
for M, N in ((256, 768), (192, 768), (256, 2304), (192, 3072), (64, 768)):
    for K in (64, 256, 768, 3072):
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
        b = torch.randn(N, device=DEV)
        res = torch.randn(M, N, device=DEV)
        out = torch.empty(M, N, device=DEV)
        us = timeit(lambda: ops.linear(x, w, b, residual=res, out_dtype=torch.float32, out=out), iters=200, warm=10)
        ob = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        us2 = timeit(lambda: ops.linear(x, w, b, out=ob), iters=200, warm=10)
        print(f"M={M} N={N} K={K}: fp32+residual {us:6.1f} us   bf16 {us2:6.1f} us", flush=True)
