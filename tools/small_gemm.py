#!/usr/bin/env python3
"""Isolated timings of the four ViT GEMM shapes at a SMALL row count (B = 1 calls): which kernel / tile the dispatcher
picks (HK_GEMM_PP_MIN, HK_GEMM_VARIANT in the environment), LDS-DMA bytes it moves, and the same with every M-tile
reading the same rows of A (L2-resident A).  usage: tools/small_gemm.py [M=2352]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2352
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("HK_"))
for name, N, K in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device=DEV)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    for bc in (False, True):
        d = ops.gemm_desc(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, bias=b.data_ptr(),
                          a_rows=(K, 128, 0) if bc else (K, 0, 0), c_rows=(N, 0, 0))
        us = timeit(lambda: ops.run_gemm(d), iters=50, warm=5)
        print(f"[{tag}] M={M} {name:5s} N={N} K={K} A-broadcast={int(bc)}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)
    if name in ("proj", "fc2") and os.environ.get("SMALL_GEMM_SPLITS"):
        res = torch.randn(M, N, device=DEV)
        for S in (int(v) for v in os.environ["SMALL_GEMM_SPLITS"].split(",")):
            xr = torch.randn(M, N, device=DEV).to(torch.bfloat16)
            def both():
                parts = ops.linear_split_k(x, w, b, None, S)
                ops.residual_add_parts(xr, parts, 1e-6)
            us = timeit(both, iters=50, warm=5)
            print(f"[{tag}] M={M} {name:5s} N={N} K={K} split_k={S} + residual_add_parts: {us:7.1f} us", flush=True)
