#!/usr/bin/env python3
"""One launch of each hot kernel at production shape, for rocprofv3 --pmc passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops
DEV = "cuda:0"
M = 100352
what = sys.argv[1:] or ["gemm", "attn"]
if "gemm" in what:
    for N, K in ((4096, 1024), (1024, 4096)):
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        for _ in range(3):
            ops.linear(x, w, None, out=out)
        torch.cuda.synchronize()
if "attn" in what:
    B, N, H = 32, 3136, 16
    qkv = (torch.randn(B, N, 3 * H * 64, device=DEV) * 0.8).to(torch.bfloat16)
    D = H * 64
    q = qkv[:, :, :D].unflatten(2, (H, 64)); k = qkv[:, :, D:2 * D].unflatten(2, (H, 64)); v = qkv[:, :, 2 * D:].unflatten(2, (H, 64))
    out = torch.empty(B, N, H, 64, device=DEV, dtype=torch.bfloat16)
    for _ in range(3):
        ops.flash_attention(q, k, v, out=out)
    torch.cuda.synchronize()
