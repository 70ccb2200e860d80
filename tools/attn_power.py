#!/usr/bin/env python3
"""ViT attention launch (B=32, N=3136, H=16) on random vs all-zero q/k/v: separates data-dependent power/clock limits
from issue-slot limits (the exp / MFMA instruction stream is identical)."""
import os, sys
import torch

torch.set_grad_enabled(False)   # inference tools: encode_img follows autograd mode like the reference module
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops
from tools.bench_kernels import timeit
DEV = "cuda:0"
B, N, H = 32, 3136, 16
D = H * 64
for kind in ("random", "zeros", "random"):
    qkv = (torch.randn(B, N, 3 * D, device=DEV) * 0.8).to(torch.bfloat16)
    if kind == "zeros":
        qkv.zero_()
    q, k, v = (qkv[:, :, i * D:(i + 1) * D].unflatten(2, (H, 64)) for i in range(3))
    out = torch.empty(B, N, H, 64, device=DEV, dtype=torch.bfloat16)
    us = timeit(lambda: ops.flash_attention(q, k, v, out=out), iters=20)
    print(f"attn {kind}: {us:8.1f} us {4.0*B*H*N*N*64/us/1e6:7.1f} TFLOP/s", flush=True)
