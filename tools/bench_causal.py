#!/usr/bin/env python3
"""Causal head-128 attention at the Cfg4 prefill shape (B=8, H=32, S=352) and longer, with / without key mask."""
import os, sys, math
import torch

torch.set_grad_enabled(False)   # inference tools: encode_img follows autograd mode like the reference module
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops
from tools.bench_kernels import timeit
DEV = "cuda:0"
for (B, H, S) in ((8, 32, 352), (8, 32, 1024), (2, 32, 2048)):
    qkv = (torch.randn(B, S, 3 * H * 128, device=DEV) * 0.5).to(torch.bfloat16)
    q, k, v = (qkv[:, :, i * H * 128:(i + 1) * H * 128].unflatten(2, (H, 128)) for i in range(3))
    out = torch.empty(B, S, H, 128, device=DEV, dtype=torch.bfloat16)
    kb = torch.zeros(B, S, device=DEV)
    fl = 4.0 * B * H * 128 * S * (S + 1) / 2
    for name, bias in (("nomask", None), ("mask", kb)):
        us = timeit(lambda: ops.flash_attention(q, k, v, key_bias=bias, out=out, causal=True), iters=20)
        print(f"causal hd128 B={B} H={H} S={S} {name}: {us:8.1f} us {fl/us/1e6:7.1f} TFLOP/s", flush=True)
