#!/usr/bin/env python3
"""Isolated timing of the key split (hk_attn_desc.key_splits) at the B = 1 shapes: ViT self-attention and the
Q-Former's cross-attention, T = 8 / 12 / 16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

DEV = "cuda:0"
for T in (8, 12, 16):
    N = 196 * T
    for name, H, Nq in (("vit", 16, N), ("cross", 12, 96)):
        q = (torch.randn(1, Nq, H, 64, device=DEV) * 0.8).to(torch.bfloat16)
        k = (torch.randn(1, N, H, 64, device=DEV) * 0.8).to(torch.bfloat16)
        v = (torch.randn(1, N, H, 64, device=DEV) * 0.8).to(torch.bfloat16)
        out = torch.empty_like(q)
        line = f"T={T:2d} {name:5s} Nq={Nq:4d} Nk={N}:"
        for S in (1, 2, 3, 4, 8):
            us = timeit(lambda: ops.flash_attention(q, k, v, out=out, key_splits=S), iters=50, warm=5)
            line += f"  S={S}: {us:6.1f} us"
        print(line, flush=True)
