#!/usr/bin/env python3
"""Q-Former masked self-attention launch (B=32, S=128, 12 heads, additive key bias) timed through the C ABI."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

B, S, H = int(os.environ.get("B", 32)), 128, 12
qkv = (torch.randn(B, S, 3 * H * 64, device="cuda") * 0.5).to(torch.bfloat16)
d = H * 64
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(3))
kb = torch.zeros(B, S, device="cuda")
kb[:, 120:] = -10000.0
out = torch.empty(B, S, H, 64, device="cuda", dtype=torch.bfloat16)
print(f"qf self-attn B={B} S={S}: {timeit(lambda: ops.flash_attention(q, k, v, key_bias=kb, out=out), iters=50):.1f} us", flush=True)
