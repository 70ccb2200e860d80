#!/usr/bin/env python3
"""Split-K of the prefill's o_proj / down_proj at a skinny M (B = 1 prompts): raw partial-product GEMM time per split factor."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops
from tools.bench_kernels import timeit
DEV = "cuda:0"
torch.set_grad_enabled(False)
M = int(os.environ.get("LLM_M", 352))
for name, N, K in (("o_proj", 4096, 4096), ("down_proj", 4096, 11008)):
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
    res = torch.randn(M, N, device=DEV)
    line = f"M={M} {name} N={N} K={K}:"
    us = timeit(lambda: ops.linear(x, w, None, residual=res, out_dtype=torch.float32, out=res), iters=30, warm=3)
    line += f" unsplit {us:6.1f} us;"
    for S in (2, 4, 8):
        if (K // 64) % S: continue
        parts = torch.empty((S, M, N), dtype=torch.float32, device=DEV)
        d = ops.gemm_desc(x.data_ptr(), w.data_ptr(), parts.data_ptr(), M, N, K, residual=res.data_ptr(), a_rows=(K, 0, 0), c_rows=(N, 0, 0),
                          r_rows=(N, 0, 0), out_f32=True, split_k=S, split_stride=M * N)
        us = timeit(lambda: ops.run_gemm(d), iters=30, warm=3)
        line += f" S={S} {us:6.1f} us;"
    print(line, flush=True)
