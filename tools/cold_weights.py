#!/usr/bin/env python3
"""Prefill GEMMs (B = 8, S = 352) with HOT weights (one weight buffer, resident in the Infinity Cache between launches) against COLD
ones (a ring of buffers larger than the cache, as a real prefill meets them: every layer's weights are read once from HBM)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops
DEV = "cuda:0"
torch.set_grad_enabled(False)
M = int(os.environ.get('LLM_M', 2816))
def timeit(fns, iters):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, N, K, kind in (("qkv", 12288, 4096, 0), ("o_proj", 4096, 4096, 1), ("gate_up+swiglu", 22016, 4096, 5), ("down_proj", 4096, 11008, 1)):
    nbuf = max(2, int(1.6e9 // (N * K * 2)))
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    res = torch.randn(M, N, device=DEV) if kind == 1 else None
    out = torch.empty(M, N // 2 if kind == 5 else N, device=DEV, dtype=torch.float32 if kind == 1 else torch.bfloat16)
    def mk(w):
        if kind == 1: return lambda: ops.linear(x, w, None, residual=res, out_dtype=torch.float32, out=out)
        if kind == 5: return lambda: ops.linear(x, w, None, act="swiglu", out=out)
        return lambda: ops.linear(x, w, None, out=out)
    hot = min(timeit([mk(ws[0])], 24) for _ in range(3))
    cold = min(timeit([mk(w) for w in ws], 3 * nbuf) for _ in range(3))
    fl = 2.0 * M * N * K
    print(f"{name:15s} N={N} K={K}: hot {hot:7.1f} us ({fl/hot/1e6:6.1f} TF)  cold ({nbuf} buffers) {cold:7.1f} us ({fl/cold/1e6:6.1f} TF)  +{(cold/hot-1)*100:.1f} %", flush=True)
    del ws
