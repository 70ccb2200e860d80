#!/usr/bin/env python3
"""Per-kernel timings at the production shapes (B=32, T=16) through the C ABI, HIP events on the
launch stream, random operands.  usage: tools/bench_kernels.py [attn] [gemm] [ln]"""
import os
import sys

import torch

torch.set_grad_enabled(False)   # inference tools: encode_img follows autograd mode like the reference module

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def bench_attn(B=32, N=3136, H=16):
    qkv = (torch.randn(B, N, 3 * H * 64, device=DEV) * 0.8).to(torch.bfloat16)
    D = H * 64
    q = qkv[:, :, :D].unflatten(2, (H, 64))
    k = qkv[:, :, D:2 * D].unflatten(2, (H, 64))
    v = qkv[:, :, 2 * D:].unflatten(2, (H, 64))
    out = torch.empty(B, N, H, 64, device=DEV, dtype=torch.bfloat16)
    us = timeit(lambda: ops.flash_attention(q, k, v, out=out))
    fl = 4.0 * B * H * N * N * 64
    print(f"attn  B={B} N={N} H={H}: {us:9.1f} us  {fl/us/1e6:8.1f} TFLOP/s", flush=True)
    # cross-attention shape (96 queries into N keys, 12 heads)
    kv = (torch.randn(B, N, 2 * 768, device=DEV) * 0.8).to(torch.bfloat16)
    qc = (torch.randn(B, 96, 12, 64, device=DEV) * 0.8).to(torch.bfloat16)
    kc = kv[:, :, :768].unflatten(2, (12, 64))
    vc = kv[:, :, 768:].unflatten(2, (12, 64))
    us = timeit(lambda: ops.flash_attention(qc, kc, vc))
    print(f"xattn B={B} Nq=96 Nk={N} H=12: {us:9.1f} us  {4.0*B*12*96*N*64/us/1e6:8.1f} TFLOP/s", flush=True)


def bench_gemm(M=100352, only=None):
    for name, N, K, kw in (("qkv", 3072, 1024, {}), ("proj+res", 1024, 1024, {"res": True}),
                           ("fc1+gelu", 4096, 1024, {"act": "gelu"}), ("fc2+res", 1024, 4096, {"res": True}),
                           ("fc1 noact", 4096, 1024, {}), ("ckv", 9216, 1024, {}), ("patch", 1024, 768, {"res": True}),
                           ("proj+res16", 1024, 1024, {"res": True, "bf16res": True}),
                           ("fc2+res16", 1024, 4096, {"res": True, "bf16res": True})):
        if only and name not in only:
            continue
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
        b = torch.randn(N, device=DEV)
        res = torch.randn(M, N, device=DEV) if kw.get("res") else None
        if kw.get("bf16res"):
            res = res.to(torch.bfloat16)
        odt = torch.float32 if (res is not None and not kw.get("bf16res")) else torch.bfloat16
        out = res if kw.get("bf16res") else torch.empty(M, N, device=DEV, dtype=odt)
        us = timeit(lambda: ops.linear(x, w, b, residual=res, act=kw.get("act", "none"), out_dtype=odt, out=out))
        print(f"gemm {name:10s} M={M} N={N} K={K}: {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
        del x, w, res, out


def bench_ln(M=100352, D=1024):
    x = torch.randn(M, D, device=DEV)
    g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    us = timeit(lambda: ops.layernorm(x, g, b, 1e-6))
    print(f"ln f32->bf16 M={M} D={D}: {us:8.1f} us  {M*D*6/us/1e6:6.2f} TB/s", flush=True)
    xb = x.to(torch.bfloat16)
    us = timeit(lambda: ops.layernorm(xb, g, b, 1e-6))
    print(f"ln bf16->bf16 M={M} D={D}: {us:8.1f} us  {M*D*4/us/1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["attn", "gemm", "ln"]
    if "attn" in what:
        bench_attn()
    if "gemm" in what:
        bench_gemm()
    if "gelu" in what:
        bench_gemm(only=("fc1+gelu", "fc1 noact"))
    if "hot" in what:   # the four ViT-block GEMMs as the engine runs them (bf16 residual stream)
        bench_gemm(only=("qkv", "proj+res16", "fc1+gelu", "fc2+res16"))
    if "ln" in what:
        bench_ln()
