"""hk_gemv_bf16 at the decode step's shapes with COLD weights (a ring of weight buffers larger than the Infinity Cache, as in a
real step where 13 GB stream by per token): us per launch and GB/s.  HK_GEMV_KS / HK_GEMV_XLDS select variants.
python tools/gemv_bench.py [M]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import lib as L, ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
lib = L.load()
shapes = [("qkv", 12288, 4096, "none", True), ("o_proj", 4096, 4096, "res", False), ("gate_up", 22016, 4096, "swiglu", True),
          ("down", 4096, 11008, "res", False), ("lm_head", 32000, 4096, "f32", True)]
for name, N, K, kind, norm in shapes:
    nbuf = max(2, int(1.2e9 // (N * K * 2)))
    ws = [torch.randn((N, K), device=dev, dtype=torch.float32).mul_(0.02).to(torch.bfloat16) for _ in range(nbuf)]
    TILED = os.environ.get("GEMV_TILED", "0") == "1"
    if TILED:
        ws = [ops.tile_gemv_weight(w) for w in ws]
    x32 = torch.randn((M, K), device=dev)
    xb = x32.to(torch.bfloat16)
    gamma = torch.ones(K, device=dev)
    res = torch.zeros((M, N), device=dev)
    out = torch.empty((M, N), device=dev, dtype=torch.float32 if kind in ("res", "f32") else torch.bfloat16)
    descs = []
    for w in ws:
        kw = dict(out_f32=kind in ("res", "f32"), act="swiglu" if kind == "swiglu" else "none", w_tiled=TILED)
        if kind == "res":
            kw["residual"] = res.data_ptr()
        if norm and M <= 4:
            d = ops.gemv_desc(x32.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, norm_weight=gamma.data_ptr(), norm_eps=1e-6, **kw)
        else:
            d = ops.gemv_desc(xb.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, **kw)
        descs.append(d)
    st = torch.cuda.current_stream().cuda_stream
    for d in descs:
        L.check(lib.hk_gemv_bf16(C.byref(d), st), "gemv")
    torch.cuda.synchronize()
    reps = 6
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for d in descs:
            lib.hk_gemv_bf16(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * nbuf)
    print(f"M={M} {name:8s} N={N:6d} K={K:6d} KS={os.environ.get('HK_GEMV_KS', 'auto'):>4s} tiled={int(TILED)}: {us:7.1f} us  {N * K * 2 / us / 1e6:6.2f} TB/s", flush=True)
    del ws
