#!/usr/bin/env python3
"""Dump per-workgroup timestamps of one GEMM launch (HK_GEMM_TRACE) and summarise epilogue durations / phase alignment."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.environ.setdefault("HK_GEMM_TRACE", "/tmp/gemm_trace.bin")
# the trace switch exists only in the diagnostic build (make -C hawkeye_amd/csrc diag)
os.environ.setdefault("HK_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "hawkeye_amd", "libhawkeye_hip_diag.so"))
from hawkeye_amd import ops
DEV = "cuda:0"
# usage: gemm_trace.py [N K [res16|gelu]]
M = 100352
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3072, 1024)
mode = sys.argv[3] if len(sys.argv) > 3 else ""
x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
w = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
b = torch.randn(N, device=DEV)
out = torch.randn(M, N, device=DEV).to(torch.bfloat16)
if mode.startswith("fold"):
    g, bt = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    wf, cc, bf = ops.fold_layernorm(w.float(), b, g, bt)
    st = torch.zeros((M + 255) // 256 * 256, 2, device=DEV)
    ops.row_stats(x, 1e-6, out=st[:M])
for _ in range(3):
    if mode.startswith("fold"):
        ops.linear(x, wf, bf, act="gelu" if mode == "foldgelu" else "none", out=out, row_stats=st[:M], col_c=cc)
    elif mode == "res16":
        ops.linear(x, w, b, residual=out, out_dtype=torch.bfloat16, out=out)
    else:
        ops.linear(x, w, b, act="gelu" if mode == "gelu" else "none", out=out)
torch.cuda.synchronize()
print(f"N={N} K={K} {mode}")
a = np.fromfile(path, dtype=np.uint64).reshape(-1, 64).astype(np.int64)
print("workgroups", a.shape[0])
t0 = a[:, 1].min()
start = (a[:, 1] - t0) / 100.0
ntile = ((a[:, 2::2] != 0).sum(1))
print("tiles per wg: min/max", ntile.min(), ntile.max())
loop_end = (a[:, 2:64:2] - t0) / 100.0
epi_end = (a[:, 3:64:2] - t0) / 100.0
epi = []
loop = []
for i in range(a.shape[0]):
    prev = start[i]
    for t in range(ntile[i]):
        loop.append(loop_end[i, t] - prev)
        epi.append(epi_end[i, t] - loop_end[i, t])
        prev = epi_end[i, t]
epi, loop = np.array(epi), np.array(loop)
print(f"main loop per tile: mean {loop.mean():.2f} us  p10 {np.percentile(loop,10):.2f} p90 {np.percentile(loop,90):.2f}")
print(f"epilogue per tile:  mean {epi.mean():.2f} us  p10 {np.percentile(epi,10):.2f} p90 {np.percentile(epi,90):.2f}")
print(f"kernel span {max(epi_end.max(), 0):.1f} us; start spread {start.max():.1f} us")
# how aligned are epilogues: fraction of workgroups in epilogue at sampled times
ts = np.linspace(20, epi_end.max() - 20, 400)
frac = []
for t in ts:
    inside = ((loop_end <= t) & (epi_end > t) & (loop_end > 0)).any(1).mean()
    frac.append(inside)
frac = np.array(frac)
print(f"fraction of WGs inside an epilogue: mean {frac.mean():.3f} max {frac.max():.3f} p95 {np.percentile(frac,95):.3f}")
