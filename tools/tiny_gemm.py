#!/usr/bin/env python3
"""Fixed vs per-K-step cost of the small Q-Former GEMMs (rows = B*S = 256, N = 768): prebuilt descriptors, 100 launches per
shape back to back; read the kernel durations from a rocprofv3 kernel trace of this script (tools/prof_tiny.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops
DEV = "cuda:0"
for M, N in ((256, 768), (192, 3072)):
    for K in (64, 256, 768, 3072):
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
        b = torch.randn(N, device=DEV)
        res = torch.randn(M, N, device=DEV)
        out = torch.empty(M, N, device=DEV)
        d = ops.gemm_desc(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, bias=b.data_ptr(), residual=res.data_ptr(),
                          a_rows=(K, 0, 0), c_rows=(N, 0, 0), r_rows=(N, 0, 0), out_f32=True)
        torch.cuda.synchronize()
        for _ in range(100):
            ops.run_gemm(d)
        torch.cuda.synchronize()
        print(f"M={M} N={N} K={K} done", flush=True)
