#!/usr/bin/env python3
"""The four launches of tools/gemm_contention.py (fc2 shape) once each after warm-up, for a rocprofv3 --pmc pass:
order = normal, A-broadcast, zeros, zeros + A-broadcast (dispatch order identifies them)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops
DEV = "cuda:0"
M, N, K = 98304, 1024, 4096
for zeros in (False, True):
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
    if zeros:
        x.zero_(); w.zero_()
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    for bc in (False, True):
        d = ops.gemm_desc(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K,
                          a_rows=(K, 256, 0) if bc else (K, 0, 0), c_rows=(N, 0, 0))
        for _ in range(4):
            ops.run_gemm(d)
        torch.cuda.synchronize()
