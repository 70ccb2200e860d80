"""Separates what limits the persistent GEMM when all 256 CUs run: the same launch with (a) normal operands,
(b) every M-tile reading the same 256 rows of A (all operand traffic L2-resident, same MFMA work),
(c) zero operands (lowest MFMA switching power), (d) both."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

DEV = "cuda:0"
for (M, N, K) in ((98304, 1024, 4096), (98304, 4096, 1024)):
    for zeros in (False, True):
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
        if zeros:
            x.zero_()
            w.zero_()
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        for bc in (False, True):
            d = ops.gemm_desc(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K,
                              a_rows=(K, 256, 0) if bc else (K, 0, 0), c_rows=(N, 0, 0))
            us = timeit(lambda: ops.run_gemm(d), iters=20)
            print(f"M={M} N={N} K={K} zeros={zeros} A-broadcast={bc}: {us:8.1f} us {2.0*M*N*K/us/1e6:8.1f} TF", flush=True)
