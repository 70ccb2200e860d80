#!/usr/bin/env python3
"""A few launches of the qkv (folded LayerNorm) and fc2 (in-place residual + row sums) GEMMs at M = 100 352 on both
large-problem kernels (256x256 one workgroup per CU; 256x128 two workgroups per CU), for rocprofv3 passes."""
import os, sys
import torch
torch.set_grad_enabled(False)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from duo_check import make, run, lib
M = 100352
for kind, N, K in ((3, 3072, 1024), (4, 1024, 4096)):
    args = make(M, N, K, kind)
    for mask in (0, 31):
        lib.hk_gemm_duo_mask(mask)
        for _ in range(3):
            run(kind, *args[:7], args[-1])
        torch.cuda.synchronize()
