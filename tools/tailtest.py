import sys, os, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
from hawkeye_amd import ops
from tools.bench_kernels import timeit
DEV="cuda:0"
for (M,N,K) in ((98304,1024,4096),(2048,1024,4096),(100352,1024,4096),(98304,1024,1024),(2048,1024,1024),(100352,1024,1024)):
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device=DEV)
    res = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    us = timeit(lambda: ops.linear(x, w, b, residual=res, out=res), iters=20)
    print(f"M={M} N={N} K={K}: {us:8.1f} us {2.0*M*N*K/us/1e6:8.1f} TF", flush=True)
