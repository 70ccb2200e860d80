#!/usr/bin/env python3
"""One launch set of the round's hot kernels at production shapes, for rocprofv3 --pmc passes (tools/pmc_r03.sh):
folded-LayerNorm GEMMs (qkv, fc1+GELU), in-place residual GEMMs (proj, fc2), ViT attention, the prefill's causal attention,
SwiGLU gate|up GEMM, 192-row-tile o_proj, and the key-split training attention (cross shape of the stage-3 step)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops, train_ops as T
from tools.duo_check import make, run
DEV = "cuda:0"
torch.set_grad_enabled(False)
M = 100352
for kind, N, K in ((3, 3072, 1024), (3, 4096, 1024), (4, 1024, 1024), (4, 1024, 4096)):
    args = make(M, N, K, kind)
    for _ in range(2):
        run(kind, *args[:7], args[-1])
    torch.cuda.synchronize()
    del args
B, N, H = 32, 3136, 16
qkv = (torch.randn(B, N, 3 * H * 64, device=DEV) * 0.8).to(torch.bfloat16)
D = H * 64
q, k, v = (qkv[:, :, i * D:(i + 1) * D].unflatten(2, (H, 64)) for i in range(3))
out = torch.empty(B, N, H, 64, device=DEV, dtype=torch.bfloat16)
for _ in range(2):
    ops.flash_attention(q, k, v, out=out)
torch.cuda.synchronize()
del qkv, q, k, v, out
# prefill shapes (B = 8, S = 352, Vicuna-7B)
Bq, S, Hl = 8, 352, 32
q = (torch.randn(Bq, S, Hl, 128, device=DEV) * 0.5).to(torch.bfloat16)
k = (torch.randn(Bq, S, Hl, 128, device=DEV) * 0.5).to(torch.bfloat16)
v = (torch.randn(Bq, S, Hl, 128, device=DEV) * 0.5).to(torch.bfloat16)
for _ in range(2):
    ops.flash_attention(q, k, v, causal=True)
x = torch.randn(Bq * S, 4096, device=DEV).to(torch.bfloat16)
wgu = (torch.randn(22016, 4096, device=DEV) * 0.02).to(torch.bfloat16)
wo = (torch.randn(4096, 4096, device=DEV) * 0.02).to(torch.bfloat16)
res = torch.randn(Bq * S, 4096, device=DEV)
for _ in range(2):
    ops.linear(x, wgu, None, act="swiglu")
    ops.linear(x, wo, None, residual=res, out_dtype=torch.float32, out=res)
torch.cuda.synchronize()
# stage-3 cross-attention of the training step (B = 2, 12 heads, 96 queries, 2353 keys)
qt = (torch.randn(2, 96, 12, 64, device=DEV) * 0.2).to(torch.bfloat16)
kt = (torch.randn(2, 2353, 12, 64, device=DEV) * 0.8).to(torch.bfloat16)
vt = (torch.randn(2, 2353, 12, 64, device=DEV) * 0.8).to(torch.bfloat16)
do = (torch.randn(2, 96, 12, 64, device=DEV) * 0.8).to(torch.bfloat16)
o32 = torch.empty((2, 96, 12, 64), dtype=torch.float32, device=DEV)
for _ in range(2):
    o, lse = T.attn_train_fwd(qt, kt, vt, None, scale=0.125, p_drop=0.1, seed=3, site=5, o_f32=o32)
    T.attn_train_bwd(qt, kt, vt, o, lse, do, None, scale=0.125, p_drop=0.1, seed=3, site=5, o_f32=o32)
torch.cuda.synchronize()
print("done")
