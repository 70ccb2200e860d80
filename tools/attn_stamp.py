#!/usr/bin/env python3
"""Where a wave of the pipelined attention kernel spends its cycles (diagnostic build `make -C hawkeye_amd/csrc stamp`,
loaded through HK_LIB_PATH): per-iteration s_memtime totals of  DMA issue | Q.K^T + softmax | check + P.V | vmcnt wait |
barrier,  averaged over all waves of one ViT-shape launch (B=32, N=3136, 16 heads)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HK_LIB_PATH", os.path.join(ROOT, "hawkeye_amd", "libhawkeye_hip_stamp.so"))
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from hawkeye_amd import ops  # noqa: E402

B, N, H = int(os.environ.get("B", 32)), int(os.environ.get("N", 3136)), 16
qkv = (torch.randn(B, N, 3 * H * 64, device="cuda") * 0.8).to(torch.bfloat16)
D = H * 64
q, k, v = (qkv[:, :, i * D:(i + 1) * D].unflatten(2, (H, 64)) for i in range(3))
out = torch.empty(B, N, H, 64, device="cuda", dtype=torch.bfloat16)
NW = 8 if os.environ.get("HK_ATTN_PIPE") == "8" else 4
nwg = B * H * ((N + NW * 32 - 1) // (NW * 32))
buf = torch.zeros(nwg * NW * 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    ops.flash_attention(q, k, v, out=out)
torch.cuda.synchronize()
os.environ["HK_ATTN_STAMP_PTR"] = hex(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.flash_attention(q, k, v, out=out)
e1.record()
torch.cuda.synchronize()
t = buf.view(nwg * NW, 8).double().cpu()
nt = t[:, 5]
ok = nt > 0
per = t[ok, :5] / nt[ok, None]
names = ["dma issue", "qk + softmax", "check + pv", "vmcnt wait", "barrier"]
print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us; waves stamped {int(ok.sum())}; cycles per iteration and wave:")
for i, n in enumerate(names):
    c = per[:, i]
    print(f"  {n:14s} mean {c.mean():8.1f}  p10 {c.quantile(0.1):8.1f}  p90 {c.quantile(0.9):8.1f}")
print(f"  total          mean {per.sum(1).mean():8.1f}")
