#!/usr/bin/env python3
"""B=1 call latency (the recursive-grounding shape); the HK_* switches in the environment are printed with each line
(run it twice in one gpurun call for a same-box A/B, e.g. HK_SPLIT_K=0 / 1, HK_GRAPH=0 / 1)."""
import os, sys, time
import torch

torch.set_grad_enabled(False)   # inference tools: encode_img follows autograd mode like the reference module
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import weights as W
from hawkeye_amd.config import FULL
from hawkeye_amd.model import HawkEyeEncoder
DEV = torch.device("cuda", 0)
cfg = FULL
model = HawkEyeEncoder(cfg, out_dtype=torch.bfloat16)
model.load_state_dict(W.make_state_dict(cfg, 1), strict=True)
model = model.to(DEV).eval()
model.device_pos_embed = True
SHAPES = ((1, 1), (1, 8), (1, 12), (1, 16), (4, 12), (4, 16))
if len(sys.argv) > 1:      # e.g. 8x12,8x16
    SHAPES = tuple(tuple(int(v) for v in a.split('x')) for a in sys.argv[1].split(','))
for B, T in SHAPES:
    image = W.make_clip(B, T, cfg.img_size, 7).to(DEV)
    ids, mask = W.make_instruction_ids(B, 32, cfg.qf_vocab, 7)
    ids, mask = ids.to(DEV), mask.to(DEV)
    for _ in range(3):
        model.encode_img(image, (ids, mask))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        out, _ = model.encode_img(image, (ids, mask))
        if os.environ.get("B1_SYNC") == "1":      # the eval scripts' pattern: the LLM consumes every result
            float(out[0, 0, 0])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("HK_") or k == "B1_SYNC")
    print(f"[{tag}] B={B} T={T}: {ms:.3f} ms/call  {B/ms*1e3:.1f} clips/s", flush=True)
