#!/usr/bin/env python3
"""In-situ per-launch durations of one encode_img call (engine.profile): usage profile_call.py B T"""
import os, sys
import torch

torch.set_grad_enabled(False)   # inference tools: encode_img follows autograd mode like the reference module
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import weights as W
from hawkeye_amd.config import FULL
from hawkeye_amd.model import HawkEyeEncoder
B, T = int(sys.argv[1]), int(sys.argv[2])
DEV = torch.device("cuda", 0)
cfg = FULL
model = HawkEyeEncoder(cfg, out_dtype=torch.bfloat16)
model.load_state_dict(W.make_state_dict(cfg, 1), strict=True)
model = model.to(DEV).eval()
model.device_pos_embed = True
image = W.make_clip(B, T, cfg.img_size, 7).to(DEV)
ids, mask = W.make_instruction_ids(B, 32, cfg.qf_vocab, 7)
ids, mask = ids.to(DEV), mask.to(DEV)
for _ in range(3):
    model.encode_img(image, (ids, mask))
eng = model.sync_engine()
table = model._table(T == 1, T * cfg.patches_per_frame, DEV)
prof = eng.profile(image, ids, mask, table, reps=5)
tot = sum(n * us for n, us in prof.values())
print(f"B={B} T={T}: sum of launches {tot/1e3:.3f} ms")
for tag, (n, us) in sorted(prof.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    print(f"  {tag:18s} x{n:4d}  {us:8.1f} us  {n*us/tot*100:5.1f} %")
