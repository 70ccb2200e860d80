#!/usr/bin/env python3
"""A few encode_img calls of one shape (for rocprofv3 --kernel-trace): usage one_call.py B T [calls]; prints the number
of device launches per call on the last line."""
import os, sys
import torch

torch.set_grad_enabled(False)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import weights as W
from hawkeye_amd.config import FULL
from hawkeye_amd.model import HawkEyeEncoder
B, T = int(sys.argv[1]), int(sys.argv[2])
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 5
DEV = torch.device("cuda", 0)
cfg = FULL
model = HawkEyeEncoder(cfg, out_dtype=torch.bfloat16)
model.load_state_dict(W.make_state_dict(cfg, 1), strict=True)
model = model.to(DEV).eval()
model.device_pos_embed = True
image = W.make_clip(B, T, cfg.img_size, 7).to(DEV)
ids, mask = W.make_instruction_ids(B, 32, cfg.qf_vocab, 7)
ids, mask = ids.to(DEV), mask.to(DEV)
for _ in range(calls):
    model.encode_img(image, (ids, mask))
    torch.cuda.synchronize()
print(len(model._engine.plans[(B, T, 32)].steps))
