#!/usr/bin/env python3
"""SURVEY.md §8d sweeps on one MI355X: encode_img clips/s over B x T (Cfg 2 sweep), the ragged-instruction variant,
the uint8-frame entry (row f3), and the Cfg 3 recursive-grounding stand-in (a stream of fresh B=1, T=12 clips, and
the same turns micro-batched across 8 / 32 independent examples).  Inputs resident in HBM; HIP events on the
launch stream; one JSON document on stdout."""
import json
import os
import sys
import time

import torch

torch.set_grad_enabled(False)   # inference tools: encode_img follows autograd mode like the reference module

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import weights as W  # noqa: E402
from hawkeye_amd.config import FULL  # noqa: E402
from hawkeye_amd.engine import flops_per_clip  # noqa: E402
from hawkeye_amd.model import HawkEyeEncoder  # noqa: E402

DEV = torch.device("cuda", 0)
cfg = FULL
model = HawkEyeEncoder(cfg, out_dtype=torch.bfloat16)
model.load_state_dict(W.make_state_dict(cfg, 1), strict=True)
model = model.to(DEV).eval()
model.device_pos_embed = True     # row f4: table per T generated on the device, cached


def timed(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, (time.perf_counter() - t0) * 1e3 / iters


res = {"sweep": [], "notes": "bf16, synthetic N(0,1) clips, Lt=32, one MI355X; ms = device time per call"}
for T in (1, 8, 12, 16):
    for B in (1, 8, 32, 64):
        image = W.make_clip(B, T, cfg.img_size, 7).to(DEV)
        ids, mask = W.make_instruction_ids(B, 32, cfg.qf_vocab, 7)
        ids, mask = ids.to(DEV), mask.to(DEV)
        iters = max(3, min(30, int(2000 / (B * max(T, 2)))))
        ms, wall = timed(lambda: model.encode_img(image, (ids, mask)), iters)
        fl = flops_per_clip(cfg, T, 32) * B
        res["sweep"].append({"B": B, "T": T, "ms": round(ms, 3), "wall_ms": round(wall, 3),
                             "clips_per_s": round(B / wall * 1e3, 2), "tflops": round(fl / ms / 1e9, 1)})
        print(res["sweep"][-1], file=sys.stderr, flush=True)
        del image
    model._engine.plans.clear()
    torch.cuda.empty_cache()

# ragged instructions (lengths 8..32, padded to 32): exercises the key mask; same shapes as Cfg 2
B, T = 32, 16
image = W.make_clip(B, T, cfg.img_size, 9).to(DEV)
ids, mask = W.make_instruction_ids(B, 32, cfg.qf_vocab, 9, ragged=True)
ids, mask = ids.to(DEV), mask.to(DEV)
ms, wall = timed(lambda: model.encode_img(image, (ids, mask)), 10)
res["cfg2_ragged"] = {"B": B, "T": T, "ms": round(ms, 3), "clips_per_s": round(B / wall * 1e3, 2)}
# uint8 frames (row f3): normalisation inside the patch gather, 4x less input traffic
g = torch.Generator(device="cpu").manual_seed(3)
frames = torch.randint(0, 256, (B, T, cfg.img_size, cfg.img_size, 3), generator=g, dtype=torch.uint8).to(DEV)
ms, wall = timed(lambda: model.encode_frames(frames, (ids, mask)), 10)
res["cfg2_uint8_frames"] = {"B": B, "T": T, "ms": round(ms, 3), "clips_per_s": round(B / wall * 1e3, 2)}
del image, frames
model._engine.plans.clear()
torch.cuda.empty_cache()

# Cfg 3 stand-in: recursive grounding = 4 turns per example, each turn a fresh T=12 clip; LLM time excluded.
T = 12
n_clips = 240
pool = [W.make_clip(1, T, cfg.img_size, 1000 + i).to(DEV) for i in range(8)]
ids1, mask1 = W.make_instruction_ids(1, 32, cfg.qf_vocab, 5)
ids1, mask1 = ids1.to(DEV), mask1.to(DEV)
k = [0]


def one_turn():
    k[0] += 1
    out, _ = model.encode_img(pool[k[0] % 8], (ids1, mask1))
    return out


ms, wall = timed(one_turn, n_clips, warm=4)
res["cfg3_sequential_b1_t12"] = {"clips": n_clips, "ms_per_clip": round(wall, 3), "clips_per_s": round(1e3 / wall, 2),
                                 "assumption": "3720 examples x 4 turns (Charades-STA test split) => "
                                               f"{3720 * 4 * wall / 1e3:.0f} s of encode_img on one GPU"}
for mb in (8, 32):
    image = W.make_clip(mb, T, cfg.img_size, 77).to(DEV)
    idb, mkb = W.make_instruction_ids(mb, 32, cfg.qf_vocab, 5, ragged=True)
    idb, mkb = idb.to(DEV), mkb.to(DEV)
    ms, wall = timed(lambda: model.encode_img(image, (idb, mkb)), 12)
    res[f"cfg3_microbatch_{mb}_t12"] = {"ms_per_call": round(wall, 3), "clips_per_s": round(mb / wall * 1e3, 2)}
    del image
print(json.dumps(res, indent=1))
