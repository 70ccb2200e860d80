#!/usr/bin/env python3
"""BASELINE config 4 (SURVEY.md §8d Cfg 4): encode_img (B=8, T=16) -> splice into a 256-token prompt -> Vicuna-7B-shaped
prefill (random weights, 32 layers, 4096, 32 heads) on one MI355X.  Reports ms, TFLOP/s and encode_img's share."""
import json
import os
import sys
import time

import torch

torch.set_grad_enabled(False)   # inference tools: encode_img follows autograd mode like the reference module

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import weights as W  # noqa: E402
from hawkeye_amd.config import FULL, VICUNA_7B  # noqa: E402
from hawkeye_amd.engine import flops_per_clip  # noqa: E402
from hawkeye_amd.llm import LlamaPrefill, prefill_flops  # noqa: E402
from hawkeye_amd.model import HawkEyeEncoder  # noqa: E402

DEV = torch.device("cuda", 0)
B, T, PROMPT = int(os.environ.get("B", 8)), 16, 256
enc = HawkEyeEncoder(FULL, out_dtype=torch.bfloat16)
enc.load_state_dict(W.make_state_dict(FULL, 1), strict=True)
enc = enc.to(DEV).eval()
enc.device_pos_embed = True
llm = LlamaPrefill(VICUNA_7B, DEV).init_random(0)
S = PROMPT + FULL.n_query
image = W.make_clip(B, T, FULL.img_size, 7).to(DEV)
ids, mask = W.make_instruction_ids(B, 32, FULL.qf_vocab, 7)
ids, mask = ids.to(DEV), mask.to(DEV)
src, amask = W.make_prompt(VICUNA_7B, B, S, FULL.n_query, 3, ragged=True)
src, amask = src.to(DEV), amask.to(DEV)


def ev():
    return torch.cuda.Event(enable_timing=True)


def step():
    e = [ev() for _ in range(3)]
    e[0].record()
    vid, _ = enc.encode_img(image, (ids, mask))
    e[1].record()
    out = llm.prefill(src=src, video=vid, attention_mask=amask)
    e[2].record()
    return e, out


for _ in range(3):
    step()
torch.cuda.synchronize()
n, te, tp = 10, 0.0, 0.0
t0 = time.perf_counter()
for _ in range(n):
    e, out = step()
    torch.cuda.synchronize()
    te += e[0].elapsed_time(e[1])
    tp += e[1].elapsed_time(e[2])
wall = (time.perf_counter() - t0) * 1e3 / n
te, tp = te / n, tp / n
fe, fp = flops_per_clip(FULL, T, 32) * B, prefill_flops(VICUNA_7B, B, S)
print(json.dumps({"workload": f"Cfg4: encode_img B={B} T={T} -> prefill S={S} (256 prompt + 96 video), Vicuna-7B shape, bf16, random weights",
                  "encode_img_ms": round(te, 3), "prefill_ms": round(tp, 3), "wall_ms_per_step": round(wall, 3),
                  "encode_img_share": round(te / (te + tp), 4),
                  "encode_img_tflops": round(fe / te / 1e9, 1), "prefill_tflops": round(fp / tp / 1e9, 1),
                  "prefill_tokens_per_s": round(B * S / tp * 1e3, 1),
                  "logits_finite": bool(torch.isfinite(out["logits"]).all())}))
