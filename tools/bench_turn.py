#!/usr/bin/env python3
"""One eval turn as the reference's scripts run it (tasks/test_recursive_grounding.py:409-466, one clip per call): encode_img
(B=1, T=12) -> splice into the prompt -> Vicuna-7B prefill -> greedy decode of NEW tokens, all on the HIP path.  Reports ms per
stage and per turn.  python tools/bench_turn.py [new_tokens=20] [prompt_tokens=256]"""
import json
import os
import sys
import time

import torch

torch.set_grad_enabled(False)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import weights as W  # noqa: E402
from hawkeye_amd.config import FULL, VICUNA_7B  # noqa: E402
from hawkeye_amd.decode import LlamaGenerator  # noqa: E402
from hawkeye_amd.llm import LlamaPrefill  # noqa: E402
from hawkeye_amd.model import HawkEyeEncoder  # noqa: E402

NEW = int(sys.argv[1]) if len(sys.argv) > 1 else 20
PROMPT = int(sys.argv[2]) if len(sys.argv) > 2 else 256
DEV = torch.device("cuda", 0)
B, T = int(os.environ.get("B", 1)), 12
enc = HawkEyeEncoder(FULL, out_dtype=torch.bfloat16)
enc.load_state_dict(W.make_state_dict(FULL, 1), strict=True)
enc = enc.to(DEV).eval()
enc.device_pos_embed = True
llm = LlamaPrefill(VICUNA_7B, DEV).init_random(0)
gen = LlamaGenerator(llm)
S = PROMPT + FULL.n_query
image = W.make_clip(B, T, FULL.img_size, 7).to(DEV)
ids, mask = W.make_instruction_ids(B, 32, FULL.qf_vocab, 7)
ids, mask = ids.to(DEV), mask.to(DEV)
src, _ = W.make_prompt(VICUNA_7B, B, S, FULL.n_query, 3, ragged=False)
src = src.to(DEV)


def turn():
    t0 = time.perf_counter()
    vid, _ = enc.encode_img(image, (ids, mask))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = gen.generate(src=src, video=vid, max_new_tokens=NEW, stop_sequences=[[VICUNA_7B.vocab_size + 1]])   # never matches: full length
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, out


for _ in range(3):
    turn()
runs = [turn() for _ in range(10)]
e = min(r[0] for r in runs) * 1e3
g = min(r[1] for r in runs) * 1e3
tot = min(r[0] + r[1] for r in runs) * 1e3
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    llm.prefill(src=src, video=torch.zeros((B * FULL.n_query, VICUNA_7B.hidden_size), dtype=torch.bfloat16, device=DEV))
torch.cuda.synchronize()
pf = (time.perf_counter() - t0) * 1e3 / 5
print(json.dumps({"workload": f"one eval turn, B={B}: encode_img T={T} -> prompt {S} tokens (96 video) -> {NEW} greedy tokens, Vicuna-7B shape, host wall clock incl. syncs",
                  "encode_img_ms": round(e, 3), "generate_ms": round(g, 3), "of_which_prefill_ms": round(pf, 3),
                  "decode_ms_per_token": round((g - pf) / max(NEW - 1, 1), 3), "turn_ms": round(tot, 3), "turns_per_s": round(B * 1e3 / tot, 2),
                  "tokens": list(runs[-1][2]["tokens"].shape)}))
