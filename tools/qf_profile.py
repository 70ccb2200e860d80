#!/usr/bin/env python3
"""In-situ durations of the Q-Former launches of one encode_img call, grouped by role (B=32, T=16, Lt=32 by default)."""
import os, sys
import torch
torch.set_grad_enabled(False)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import posembed, weights as W
from hawkeye_amd.config import FULL
from hawkeye_amd.model import HawkEyeEncoder
B, T, Lt = (int(a) for a in (sys.argv[1:4] + ["32", "16", "32"][len(sys.argv) - 1:]))
cfg, dev = FULL, torch.device("cuda", 0)
model = HawkEyeEncoder(cfg, out_dtype=torch.bfloat16)
model.load_state_dict(W.make_state_dict(cfg, 1), strict=True)
model = model.to(dev).eval()
model.vision_encoder.encoder.pos_embed = posembed.video_pos_embed(T, cfg.patches_per_frame, cfg.encoder_embed_dim, cfg.ckpt_num_frame)
image = W.make_clip(B, T, cfg.img_size, 1).to(dev)
ids, mask = W.make_instruction_ids(B, Lt, cfg.qf_vocab, 1)
ids, mask = ids.to(dev), mask.to(dev)
model.encode_img(image, (ids, mask))
eng = model.sync_engine()
plan = eng.plans[(B, T, Lt)]
cross = ["qkv", "self_attn", "ao", "ao_ln", "cq", "cross_attn", "co", "co_ln", "qi_gelu", "qo", "qo_ln", "ti_gelu", "to", "to_ln"]
plain = ["qkv", "self_attn", "ao", "ao_ln", "qi_gelu", "qo", "qo_ln", "ti_gelu", "to", "to_ln"]
roles = ["embed_ln"]
for i in range(cfg.qf_layers):
    r = cross if i % cfg.cross_attention_freq == 0 else plain
    if i == cfg.qf_layers - 1 or Lt == 0:
        r = r[:-3]
    roles += r
steps, k = [], 0
for fn, args, name, tag in plan.steps:
    if tag == "qformer":
        tag = "qf." + roles[k]
        k += 1
    steps.append((fn, args, name, tag))
assert k == len(roles), (k, len(roles))
plan.steps = steps
table = model._table(T == 1, T * cfg.patches_per_frame, dev)
prof = eng.profile(image, ids, mask, table, reps=3)
tot = sum(n * us for t, (n, us) in prof.items() if t.startswith("qf.") or t in ("llama_proj", "qformer.cross_kv"))
for t, (n, us) in sorted(prof.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if t.startswith("qf.") or t in ("llama_proj", "qformer.cross_kv"):
        print(f"{t:18s} x{n:3d}  {us:8.1f} us  total {n * us / 1e3:6.3f} ms  {n * us / tot * 100:5.1f} %")
print(f"Q-Former side total {tot / 1e3:.3f} ms")
allt = sum(n * us for n, us in prof.values())
print(f"--- whole call: {len(plan.steps)} host launches, sum of launch durations {allt / 1e3:.3f} ms")
for t, (n, us) in sorted(prof.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if not t.startswith("qf."):
        print(f"{t:18s} x{n:3d}  {us:8.1f} us  total {n * us / 1e3:6.3f} ms  {n * us / allt * 100:5.1f} %")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    model.encode_img(image, (ids, mask))
e0.record()
for _ in range(20):
    model.encode_img(image, (ids, mask))
e1.record()
torch.cuda.synchronize()
print(f"device time per call (no per-launch events): {e0.elapsed_time(e1) / 20:.3f} ms")
