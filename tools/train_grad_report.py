#!/usr/bin/env python3
"""Per-tensor gradient agreement of the HIP training step with autograd through the fp32 oracle (row f2).
usage: tools/train_grad_report.py [T] [train|eval]   -> one line per parameter: cosine, max|err|/max|ref|"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import posembed, weights as W  # noqa: E402
from hawkeye_amd.config import FULL  # noqa: E402
from hawkeye_amd.model import HawkEyeEncoder  # noqa: E402
from oracle.train_pass import dropper, train_pass  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
train = (sys.argv[2] if len(sys.argv) > 2 else "train") == "train"
cfg, seed, B, Lt, DEV = FULL, 9, 2, 7, "cuda:0"
sd = W.make_state_dict(cfg, seed)
image = W.make_clip(B, T, cfg.img_size, seed)
ids, mask = W.make_instruction_ids(B, Lt, cfg.qf_vocab, seed, True)
table = posembed.video_pos_embed(T, cfg.patches_per_frame, cfg.encoder_embed_dim, cfg.ckpt_num_frame)
dout = W._normal("train_dout", seed, (B, cfg.n_query, cfg.llama_hidden), 1.0)
m = HawkEyeEncoder(cfg, out_dtype=torch.float32, freeze_qformer=False)
m.load_state_dict(sd, strict=True)
m = m.to(DEV).eval()
if train:
    m.qformer.train()
m.vision_encoder.encoder.pos_embed = table
m.set_dropout_seed(77)
out, _ = m.encode_img(image.to(DEV), (ids.to(DEV), mask.to(DEV)))
(out * dout.to(DEV)).sum().backward()
ref_out, ref = train_pass(cfg, sd, image, ids, mask, table, dout, dropper(cfg, 77) if train else None)
o = out.detach().cpu()
print("out cos", float((o.double().flatten() @ ref_out.double().flatten()) / (o.double().norm() * ref_out.double().norm())))
named = dict(m.named_parameters())
for k, r in ref.items():
    g = named[k].grad.detach().float().cpu()
    rmax = r.abs().max().item()
    cos = float((g.double().flatten() @ r.double().flatten()) / (g.double().norm() * r.double().norm() + 1e-30))
    print(f"{k:75s} cos {cos:.5f}  rel {((g - r).abs().max().item() / max(rmax, 1e-30)):.4f}  max {rmax:.3e}")
