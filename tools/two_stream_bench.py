#!/usr/bin/env python3
"""Experiment: one encode_img over 32 clips vs two encoders over 16 clips each on two HIP streams (the tail of every
kernel of one half overlapping the head of the other half's).  Prints ms per 32 clips for both."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hawkeye_amd import posembed, weights as W  # noqa: E402
from hawkeye_amd.config import FULL  # noqa: E402
from hawkeye_amd.model import HawkEyeEncoder  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
cfg, seed, T, Lt = FULL, 1, 16, 32
sd = W.make_state_dict(cfg, seed)


def make():
    m = HawkEyeEncoder(cfg, out_dtype=torch.bfloat16)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    m.vision_encoder.encoder.pos_embed = posembed.video_pos_embed(T, cfg.patches_per_frame, cfg.encoder_embed_dim, cfg.ckpt_num_frame)
    return m


def inputs(B, s):
    image = W.make_clip(B, T, cfg.img_size, s).to(dev)
    ids, mask = W.make_instruction_ids(B, Lt, cfg.qf_vocab, seed)
    return image, (ids.to(dev), mask.to(dev))


def timed(fn, steps=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


m0 = make()
img32, ins32 = inputs(32, 1)
print(f"one stream, B=32: {timed(lambda: m0.encode_img(img32, ins32)):.2f} ms per 32 clips", flush=True)
m1 = make()
ia, na = inputs(16, 1)
ib, nb = inputs(16, 2)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()


def two():
    with torch.cuda.stream(s0):
        m0.encode_img(ia, na)
    with torch.cuda.stream(s1):
        m1.encode_img(ib, nb)


print(f"two streams, 2 x B=16: {timed(two):.2f} ms per 32 clips", flush=True)
print(f"one stream, B=16 twice: {timed(lambda: (m0.encode_img(ia, na), m1.encode_img(ib, nb))):.2f} ms per 32 clips", flush=True)
