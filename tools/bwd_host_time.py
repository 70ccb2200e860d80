#!/usr/bin/env python3
"""Host enqueue time vs device time of the training backward (B=2, T=12), weight gradients in line / on the side stream."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import posembed, weights as W
from hawkeye_amd.config import FULL
from hawkeye_amd.model import HawkEyeEncoder
dev = torch.device("cuda", 0)
cfg, seed, B, T, Lt = FULL, 1, 2, 12, 32
m = HawkEyeEncoder(cfg, out_dtype=torch.float32, freeze_qformer=False)
m.load_state_dict(W.make_state_dict(cfg, seed), strict=True)
m = m.to(dev).train()
m.vision_encoder.encoder.pos_embed = posembed.video_pos_embed(T, cfg.patches_per_frame, cfg.encoder_embed_dim, cfg.ckpt_num_frame)
image = W.make_clip(B, T, cfg.img_size, seed).to(dev)
ids, mask = W.make_instruction_ids(B, Lt, cfg.qf_vocab, seed, True)
ids, mask = ids.to(dev), mask.to(dev)
dout = W._normal("train_dout", seed, (B, cfg.n_query, cfg.llama_hidden), 1.0).to(dev)
te = m.sync_train_engine()
table = m._table(False, T * cfg.patches_per_frame, dev)
for side in (False, True, False, True):
    te.wgrad_side = side
    host, devt = [], []
    for it in range(12):
        _, plan = te.forward(image, ids, mask, table, train=True, seed=it)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        te.backward(plan, dout, None)
        e1.record(); t1 = time.perf_counter()
        torch.cuda.synchronize()
        if it >= 2:
            host.append((t1 - t0) * 1e3); devt.append(e0.elapsed_time(e1))
    print(f"side={int(side)}: host enqueue {sum(host)/len(host):.3f} ms, device span {sum(devt)/len(devt):.3f} ms", flush=True)
