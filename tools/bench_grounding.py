#!/usr/bin/env python3
"""BASELINE config 3 (SURVEY.md §8d Cfg 3): recursive grounding, synthetic stand-in for Charades-STA -- `--examples`
examples x 4 turns, every turn a fresh T=12 clip, uint8 frames staged from the host.  Reports clips/s for the
reference's access pattern (one clip per encode call, micro_batch 1) and through RecursiveGroundingBatcher at
micro-batches of 8 / 32, H2D and host frame generation INCLUDED (that is the point of the pinned double buffer).
LLM generation between turns is excluded (answer_fn is a table lookup that forces a host sync on the result).

    python tools/bench_grounding.py [--examples 64] [--turns 4]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_grounding.py   (examples sharded)
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import dist as hdist  # noqa: E402
from hawkeye_amd import grounding as G  # noqa: E402
from hawkeye_amd import posembed, weights as W  # noqa: E402
from hawkeye_amd.config import FULL  # noqa: E402
from hawkeye_amd.model import HawkEyeEncoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--examples", type=int, default=64)
    ap.add_argument("--turns", type=int, default=4)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--lt", type=int, default=32)
    ap.add_argument("--micro", type=str, default="1,8,32")
    args = ap.parse_args()
    rank, local_rank, world = hdist.init("nccl")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    cfg, seed, T = FULL, 1, args.frames
    m = HawkEyeEncoder(cfg, out_dtype=torch.bfloat16)
    m.load_state_dict(W.make_state_dict(cfg, seed), strict=True)
    m = m.to(dev).eval()
    m.vision_encoder.encoder.pos_embed = posembed.video_pos_embed(T, cfg.patches_per_frame, cfg.encoder_embed_dim, cfg.ckpt_num_frame)
    ids, mask = W.make_instruction_ids(1, args.lt, cfg.qf_vocab, seed)
    ins = (ids.to(dev), mask.to(dev))
    src = G.SyntheticFrameSource(size=cfg.img_size, seed=2)
    examples = list(range(args.examples))

    def answer(example, turn, out, bound):
        float(out[0, 0])                      # the LLM needs the result on the host side of the loop: a real sync
        return G.CHOICES[(example + turn) % 3]      # never 'throughout': every example runs all turns

    res = {}
    with torch.no_grad():
        for mb in [int(x) for x in args.micro.split(",")]:
            b = G.RecursiveGroundingBatcher(m.encode_frames, src, answer, ins, max_turns=args.turns, num_segments=T,
                                            micro_batch=mb, size=cfg.img_size, device=dev)
            b.run(examples[: max(2 * mb, 4)], rank, world)          # warm-up: plans, pinned buffers, base images
            hdist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b.stats = {k: 0 for k in b.stats}
            b.run(examples, rank, world)
            torch.cuda.synchronize()
            dt = hdist.max_over_ranks(time.perf_counter() - t0, dev)
            clips = args.examples * args.turns
            res[f"micro_batch_{mb}"] = {"clips_per_s": round(clips / dt, 1), "ms_per_clip": round(dt / clips * 1e3, 3),
                                        "encode_calls_rank0": b.stats["encode_calls"], "h2d_MB_rank0": round(b.stats["h2d_bytes"] / 1e6, 1)}
        # host-side cost alone (frame generation into the pinned buffer), for the split
        buf = torch.empty((T, cfg.img_size, cfg.img_size, 3), dtype=torch.uint8).numpy()
        t0 = time.perf_counter()
        for ex in examples[:32]:
            src.frames(ex, G.frame_indices(None, 3.0, 100, T), buf)
        res["host_frame_source_ms_per_clip"] = round((time.perf_counter() - t0) / 32 * 1e3, 3)
    if rank == 0:
        print(json.dumps({"workload": f"Cfg3 recursive grounding, synthetic: {args.examples} examples x {args.turns} turns, B=1 T={T} "
                                      f"clips re-encoded per turn, uint8 frames from host, Lt={args.lt}", "n_gpus": world, **res}), flush=True)
    hdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
