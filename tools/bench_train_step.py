#!/usr/bin/env python3
"""BASELINE config 5 (SURVEY.md §8d Cfg 5), the encode_img side of the stage-3 instruction-tuning step:
frozen ViT on the inference kernels, Q-Former / query tokens / llama_proj in train mode (dropout 0.1, drop-path up to
0.2), backward from a synthetic upstream gradient, gradient all-reduce of the flat arena (RCCL when launched with
torchrun on > 1 GPU), refresh of the bf16 operands after an optimizer update.

    python tools/bench_train_step.py [--batch 2] [--frames 12] [--lt 32] [--steps 20]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train_step.py ...

The Vicuna-7B forward/backward with LoRA that sits between encode_img's output and the loss in the reference's step
(hawkeye_it.py:287-375) is NOT part of this measurement: it is not built (SURVEY.md §8f lists only its prefill, row f1).
The upstream gradient d(loss)/d(inputs_llama) is synthetic.  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import dist as hdist  # noqa: E402
from hawkeye_amd import posembed, weights as W  # noqa: E402
from hawkeye_amd.config import FULL  # noqa: E402
from hawkeye_amd.model import HawkEyeEncoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)        # scripts/train/config_7b_stage3.py:15
    ap.add_argument("--frames", type=int, default=12)      # :13
    ap.add_argument("--lt", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--autograd", action="store_true", help="go through encode_img + Tensor.backward instead of the engine API")
    args = ap.parse_args()
    rank, local_rank, world = hdist.init("nccl")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    cfg, seed, B, T, Lt = FULL, 1, args.batch, args.frames, args.lt
    sd = W.make_state_dict(cfg, seed)
    m = HawkEyeEncoder(cfg, out_dtype=torch.float32, freeze_qformer=False)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    m.set_dropout_seed(1000 + rank)
    m.vision_encoder.encoder.pos_embed = posembed.video_pos_embed(T, cfg.patches_per_frame, cfg.encoder_embed_dim,
                                                                  cfg.ckpt_num_frame)
    image = W.make_clip(B, T, cfg.img_size, seed + 100 * rank).to(dev)
    ids, mask = W.make_instruction_ids(B, Lt, cfg.qf_vocab, seed, True)
    ids, mask = ids.to(dev), mask.to(dev)
    dout = W._normal("train_dout", seed, (B, cfg.n_query, cfg.llama_hidden), 1.0).to(dev)
    te = m.sync_train_engine()
    table = m._table(False, T * cfg.patches_per_frame, dev)
    reducer = hdist.BucketReducer(te.arena.grad, average=True, bucket_bytes=args.bucket_mb << 20)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    acc = {"pack": 0.0, "vit_fwd": 0.0, "qformer_fwd": 0.0, "backward+allreduce": 0.0}

    def step(timed):
        if args.autograd:
            for p in te.arena.params:
                p.grad = None
            out, _ = m.encode_img(image, (ids, mask))
            out.backward(dout)
            m.all_reduce_gradients(bucket_mb=args.bucket_mb)
            return
        e = [ev() for _ in range(5)]
        e[0].record()
        te.pack(force=True)                       # what follows an optimizer update: arena -> bf16 + transposed operands
        e[1].record()
        if timed:                                 # the frozen half alone, only to report its share
            te.enc.forward_vit(image, ids, mask, table)
        e[2].record()
        _, plan = te.forward(image, ids, mask, table, train=True, seed=step.n)
        e[3].record()
        te.backward(plan, dout, reducer)
        e[4].record()
        step.n += 1
        if timed:
            torch.cuda.synchronize()
            t = [e[i].elapsed_time(e[i + 1]) for i in range(4)]
            acc["pack"] += t[0]
            acc["vit_fwd"] += t[1]
            acc["qformer_fwd"] += t[2] - t[1]     # te.forward runs the ViT again; the share of the Q-Former half
            acc["backward+allreduce"] += t[3]
    step.n = 0

    for _ in range(args.warmup):
        step(False)
    hdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    hdist.barrier()
    torch.cuda.synchronize()
    wall = hdist.max_over_ranks(time.perf_counter() - t0, dev) / args.steps * 1e3
    if not args.autograd:
        for _ in range(5):
            step(True)
    if rank == 0:
        n_par = te.arena.total
        res = {"workload": f"Cfg5 encode_img side of the stage-3 step: B={B}/GPU T={T} Lt={Lt}, frozen ViT + Q-Former/proj "
                           f"train forward+backward, {'autograd node' if args.autograd else 'engine API'}",
               "n_gpus": world, "ms_per_step": round(wall, 3), "clips_per_s": round(world * B / wall * 1e3, 2),
               "trainable_params": n_par, "grad_bytes_fp32": 4 * n_par,
               "allreduce": {"buckets": len(reducer.bounds), "bucket_mb": args.bucket_mb, "active": bool(reducer.active)},
               "fwd_launches": len(te.plans[(B, T, Lt)].fwd), "bwd_launches": len(te.plans[(B, T, Lt)].bwd)}
        if not args.autograd:
            # step(True) runs the ViT twice (once alone to time it): subtract it from the reported split
            res["split_ms"] = {k: round(v / 5, 3) for k, v in acc.items()}
        print(json.dumps(res), flush=True)
    hdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
