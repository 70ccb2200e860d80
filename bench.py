#!/usr/bin/env python3
"""Headline benchmark: encode_img clips/s (UMT-L/16 23 blocks + Q-Former + llama_proj, 16 frames @ 224).

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver through torch.distributed.run, one rank per GPU.  A step is one
`encode_img` over a batch of 32 synthetic clips already resident in HBM (BASELINE.json configs[1]);
per-GPU work is fixed as N grows (weak scaling), clips are independent so there is no data-path
collective; for N > 1 the step ends with the all_gather of the [32,96,4096] outputs.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from hawkeye_amd import dist as hdist  # noqa: E402
from hawkeye_amd import ops, posembed, weights as W  # noqa: E402
from hawkeye_amd.config import FULL  # noqa: E402
from hawkeye_amd.engine import flops_per_clip  # noqa: E402
from hawkeye_amd.model import HawkEyeEncoder  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd, frames, lt, seed):
    """The oracle (CPU fp32 restatement of the reference path) timed on this box's host cores:
    one clip at the headline frame count and one at Cfg1 (B=1, T=8), each 1 warm-up + 3 timed runs, median.
    Threads: the box's CPU share for one GPU (OMP_NUM_THREADS there; os.cpu_count() where it is unset)."""
    import statistics
    from oracle import encode_img_oracle as O
    cores = int(os.environ.get("OMP_NUM_THREADS", 0)) or (os.cpu_count() or 1)
    torch.set_num_threads(cores)
    model = _cpu_model()

    def run(T, Lt_):
        image = W.make_clip(1, T, FULL.img_size, seed)
        ids, mask = W.make_instruction_ids(1, Lt_, FULL.qf_vocab, seed)
        table = posembed.video_pos_embed(T, FULL.patches_per_frame, FULL.encoder_embed_dim, FULL.ckpt_num_frame)
        times, out = [], None
        for k in range(4):
            t0 = time.perf_counter()
            with torch.no_grad():
                out, _ = O.encode_img(sd, FULL.to_dict(), image, ids, mask, table)
            if k > 0:
                times.append(time.perf_counter() - t0)
        return statistics.median(times), times, out

    dt, times, out = run(frames, lt)
    dt8, times8, _ = run(8, 12)
    return {"value": round(1.0 / dt, 4), "unit": "clips/s", "cores": cores, "kind": "port", "cpu_model": model,
            "logical_cpus_visible": os.cpu_count(),
            "sample": f"1 clip, T={frames}, Lt={lt}, fp32 oracle (torch CPU, {cores} threads on {model}); 1 warm-up + 3 timed, "
                      f"median {dt:.2f} s (runs {', '.join(f'{t:.2f}' for t in times)})",
            "cfg1_B1_T8": {"value": round(1.0 / dt8, 4), "unit": "clips/s", "median_s": round(dt8, 3),
                           "runs_s": [round(t, 3) for t in times8], "sample": "1 clip, T=8, Lt=12 (BASELINE configs[0])"}}, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU per step")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--lt", type=int, default=32, help="instruction tokens")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-insitu", action="store_true", help="skip the per-kernel in-situ timing and the MFMA probe (PMC passes)")
    ap.add_argument("--overlap-gather", action="store_true",
                    help="all_gather of call k on RCCL's stream beside the kernels of call k + 1 (default: in line on the compute "
                         "stream; unmeasured at N > 1 -- RCCL's workgroups then share CUs with the persistent GEMMs)")
    ap.add_argument("--force-gather", action="store_true", help="run the output all_gather even with one rank (plumbing test)")
    args = ap.parse_args()

    # RCCL prints a version banner on stdout when its first communicator comes up; the contract is ONE
    # JSON line on stdout, so stdout points at stderr until the process group has done a collective.
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        rank, local_rank, world = hdist.init("nccl")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            hdist.max_over_ranks(0.0, torch.device("cuda", local_rank))
            hdist.barrier()
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the encode_img path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg, seed = FULL, 1
    B, T, Lt = args.batch, args.frames, args.lt
    sd = W.make_state_dict(cfg, seed)
    model = HawkEyeEncoder(cfg, out_dtype=torch.bfloat16)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    model.vision_encoder.encoder.pos_embed = posembed.video_pos_embed(
        T, cfg.patches_per_frame, cfg.encoder_embed_dim, cfg.ckpt_num_frame)
    # synthetic inputs resident in HBM before the timed region; different clips per rank
    image = W.make_clip(B, T, cfg.img_size, seed + 100 * rank).to(dev)
    ids, mask = W.make_instruction_ids(B, Lt, cfg.qf_vocab, seed)
    ids, mask = ids.to(dev), mask.to(dev)
    gather = (world > 1 and not args.no_gather) or args.force_gather

    # --overlap-gather: the all_gather of call k runs on RCCL's stream beside the kernels of call k + 1
    # (hawkeye_amd.dist.OutputGatherer); the last one is collected inside the timed region
    gatherer = hdist.OutputGatherer()

    @torch.no_grad()
    def step():
        out, _ = model.encode_img(image, (ids, mask))
        if gather and args.overlap_gather:
            return gatherer.submit(out)
        if gather:
            out = hdist.gather_outputs(out)
        return out

    for _ in range(args.warmup):
        step()
    gatherer.collect()
    hdist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        out = step()
    out = gatherer.collect() if gatherer.pending is not None else out
    ev1.record()
    hdist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    wall = hdist.max_over_ranks(wall, dev)
    seen = hdist.ranks_seen(dev)        # every rank reports itself over the job's own (RCCL) group

    if rank == 0:
        clips = world * B * args.steps
        value = clips / wall
        fl = flops_per_clip(cfg, T, Lt)
        # roofline of the step (the whole launch sequence of one encode_img): algorithmic FLOPs of
        # the B clips one step processes / the step's device time from HIP events on the launch stream
        step_s = dev_ms / 1e3 / args.steps
        achieved = fl * B / step_s / 1e12
        # HBM-side bytes per step from the committed rocprofv3 PMC passes (profiles/rNN_hbm_traffic.txt -> traffic.json);
        # PMC collection needs the profiler, so the live run quotes the profiled figure for this workload, and only
        # while the kernel sources are the ones that were profiled (traffic.json carries their digest)
        tj = {}
        try:
            from hawkeye_amd.lib import source_digest
            tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
            # another workload, or counters taken with another kernel tree: no figure rather than a stale one
            if tj.get("workload") != f"B={B} T={T} Lt={Lt}" or tj.get("csrc_digest") != source_digest():
                tj = {}
        except (OSError, ValueError):
            tj = {}
        step_roof = {"kernel": "encode_img step (all launches of one call)", "achieved": round(achieved, 2),
                     "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                     "traffic": tj.get("bytes_per_step"), "traffic_source": tj.get("source"), "gflop_per_clip": round(fl / 1e9, 1),
                     "device_ms_per_step": round(step_s * 1e3, 3)}
        res = {
            "metric": "encode_img clips/s (UMT-L/16+QFormer,16f@224)", "value": round(value, 3), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"encode_img B={B}/GPU T={T} 224px Lt={Lt} (BASELINE configs[1])",
                       "clips_per_gpu": B, "frames": T, "tokens_per_clip": T * cfg.patches_per_frame,
                       "vit_blocks": cfg.run_depth, "parallelism": f"clip-sharded x{world}",
                       "output_all_gather": bool(gather),
                       "gather_overlapped": bool(gather and args.overlap_gather)},
            "roofline": dict(step_roof, bound="mfma", step=step_roof),
            # one entry per rank, gathered over RCCL: N entries with N distinct (local_rank, device_index) = the
            # collective spanned N ranks on N GPUs
            "ranks_seen": seen,
        }
        if not args.no_insitu:
            # Dominant kernel, in situ: one HIP event pair around every launch of 3 more calls (same stream and order as
            # the timed steps); algorithmic flops per launch from SURVEY.md 8d's formula terms; roofline.achieved is that
            # kernel's flops per launch / its average launch duration, roofline.step the whole launch sequence.
            N_, D_, R_ = T * cfg.patches_per_frame, cfg.encoder_embed_dim, B * T * cfg.patches_per_frame
            kflops = {"vit.qkv": 2.0 * R_ * D_ * 3 * D_, "vit.attn": 4.0 * B * N_ * N_ * D_, "vit.proj": 2.0 * R_ * D_ * D_,
                      "vit.fc1_gelu": 2.0 * R_ * D_ * 4 * D_, "vit.fc2": 2.0 * R_ * D_ * 4 * D_,
                      "vit.patch_embed": 2.0 * R_ * cfg.patch_k * D_,
                      "qformer.cross_kv": 2.0 * R_ * D_ * 2 * cfg.qf_hidden * 6}
            kname = {"vit.attn": "hk_attn_kernel<false, 1, 3, 1, false>", "vit.qkv": "hk_gemm_pp_kernel<3, 4>",
                     "vit.fc1_gelu": "hk_gemm_pp_kernel<3, 4>", "vit.fc2": "hk_gemm_pp_kernel<4, 4>", "vit.proj": "hk_gemm_pp_kernel<4, 4>"}
            eng = model.sync_engine()
            table = model._table(T == 1, N_, dev)
            prof = eng.profile(image, ids, mask, table, reps=3)
            kern, tot_us = {}, sum(n * us for n, us in prof.values())
            for tag, (n, us) in sorted(prof.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
                e = {"launches": n, "us": round(us, 1), "share": round(n * us / tot_us, 4)}
                if tag in kflops:
                    e["tflops"] = round(kflops[tag] / us / 1e6, 1)
                    e["frac"] = round(kflops[tag] / us / 1e6 / PEAK_BF16_TFLOPS, 4)
                elif tag in ("vit.ln", "vision_layernorm", "vit.ln_stats"):
                    nbytes = R_ * D_ * ((2 if tag != "vit.ln_stats" else 0) + (2 if eng.residual_dtype == torch.bfloat16 else 4))
                    if tag == "vit.ln_stats" and eng.epi_stats:
                        nbytes = R_ * (D_ // 64) * 8      # the epilogue partial sums the finalize pass reads
                    e["tb_s"] = round(nbytes / us / 1e6, 2)
                    e["frac_hbm"] = round(nbytes / us / 1e6 / 8.0, 4)
                kern[tag] = e
            dom = next(t for t in kern if t in kflops)
            ktraffic = None
            for k, v in (tj.get("per_kernel") or {}).items():      # keys: "<kernel> [<workgroups> wg]"
                if k.startswith(kname.get(dom, "?")) and v.get("launches_per_step") == kern[dom]["launches"]:
                    ktraffic = v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]
            res["roofline"].update({
                "kernel": f"{kname.get(dom, dom)} ({dom}, {kern[dom]['launches']} launches per step, {kern[dom]['share'] * 100:.1f} % of it)",
                "achieved": kern[dom]["tflops"], "frac": kern[dom]["frac"], "traffic": ktraffic,
                "flop_per_launch": kflops[dom], "us_per_launch": kern[dom]["us"]})
            for k in ("gflop_per_clip", "device_ms_per_step"):
                res["roofline"].pop(k, None)
            res["roofline"]["kernels"] = kern
            # register-resident MFMA probe on this chip: what the silicon sustains with random / zero data
            mp = {"16x16x32_random": round(ops.mfma_probe_tflops(0), 1), "16x16x32_zeros": round(ops.mfma_probe_tflops(0, zeros=True), 1),
                  "32x32x16_random": round(ops.mfma_probe_tflops(1), 1)}
            res["roofline"]["measured_mfma_peak"] = mp
            res["roofline"]["step"] = dict(step_roof, frac_of_measured_peak=round(achieved / max(mp["16x16x32_random"], 1e-9), 4))
        if not args.no_cpu_baseline and world == 1:      # the CPU leg runs at N = 1 only (the other ranks would idle at the barrier)
            base, ref = cpu_baseline(sd, T, Lt, seed)
            res["cpu_baseline"] = base
            # the same clip through the HIP path, checked against the oracle run that was just timed
            img1 = W.make_clip(1, T, cfg.img_size, seed).to(dev)
            i1, m1 = W.make_instruction_ids(1, Lt, cfg.qf_vocab, seed)
            model.out_dtype = torch.float32
            with torch.no_grad():
                got, _ = model.encode_img(img1, (i1.to(dev), m1.to(dev)))
            a, b = got.double().flatten().cpu(), ref.double().flatten()
            res["parity"] = {"cosine": round(float(a @ b / (a.norm() * b.norm())), 6),
                             "max_rel_err": round(float((a - b).abs().max() / b.abs().max()), 5)}
        print(json.dumps(res), flush=True)
    hdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
