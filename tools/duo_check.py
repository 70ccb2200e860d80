#!/usr/bin/env python3
"""The two-workgroups-per-CU GEMM (hk_gemm_duo_kernel) against the 256x256 kernel: same bits expected (same MFMA order
per output element), then interleaved timings at the ViT shapes.  usage: tools/duo_check.py [check] [time] [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import lib as L, ops  # noqa: E402

DEV = "cuda:0"
lib = L.load()


def run(kind, x, w, b, res, st, cc, rp, out):
    if kind == 3:
        return ops.linear(x, w, b, act="gelu" if out.shape[1] == 4096 else "none", out=out, row_stats=st, col_c=cc)
    if kind == 0:
        return ops.linear(x, w, b, act="gelu" if out.shape[1] == 4096 else "none", out=out)
    if kind == 1:
        return ops.linear(x, w, b, residual=res, out_dtype=torch.float32, out=out)
    if kind == 2:
        return ops.linear(x, w, b, residual=out, out_dtype=torch.bfloat16, out=out)
    if kind == 4:
        return ops.linear(x, w, b, residual=out, out_dtype=torch.bfloat16, out=out, row_part=rp)


def make(M, N, K, kind):
    g = torch.Generator(device=DEV).manual_seed(M + N + K + kind)
    x = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g) if kind == 1 else None
    st = cc = rp = None
    if kind == 3:
        st = torch.zeros((M + 255) // 256 * 256, 2, device=DEV)
        ops.row_stats(x, 1e-6, out=st[:M])
        cc = w.float().sum(1).contiguous()
        st = st[:M]
    if kind == 4:
        rp = torch.zeros(M, N // 64, 2, device=DEV)
    odt = torch.float32 if kind == 1 else torch.bfloat16
    out0 = torch.randn(M, N, device=DEV, generator=g).to(odt)
    return x, w, b, res, st, cc, rp, out0


def check():
    ok = True
    for kind in (0, 3, 2, 4, 1):
        for (M, N, K) in ((25088, 1024 if kind in (2, 4) else 3072, 1024), (25088 + 77, 1024, 4096 if kind != 3 else 1024),
                          (16384, 4096 if kind in (0, 3) else 1024, 1024), (32768 + 256, 1024, 768)):
            args = make(M, N, K, kind)
            outs = []
            for mask in (0, 31):
                lib.hk_gemm_duo_mask(mask)
                out = args[-1].clone()
                rp = args[6].clone() if args[6] is not None else None
                run(kind, *args[:6], rp, out)
                torch.cuda.synchronize()
                outs.append((out, rp))
            same = torch.equal(outs[0][0], outs[1][0])
            # rows past the last whole 256-row tile go to the multi-stage kernel on the duo path: its row sums are added in
            # another order (fp32 rounding), everything else must be the same bits
            rps = True if outs[0][1] is None else (torch.equal(outs[0][1], outs[1][1]) if M % 256 == 0 else
                                                   torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-4))
            d = (outs[0][0].float() - outs[1][0].float()).abs().max().item()
            print(f"epi {kind} M={M} N={N} K={K}: identical={same} row_part identical={rps} max|diff|={d:.3e}", flush=True)
            ok = ok and same and rps
    lib.hk_gemm_duo_mask(0)
    print("CHECK", "OK" if ok else "FAILED", flush=True)
    return ok


def digest():
    """Hashes of the default kernel's outputs at shapes with whole, partial and mapped tiles: two library builds
    (HK_LIB_PATH) that should produce the same bits print the same lines."""
    import hashlib
    lib.hk_gemm_duo_mask(0)
    for kind in (0, 3, 2, 4, 1):
        for (M, N, K) in ((25088, 1024 if kind in (2, 4) else 3072, 1024), (25088 + 77, 1024, 4096 if kind != 3 else 1024),
                          (16384, 4096 if kind in (0, 3) else 1024, 1024), (32768 + 256, 1024, 768), (512, 4096, 64), (1024, 2048, 128),
                          (2816, 4096, 1024), (2716, 4096, 512)):    # one partial round of 256-row tiles: the 192-row tile's shapes
            args = make(M, N, K, kind)
            out = args[-1].clone()
            rp = args[6].clone() if args[6] is not None else None
            run(kind, *args[:6], rp, out)
            torch.cuda.synchronize()
            h = hashlib.sha256(out.view(torch.uint8).cpu().numpy().tobytes())
            if rp is not None:
                h.update(rp.cpu().numpy().tobytes())
            print(f"DIGEST epi {kind} M={M} N={N} K={K}: {h.hexdigest()[:24]}", flush=True)


def timeit(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def time_pp(M=100352):
    """The default (256x256) kernel alone at the ViT shapes: for same-box A/B of library builds (HK_LIB_PATH)."""
    shapes = (("qkv (fold)", 3, 3072, 1024), ("fc1+gelu (fold)", 3, 4096, 1024), ("proj (row_part)", 4, 1024, 1024),
              ("fc2 (row_part)", 4, 1024, 4096), ("ckv", 0, 9216, 1024))
    lib.hk_gemm_duo_mask(0)
    tag = os.path.basename(os.environ.get("HK_LIB_PATH", "default"))
    for name, kind, N, K in shapes:
        args = make(M, N, K, kind)
        out = args[-1]
        t = min(timeit(lambda: run(kind, *args[:7], out)) for _ in range(3))
        print(f"[{tag}] {name:18s}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s", flush=True)
        del args, out


def time_llm(M=2816):
    """Prefill GEMM shapes at B = 8, S = 352 (Vicuna-7B): for A/B of dispatch rules through the environment."""
    shapes = (("qkv", 0, 12288, 4096), ("o_proj", 1, 4096, 4096), ("gate_up", 0, 22016, 4096), ("down_proj", 1, 4096, 11008))
    lib.hk_gemm_duo_mask(0)
    tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("HK_GEMM"))
    for name, kind, N, K in shapes:
        args = make(M, N, K, kind)
        out = args[-1]
        t = min(timeit(lambda: run(kind, *args[:7], out), iters=16) for _ in range(3))
        print(f"[{tag}] {name:10s} M={M} N={N} K={K}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s", flush=True)
        del args, out


def time_all(M=100352):
    shapes = (("qkv (fold)", 3, 3072, 1024), ("fc1+gelu (fold)", 3, 4096, 1024), ("proj (row_part)", 4, 1024, 1024),
              ("fc2 (row_part)", 4, 1024, 4096), ("ckv", 0, 9216, 1024), ("patch", 2, 1024, 768))
    for name, kind, N, K in shapes:
        args = make(M, N, K, kind)
        out = args[-1]
        res = {0: [], 31: []}
        for rnd in range(3):
            for mask in (0, 31):
                lib.hk_gemm_duo_mask(mask)
                res[mask].append(timeit(lambda: run(kind, *args[:7], out)))
        fl = 2.0 * M * N * K
        a, b = min(res[0]), min(res[31])
        print(f"{name:18s} M={M} N={N} K={K}: 256x256 {a:8.1f} us ({fl/a/1e6:7.1f} TF)   duo {b:8.1f} us ({fl/b/1e6:7.1f} TF)   "
              f"ratio {b/a:.3f}   all: {[round(v) for v in res[0]]} vs {[round(v) for v in res[31]]}", flush=True)
        del args, out
    lib.hk_gemm_duo_mask(0)


if __name__ == "__main__":
    torch.set_grad_enabled(False)      # (not at import: tests import this module)
    what = sys.argv[1:] or ["check", "time"]
    if "check" in what:
        if not check():
            sys.exit(1)
    if "time" in what:
        time_all()
    if "llm" in what:
        time_llm(int(os.environ.get("LLM_M", 2816)))
    if "digest" in what:
        digest()
    if "pp" in what:
        time_pp()
