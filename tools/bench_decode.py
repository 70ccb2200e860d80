"""Decode-step timing at Vicuna-7B shape (row f1 continued): ms per token and achieved HBM GB/s against the algorithmic
bytes of one step (hawkeye_amd.decode.decode_bytes_per_token).  python tools/bench_decode.py [B] [S] [new_tokens]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd.config import VICUNA_7B  # noqa: E402
from hawkeye_amd.decode import LlamaGenerator, decode_bytes_per_token  # noqa: E402
from hawkeye_amd.llm import LlamaPrefill  # noqa: E402


def torch_eager_step_ms(pre, cfg, B, S, T, dev):
    """The same decode step written with PyTorch ops on the same weights (F.linear = hipBLASLt / rocBLAS, sdpa, elementwise
    ops): what an eager bf16 port of modeling_llama.py's one-token forward costs on this GPU.  Baseline only."""
    import torch.nn.functional as F
    w, D, H, hd, I = pre.w, cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.intermediate_size
    Lyr = cfg.num_hidden_layers
    gate = [w[f"l{i}.gu"][0::2].contiguous() for i in range(Lyr)]
    up = [w[f"l{i}.gu"][1::2].contiguous() for i in range(Lyr)]
    kc = torch.randn((Lyr, B, H, S + T, hd), device=dev, dtype=torch.bfloat16)
    vc = torch.randn_like(kc)
    cos = torch.cat([w["cos"], w["cos"]], -1).to(torch.bfloat16)
    sin = torch.cat([w["sin"], w["sin"]], -1).to(torch.bfloat16)

    def rms(x, g):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + cfg.rms_norm_eps) * g).to(torch.bfloat16)

    def rot(x):
        return torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], -1)

    @torch.no_grad()
    def step(tok, t):
        x = w["embed"][tok].float()
        n = S + t
        for i in range(Lyr):
            o = f"l{i}."
            qkv = F.linear(rms(x, w[o + "n1"]), w[o + "qkv"]).view(B, 3, H, hd)
            q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
            q, k = q * cos[n] + rot(q) * sin[n], k * cos[n] + rot(k) * sin[n]
            kc[i, :, :, n], vc[i, :, :, n] = k, v
            a = F.scaled_dot_product_attention(q[:, :, None], kc[i, :, :, :n + 1], vc[i, :, :, :n + 1]).reshape(B, D)
            x = x + F.linear(a, w[o + "o"]).float()
            h = rms(x, w[o + "n2"])
            x = x + F.linear(F.silu(F.linear(h, gate[i])) * F.linear(h, up[i]), w[o + "down"]).float()
        return F.linear(rms(x, w["norm"]), w["lm_head"]).float().argmax(-1)

    tok = torch.zeros(B, dtype=torch.int64, device=dev)
    for t in range(3):
        tok = step(tok, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(3, 3 + 20):
        tok = step(tok, t)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / 20


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 352
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    cfg, dev = VICUNA_7B, torch.device("cuda:0")
    pre = LlamaPrefill(cfg, dev).init_random(0)
    gen = LlamaGenerator(pre)
    g = torch.Generator(device="cpu").manual_seed(1)
    src = torch.randint(3, cfg.vocab_size, (B, S), generator=g).to(dev)
    gen.generate(src=src, max_new_tokens=4)                     # plans, pools
    torch.cuda.synchronize()
    res = {}
    for name, kw in (("free_running", {}), ("per_token_host_check", {"stop_sequences": [[cfg.vocab_size + 5]]})):
        times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pre.prefill(src=src)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out = gen.generate(src=src, max_new_tokens=T, **kw)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            times.append(((t2 - t1) - (t1 - t0), t1 - t0))      # generate() includes its own prefill
        dec, pf = min(times)
        ms_tok = dec * 1e3 / (T - 1)
        nbytes = decode_bytes_per_token(cfg, B, S + T // 2)
        res[name] = {"ms_per_token": round(ms_tok, 3), "tokens_per_s": round(B * 1e3 / ms_tok, 1),
                     "achieved_GBps": round(nbytes / ms_tok / 1e6, 1), "frac_of_8TBps": round(nbytes / ms_tok / 1e6 / 8000, 4),
                     "prefill_ms": round(pf * 1e3, 3)}
        assert out["tokens"].shape == (B, T)
    if os.environ.get("DECODE_TORCH_BASELINE", "0") == "1":
        ms = torch_eager_step_ms(pre, cfg, B, S, T, dev)
        res["torch_eager_same_gpu"] = {"ms_per_token": round(ms, 3), "tokens_per_s": round(B * 1e3 / ms, 1)}
    print(json.dumps({"workload": f"Vicuna-7B greedy decode B={B} prompt={S} new_tokens={T}", "bytes_per_token": nbytes, **res}))


if __name__ == "__main__":
    main()
