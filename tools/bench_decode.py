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
    print(json.dumps({"workload": f"Vicuna-7B greedy decode B={B} prompt={S} new_tokens={T}", "bytes_per_token": nbytes, **res}))


if __name__ == "__main__":
    main()
