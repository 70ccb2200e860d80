#!/usr/bin/env python3
"""Interleaved A/B of the attention variants (env is read once per process, so each variant runs in
a child process; rounds are interleaved on the same device)."""
import os, subprocess, sys
variants = [("0", "3", "0"), ("1", "2", "0"), ("0", "3", "1"), ("1", "2", "1"), ("1", "3", "1")]
for rnd in range(2):
    for m, o, l in variants:
        env = dict(os.environ, HK_ATTN_MODE=m, HK_ATTN_OCC=o, HK_ATTN_LOADER=l)
        r = subprocess.run([sys.executable, "-u", "tools/bench_kernels.py", "attn"], env=env, capture_output=True, text=True)
        out = r.stdout
        print(f"mode={m} occ={o} loader={l} | " + " | ".join(x.split(":")[1].strip() for x in out.strip().splitlines()) + (r.stderr[-300:] if r.returncode else ""), flush=True)
