#!/usr/bin/env python3
"""Interleaved A/B of two builds of libhawkeye_hip.so on the same device (devices differ by several per cent and the
clock depends on load history, so only same-box, interleaved rounds compare).
usage: tools/ab_lib.py <rounds> <lib_a.so> <lib_b.so> [...] -- <tool.py> [args]     (each run is a child process)"""
import os, subprocess, sys
i = sys.argv.index("--")
rounds, libs = int(sys.argv[1]), sys.argv[2:i]
cmd = [sys.executable, "-u"] + sys.argv[i + 1:]
for rnd in range(rounds):
    for tag, lib in zip("ABCDEFGH", libs):
        r = subprocess.run(cmd, env=dict(os.environ, HK_LIB_PATH=os.path.abspath(lib)), capture_output=True, text=True)
        for line in r.stdout.strip().splitlines():
            print(f"[{tag} r{rnd}] {line}", flush=True)
        if r.returncode:
            print(f"[{tag} r{rnd}] rc={r.returncode} {r.stderr[-400:]}", flush=True)
            sys.exit(1)
