#!/usr/bin/env python3
"""Host-side cost of one small encode_img call: time until the call returns (everything enqueued), and a cProfile of it.
In the eval scripts every call is followed by a host sync (the LLM consumes the result), so this time is not hidden behind
the previous call's device work.  usage: tools/host_profile.py [B] [T]"""
import cProfile, os, pstats, sys, time
import torch

torch.set_grad_enabled(False)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import weights as W
from hawkeye_amd.config import FULL
from hawkeye_amd.model import HawkEyeEncoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
DEV = torch.device("cuda", 0)
cfg = FULL
model = HawkEyeEncoder(cfg, out_dtype=torch.bfloat16)
model.load_state_dict(W.make_state_dict(cfg, 1), strict=True)
model = model.to(DEV).eval()
model.device_pos_embed = True
image = W.make_clip(B, T, cfg.img_size, 7).to(DEV)
ids, mask = W.make_instruction_ids(B, 32, cfg.qf_vocab, 7)
ids, mask = ids.to(DEV), mask.to(DEV)
for _ in range(5):
    out, _ = model.encode_img(image, (ids, mask))
    float(out[0, 0, 0])
host, total = [], []
for _ in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, _ = model.encode_img(image, (ids, mask))
    t1 = time.perf_counter()
    float(out[0, 0, 0])
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print(f"B={B} T={T}: host enqueue {host[len(host)//2]:.3f} ms (median), call + sync {total[len(total)//2]:.3f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    out, _ = model.encode_img(image, (ids, mask))
    float(out[0, 0, 0])
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
