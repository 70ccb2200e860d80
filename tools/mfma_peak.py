"""Register-resident MFMA peak of the box's MI355X for both MFMA shapes, random and zero operands,
at 1, 2 and 8 waves per SIMD."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import ops  # noqa: E402

res = {}
for grid in (256, 512, 2048):
    for kind, name in ((0, "16x16x32"), (1, "32x32x16")):
        for zeros in (False, True):
            res[f"{name}_{'zeros' if zeros else 'random'}_wg{grid}"] = round(
                ops.mfma_probe_tflops(kind, zeros=zeros, grid=grid, iters=4000 * 2048 // grid), 1)
print(json.dumps(res, indent=1))
