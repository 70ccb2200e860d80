"""Row f2 pin: autograd through the oracle (eval mode, and train mode with the injected masks) reproduces the
outputs and parameter gradients the reference's own modules produced (tests/golden/train_small.npz, written by
oracle/make_goldens.py train).  fp32 on both sides; tolerance 2e-4 of each tensor's max."""
import os

import numpy as np
import pytest
import torch

from hawkeye_amd import posembed, weights as W
from hawkeye_amd.config import SMALL_TRAIN
from oracle import dropout_masks as DM
from oracle.train_pass import dropper as _dropper, train_pass as oracle_train_pass


def load_case(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_small.npz"))
    cfg = SMALL_TRAIN
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    sd = W.make_state_dict(cfg, seed)
    image = W.make_clip(B, T, cfg.img_size, seed)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["text_mask"])
    table = posembed.video_pos_embed(T, cfg.patches_per_frame, cfg.encoder_embed_dim, cfg.ckpt_num_frame)
    return g, cfg, sd, image, ids, mask, table, torch.from_numpy(g["dout"])


def dropper(cfg, seed):
    return _dropper(cfg, seed)


def check_against_golden(g, mode, out, grads, tol, floor=1e-3):
    ref = torch.from_numpy(g[f"{mode}.out"])
    assert (out - ref).abs().max() <= tol * ref.abs().max()
    used = torch.from_numpy(g["used_ids"])
    n_checked = 0
    for key in g.files:
        if not key.startswith(mode + "."):
            continue
        kind, _, name = key[len(mode) + 1:].partition(".")
        if kind not in ("grad", "slice", "rows", "l2"):
            continue
        gr = grads[name].detach().float().cpu()
        want = g[key]
        if kind == "l2":
            assert abs(gr.double().norm().item() - float(want)) <= 5 * tol * max(float(want), 1e-2), name
            continue
        want = torch.from_numpy(want)
        if kind == "grad":
            got = gr
        elif kind == "rows":
            got = gr[used]
        else:
            flat = gr.reshape(gr.shape[0], -1)
            rows = np.linspace(0, flat.shape[0] - 1, min(8, flat.shape[0])).astype(np.int64)
            cols = np.linspace(0, flat.shape[1] - 1, min(32, flat.shape[1])).astype(np.int64)
            got = flat[torch.from_numpy(rows)][:, torch.from_numpy(cols)]
        scale = max(gr.abs().max().item(), floor)    # key biases have a mathematically zero gradient (softmax shift)
        assert (got - want).abs().max().item() <= tol * scale, (name, kind)
        n_checked += 1
    assert n_checked > 40


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_oracle_gradients_equal_reference_gradients(golden_dir, mode):
    g, cfg, sd, image, ids, mask, table, dout = load_case(golden_dir)
    drop = dropper(cfg, int(g["drop_seed"])) if mode == "train" else None
    out, grads = oracle_train_pass(cfg, sd, image, ids, mask, table, dout, drop)
    check_against_golden(g, mode, out, grads, 2e-4)


def test_train_mode_differs_from_eval_mode(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_small.npz"))
    a, b = g["eval.out"], g["train.out"]
    assert np.abs(a - b).max() > 0.05 * np.abs(a).max()
