"""ORACLE -- test infrastructure, not product code.

Reference gradients of the stage-3 step for the encode_img path: autograd through the fp32 oracle
(oracle/encode_img_oracle.py) with the ViT frozen (hawkeye_it.py:82-91) and loss = sum(out * dout), which is what
`scaler.scale(loss).backward()` (tasks/train_it.py:70) sends into encode_img's output for an upstream gradient dout.
Pinned against the reference's own modules by tests/test_train_oracle_vs_golden.py.
"""
import torch

from . import dropout_masks as DM
from . import encode_img_oracle as O


def trainable_keys(sd):
    keys = [k for k in sd if k.startswith(("qformer.", "llama_proj.")) or k in ("query_tokens", "extra_query_tokens")]
    return [k for k in keys if sd[k].is_floating_point()]


def train_pass(cfg, sd, image, ids, mask, table, dout, drop):
    """Returns (out, {name: grad}); drop = None (eval mode) or a DM.MaskDropper (train mode)."""
    sd = {k: v.clone() for k, v in sd.items()}
    keys = trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    out, _ = O.encode_img(sd, cfg.to_dict(), image, ids, mask, table, drop=drop)
    (out * dout).sum().backward()
    return out.detach(), {k: (sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in keys}


def dropper(cfg, seed, p_hidden=0.1, p_attn=0.1, path_rate=0.2):
    return DM.MaskDropper(seed, p_hidden, p_attn, DM.path_rates(cfg.qf_layers, path_rate))
