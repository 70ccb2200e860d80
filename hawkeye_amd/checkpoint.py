"""Checkpoint ingestion for the encode_img path.

The reference restores its weights by stacking up to three files with `load_state_dict(strict=False)`:
`vit_blip_model_path` (UMT-L ViT + Q-Former), `videochat2_model_path` (stage 2, possibly wrapped in
{"model": ...}) (models/hawkeye_it.py:154-168) and finally the instruction-tuned `hawkeye.pth`
(tasks/test.py:659-664), whose "model" entry only holds the parameters that were trainable when it was
saved (tasks/train_it.py:100-107).  `load_checkpoints` does the same for the hot-path half of the model and
reports what is still unset, which the reference only prints.

Files are opened with `torch.load(..., weights_only=True, mmap=True)`: nothing in the file is executed and the
7B-parameter LLM tensors that share the file are never paged in.

`save_packed` / `load_packed` store the engine's device operand set (bf16 GEMM weights with the fused QKV and the
fused 6-layer cross-K/V matrix, fp32 biases / LayerNorm terms) as one safetensors file, so a serving process
starts without the fp32 checkpoint, the casts and the concatenations.
"""
import json
import os
from typing import Dict, Iterable, List, Sequence, Union

import torch

from .engine import HipEncoderEngine

HOT_PATH_PREFIXES = ("vision_encoder.", "vision_layernorm.", "qformer.", "query_tokens", "extra_query_tokens",
                     "llama_proj.")
PACKED_FORMAT = "hawkeye_amd.packed.v2"   # v2: norm1/norm2 may be folded into qkv/fc1 (meta "fold_ln")


def hot_path_items(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The entries of a HawkEye_it state dict that encode_img reads (everything else is the LLM / LoRA)."""
    return {k: v for k, v in state.items() if k.startswith(HOT_PATH_PREFIXES) and torch.is_tensor(v)}


def read_state(path: str) -> Dict[str, torch.Tensor]:
    """One checkpoint file -> flat state dict; unwraps {"model": ...} (hawkeye_it.py:163-166, test.py:660-663)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True, mmap=True)
    if isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict):
        ckpt = ckpt["model"]
    if not isinstance(ckpt, dict):
        raise RuntimeError(f"{path}: expected a state dict, got {type(ckpt).__name__}")
    return ckpt


def load_checkpoints(model, paths: Union[str, Sequence[str]], strict: bool = False) -> Dict[str, List[str]]:
    """Apply checkpoint files in order (later files overwrite earlier ones), like the reference's stacked
    `load_state_dict(strict=False)` calls.  Returns {"loaded": [...], "missing": [...], "ignored": [...]}:
    `missing` = hot-path parameters no file provided (left at their initial values),
    `ignored` = hot-path-prefixed keys the model does not have (e.g. the 24th ViT block that
    return_index = -2 never builds, models/blip2/vit.py:350).  A shape mismatch always raises, naming the key;
    strict=True also raises on missing keys."""
    if isinstance(paths, (str, os.PathLike)):
        paths = [paths]
    own = model.state_dict()
    loaded, ignored = set(), []
    for path in paths:
        state = hot_path_items(read_state(str(path)))
        take = {}
        for k, v in state.items():
            if k not in own:
                ignored.append(k)
                continue
            if tuple(v.shape) != tuple(own[k].shape):
                raise RuntimeError(f"{path}: shape mismatch for {k}: checkpoint {tuple(v.shape)} vs model "
                                   f"{tuple(own[k].shape)} (was qformer.resize_token_embeddings applied? "
                                   "models/hawkeye_it.py:72)")
            take[k] = v
        model.load_state_dict(take, strict=False)
        loaded.update(take)
    missing = [k for k in own if k not in loaded and not k.endswith("position_ids")]
    if strict and missing:
        raise RuntimeError(f"checkpoints {list(map(str, paths))} leave {len(missing)} hot-path tensors unset, "
                           f"first: {missing[:5]}")
    return {"loaded": sorted(loaded), "missing": missing, "ignored": sorted(set(ignored))}


def save_packed(model, path: str) -> int:
    """Write the engine's packed device operands (model must be on the GPU) to a safetensors file; returns bytes."""
    from safetensors.torch import save_file
    eng: HipEncoderEngine = model.sync_engine()
    tensors = {k: v.detach().to("cpu").contiguous() for k, v in eng.w.items()}
    meta = {"format": PACKED_FORMAT, "config": json.dumps(model.cfg.to_dict(), sort_keys=True),
            "fold_ln": "1" if eng.fold_ln else "0"}
    save_file(tensors, path, metadata=meta)
    return os.path.getsize(path)


def load_packed(model, path: str) -> HipEncoderEngine:
    """Install a packed operand file as the model's engine weights (no fp32 state dict needed).  The model's
    nn.Parameters keep whatever values they have; the engine is marked in sync with them, so change a parameter
    (or call model.sync_engine(force=True)) to repack from the module tree instead."""
    from safetensors import safe_open
    dev = model.device
    if dev.type != "cuda":
        raise RuntimeError("load_packed needs the model on an MI355X (model.to('cuda')); there is no CPU path")
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        if meta.get("format") != PACKED_FORMAT:
            raise RuntimeError(f"{path}: not a {PACKED_FORMAT} file")
        if json.loads(meta.get("config", "{}")) != json.loads(json.dumps(model.cfg.to_dict(), sort_keys=True)):
            raise RuntimeError(f"{path}: packed for a different EncoderConfig")
        w = {k: f.get_tensor(k).to(dev) for k in f.keys()}
    eng = HipEncoderEngine(model.cfg, dev, model.residual_dtype)
    eng.fold_ln = meta.get("fold_ln") == "1"   # the file's qkv / fc1 operands are folded (or not); the plan follows
    if eng.fold_ln and model.residual_dtype != torch.bfloat16:
        raise RuntimeError(f"{path}: packed with norm1/norm2 folded into qkv/fc1, which needs the bf16 residual stream")
    eng.w = w
    model._engine = eng
    model._engine_key = {part: (str(dev), model._weights_key(part)) for part in ("vit", "qformer")}
    model._train_engine = None
    return eng
