"""Checkpoint ingestion (row f4) on CPU: stacked strict=False loads like models/hawkeye_it.py:154-168 and
tasks/test.py:659-664, through the safe loader (weights_only=True)."""
import collections

import pytest
import torch

from hawkeye_amd import checkpoint as CK, weights as W
from hawkeye_amd.config import SMALL
from hawkeye_amd.model import HawkEyeEncoder


def _files(tmp_path):
    base = W.make_state_dict(SMALL, 1, depth_in_ckpt=True)      # checkpoint carries all encoder_depth ViT blocks
    tuned = W.make_state_dict(SMALL, 2)
    # file A: "umt_l16_qformer.pth"-like, a bare state dict with the ViT and the Q-Former
    a = collections.OrderedDict((k, v) for k, v in base.items() if k.startswith(("vision_encoder.", "qformer.")))
    # file B: "hawkeye.pth"-like: {"model": trainable parameters only (no ViT), LLM/LoRA keys, ...}
    b = collections.OrderedDict((k, v) for k, v in tuned.items() if not k.startswith("vision_encoder."))
    b["llama_model.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight"] = torch.zeros(16, 32)
    b["llama_model.base_model.model.lm_head.weight"] = torch.zeros(8, 32)
    pa, pb = str(tmp_path / "a.pth"), str(tmp_path / "b.pth")
    torch.save(a, pa)
    torch.save({"model": b, "optimizer": {"state": {}}, "epoch": 3, "global_step": 1234, "config": {"lr": 1e-4}}, pb)
    return base, tuned, pa, pb


def test_stacked_checkpoints_like_the_reference(tmp_path):
    base, tuned, pa, pb = _files(tmp_path)
    m = HawkEyeEncoder(SMALL)
    rep = CK.load_checkpoints(m, [pa, pb])
    sd = m.state_dict()
    # ViT from file A, everything else overwritten by file B
    k_vit = "vision_encoder.encoder.blocks.1.mlp.fc1.weight"
    assert torch.equal(sd[k_vit], base[k_vit])
    for k in ("qformer.bert.encoder.layer.0.attention.self.query.weight", "llama_proj.bias", "query_tokens",
              "extra_query_tokens", "vision_layernorm.weight"):
        assert torch.equal(sd[k], tuned[k]), k
    assert rep["missing"] == []
    # the checkpoint's last ViT block is not built (return_index = -2) and is dropped, as strict=False does
    last = f"vision_encoder.encoder.blocks.{SMALL.encoder_depth - 1}."
    assert rep["ignored"] and all(k.startswith(last) for k in rep["ignored"])
    assert not any(k.startswith("llama_model") for k in rep["loaded"] + rep["ignored"])


def test_missing_and_mismatch_are_reported(tmp_path):
    base, tuned, pa, pb = _files(tmp_path)
    m = HawkEyeEncoder(SMALL)
    rep = CK.load_checkpoints(m, pb)                      # hawkeye.pth alone: the frozen ViT is not in it
    assert rep["missing"] and all(k.startswith("vision_encoder.") for k in rep["missing"])
    with pytest.raises(RuntimeError, match="unset"):
        CK.load_checkpoints(HawkEyeEncoder(SMALL), pb, strict=True)
    bad = dict(torch.load(pa, weights_only=True))
    k = "qformer.bert.embeddings.word_embeddings.weight"
    bad[k] = bad[k][:-1]                                  # 30522-row table: resize_token_embeddings not applied
    pc = str(tmp_path / "c.pth")
    torch.save(bad, pc)
    with pytest.raises(RuntimeError, match="shape mismatch for " + k):
        CK.load_checkpoints(HawkEyeEncoder(SMALL), pc)


def test_packed_needs_a_gpu():
    m = HawkEyeEncoder(SMALL)
    with pytest.raises(RuntimeError, match="no CPU path"):
        CK.save_packed(m, "/tmp/never_written.safetensors")
    with pytest.raises(RuntimeError, match="no CPU path"):
        CK.load_packed(m, "/tmp/never_written.safetensors")
