"""Deterministic synthetic weights for the encode_img path.

There are no checkpoints offline, so parity is checked on seeded random
weights.  Keys and shapes are exactly the reference's state-dict surface
(models/hawkeye_it.py:57-80,183-185, models/blip2/vit.py:256-292,
models/blip2/Qformer.py:52-78,124-160,305-433).  The generator only depends on
numpy's PCG64 stream so the GPU box, which has no reference checkout, can
regenerate bit-identical tensors from (config, seed).

Init laws follow the reference (xavier-uniform ViT linears vit.py:386-393,
N(0, 0.02) BERT weights Qformer.py:697-707) except that biases, LayerNorm
affine terms and extra_query_tokens -- all zeros/ones in the reference -- are
drawn non-degenerate so that a kernel that drops a bias or a gamma fails parity.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

from .config import EncoderConfig, LlmConfig


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed]))


def _normal(name, seed, shape, std, mean=0.0):
    a = _rng(name, seed).standard_normal(shape, dtype=np.float32)
    a *= np.float32(std)
    if mean != 0.0:
        a += np.float32(mean)
    return torch.from_numpy(a)


def _uniform(name, seed, shape, bound):
    a = _rng(name, seed).random(shape, dtype=np.float32)
    a = (a * np.float32(2.0) - np.float32(1.0)) * np.float32(bound)
    return torch.from_numpy(a)


def make_state_dict(cfg: EncoderConfig, seed: int = 0, depth_in_ckpt: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """fp32 CPU state dict for the hot-path modules of HawkEye_it.

    depth_in_ckpt=True also emits the checkpoint's extra ViT blocks
    (encoder_depth of them) that strict=False loading drops (SURVEY.md top table).
    """
    sd = OrderedDict()
    D = cfg.encoder_embed_dim
    Hm = D * cfg.mlp_ratio
    p = "vision_encoder.encoder."

    def lin_xavier(key, out_f, in_f):
        bound = float(np.sqrt(6.0 / (in_f + out_f)))
        sd[key] = _uniform(key, seed, (out_f, in_f), bound)

    def ln(prefix, n):
        sd[prefix + ".weight"] = _normal(prefix + ".weight", seed, (n,), 0.1, 1.0)
        sd[prefix + ".bias"] = _normal(prefix + ".bias", seed, (n,), 0.05)

    kb = float(1.0 / np.sqrt(cfg.patch_k))
    sd[p + "patch_embed.proj.weight"] = _uniform(
        p + "patch_embed.proj.weight", seed, (D, 3, 1, cfg.patch_size, cfg.patch_size), kb)
    sd[p + "patch_embed.proj.bias"] = _uniform(p + "patch_embed.proj.bias", seed, (D,), kb)
    nblk = cfg.encoder_depth if depth_in_ckpt else cfg.run_depth
    for i in range(nblk):
        b = f"{p}blocks.{i}."
        ln(b + "norm1", D)
        sd[b + "attn.q_bias"] = _normal(b + "attn.q_bias", seed, (D,), 0.02)
        sd[b + "attn.v_bias"] = _normal(b + "attn.v_bias", seed, (D,), 0.02)
        lin_xavier(b + "attn.qkv.weight", 3 * D, D)
        lin_xavier(b + "attn.proj.weight", D, D)
        sd[b + "attn.proj.bias"] = _normal(b + "attn.proj.bias", seed, (D,), 0.02)
        ln(b + "norm2", D)
        lin_xavier(b + "mlp.fc1.weight", Hm, D)
        sd[b + "mlp.fc1.bias"] = _normal(b + "mlp.fc1.bias", seed, (Hm,), 0.02)
        lin_xavier(b + "mlp.fc2.weight", D, Hm)
        sd[b + "mlp.fc2.bias"] = _normal(b + "mlp.fc2.bias", seed, (D,), 0.02)
    ln("vision_layernorm", D)

    d = cfg.qf_hidden
    q = "qformer.bert."

    def lin_bert(prefix, out_f, in_f):
        sd[prefix + ".weight"] = _normal(prefix + ".weight", seed, (out_f, in_f), 0.02)
        sd[prefix + ".bias"] = _normal(prefix + ".bias", seed, (out_f,), 0.02)

    sd[q + "embeddings.position_ids"] = torch.arange(cfg.qf_max_pos).expand((1, -1)).clone()
    sd[q + "embeddings.word_embeddings.weight"] = _normal(
        q + "embeddings.word_embeddings.weight", seed, (cfg.qf_vocab, d), 0.02)
    sd[q + "embeddings.position_embeddings.weight"] = _normal(
        q + "embeddings.position_embeddings.weight", seed, (cfg.qf_max_pos, d), 0.02)
    ln(q + "embeddings.LayerNorm", d)
    for i in range(cfg.qf_layers):
        l = f"{q}encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            lin_bert(l + "attention.self." + nm, d, d)
        lin_bert(l + "attention.output.dense", d, d)
        ln(l + "attention.output.LayerNorm", d)
        if i % cfg.cross_attention_freq == 0:
            lin_bert(l + "crossattention.self.query", d, d)
            lin_bert(l + "crossattention.self.key", d, D)
            lin_bert(l + "crossattention.self.value", d, D)
            lin_bert(l + "crossattention.output.dense", d, d)
            ln(l + "crossattention.output.LayerNorm", d)
        lin_bert(l + "intermediate.dense", cfg.qf_intermediate, d)
        lin_bert(l + "output.dense", d, cfg.qf_intermediate)
        ln(l + "output.LayerNorm", d)
        lin_bert(l + "intermediate_query.dense", cfg.qf_intermediate, d)
        lin_bert(l + "output_query.dense", d, cfg.qf_intermediate)
        ln(l + "output_query.LayerNorm", d)

    sd["query_tokens"] = _normal("query_tokens", seed, (1, cfg.num_query_token, d), 0.02)
    if cfg.extra_num_query_token > 0:
        sd["extra_query_tokens"] = _normal(
            "extra_query_tokens", seed, (1, cfg.extra_num_query_token, d), 0.02)
    pb = float(1.0 / np.sqrt(d))
    sd["llama_proj.weight"] = _uniform("llama_proj.weight", seed, (cfg.llama_hidden, d), pb)
    sd["llama_proj.bias"] = _uniform("llama_proj.bias", seed, (cfg.llama_hidden,), pb)
    return sd


def make_clip(batch: int, frames: int, img_size: int, seed: int = 0) -> torch.Tensor:
    """Synthetic normalised clip [B,T,3,H,W] ~ N(0,1) (Cfg 1/2)."""
    return _normal("clip", seed, (batch, frames, 3, img_size, img_size), 1.0)


def make_instruction_ids(batch: int, max_len: int, vocab: int, seed: int = 0, ragged: bool = False):
    """Synthetic WordPiece ids standing in for the BertTokenizer:
    [CLS]=101 first, [SEP]=102 last, pad=0; returns (input_ids, attention_mask) int64."""
    rng = _rng("instruction", seed)
    ids = np.zeros((batch, max_len), dtype=np.int64)
    mask = np.zeros((batch, max_len), dtype=np.int64)
    lo = min(999, vocab // 2)
    for b in range(batch):
        n = max_len if (not ragged or b == 0) else int(rng.integers(max(3, max_len // 4), max_len + 1))
        ids[b, :n] = rng.integers(lo, vocab - 1, size=n)
        ids[b, 0] = 101 if vocab > 102 else 1
        ids[b, n - 1] = 102 if vocab > 102 else 2
        mask[b, :n] = 1
    return torch.from_numpy(ids), torch.from_numpy(mask)


def make_llm_state_dict(cfg: LlmConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """fp32 CPU state dict of LlamaForCausalLM (models/blip2/modeling_llama.py:418-433,612-622 key names).
    Linear / embedding weights N(0, 0.02) (LlamaPreTrainedModel._init_weights, :392-402); the RMSNorm gains, all ones in
    the reference, are drawn around 1 so a kernel that drops one fails parity."""
    sd = OrderedDict()
    D, I = cfg.hidden_size, cfg.intermediate_size
    sd["model.embed_tokens.weight"] = _normal("model.embed_tokens.weight", seed, (cfg.vocab_size, D), 0.02)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[p + f"self_attn.{n}.weight"] = _normal(p + f"self_attn.{n}.weight", seed, (D, D), 0.02)
        sd[p + "mlp.gate_proj.weight"] = _normal(p + "mlp.gate_proj.weight", seed, (I, D), 0.02)
        sd[p + "mlp.down_proj.weight"] = _normal(p + "mlp.down_proj.weight", seed, (D, I), 0.02)
        sd[p + "mlp.up_proj.weight"] = _normal(p + "mlp.up_proj.weight", seed, (I, D), 0.02)
        sd[p + "input_layernorm.weight"] = _normal(p + "input_layernorm.weight", seed, (D,), 0.1, 1.0)
        sd[p + "post_attention_layernorm.weight"] = _normal(p + "post_attention_layernorm.weight", seed, (D,), 0.1, 1.0)
    sd["model.norm.weight"] = _normal("model.norm.weight", seed, (D,), 0.1, 1.0)
    sd["lm_head.weight"] = _normal("lm_head.weight", seed, (cfg.vocab_size, D), 0.02)
    return sd


def make_prompt(cfg: LlmConfig, B: int, S: int, n_video: int, seed: int, ragged: bool = False):
    """A spliced prompt: int64 [B, S] source indices (>= 0 token id, < 0 video row -(b*n_video + j) - 1 placed after
    the first quarter of the prompt, like `###Human: <Video>` + video + `</Video> ...`, tasks/test.py:79-97) and a
    right-padded attention mask [B, S] (hawkeye_it.py:340-368 pads on the right)."""
    rng = _rng("prompt", seed)
    ids = torch.from_numpy(rng.integers(3, cfg.vocab_size, size=(B, S), dtype=np.int64))
    ids[:, 0] = 1                                       # <s>
    start = max(1, (S - n_video) // 4)
    for b in range(B):
        for j in range(n_video):
            ids[b, start + j] = -(b * n_video + j) - 1
    mask = torch.ones(B, S, dtype=torch.int64)
    if ragged:
        for b in range(1, B):
            n_pad = int(rng.integers(1, max(2, (S - start - n_video) // 2)))
            mask[b, S - n_pad:] = 0
            ids[b, S - n_pad:] = 2                      # pad = eos (hawkeye_it.py:104)
    return ids, mask


# ---------------------------------------------------------------------------------------------------------------------
# "Massive activation" statistics (missing item 3).  A trained UMT-L, unlike an init-law model, carries a few
# residual-stream channels two orders of magnitude above the row RMS, LayerNorm gains far from 1 and attention heads with
# sharply peaked logits -- exactly where a bf16 residual stream and a folded LayerNorm lose bits.  Real checkpoints are
# not available offline, so the statistics are produced by a deterministic edit of the seeded state dict; the same edit
# goes through the reference modules (oracle/make_goldens.py outlier -> tests/golden/full_T8_outlier.npz).
# ---------------------------------------------------------------------------------------------------------------------
OUTLIER_CHANNELS = (77, 412, 901)      # residual channels driven to ~100x the row RMS
OUTLIER_GAIN_CHANNELS = (5, 600)       # norm1.weight entries x 20 in every block
OUTLIER_HEAD = 3                       # the attention head whose logits are spread > 30 (every 4th block)


def make_outlier_state_dict(cfg: EncoderConfig, seed: int = 0, depth_in_ckpt: bool = False):
    sd = make_state_dict(cfg, seed, depth_in_ckpt)
    D, hd = cfg.encoder_embed_dim, cfg.encoder_embed_dim // cfg.encoder_num_heads
    p = "vision_encoder.encoder."
    nblk = cfg.encoder_depth if depth_in_ckpt else cfg.run_depth
    ch = [c % D for c in OUTLIER_CHANNELS]
    sign = (1.0, -1.0, 1.0)
    # the patch-embed output of these weights has row RMS ~0.9 over the remaining channels (unit-normal pixels,
    # kaiming-uniform Conv3d) and the blocks grow it to ~4; the channels start at +-100 and each fc2 bias pushes them 15
    # further (~450 after block 22), which keeps them at ~100x the RMS of the remaining channels through all 23 blocks
    for c, s in zip(ch, sign):
        sd[p + "patch_embed.proj.bias"][c] += 100.0 * s
    for i in range(nblk):
        b = f"{p}blocks.{i}."
        for c, s in zip(ch, sign):
            sd[b + "mlp.fc2.bias"][c] += 15.0 * s
        for c in OUTLIER_GAIN_CHANNELS:
            sd[b + "norm1.weight"][c % D] *= 20.0
        if i % 4 == 1:
            # one head with sharply peaked attention: its q and k rows x 12 => logits x 144
            h = OUTLIER_HEAD % cfg.encoder_num_heads
            w = sd[b + "attn.qkv.weight"]
            w[h * hd:(h + 1) * hd] *= 12.0
            w[D + h * hd:D + (h + 1) * hd] *= 12.0
            sd[b + "attn.q_bias"][h * hd:(h + 1) * hd] *= 12.0
    return sd
