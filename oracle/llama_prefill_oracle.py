"""ORACLE -- test infrastructure, not product code (see oracle/encode_img_oracle.py for the rules).

CPU fp32 restatement of the LLaMA prefill that consumes encode_img's output:
LlamaForCausalLM.forward with inputs_embeds (models/blip2/modeling_llama.py:615-742) -> LlamaModel.forward
(:482-611) -> LlamaDecoderLayer (:232-300) -> LlamaAttention (:143-231), LlamaMLP (:125-140), LlamaRMSNorm
(:57-75), rotary embedding (:78-123), plus embed_tokens_ (:444-455) and the splice of the video tokens into
the prompt (tasks/test.py:79-97).  Plain functions over a state dict with the reference's key names.

Pinned by tests/golden/llm_small.npz, produced by oracle/make_goldens.py from the reference's own
modeling_llama.py classes (the file imports unmodified under transformers 5.x).
"""
import math

import torch
import torch.nn.functional as F


def rmsnorm(x, weight, eps):
    """modeling_llama.py:66-75"""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return weight * (x * torch.rsqrt(var + eps))


def rotary_tables(head_dim, n_pos, base=10000.0):
    """modeling_llama.py:79-89: inv_freq, freqs = t x inv_freq, emb = cat(freqs, freqs); cos/sin fp32.
    Returns the first halves [n_pos, head_dim/2] (the two halves are equal)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(n_pos, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return freqs.cos(), freqs.sin()


def rotate_half(x):
    """modeling_llama.py:108-113"""
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary(q, k, cos_h, sin_h, position_ids):
    """modeling_llama.py:116-123; q,k [B,H,S,hd]; cos_h/sin_h [P, hd/2]; position_ids [B,S]."""
    cos = torch.cat((cos_h, cos_h), -1)[position_ids][:, None]     # [B,1,S,hd]
    sin = torch.cat((sin_h, sin_h), -1)[position_ids][:, None]
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def decoder_mask(attention_mask, S, dtype=torch.float32):
    """modeling_llama.py:25-55,457-480: causal mask + expanded padding mask, additive, finfo.min."""
    B = attention_mask.shape[0]
    neg = torch.finfo(dtype).min
    causal = torch.full((S, S), neg, dtype=dtype)
    causal = causal.masked_fill(torch.arange(S)[None, :] <= torch.arange(S)[:, None], 0.0)
    inv = 1.0 - attention_mask[:, None, None, :].to(dtype).expand(B, 1, S, S)
    pad = inv.masked_fill(inv.bool(), neg)
    return pad + causal[None, None]


def attention(sd, pre, x, mask, position_ids, heads, cos_h, sin_h, kv_out=None):
    """modeling_llama.py:176-231"""
    B, S, D = x.shape
    hd = D // heads
    q = F.linear(x, sd[pre + "q_proj.weight"]).view(B, S, heads, hd).transpose(1, 2)
    k = F.linear(x, sd[pre + "k_proj.weight"]).view(B, S, heads, hd).transpose(1, 2)
    v = F.linear(x, sd[pre + "v_proj.weight"]).view(B, S, heads, hd).transpose(1, 2)
    q, k = apply_rotary(q, k, cos_h, sin_h, position_ids)
    if kv_out is not None:
        kv_out.append((k, v))
    w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(hd)
    w = w + mask
    w = torch.max(w, torch.tensor(torch.finfo(w.dtype).min))
    w = F.softmax(w, dim=-1, dtype=torch.float32)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, S, D)
    return F.linear(o, sd[pre + "o_proj.weight"])


def mlp(sd, pre, x):
    """modeling_llama.py:139-140 (hidden_act = silu)"""
    return F.linear(F.silu(F.linear(x, sd[pre + "gate_proj.weight"])) * F.linear(x, sd[pre + "up_proj.weight"]),
                    sd[pre + "down_proj.weight"])


def decoder_layer(sd, i, x, mask, position_ids, cfg, cos_h, sin_h, kv_out=None):
    """modeling_llama.py:268-300"""
    p = f"model.layers.{i}."
    h = x + attention(sd, p + "self_attn.", rmsnorm(x, sd[p + "input_layernorm.weight"], cfg["rms_norm_eps"]), mask,
                      position_ids, cfg["num_attention_heads"], cos_h, sin_h, kv_out)
    return h + mlp(sd, p + "mlp.", rmsnorm(h, sd[p + "post_attention_layernorm.weight"], cfg["rms_norm_eps"]))


def embed_splice(sd, src, video):
    """embed_tokens_ (modeling_llama.py:444-455, num_frame_tokens = 0) + the torch.cat splice of tasks/test.py:79-97,
    written as one gather: src >= 0 token id, src < 0 row -src-1 of video [*, D]."""
    table = sd["model.embed_tokens.weight"]
    tok = table[src.clamp_min(0)]
    if video is None:
        return tok
    vid = video.reshape(-1, table.shape[1]).float()[(-src - 1).clamp_min(0)]
    return torch.where((src >= 0)[..., None], tok, vid)


def generation_position_ids(attention_mask):
    """prepare_inputs_for_generation (:751-754): positions count the valid tokens, padded positions get 1."""
    pos = attention_mask.long().cumsum(-1) - 1
    return pos.masked_fill(attention_mask == 0, 1)


def last_valid_index(attention_mask):
    """Index of the last 1 of every row (right padding: sum - 1; left padding: S - 1)."""
    S = attention_mask.shape[1]
    return S - 1 - attention_mask.long().flip(1).argmax(1)


def prefill(sd, cfg, inputs_embeds, attention_mask, stages=None, position_ids=None):
    """LlamaModel.forward (:482-611) with inputs_embeds + LlamaForCausalLM's lm_head (:688-700).
    position_ids None = the forward's default arange (:526-533); generate passes generation_position_ids(mask).
    Returns (final-norm hidden [B,S,D], logits at each row's last valid position [B,V], [(k, v)] per layer)."""
    B, S, D = inputs_embeds.shape
    if position_ids is None:
        position_ids = torch.arange(S)[None].expand(B, S)                   # :526-533
    mask = decoder_mask(attention_mask, S)
    cos_h, sin_h = rotary_tables(D // cfg["num_attention_heads"], cfg["max_position_embeddings"], cfg["rope_base"])
    x = inputs_embeds.float()
    kv = []
    for i in range(cfg["num_hidden_layers"]):
        x = decoder_layer(sd, i, x, mask, position_ids, cfg, cos_h, sin_h, kv)
        if stages is not None:
            stages[f"layer{i}"] = x
    x = rmsnorm(x, sd["model.norm.weight"], cfg["rms_norm_eps"])
    last = last_valid_index(attention_mask)
    logits = F.linear(x[torch.arange(B), last], sd["lm_head.weight"])
    return x, logits, kv


def greedy_decode(sd, cfg, inputs_embeds, attention_mask, n_tokens, forced=None):
    """generate(do_sample=False, num_beams=1) as the eval scripts call it (tasks/test.py:121-133): the arg-max of the
    last-position logits is appended, its embedding (embed_tokens_, :444-455) extends the sequence, the mask grows by a
    one and positions stay cumsum(mask) - 1 (prepare_inputs_for_generation :744-773).  Restated WITHOUT a cache: by
    causality the last row of a forward over the whole extended sequence equals the reference's one-token step with
    past_key_values (:199-206), so this checks the cached implementations from the other side.
    forced int64 [B, n_tokens] or None: tokens to feed instead of the arg-max.
    Returns (logits [n_tokens, B, V] -- the logits that chose each token --, tokens [B, n_tokens])."""
    emb, mask = inputs_embeds.float(), attention_mask.clone()
    logits, tokens = [], []
    for t in range(n_tokens):
        _, lg, _ = prefill(sd, cfg, emb, mask, position_ids=generation_position_ids(mask))
        nxt = lg.argmax(-1)
        logits.append(lg)
        tokens.append(nxt)
        feed = nxt if forced is None else forced[:, t]
        emb = torch.cat([emb, sd["model.embed_tokens.weight"][feed][:, None].float()], 1)
        mask = torch.cat([mask, torch.ones_like(mask[:, :1])], 1)
    return torch.stack(logits), torch.stack(tokens, 1)
