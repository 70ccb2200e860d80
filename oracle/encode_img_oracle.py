"""ORACLE -- test infrastructure, not product code.

A CPU fp32 restatement of the reference's `HawkEye_it.encode_img` arithmetic
(models/hawkeye_it.py:219-270) written as plain functions over a flat state
dict that uses the reference's own key names.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file;
the product path (hawkeye_amd/) never does.

Pinning: the reference holds no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference's own
modules run in the build container: tests/golden/*.npz, produced by
oracle/make_goldens.py (which imports /root/reference/models/blip2/{vit,Qformer}.py).
tests/test_oracle_vs_golden.py re-checks that pin on every CPU test run.

Every function cites the reference lines it restates.  Floating point; the
arithmetic type is fp32 throughout (the reference's CPU path: no autocast,
models/blip2/blip2.py:36-44).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# Frame transform feeding the path (row f3)
# ----------------------------------------------------------------------------

def normalize_frames(frames_u8, mean, std):
    """uint8 frames [B,T,H,W,3] -> the fp32 clip [B,T,3,H,W] the eval scripts hand to encode_img.

    dataset/video_transforms.py:367-389 ToTorchFormatTensor (HWC -> CHW, .float().div(255)) and :94-107
    GroupNormalize (per channel t.sub_(mean).div_(std), fp32 tensor with Python-float scalars), then the caller's
    reshape to [1,T,3,H,W] (tasks/test.py:490-492).  Pinned by tests/golden/frame_transform.npz."""
    x = frames_u8.permute(0, 1, 4, 2, 3).contiguous().float().div(255)
    for c in range(3):
        x[:, :, c].sub_(float(mean[c])).div_(float(std[c]))
    return x


# ----------------------------------------------------------------------------
# ViT  (models/blip2/vit.py)
# ----------------------------------------------------------------------------

def patch_embed(sd, image_bcthw, patch):
    """vit.py:146-158 -- Conv3d(3, D, k=s=(1,p,p)) then flatten(2).transpose(1,2).

    Restated as the per-frame patch GEMM it is (tubelet_size=1): rows ordered
    (t, h/p, w/p), K ordered (c, ky, kx) -- the Conv3d weight's own flattening.
    """
    w = sd["vision_encoder.encoder.patch_embed.proj.weight"]
    b = sd["vision_encoder.encoder.patch_embed.proj.bias"]
    B, C, T, H, W = image_bcthw.shape
    D = w.shape[0]
    gh, gw = H // patch, W // patch
    x = image_bcthw.reshape(B, C, T, gh, patch, gw, patch)
    x = x.permute(0, 2, 3, 5, 1, 4, 6).reshape(B, T * gh * gw, C * patch * patch)
    return x @ w.reshape(D, -1).t() + b


def vit_attention(sd, pre, x, heads):
    """vit.py:81-100 -- qkv bias = cat(q_bias, 0, v_bias); q scaled BEFORE q k^T."""
    B, N, C = x.shape
    hd = C // heads
    qb, vb = sd[pre + "q_bias"], sd[pre + "v_bias"]
    bias = torch.cat((qb, torch.zeros_like(vb), vb))
    qkv = F.linear(x, sd[pre + "qkv.weight"], bias)
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (hd ** -0.5)
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def vit_mlp(sd, pre, x):
    """vit.py:48-54 -- fc2(GELU_erf(fc1(x)))."""
    h = F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"])
    h = F.gelu(h)  # nn.GELU default = exact erf
    return F.linear(h, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def vit_block(sd, i, x, heads, eps):
    """vit.py:124-127 -- pre-LN residual block, gamma_1/2 = None (init_values=0)."""
    b = f"vision_encoder.encoder.blocks.{i}."
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[b + "norm1.weight"], sd[b + "norm1.bias"], eps)
    x = x + vit_attention(sd, b + "attn.", h, heads)
    h = F.layer_norm(x, (C,), sd[b + "norm2.weight"], sd[b + "norm2.bias"], eps)
    x = x + vit_mlp(sd, b + "mlp.", h)
    return x


def vit_forward(sd, cfg, image_bcthw, pos_embed, stages=None):
    """vit.py:301-320, 399-405 -- patch-embed + pos table, run_depth blocks, norm=Identity."""
    x = patch_embed(sd, image_bcthw, cfg["patch_size"])
    if stages is not None:
        stages["patch_embed"] = x
    x = x + pos_embed.to(x.dtype)
    run_depth = cfg["encoder_depth"] + cfg["return_index"] + 1
    for i in range(run_depth):
        x = vit_block(sd, i, x, cfg["encoder_num_heads"], cfg["vit_ln_eps"])
        if stages is not None and (i == 0 or i == run_depth - 1):
            stages[f"vit_block_{i}"] = x
    return x


# ----------------------------------------------------------------------------
# Q-Former  (models/blip2/Qformer.py)
# ----------------------------------------------------------------------------

# kinds of dropout site inside one BertLayer, shared with oracle/dropout_masks.py and hawkeye_amd/train.py
SELF_PROB, SELF_HID, SELF_PATH, CROSS_PROB, CROSS_HID, CROSS_PATH, QFFN_HID, QFFN_PATH, TFFN_HID, TFFN_PATH = range(10)
EMBED_SITE = 4095


def _site(layer, kind):
    return layer * 16 + kind


def qf_embeddings(sd, cfg, input_ids, query_embeds, drop=None):
    """Qformer.py:80-121 -- word + absolute position embeddings for the text
    tokens only, cat(query, text), LayerNorm; dropout (:120) only in train mode.

    `drop` (all functions below): None = eval mode.  Otherwise an object with elem(site, x, p) / path(site, x, p)
    (element dropout / per-sample drop-path with an injected mask, oracle/dropout_masks.MaskDropper) and the
    probabilities p_hidden, p_attn, path_rates[layer] (blip2.py:62-64)."""
    q = "qformer.bert.embeddings."
    d = cfg["qf_hidden"]
    if input_ids is not None:
        Lt = input_ids.shape[1]
        emb = sd[q + "word_embeddings.weight"][input_ids]
        emb = emb + sd[q + "position_embeddings.weight"][:Lt].unsqueeze(0)
        emb = torch.cat((query_embeds, emb), dim=1)
    else:
        emb = query_embeds
    out = F.layer_norm(emb, (d,), sd[q + "LayerNorm.weight"], sd[q + "LayerNorm.bias"], cfg["qf_ln_eps"])
    if drop is not None:
        out = drop.elem(EMBED_SITE, out, drop.p_hidden)                                   # :120
    return out


def qf_attention(sd, pre, cfg, hidden, kv_source, additive_mask, drop=None, layer=0, kinds=(SELF_PROB, SELF_HID, SELF_PATH)):
    """Qformer.py:182-288 (BertSelfAttention) + 313-318 (BertSelfOutput).

    scores = q k^T / sqrt(hd) + mask (scaled AFTER q k^T, :257-260); softmax;
    context; dense; post-LN LayerNorm(dense + input)."""
    d = cfg["qf_hidden"]
    H = cfg["qf_heads"]
    hd = d // H

    def split(t):
        return t.view(t.shape[0], t.shape[1], H, hd).permute(0, 2, 1, 3)

    s = pre + "self."
    q = split(F.linear(hidden, sd[s + "query.weight"], sd[s + "query.bias"]))
    k = split(F.linear(kv_source, sd[s + "key.weight"], sd[s + "key.bias"]))
    v = split(F.linear(kv_source, sd[s + "value.weight"], sd[s + "value.bias"]))
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
    if additive_mask is not None:
        scores = scores + additive_mask
    probs = scores.softmax(dim=-1)
    if drop is not None:
        probs = drop.elem(_site(layer, kinds[0]), probs, drop.p_attn)                     # :271
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
    ctx = ctx.view(ctx.shape[0], ctx.shape[1], d)
    o = pre + "output."
    out = F.linear(ctx, sd[o + "dense.weight"], sd[o + "dense.bias"])
    if drop is not None:
        out = drop.elem(_site(layer, kinds[1]), out, drop.p_hidden)                       # :315
        out = drop.path(_site(layer, kinds[2]), out, drop.path_rates[layer])              # :316
    return F.layer_norm(out + hidden, (d,), sd[o + "LayerNorm.weight"], sd[o + "LayerNorm.bias"],
                        cfg["qf_ln_eps"])


def qf_ffn(sd, pre_i, pre_o, cfg, x, drop=None, layer=0, kinds=(QFFN_HID, QFFN_PATH)):
    """Qformer.py:387-390, 401-406, 509-517 -- GELU(erf) FFN with post-LN residual."""
    d = cfg["qf_hidden"]
    h = F.gelu(F.linear(x, sd[pre_i + "dense.weight"], sd[pre_i + "dense.bias"]))
    h = F.linear(h, sd[pre_o + "dense.weight"], sd[pre_o + "dense.bias"])
    if drop is not None:
        h = drop.elem(_site(layer, kinds[0]), h, drop.p_hidden)                           # :403
        h = drop.path(_site(layer, kinds[1]), h, drop.path_rates[layer])                  # :404
    return F.layer_norm(h + x, (d,), sd[pre_o + "LayerNorm.weight"], sd[pre_o + "LayerNorm.bias"],
                        cfg["qf_ln_eps"])


def qf_layer(sd, cfg, i, hidden, self_mask, image_embeds, cross_mask, query_length, drop=None):
    """Qformer.py:435-507 (BertLayer.forward)."""
    l = f"qformer.bert.encoder.layer.{i}."
    att = qf_attention(sd, l + "attention.", cfg, hidden, hidden, self_mask, drop, i)
    q_out = att[:, :query_length, :]
    if i % cfg["cross_attention_freq"] == 0:
        q_out = qf_attention(sd, l + "crossattention.", cfg, q_out, image_embeds, cross_mask, drop, i,
                             (CROSS_PROB, CROSS_HID, CROSS_PATH))
    out = qf_ffn(sd, l + "intermediate_query.", l + "output_query.", cfg, q_out, drop, i)
    if att.shape[1] > query_length:
        out_t = qf_ffn(sd, l + "intermediate.", l + "output.", cfg, att[:, query_length:, :], drop, i,
                       (TFFN_HID, TFFN_PATH))
        out = torch.cat([out, out_t], dim=1)
    return out


def qformer_forward(sd, cfg, input_ids, attention_mask, query_embeds, image_embeds, stages=None, drop=None):
    """Qformer.py:837-998 (BertModel.forward), encoder Qformer.py:550-592.

    Self mask: (1 - m) * -10000 broadcast over heads and query rows (:818,:834).
    Cross mask: invert_attention_mask(ones) = all zeros (hawkeye_it.py:235)."""
    hidden = qf_embeddings(sd, cfg, input_ids, query_embeds, drop)
    if stages is not None:
        stages["qf_embeddings"] = hidden
    self_mask = (1.0 - attention_mask[:, None, None, :].to(hidden.dtype)) * -10000.0
    cross_mask = None  # all-zero additive mask
    ql = query_embeds.shape[1]
    for i in range(cfg["qf_layers"]):
        hidden = qf_layer(sd, cfg, i, hidden, self_mask, image_embeds, cross_mask, ql, drop)
        if stages is not None:
            stages[f"qf_layer_{i}"] = hidden
    return hidden


# ----------------------------------------------------------------------------
# encode_img  (models/hawkeye_it.py:219-270)
# ----------------------------------------------------------------------------

def encode_img(sd, cfg, image, input_ids, text_mask, pos_embed, stages=None, drop=None):
    """image [B,T,3,H,W] fp32; input_ids/text_mask [B,Lt] int64 (stand-in for the
    BertTokenizer output, hawkeye_it.py:243-249); pos_embed [1,T*L,D] is the
    table the encoder holds at call time (pos_embed, or img_pos_embed when T==1,
    vit.py:304-307).  Returns (inputs_llama [B,nq,llama_hidden], use_image)."""
    T = image.shape[1]
    use_image = T == 1
    x = image.permute(0, 2, 1, 3, 4)                                  # :228
    emb = vit_forward(sd, cfg, x, pos_embed, stages)                  # :231
    D = emb.shape[-1]
    emb = F.layer_norm(emb, (D,), sd["vision_layernorm.weight"], sd["vision_layernorm.bias"],
                       cfg["vision_ln_eps"])                          # :234
    if stages is not None:
        stages["vision_layernorm"] = emb
    B = emb.shape[0]
    qt = sd["query_tokens"]
    if "extra_query_tokens" in sd:
        qt = torch.cat([qt, sd["extra_query_tokens"]], dim=1)         # :237-238
    qt = qt.expand(B, -1, -1)                                         # :241
    nq = qt.shape[1]
    if input_ids is not None:
        atts = torch.cat([torch.ones(B, nq, dtype=torch.long), text_mask], dim=1)   # :250-251
    else:
        # qformer_text_input=False (:261-267): no ids, no mask -> BertModel.forward builds an all-ones mask over
        # the query rows (Qformer.py:912-915)
        atts = torch.ones(B, nq, dtype=torch.long)
    hidden = qformer_forward(sd, cfg, input_ids, atts, qt, emb, stages, drop)    # :253-260
    out = F.linear(hidden[:, :nq, :], sd["llama_proj.weight"], sd["llama_proj.bias"])  # :269
    return out, use_image
