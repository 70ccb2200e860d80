"""Row f1 on a real MI355X: the LLaMA-prefill kernels through the C ABI against fp32 PyTorch references, and the
whole prefill against the reference-module golden (tests/golden/llm_small.npz) and the oracle.

Tolerances: bf16 operands / fp32 accumulation -> outputs differ from fp32 references by the bf16 rounding of the
stored tensors (rel 2^-8 of the max); end to end cosine >= 0.999 (north_star tolerance), measured 0.9999+."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hawkeye_amd import ops, weights as W  # noqa: E402
from hawkeye_amd.config import SMALL_LLM  # noqa: E402
from hawkeye_amd.llm import LlamaPrefill, merge_lora, prefill_flops  # noqa: E402

DEV = "cuda:0"


def _rand(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV).to(dtype)


def _rel(a, ref):
    return ((a.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)).item()


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm())).item()


@pytest.mark.parametrize("D,in_bf16", [(256, False), (4096, False), (4096, True), (1024, True)])
def test_rmsnorm(D, in_bf16):
    x = _rand((1031, D), 1, 2.0)
    if in_bf16:
        x = x.to(torch.bfloat16)
    w = _rand((D,), 2, 0.1) + 1.0
    out = ops.rmsnorm(x, w, 1e-6)
    xf = x.float()
    ref = w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert _rel(out.float(), ref) < 5e-3


@pytest.mark.parametrize("P,rows,D", [(4, 352, 4096), (2, 37, 1024), (1, 5, 256), (3, 100, 5120), (5, 9, 768)])
def test_rmsnorm_of_summed_split_k_slices(P, rows, D):
    """hk_rmsnorm_add_parts: the residual add of o_proj / down_proj and the norm after it, on split-K slices (slice order, fp32)."""
    parts = _rand((P, rows, D), 5, 1.5)
    w = _rand((D,), 6, 0.1) + 1.0
    x, out = ops.rmsnorm_add_parts(parts, w, 1e-6)
    xs = parts[0].clone()
    for q in range(1, P):
        xs += parts[q]
    assert torch.equal(x, xs)
    ref = w * (xs * torch.rsqrt(xs.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert _rel(out.float(), ref) < 5e-3
    assert torch.equal(out, ops.rmsnorm(xs, w, 1e-6)) or _rel(out.float(), ops.rmsnorm(xs, w, 1e-6).float()) < 8e-3


def test_prefill_split_k_route_matches_the_unsplit_one():
    """Skinny prompts send o_proj / down_proj through split-K + hk_rmsnorm_add_parts: same logits and hidden states as the
    unsplit route to fp32 summation-order noise (bf16 rounding of the normalised stream can flip a last bit)."""
    from hawkeye_amd.config import LlmConfig
    from hawkeye_amd.llm import LlamaPrefill
    cfg = LlmConfig(hidden_size=1024, num_hidden_layers=3, num_attention_heads=8, intermediate_size=4096, vocab_size=512,
                    max_position_embeddings=512)
    B, S = 1, 200
    src = torch.randint(0, cfg.vocab_size, (B, S), device=DEV)
    outs = []
    for sk in (1, 4):
        m = LlamaPrefill(cfg, DEV).init_random(3)
        m.split_k_for = lambda M, N, K, sk=sk: sk if (K // 64) % sk == 0 else 1
        r = m.prefill(src=src)
        outs.append((r["logits"].clone(), r["hidden"].float().clone()))
        if sk > 1:
            assert any(name == "hk_rmsnorm_add_parts" for _, _, name in next(iter(m.plans.values())).steps)
    assert _rel(outs[1][0], outs[0][0]) < 2e-2 and _rel(outs[1][1], outs[0][1]) < 2e-2


def test_rope_matches_reference_formula():
    from oracle import llama_prefill_oracle as LO
    B, S, H, hd = 2, 37, 3, 128
    qkv = _rand((B * S, 3 * H * hd), 3, dtype=torch.bfloat16)
    cos_h, sin_h = LO.rotary_tables(hd, 64)
    q0 = qkv.float().view(B, S, 3, H, hd)
    q, k, v = (q0[:, :, i].transpose(1, 2).cpu() for i in range(3))
    pos = torch.arange(S)[None].expand(B, S)
    qr, kr = LO.apply_rotary(q, k, cos_h, sin_h, pos)
    got = ops.rope_(qkv.clone(), cos_h.to(DEV), sin_h.to(DEV), H, hd, S, q_scale=0.25).float().view(B, S, 3, H, hd)
    assert _rel(got[:, :, 0].transpose(1, 2).cpu(), qr * 0.25) < 5e-3
    assert _rel(got[:, :, 1].transpose(1, 2).cpu(), kr) < 5e-3
    assert torch.equal(got[:, :, 2], q0[:, :, 2])                 # v untouched
    # explicit position ids (reversed) select other table rows
    pid = torch.arange(S - 1, -1, -1, dtype=torch.int32).repeat(B).to(DEV)
    got2 = ops.rope_(qkv.clone(), cos_h.to(DEV), sin_h.to(DEV), H, hd, S, position_ids=pid).float().view(B, S, 3, H, hd)
    qr2, _ = LO.apply_rotary(q, k, cos_h, sin_h, pos.flip(1))
    assert _rel(got2[:, :, 0].transpose(1, 2).cpu(), qr2) < 5e-3


def test_silu_mul_and_embed_splice():
    gu = _rand((513, 2 * 1376), 4, 2.0, dtype=torch.bfloat16)
    out = ops.silu_mul(gu)
    g, u = gu.float()[:, :1376], gu.float()[:, 1376:]
    assert _rel(out.float(), torch.nn.functional.silu(g) * u) < 5e-3
    table = _rand((100, 256), 5, dtype=torch.bfloat16)
    for vdt in (torch.bfloat16, torch.float32):
        video = _rand((6, 256), 6, dtype=vdt)
        src = torch.tensor([5, 99, -1, -6, 0, -3, 7], dtype=torch.int64, device=DEV)
        got = ops.embed_splice(src, table, video)
        want = torch.stack([table[5], table[99], video[0], video[5], table[0], video[2], table[7]]).float()
        assert torch.equal(got, want)
    with pytest.raises(RuntimeError, match="outside"):
        ops.embed_splice(torch.tensor([100], device=DEV), table, None)
    with pytest.raises(RuntimeError, match="video row"):
        ops.embed_splice(torch.tensor([-7], device=DEV), table, video)


def _ref_attn(q, k, v, scale, causal, key_bias=None):
    # q [B,Nq,H,hd] etc.; fp32 softmax reference with the reference's additive masks
    qf, kf, vf = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))
    s = qf @ kf.transpose(2, 3) * scale
    Nq, Nk = q.shape[1], k.shape[1]
    if causal:
        i = torch.arange(Nq, device=q.device)[:, None] + (Nk - Nq)
        j = torch.arange(Nk, device=q.device)[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    return (torch.softmax(s, -1) @ vf).permute(0, 2, 1, 3)


@pytest.mark.parametrize("hd,B,H,Nq,Nk,causal,bias", [
    (128, 2, 3, 352, 352, True, False), (128, 1, 2, 200, 200, True, True), (128, 2, 2, 96, 300, False, False),
    (128, 1, 2, 70, 198, True, False), (64, 2, 4, 257, 257, True, False), (64, 1, 3, 130, 130, True, True),
    (128, 1, 32, 1024, 1024, True, False)])
def test_causal_attention_head128_and_64(hd, B, H, Nq, Nk, causal, bias):
    q = _rand((B, Nq, H, hd), 7, dtype=torch.bfloat16)
    k = _rand((B, Nk, H, hd), 8, dtype=torch.bfloat16)
    v = _rand((B, Nk, H, hd), 9, dtype=torch.bfloat16)
    scale = hd ** -0.5
    qs = (q.float() * (scale * math.log2(math.e))).to(torch.bfloat16)
    kb = None
    if bias:      # right-padded row: the last 40 keys masked like (1 - mask) * -10000
        kb = torch.zeros((B, Nk), device=DEV)
        kb[0, Nk - 40:] = -10000.0
    out = ops.flash_attention(qs, k, v, key_bias=kb, causal=causal)
    ref = _ref_attn(qs.float() / (scale * math.log2(math.e)), k, v, scale, causal, kb)
    rows = slice(0, Nq - 40) if bias else slice(None)     # rows whose own position is padding are unspecified
    assert _rel(out.float()[:, rows], ref[:, rows]) < 1.5e-2


def test_causal_attention_left_padding_keeps_precision():
    """First key tiles fully masked for every row (left padding > 64 keys): the running maximum starts at the mask
    value and must not cost precision once real keys arrive (scores are recomputed, not shifted)."""
    B, H, N, hd = 1, 2, 300, 128
    q, k, v = (_rand((B, N, H, hd), s, dtype=torch.bfloat16) for s in (11, 12, 13))
    scale = hd ** -0.5
    qs = (q.float() * (scale * math.log2(math.e))).to(torch.bfloat16)
    kb = torch.zeros((B, N), device=DEV)
    kb[0, :150] = -10000.0
    out = ops.flash_attention(qs, k, v, key_bias=kb, causal=True)
    ref = _ref_attn(qs.float() / (scale * math.log2(math.e)), k, v, scale, True, kb)
    assert _rel(out.float()[:, 150:], ref[:, 150:]) < 1.5e-2


def test_prefill_matches_reference_golden_and_oracle(golden_dir):
    from oracle import llama_prefill_oracle as LO
    g = np.load(os.path.join(golden_dir, "llm_small.npz"))
    cfg = SMALL_LLM
    B, S, nv, seed = int(g["B"]), int(g["S"]), int(g["n_video"]), int(g["seed"])
    sd = W.make_llm_state_dict(cfg, seed)
    m = LlamaPrefill(cfg, DEV).load_state_dict(sd)
    src, mask = torch.from_numpy(g["src"]).to(DEV), torch.from_numpy(g["mask"]).to(DEV)
    video = torch.from_numpy(g["video"]).to(DEV)
    out = m.prefill(src=src, video=video, attention_mask=mask)
    valid = mask.bool()
    ref_h = torch.from_numpy(g["final_norm"]).to(DEV)
    assert _cos(out["hidden"].float()[valid], ref_h[valid]) > 0.999
    assert _rel(out["hidden"].float()[valid], ref_h[valid]) < 3e-2
    ref_l = torch.from_numpy(g["logits_last"]).to(DEV)
    assert _cos(out["logits"], ref_l) > 0.999 and _rel(out["logits"], ref_l) < 3e-2
    k0, v0 = out["kv"][0]
    rk = torch.from_numpy(g["k0"]).to(DEV).permute(0, 2, 1, 3)      # [B,S,H,hd]
    rv = torch.from_numpy(g["v0"]).to(DEV).permute(0, 2, 1, 3)
    assert _rel(k0.float()[valid], rk[valid]) < 1e-2 and _rel(v0.float()[valid], rv[valid]) < 1e-2
    # inputs_embeds entry (what generate(inputs_embeds=...) receives, tasks/test.py:119): same result
    out2 = m.prefill(inputs_embeds=torch.from_numpy(g["inputs_embeds"]).to(DEV), attention_mask=mask)
    assert _cos(out2["logits"], ref_l) > 0.999
    # no mask == all-ones mask; fresh seed vs the oracle run here
    src3, _ = W.make_prompt(cfg, 1, 200, 32, 9)
    vid3 = W._normal("v3", 9, (32, cfg.hidden_size), 0.05)
    sd_cpu = sd
    with torch.no_grad():
        h3, l3, _ = LO.prefill(sd_cpu, cfg.to_dict(), LO.embed_splice(sd_cpu, src3, vid3), torch.ones(1, 200, dtype=torch.long))
    o3 = m.prefill(src=src3.to(DEV), video=vid3.to(DEV).to(torch.bfloat16))
    # video tokens arrive in bf16 here (encode_img's output dtype): compare against the oracle on the same rounding
    with torch.no_grad():
        h3b, l3b, _ = LO.prefill(sd_cpu, cfg.to_dict(), LO.embed_splice(sd_cpu, src3, vid3.to(torch.bfloat16).float()),
                                 torch.ones(1, 200, dtype=torch.long))
    assert _cos(o3["logits"].cpu(), l3b) > 0.999 and _cos(o3["hidden"].float().cpu(), h3b) > 0.999
    assert prefill_flops(cfg, 1, 200) > 0


def test_prefill_left_padded_batch_matches_reference_generate_prefill(golden_dir):
    """A left-padded batch as `generate` prefills it (position ids cumsum(mask) - 1, logits at the last position):
    against the reference-module fixture; the same mask with default position ids is refused."""
    g = np.load(os.path.join(golden_dir, "llm_small_leftpad.npz"))
    cfg, seed = SMALL_LLM, int(g["seed"])
    m = LlamaPrefill(cfg, DEV).load_state_dict(W.make_llm_state_dict(cfg, seed))
    src, mask = torch.from_numpy(g["src"]).to(DEV), torch.from_numpy(g["mask"]).to(DEV)
    video = torch.from_numpy(g["video"]).to(DEV)
    ref_l = torch.from_numpy(g["logits_last"]).to(DEV)
    ref_h = torch.from_numpy(g["final_norm"]).to(DEV)
    valid = mask.bool()
    for pos in ("generate", torch.from_numpy(g["position_ids"]).to(DEV)):
        out = m.prefill(src=src, video=video, attention_mask=mask, position_ids=pos)
        assert _cos(out["logits"], ref_l) > 0.999 and _rel(out["logits"], ref_l) < 3e-2
        for b in range(mask.shape[0]):          # row by row: a wrong last index / shifted positions shows per row
            assert _cos(out["logits"][b], ref_l[b]) > 0.999, b
        assert _cos(out["hidden"].float()[valid], ref_h[valid]) > 0.999
        k0 = out["kv"][0][0]
        rk = torch.from_numpy(g["k0"]).to(DEV).permute(0, 2, 1, 3)
        assert _rel(k0.float()[valid], rk[valid]) < 1e-2
    with pytest.raises(RuntimeError, match="left-padded"):
        m.prefill(src=src, video=video, attention_mask=mask)


def test_lora_merge_equals_adapter_forward():
    cfg, seed = SMALL_LLM, 4
    sd = W.make_llm_state_dict(cfg, seed)
    r, alpha = 16, 32.0
    wrapped = {"llama_model.base_model.model." + k: v for k, v in sd.items()}
    for i in range(cfg.num_hidden_layers):
        for n in ("q_proj", "v_proj"):
            p = f"llama_model.base_model.model.model.layers.{i}.self_attn.{n}."
            wrapped[p + "lora_A.default.weight"] = W._normal(p + "A", seed, (r, cfg.hidden_size), 0.05)
            wrapped[p + "lora_B.default.weight"] = W._normal(p + "B", seed, (cfg.hidden_size, r), 0.05)
    from hawkeye_amd.llm import strip_prefix
    merged = merge_lora(strip_prefix(wrapped), alpha)
    k = "model.layers.1.self_attn.v_proj.weight"
    p = "llama_model.base_model.model.model.layers.1.self_attn.v_proj."
    want = sd[k] + (alpha / r) * wrapped[p + "lora_B.default.weight"] @ wrapped[p + "lora_A.default.weight"]
    assert torch.allclose(merged[k], want, atol=1e-6) and not any("lora" in x for x in merged)
    m = LlamaPrefill(cfg, DEV).load_state_dict(wrapped, lora_alpha=alpha)
    m2 = LlamaPrefill(cfg, DEV).load_state_dict(merged)
    src, _ = W.make_prompt(cfg, 1, 64, 8, 3)
    vid = _rand((8, cfg.hidden_size), 5, 0.05)
    a = m.prefill(src=src.to(DEV), video=vid)["logits"].clone()
    b = m2.prefill(src=src.to(DEV), video=vid)["logits"]
    assert torch.equal(a, b)


def test_prefill_workspace_is_pooled_across_prompt_lengths():
    """prompt length differs per example, and every distinct (B, S) used to allocate a workspace of
    its own (qkv [32, M, 3D] alone is 2.2 GB at B = 8, S = 352 with the 7B shape).  Now the workspace is one pool of
    grow-only buffers: 12 distinct S after the first (largest) one grow the allocation by < 0.5 GB -- here by nothing --
    and a pooled plan gives the bits of a fresh engine that only ever saw that shape."""
    from hawkeye_amd.config import LlmConfig
    cfg = LlmConfig(num_hidden_layers=4)          # 7B widths (4096 / 11008 / 32 heads x 128), 4 layers
    m = LlamaPrefill(cfg, DEV).init_random(seed=3)
    B = 8
    g = torch.Generator(device=DEV).manual_seed(5)
    lengths = [352] + list(range(340, 220, -10))   # 13 distinct prompt lengths, the largest first
    outs, base = {}, None
    for S in lengths:
        src = torch.randint(0, cfg.vocab_size, (B, S), generator=g, device=DEV)
        o = m.prefill(src=src)
        outs[S] = (src, o["logits"].clone())
        torch.cuda.synchronize()
        if base is None:
            base = torch.cuda.memory_allocated()
    grown = torch.cuda.memory_allocated() - base - sum(v[1].numel() * 4 + v[0].numel() * 8 for v in list(outs.values())[1:])
    print(f"[pool] prefill memory growth over 12 more prompt lengths: {grown / 2**20:.1f} MiB; pooled {m.pooled_bytes() / 2**30:.2f} GiB; "
          f"plans {len(m.plans)}")
    assert grown < (1 << 29)
    assert len(m.plans) == len(lengths)
    fresh = LlamaPrefill(cfg, DEV).init_random(seed=3)
    for S in (lengths[4], lengths[0]):             # a smaller shape first, so the fresh engine's pool has to grow
        src, want = outs[S]
        assert torch.equal(fresh.prefill(src=src)["logits"], want), S
        assert torch.equal(m.prefill(src=src)["logits"], want), S      # revisited after the others used the same buffers


def test_bare_cuda_device_string_is_the_current_device():
    """INTEGRATION.md builds LlamaPrefill(device="cuda"): inputs on cuda:0 must be accepted."""
    pre = LlamaPrefill(SMALL_LLM, "cuda").load_state_dict(W.make_llm_state_dict(SMALL_LLM, 2))
    src, _ = W.make_prompt(SMALL_LLM, 1, 24, 8, 2, ragged=False)
    video = W._normal("video_tokens", 2, (8, SMALL_LLM.hidden_size), 0.05)
    out = pre.prefill(src=src.to("cuda:0"), video=video.to("cuda:0"))
    assert torch.isfinite(out["logits"]).all()
