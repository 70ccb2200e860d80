"""LLaMA / Vicuna prefill on MI355X: the consumer of encode_img's [B, 96, 4096] output.

Mirrors what the reference runs between `encode_img` and the first generated token:
embed_tokens_ on the prompt segments + splice of the video tokens (tasks/test.py:79-97, hawkeye_it.py:287-375),
then LlamaForCausalLM.forward with inputs_embeds (models/blip2/modeling_llama.py:482-611, 615-742): 32 pre-norm
decoder layers (RMSNorm, q/k/v projections, rotary embedding, causal attention with fp32 softmax, o_proj,
SwiGLU MLP), final RMSNorm, lm_head on the last position of every row, and the K/V cache generation continues from.

Numerics: bf16 MFMA operands, fp32 accumulation, fp32 residual stream, fp32 norm statistics and softmax (the
reference runs the LLM in fp16 with fp32 softmax / variance).  LoRA adapters (hawkeye_it.py:170-177, r=16,
alpha=32 on q_proj / v_proj) are merged into the dense weights at load: W + (alpha / r) B A.

There is no CPU path: everything below enqueues C-ABI calls of libhawkeye_hip.so (include/hawkeye_hip.h).
"""
import ctypes as C
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L
from . import ops
from .config import LlmConfig
from .engine import _Pool

LOG2E = 1.4426950408889634
_PREFIXES = ("llama_model.base_model.model.", "llama_model.", "base_model.model.", "")


def merge_lora(sd: Dict[str, torch.Tensor], lora_alpha: float = 32.0, r: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Fold PEFT LoRA pairs (`X.lora_A[.default].weight` [r, in], `X.lora_B[.default].weight` [out, r]) into
    `X.weight` and drop them: W' = W + (lora_alpha / r) * B @ A  (what PEFT's merged inference computes)."""
    out = dict(sd)
    for ka in [k for k in sd if ".lora_A." in k and k.endswith(".weight")]:
        kb = ka.replace(".lora_A.", ".lora_B.")
        base = ka.split(".lora_A.")[0] + ".weight"
        if kb not in sd or base not in sd:
            raise RuntimeError(f"merge_lora: {ka} has no matching lora_B / base weight")
        A, Bm = sd[ka].float(), sd[kb].float()
        rank = A.shape[0] if r is None else r
        out[base] = (sd[base].float() + (lora_alpha / rank) * (Bm @ A)).to(sd[base].dtype)
        del out[ka], out[kb]
    return out


def strip_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept HawkEye_it-level keys (`llama_model.base_model.model.model.layers...` under PEFT, hawkeye_it.py:176)
    as well as bare LlamaForCausalLM keys (`model.layers...`, `lm_head.weight`)."""
    for p in _PREFIXES:
        if any(k.startswith(p + "model.embed_tokens.") for k in sd):
            return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    raise RuntimeError("no LlamaForCausalLM weights (model.embed_tokens.weight) found in the state dict")


MAX_PLANS = 64      # host-side descriptor sets kept (LRU); device memory is pooled, not per plan


class _Plan:
    def __init__(self):
        self.ws: Dict[str, torch.Tensor] = {}
        self.steps: List[Tuple] = []
        self.keep: List = []
        self.pool_epoch = 0


class LlamaPrefill:
    """prefill(src | inputs_embeds, attention_mask, video) -> dict(hidden, logits, kv)."""

    def __init__(self, cfg: LlmConfig = LlmConfig(), device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("LlamaPrefill runs on an MI355X only (device must be cuda:N)")
        if self.device.index is None:       # "cuda" -> the current device, so that tensors on cuda:N compare equal to it
            self.device = torch.device("cuda", torch.cuda.current_device())
        if cfg.head_dim not in (64, 128):
            raise RuntimeError(f"head size {cfg.head_dim} not supported by the HIP attention kernels (64 or 128)")
        if cfg.hidden_size % 256 or cfg.intermediate_size % 64:
            raise RuntimeError("hidden_size must be a multiple of 256 and intermediate_size of 64")
        self.lib = L.load()
        self.w: Dict[str, torch.Tensor] = {}
        # gate | up projections with silu(gate) * up in the GEMM epilogue (HK_ACT_SWIGLU, interleaved weight rows) instead of
        # a [M, 2I] matrix and an elementwise pass over it.  HK_LLM_SWIGLU=0: the two-launch form (A/B aid); read at
        # construction because it decides the packed weight layout.
        self.fused_swiglu = os.environ.get("HK_LLM_SWIGLU", "1") != "0"
        # Prompt length differs per example, so an eval run meets a new (B, S) on almost every call.  Workspace tensors
        # are views of ONE pool of grow-only buffers (the engine's _Pool): the memory held is the largest size each
        # buffer was ever asked for (qkv alone is 2.2 GB at B = 8, S = 352), not the sum over shapes; plans are host-side
        # descriptor sets, kept LRU and rebuilt when a buffer had to grow.
        self.pool = _Pool(self.device)
        self.plans: "OrderedDict[Tuple[int, int, bool, bool], _Plan]" = OrderedDict()   # (B, S, masked, explicit position ids)
        self.last_ws: Dict[str, torch.Tensor] = {}
        # projection weights streamed from a 64 x 64-block copy (hk_gemm_desc.w_tiled): HK_LLM_TILED=0 = row-major (A/B aid)
        self.tiled = os.environ.get("HK_LLM_TILED", "1") != "0"
        self.w64: Dict[int, torch.Tensor] = {}

    def pooled_bytes(self) -> int:
        return self.pool.nbytes()

    def plan_for(self, B: int, S: int, masked: bool, with_pos: bool) -> "_Plan":
        key = (B, S, masked, with_pos)
        plan = self.plans.get(key)
        if plan is not None and plan.pool_epoch == self.pool.epoch:
            self.plans.move_to_end(key)
            return plan
        if plan is not None:
            del self.plans[key]
        epoch = self.pool.epoch
        plan = self._build_plan(B, S, masked, with_pos)
        if self.pool.epoch != epoch:
            self.plans.clear()              # a buffer grew: every older plan points into the released allocation
        plan.pool_epoch = self.pool.epoch
        self.plans[key] = plan
        while len(self.plans) > MAX_PLANS:
            self.plans.popitem(last=False)
        return plan

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor], lora_alpha: float = 32.0):
        cfg, dev = self.cfg, self.device
        sd = merge_lora(strip_prefix(sd), lora_alpha)
        w: Dict[str, torch.Tensor] = {}

        def bf(t):
            return ops.cast_bf16(t.detach().to(device=dev, dtype=torch.float32).contiguous())

        def f32(t):
            return t.detach().to(device=dev, dtype=torch.float32).contiguous()

        with torch.cuda.device(dev):
            w["embed"] = bf(sd["model.embed_tokens.weight"])
            for i in range(cfg.num_hidden_layers):
                p, o = f"model.layers.{i}.", f"l{i}."
                w[o + "qkv"] = bf(torch.cat([sd[p + f"self_attn.{n}_proj.weight"].detach().float() for n in "qkv"]))
                w[o + "o"] = bf(sd[p + "self_attn.o_proj.weight"])
                gate, up = sd[p + "mlp.gate_proj.weight"].detach().float(), sd[p + "mlp.up_proj.weight"].detach().float()
                # fused SwiGLU epilogue (HK_ACT_SWIGLU): rows interleaved (gate_0, up_0, gate_1, up_1, ...); else gate | up
                w[o + "gu"] = bf(torch.stack([gate, up], 1).reshape(2 * gate.shape[0], gate.shape[1]) if self.fused_swiglu
                                 else torch.cat([gate, up]))
                w[o + "down"] = bf(sd[p + "mlp.down_proj.weight"])
                w[o + "n1"], w[o + "n2"] = f32(sd[p + "input_layernorm.weight"]), f32(sd[p + "post_attention_layernorm.weight"])
            w["norm"] = f32(sd["model.norm.weight"])
            w["lm_head"] = bf(sd["lm_head.weight"])
            # rotary tables exactly as LlamaRotaryEmbedding builds them (modeling_llama.py:79-89), first half
            hd = cfg.head_dim
            inv_freq = 1.0 / (cfg.rope_base ** (torch.arange(0, hd, 2).float() / hd))
            freqs = torch.einsum("i,j->ij", torch.arange(cfg.max_position_embeddings, dtype=inv_freq.dtype), inv_freq)
            w["cos"], w["sin"] = freqs.cos().to(dev).contiguous(), freqs.sin().to(dev).contiguous()
        self.w, self.w64 = w, {}
        self.plans.clear()
        return self

    def init_random(self, seed: int = 0, std: float = 0.02):
        """Random weights of this architecture generated on the device (bench / smoke aid: a 7B state dict takes
        minutes to draw on the host).  Same packed layout as load_state_dict produces."""
        cfg, dev = self.cfg, self.device
        D, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
        g = torch.Generator(device=dev).manual_seed(seed)
        w: Dict[str, torch.Tensor] = {}

        def rnd(*shape):
            return (torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * std).to(torch.bfloat16)

        with torch.cuda.device(dev):
            w["embed"] = rnd(V, D)
            for i in range(cfg.num_hidden_layers):
                o = f"l{i}."
                w[o + "qkv"], w[o + "o"], w[o + "gu"], w[o + "down"] = rnd(3 * D, D), rnd(D, D), rnd(2 * I, D), rnd(D, I)
                w[o + "n1"] = torch.ones(D, device=dev)
                w[o + "n2"] = torch.ones(D, device=dev)
            w["norm"] = torch.ones(D, device=dev)
            w["lm_head"] = rnd(V, D)
            hd = cfg.head_dim
            inv_freq = 1.0 / (cfg.rope_base ** (torch.arange(0, hd, 2).float() / hd))
            freqs = torch.einsum("i,j->ij", torch.arange(cfg.max_position_embeddings, dtype=inv_freq.dtype), inv_freq)
            w["cos"], w["sin"] = freqs.cos().to(dev).contiguous(), freqs.sin().to(dev).contiguous()
        self.w, self.w64 = w, {}
        self.plans.clear()
        return self

    # ------------------------------------------------------------------ plan
    @staticmethod
    def split_k_for(M: int, N: int, K: int) -> int:
        """Split-K factor of an o_proj / down_proj launch: 1 unless its 256x128 tiles fill well under the chip (M <= ~1000
        rows); then the largest of 4 / 2 whose slices make at most one workgroup per CU and divide the K-tiles.  Isolated,
        M = 352 (tools/skinny_split.py): o_proj 37.1 -> 28.4 us, down_proj 81.1 -> 52.0 us at 4 slices; M = 704: 54.3 -> 43.7,
        121.7 -> 92.1 at 2.  HK_LLM_SPLIT_K=0: off."""
        if os.environ.get("HK_LLM_SPLIT_K", "1") == "0":
            return 1
        tmid = -(-M // 256) * -(-N // 128)
        kt = K // 64
        for s in (4, 2):
            if tmid * s <= 256 and kt % s == 0 and kt // s >= 16:
                return s
        return 1

    def _tiled(self, W: torch.Tensor):
        """(tensor to stream, w_tiled flag) for a packed projection weight: its 64 x 64-block copy when HK_LLM_TILED is on
        (hk_gemm_desc.w_tiled; made on first use, kept beside the row-major tensor the decode tiling and tests read)."""
        if not self.tiled or W.shape[0] % 64 or W.shape[1] % 64:
            return W, False
        key = W.data_ptr()
        t = self.w64.get(key)
        if t is None:
            with torch.cuda.device(self.device):
                t = self.w64[key] = ops.tile_gemm_weight(W)
        return t, True

    def _gemm(self, plan, A, W, Cout, M, N, K, **kw):
        W, tiled = self._tiled(W) if M > 16 else (W, False)       # lm_head on B rows stays row-major
        d = ops.gemm_desc(A, W.data_ptr(), Cout, M, N, K, a_rows=(K, 0, 0), c_rows=(N, 0, 0), w_tiled=tiled, **kw)
        plan.keep.append(d)
        plan.steps.append((self.lib.hk_gemm_bf16, (C.byref(d),), "hk_gemm_bf16"))

    def _build_plan(self, B: int, S: int, masked: bool, with_pos: bool = False) -> _Plan:
        cfg, dev, w = self.cfg, self.device, self.w
        D, I, H, hd, Lyr, V = (cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim,
                               cfg.num_hidden_layers, cfg.vocab_size)
        M = B * S
        plan = _Plan()
        bf16, f32 = torch.bfloat16, torch.float32
        with torch.cuda.device(dev):
            ws = plan.ws
            buf = self.pool.view
            ws["src"] = buf("src", (M,), torch.int64)
            ws["mask"] = buf("mask", (B, S), torch.int64)
            ws["kbias"] = buf("kbias", (B, S), f32)
            ws["x"] = buf("x", (M, D), f32)           # residual stream
            ws["xn"] = buf("xn", (M, D), bf16)
            ws["qkv"] = buf("qkv", (Lyr, M, 3 * D), bf16)  # per layer: q | k | v; k, v = the KV cache
            ws["att"] = buf("att", (M, D), bf16)
            ws["gu"] = None if self.fused_swiglu else buf("gu", (M, 2 * I), bf16)
            ws["act"] = buf("act", (M, I), bf16)
            ws["last"] = buf("last", (B,), torch.int64)
            ws["pos"] = buf("pos", (M,), torch.int32) if with_pos else None
            ws["hl_f"] = buf("hl_f", (B, D), f32)
            ws["hl_b"] = buf("hl_b", (B, D), bf16)
            ws["logits"] = buf("logits", (B, V), f32)
            # skinny prompts (B = 1 eval calls: M = 352 is 64 workgroups walking 64-172 K-tiles): o_proj / down_proj split over K,
            # fp32 slices summed by the norm that follows (hk_rmsnorm_add_parts)
            sk_o, sk_d = self.split_k_for(M, D, D), self.split_k_for(M, D, I)
            ws["parts"] = buf("parts", (max(sk_o, sk_d), M, D), f32) if max(sk_o, sk_d) > 1 else None
        P = lambda t: t.data_ptr()  # noqa: E731
        x, xn, att, gu, act = ws["x"], ws["xn"], ws["att"], ws["gu"], ws["act"]
        qscale = (hd ** -0.5) * LOG2E
        eb = 2
        def norm(gamma, sk):
            """RMSNorm of the residual stream -> xn; after a split-K GEMM it first sums the slices into the stream."""
            if sk > 1:
                plan.steps.append((self.lib.hk_rmsnorm_add_parts, (P(ws["parts"]), sk, M * D, P(x), P(gamma), P(xn), M, D,
                                                                   cfg.rms_norm_eps), "hk_rmsnorm_add_parts"))
            else:
                plan.steps.append((self.lib.hk_rmsnorm, (P(x), L.HK_DT_F32, P(gamma), P(xn), M, D, cfg.rms_norm_eps), "hk_rmsnorm"))

        def residual_gemm(A, W, K, sk):
            if sk > 1:
                self._gemm(plan, A, W, P(ws["parts"]), M, D, K, residual=P(x), r_rows=(D, 0, 0), out_f32=True, split_k=sk,
                           split_stride=M * D)
            else:
                self._gemm(plan, A, W, P(x), M, D, K, residual=P(x), r_rows=(D, 0, 0), out_f32=True)

        for i in range(Lyr):
            o = f"l{i}."
            qkv = ws["qkv"][i]
            norm(w[o + "n1"], sk_d if i > 0 else 1)
            self._gemm(plan, P(xn), w[o + "qkv"], P(qkv), M, 3 * D, D)
            plan.steps.append((self.lib.hk_rope_bf16, (P(qkv), P(w["cos"]), P(w["sin"]), P(ws["pos"]) if with_pos else None, M, H, hd, S,
                                                       cfg.max_position_embeddings, qscale), "hk_rope_bf16"))
            ad = ops.attn_desc(P(qkv), P(qkv) + D * eb, P(qkv) + 2 * D * eb, P(att), B, H, S, S,
                               3 * D, 3 * D, 3 * D, D, S * 3 * D, S * 3 * D, S * 3 * D, S * D,
                               key_bias=P(ws["kbias"]) if masked else None, head_dim=hd, causal=True)
            plan.keep.append(ad)
            plan.steps.append((self.lib.hk_flash_attn_bf16, (C.byref(ad),), "hk_flash_attn_bf16"))
            residual_gemm(P(att), w[o + "o"], D, sk_o)
            norm(w[o + "n2"], sk_o)
            if self.fused_swiglu:
                # gate | up projections and silu(gate) * up in one launch: the [M, 2I] matrix is never written
                wgu, tl = self._tiled(w[o + "gu"])
                d = ops.gemm_desc(P(xn), wgu.data_ptr(), P(act), M, 2 * I, D, a_rows=(D, 0, 0), c_rows=(I, 0, 0),
                                  act=L.HK_ACT_SWIGLU, w_tiled=tl)
                plan.keep.append(d)
                plan.steps.append((self.lib.hk_gemm_bf16, (C.byref(d),), "hk_gemm_bf16"))
            else:
                self._gemm(plan, P(xn), w[o + "gu"], P(gu), M, 2 * I, D)
                plan.steps.append((self.lib.hk_silu_mul_bf16, (P(gu), P(act), M, I), "hk_silu_mul_bf16"))
            residual_gemm(P(act), w[o + "down"], I, sk_d)
        norm(w["norm"], sk_d)
        # last valid position of every row -> lm_head (LlamaForCausalLM.forward :688-700; generation reads logits[:, -1])
        plan.steps.append((self.lib.hk_embed_splice, (P(ws["last"]), P(xn), None, L.HK_DT_BF16, P(ws["hl_f"]), B, D),
                           "hk_embed_splice"))
        plan.steps.append((self.lib.hk_cast_f32_bf16, (P(ws["hl_f"]), P(ws["hl_b"]), B * D), "hk_cast_f32_bf16"))
        self._gemm(plan, P(ws["hl_b"]), w["lm_head"], P(ws["logits"]), B, V, D, out_f32=True)
        return plan

    # ------------------------------------------------------------------ run
    @torch.no_grad()
    def prefill(self, src: Optional[torch.Tensor] = None, video: Optional[torch.Tensor] = None,
                inputs_embeds: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                position_ids=None):
        """Either `src` int64 [B,S] (token id >= 0, or -(video row)-1 for a spliced video token) with `video`
        [B*n, D] bf16/fp32 (encode_img's output), or ready `inputs_embeds` [B,S,D].  attention_mask [B,S] (1 =
        token, 0 = padding on either side) or None; logits are read at the LAST 1 of every row.
        position_ids: None = 0..S-1 for every row, what LlamaModel.forward does when none are passed
        (modeling_llama.py:526-533: the training forward, hawkeye_it.py:369-375, right padding); "generate" = the ids
        `generate` builds for a (left-)padded batch, cumsum(mask) - 1 with padded positions at 1
        (prepare_inputs_for_generation :751-754); or an explicit integer tensor [B,S].  A left-padded mask with the
        default ids is refused: the reference never runs that combination and its positions would be shifted.
        Returns {"hidden": bf16 [B,S,D] after the final RMSNorm,
        "logits": fp32 [B,V] at each row's last valid position, "kv": [(k, v)] per layer as [B,S,H,hd] bf16 views
        (k post-rotary)}.  Every returned tensor is a view of the pooled workspace, valid until the NEXT prefill call of this
        object (whatever its shape): clone what generation keeps.  Padded positions hold unspecified values."""
        cfg = self.cfg
        if not self.w:
            raise RuntimeError("LlamaPrefill: load_state_dict() has not been called")
        if (src is None) == (inputs_embeds is None):
            raise RuntimeError("prefill: pass exactly one of src / inputs_embeds")
        ref = src if src is not None else inputs_embeds
        B, S = int(ref.shape[0]), int(ref.shape[1])
        if S > cfg.max_position_embeddings:
            raise RuntimeError(f"sequence length {S} exceeds max_position_embeddings {cfg.max_position_embeddings}")
        if ref.device != self.device:
            raise RuntimeError(f"inputs are on {ref.device}, model on {self.device}")
        masked = attention_mask is not None
        with_pos = position_ids is not None
        if isinstance(position_ids, str):
            if position_ids != "generate" or not masked:
                raise RuntimeError('position_ids must be None, "generate" (needs attention_mask) or an int tensor [B,S]')
        elif with_pos and (tuple(position_ids.shape) != (B, S) or position_ids.device != self.device):
            raise RuntimeError(f"position_ids must be an integer tensor [{B},{S}] on {self.device}")
        plan = self.plan_for(B, S, masked, with_pos)
        ws = plan.ws
        D, H, hd = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            if masked:
                if attention_mask.shape != (B, S):
                    raise RuntimeError(f"attention_mask must be [{B},{S}]")
                ws["mask"].copy_(attention_mask, non_blocking=True)
                L.check(self.lib.hk_qformer_key_bias(ws["mask"].data_ptr(), ws["kbias"].data_ptr(), B, 0, S, stream),
                        "hk_qformer_key_bias")
                # index of the last valid token per row = position of the last 1 (right OR left padding); index
                # bookkeeping over a [B,S] int mask, like the reference's own cumsum
                m = ws["mask"]
                if not with_pos and bool((m[:, 0] == 0).any()):
                    raise RuntimeError('prefill: the mask is left-padded; pass position_ids="generate" (what generate() '
                                       "uses, modeling_llama.py:751-754) or explicit position_ids")
                last = S - 1 - m.flip(1).argmax(1) + torch.arange(B, device=self.device) * S
                ws["last"].copy_(last)
                if with_pos:
                    if isinstance(position_ids, str):
                        pos = (m.cumsum(1) - 1).masked_fill_(m == 0, 1)
                    else:
                        pos = position_ids
                    if int(pos.min()) < 0 or int(pos.max()) >= cfg.max_position_embeddings:
                        raise RuntimeError("position_ids outside [0, max_position_embeddings)")
                    ws["pos"].copy_(pos.reshape(-1))
            else:
                ws["last"].copy_(torch.arange(B, device=self.device) * S + (S - 1))
                if with_pos:
                    if int(position_ids.min()) < 0 or int(position_ids.max()) >= cfg.max_position_embeddings:
                        raise RuntimeError("position_ids outside [0, max_position_embeddings)")
                    ws["pos"].copy_(position_ids.reshape(-1))
            if src is not None:
                ws["src"].copy_(src.reshape(-1), non_blocking=True)
                vid = None
                if video is not None:
                    if video.device != self.device or video.dtype not in (torch.float32, torch.bfloat16):
                        raise RuntimeError("video must be an fp32/bf16 tensor on the model's GPU")
                    vid = video.contiguous().reshape(-1, D)
                if int(ws["src"].min()) < -(0 if vid is None else vid.shape[0]) or int(ws["src"].max()) >= cfg.vocab_size:
                    raise RuntimeError("prefill: src index outside the vocabulary / video tensor")
                L.check(self.lib.hk_embed_splice(ws["src"].data_ptr(), self.w["embed"].data_ptr(),
                                                 None if vid is None else vid.data_ptr(),
                                                 L.HK_DT_F32 if (vid is not None and vid.dtype == torch.float32) else L.HK_DT_BF16,
                                                 ws["x"].data_ptr(), B * S, D, stream), "hk_embed_splice")
            else:
                if inputs_embeds.shape != (B, S, D):
                    raise RuntimeError(f"inputs_embeds must be [{B},{S},{D}]")
                ws["x"].view(B, S, D).copy_(inputs_embeds, non_blocking=True)
            for fn, args, name in plan.steps:
                rc = fn(*args, stream)
                if rc != 0:
                    L.check(rc, name)
        self.last_ws = ws       # the generator (hawkeye_amd/decode.py) copies K / V of every layer out of it
        kv = [(ws["qkv"][i].view(B, S, 3, H, hd)[:, :, 1], ws["qkv"][i].view(B, S, 3, H, hd)[:, :, 2])
              for i in range(cfg.num_hidden_layers)]
        return {"hidden": ws["xn"].view(B, S, D), "logits": ws["logits"], "kv": kv}


def prefill_flops(cfg: LlmConfig, B: int, S: int) -> float:
    """Algorithmic FLOPs of one prefill (2 flops per MAC): projections + SwiGLU MLP for every token, causal attention
    counted on the lower triangle, lm_head on one position per row; norms / rotary / softmax excluded."""
    D, I, Lyr = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    per_tok = 2.0 * (4 * D * D + 3 * D * I)
    attn = 4.0 * D * S * (S + 1) / 2.0
    return Lyr * (B * S * per_tok + B * attn) + 2.0 * B * D * cfg.vocab_size
