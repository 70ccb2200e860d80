"""Greedy generation after the prefill (continued): what `llama_model.generate` does in the
reference's eval scripts (tasks/test.py:111-145, 426-470; test_recursive_grounding.py:112-145, 199-245 -- every call
site passes do_sample=False, num_beams=1), on the K/V cache the prefill leaves.

One step = one token per sequence through the 32 decoder layers (models/blip2/modeling_llama.py:176-231 with
past_key_value, :268-307, :482-611): RMSNorm, fused q|k|v projection, rotary embedding at the new position + append to the
cache, one-row attention over the cache, o_proj into the fp32 residual stream, RMSNorm, gate|up with SwiGLU, down_proj into
the stream; then the final norm, lm_head and the arg-max.  Positions follow prepare_inputs_for_generation (:744-773):
cumsum(attention_mask) - 1, so a left-padded row continues at its own length while every row's new key lands in the same
cache row.  With 1-16 rows each projection streams its weights once (13.2 GB per token for Vicuna-7B): the kernels
(csrc/hk_decode.hip) are HBM-bound, not MFMA-bound.

There is no CPU path: everything enqueues C-ABI calls of libhawkeye_hip.so.
"""
import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import lib as L
from . import ops
from .engine import _Pool
from .llm import LOG2E, LlamaPrefill

MAX_ROWS = 16          # sequences per step hk_gemv_bf16 takes
ROW_GRANULE = 64       # cache rows are allocated in multiples of this, so nearby prompt lengths share one plan

# how a step's run-time integers enter the argument tuple of a planned call
_STATIC, _ROPE, _ATTN, _FUSED = 0, 1, 2, 3


class _DecodePlan:
    def __init__(self):
        self.ws: Dict[str, torch.Tensor] = {}
        self.steps: List[Tuple] = []       # (fn, args list, kind, name)
        self.keep: List = []
        self.pool_epoch = 0


class LlamaGenerator:
    """generate(src | inputs_embeds, video, attention_mask, max_new_tokens, ...) -> {"tokens": int64 [B, T], ...}.

    Built on a loaded LlamaPrefill (shares its packed weights).  Batches are unpadded or LEFT-padded (what `generate`
    supports for decoder-only models); B <= 16."""

    def __init__(self, prefill: LlamaPrefill):
        if not prefill.w:
            raise RuntimeError("LlamaGenerator: the LlamaPrefill has no weights loaded")
        if not prefill.fused_swiglu:
            raise RuntimeError("LlamaGenerator needs the interleaved gate|up weight layout (HK_LLM_SWIGLU=0 is a prefill A/B aid)")
        self.pre, self.cfg, self.device, self.lib = prefill, prefill.cfg, prefill.device, prefill.lib
        self.pool = _Pool(self.device)
        # HK_DECODE_FUSED_ATTN=0: rotary / cache append and the attention as two launches (A/B aid; same results)
        self.fused_attn = os.environ.get("HK_DECODE_FUSED_ATTN", "1") != "0"
        # HK_DECODE_FUSED_NORM=0: every RMSNorm as its own launch (A/B aid)
        self.fused_norm = os.environ.get("HK_DECODE_FUSED_NORM", "1") != "0"
        # HK_DECODE_KEY_SPLIT=0: one workgroup per (sequence, head) whatever the cache length (A/B aid)
        self.key_split = os.environ.get("HK_DECODE_KEY_SPLIT", "1") != "0"
        # Decode-only copy of the projection weights in hk_gemv_desc.w_tiled's block layout: a wave streams one contiguous
        # region instead of 16 rows 8-22 KB apart (cold-weight sweep, row-major -> tiled: gate|up 39.5 -> 31.1 us = 5.8 TB/s,
        # down_proj 21.0 -> 16.8, qkv 21.9 -> 18.9, lm_head 55 -> 43 = 6.1 TB/s).  Costs the weights' size again (13.5 GB for
        # Vicuna-7B, of 288 GB); HK_DECODE_TILED=0 streams the prefill's row-major weights instead.
        self.tiled = os.environ.get("HK_DECODE_TILED", "1") != "0"
        self.wt: Dict[str, torch.Tensor] = {}
        if self.tiled:
            with torch.cuda.device(self.device):
                for k, v in prefill.w.items():
                    if k == "lm_head" or k.endswith((".qkv", ".o", ".gu", ".down")):
                        self.wt[k] = ops.tile_gemv_weight(v)
        self.plans: Dict[Tuple[int, int, bool], _DecodePlan] = {}
        self.k_cache: Optional[torch.Tensor] = None     # [layers, B, rows, D] views of the last call (inspection / tests)
        self.v_cache: Optional[torch.Tensor] = None

    def pooled_bytes(self) -> int:
        return self.pool.nbytes()

    # ------------------------------------------------------------------ plan
    def _plan_for(self, B: int, rows: int, masked: bool) -> _DecodePlan:
        key = (B, rows, masked)
        plan = self.plans.get(key)
        if plan is not None and plan.pool_epoch == self.pool.epoch:
            return plan
        epoch = self.pool.epoch
        plan = self._build_plan(B, rows, masked)
        if self.pool.epoch != epoch:
            self.plans.clear()
        plan.pool_epoch = self.pool.epoch
        self.plans[key] = plan
        return plan

    def _build_plan(self, B: int, rows: int, masked: bool) -> _DecodePlan:
        cfg, w = self.cfg, self.pre.w
        D, I, H, hd, Lyr, V = (cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim,
                               cfg.num_hidden_layers, cfg.vocab_size)
        plan = _DecodePlan()
        bf16, f32 = torch.bfloat16, torch.float32
        with torch.cuda.device(self.device):
            buf, ws = self.pool.view, plan.ws
            ws["kc"] = buf("kc", (Lyr, B, rows, D), bf16)
            ws["vc"] = buf("vc", (Lyr, B, rows, D), bf16)
            ws["kbias"] = buf("kbias", (B, rows), f32)
            ws["pos0"] = buf("pos0", (B,), torch.int32)
            ws["tok"] = buf("tok", (B,), torch.int64)
            ws["x"] = buf("x", (B, D), f32)
            ws["xn"] = buf("xn", (B, D), bf16)
            ws["qkv"] = buf("qkv", (B, 3 * D), bf16)
            ws["q"] = buf("q", (B, D), bf16)
            ws["att"] = buf("att", (B, D), bf16)
            ws["act"] = buf("act", (B, I), bf16)
            ws["logits"] = buf("logits", (B, (V + 3) // 4 * 4), f32)[:, :V]      # rows 16-byte aligned whatever V is
        P = lambda t: t.data_ptr()  # noqa: E731
        x, xn, qkv, q, att, act = (P(ws[k]) for k in ("x", "xn", "qkv", "q", "att", "act"))
        qscale = (hd ** -0.5) * LOG2E
        steps, lib = plan.steps, self.lib
        # long caches with few (sequence, head) pairs: the keys of one head go to several workgroups + a merge launch
        # (hk_decode_attn_*'s key_splits; pays from ~1000 keys on: a 1500-token prompt 3.7 -> see profiles/r03_decode.jsonl)
        ksplit = min(8, rows // 512) if (self.key_split and B * H <= 512) else 1
        part = P(self.pool.view("attn_part", (B * H * max(ksplit, 1) * (hd + 2),), f32)) if ksplit > 1 else None

        def gemv(xp, name, out, N, K, **kw):
            W = self.wt[name] if self.tiled else w[name]
            d = ops.gemv_desc(xp, W.data_ptr(), out, B, N, K, w_tiled=self.tiled, **kw)
            plan.keep.append(d)
            steps.append((lib.hk_gemv_bf16, [C.byref(d)], _STATIC, "hk_gemv_bf16"))

        # RMSNorm fused into the projection that consumes it (hk_gemv_desc.norm_weight, up to 4 sequences): the stream's fp32 rows
        # are the operand; otherwise a hk_rmsnorm launch writes xn first
        fuse_norm = self.fused_norm and B <= 4 and D <= 4096

        def normed_gemv(gamma, W, out, N, **kw):
            if fuse_norm:
                gemv(x, W, out, N, D, norm_weight=P(gamma), norm_eps=cfg.rms_norm_eps, **kw)
            else:
                steps.append((lib.hk_rmsnorm, [x, L.HK_DT_F32, P(gamma), xn, B, D, cfg.rms_norm_eps], _STATIC, "hk_rmsnorm"))
                gemv(xn, W, out, N, D, **kw)

        # the token chosen by the previous step (or by the prefill) -> its embedding = the new residual stream
        steps.append((lib.hk_embed_splice, [P(ws["tok"]), P(w["embed"]), None, L.HK_DT_BF16, x, B, D], _STATIC, "hk_embed_splice"))
        for i in range(Lyr):
            o = f"l{i}."
            kc, vc = P(ws["kc"][i]), P(ws["vc"][i])
            normed_gemv(w[o + "n1"], o + "qkv", qkv, 3 * D)
            if self.fused_attn:
                # rotary + cache append + attention in one launch; args 4 (pos_add) and 12 (slot) are set per step
                steps.append((lib.hk_decode_attn_rope_bf16, [qkv, P(w["cos"]), P(w["sin"]), P(ws["pos0"]), 0, kc, vc,
                                                             P(ws["kbias"]) if masked else None, att, B, H, hd, 0, rows, rows,
                                                             cfg.max_position_embeddings, qscale, ksplit, part], _FUSED,
                              "hk_decode_attn_rope_bf16"))
            else:
                # args 4 (pos_add) and 11 (slot) are set per step
                steps.append((lib.hk_decode_rope_append, [qkv, P(w["cos"]), P(w["sin"]), P(ws["pos0"]), 0, q, kc, vc, B, H, hd, 0, rows,
                                                          cfg.max_position_embeddings, qscale], _ROPE, "hk_decode_rope_append"))
                # arg 8 (n_keys) is set per step
                steps.append((lib.hk_decode_attn_bf16, [q, kc, vc, P(ws["kbias"]) if masked else None, att, B, H, hd, 0, rows, rows,
                                                        ksplit, part], _ATTN, "hk_decode_attn_bf16"))
            gemv(att, o + "o", x, D, D, residual=x, out_f32=True)
            normed_gemv(w[o + "n2"], o + "gu", act, 2 * I, act="swiglu")
            gemv(act, o + "down", x, D, I, residual=x, out_f32=True)
        normed_gemv(w["norm"], "lm_head", P(ws["logits"]), V, out_f32=True, ldo=ws["logits"].stride(0))
        steps.append((lib.hk_argmax_f32, [P(ws["logits"]), P(ws["tok"]), B, V, ws["logits"].stride(0)], _STATIC, "hk_argmax_f32"))
        return plan

    def _step(self, plan: _DecodePlan, t: int, S: int, stream: int):
        """Decode step t: consumes token t (ws['tok']) at cache row S + t, leaves logits and token t + 1."""
        for fn, args, kind, name in plan.steps:
            if kind == _ROPE:
                args[4], args[11] = t, S + t
            elif kind == _ATTN:
                args[8] = S + t + 1
            elif kind == _FUSED:
                args[4], args[12] = t, S + t
            rc = fn(*args, stream)
            if rc != 0:
                L.check(rc, name)

    # ------------------------------------------------------------------ run
    @torch.no_grad()
    def generate(self, src: Optional[torch.Tensor] = None, video: Optional[torch.Tensor] = None,
                 inputs_embeds: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                 max_new_tokens: int = 20, stop_sequences: Optional[Sequence[Sequence[int]]] = None,
                 eos_token_id: Optional[int] = None, pad_token_id: int = 0, forced_tokens: Optional[torch.Tensor] = None,
                 next_token_fn=None,
                 return_logits: bool = False, sync_every: int = 1):
        """Prompt arguments as LlamaPrefill.prefill (a mask, if given, must be all ones or LEFT-padded).  Greedy decoding:
        token 0 is the arg-max of the prefill's last-position logits, token t + 1 of decode step t's.
        stop_sequences: generation ends once the generated tokens of ROW 0 end with one of them (StoppingCriteriaSub,
        tasks/test.py:100-109 -- the reference's criterion looks at input_ids[0] only); the stop tokens stay in the output,
        as in the reference, whose callers cut the text at '###' afterwards (:140-141).
        eos_token_id: a row that emits it is finished and gets pad_token_id from then on; generation ends when every row
        is finished (generate's unfinished_sequences bookkeeping).
        forced_tokens int64 [B, >= max_new_tokens]: feed these instead of the arg-max (teacher forcing: parity tests).
        next_token_fn(logits fp32 [B, V], tokens_so_far int64 [B, t]) -> int64 [B] on the device: replaces the arg-max, e.g.
        the reference's do_sample=True path (answer()'s default, tasks/test.py:111-112: temperature, top_p, repetition
        penalty as transformers' LogitsProcessors + torch.multinomial), which stays PyTorch code of the caller's; the
        forward of every step is still the HIP path.
        sync_every: the host looks at the new tokens every this many steps (1 = the reference's per-token check; larger
        values decode up to sync_every - 1 tokens past a stop and cut them off -- same output, fewer host round trips).
        Returns {"tokens": int64 [B, T] (T <= max_new_tokens), "logits": fp32 [T, B, V] if return_logits}."""
        cfg = self.cfg
        if max_new_tokens < 1:
            raise RuntimeError("generate: max_new_tokens must be >= 1")
        ref = src if src is not None else inputs_embeds
        if ref is None:
            raise RuntimeError("generate: pass src or inputs_embeds")
        B, S = int(ref.shape[0]), int(ref.shape[1])
        if B > MAX_ROWS:
            raise RuntimeError(f"generate: {B} sequences; the decode step takes at most {MAX_ROWS}")
        if S + max_new_tokens > cfg.max_position_embeddings:
            raise RuntimeError(f"generate: prompt {S} + {max_new_tokens} new tokens exceeds max_position_embeddings "
                               f"{cfg.max_position_embeddings}")
        masked = attention_mask is not None
        if masked:
            m = attention_mask.to(torch.int64)
            if tuple(m.shape) != (B, S) or bool((m[:, -1] == 0).any()) or bool((m[:, 1:] < m[:, :-1]).any()):
                raise RuntimeError("generate: attention_mask must be [B, S], all ones or LEFT-padded (zeros first)")
        if forced_tokens is not None and (forced_tokens.dtype != torch.int64 or forced_tokens.shape[0] != B
                                          or forced_tokens.shape[1] < max_new_tokens or not forced_tokens.is_cuda):
            raise RuntimeError(f"generate: forced_tokens must be int64 [{B}, >= {max_new_tokens}] on {self.device}")
        out = self.pre.prefill(src=src, video=video, inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                               position_ids="generate" if masked else None)
        rows = -(-(S + max_new_tokens) // ROW_GRANULE) * ROW_GRANULE
        plan = self._plan_for(B, rows, masked)
        ws = plan.ws
        D, Lyr, V = cfg.hidden_size, cfg.num_hidden_layers, cfg.vocab_size
        stops = [list(map(int, s)) for s in (stop_sequences or []) if len(s)]
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            # the prefill's K (post-rotary) / V of every layer -> cache rows [0, S): one strided copy each
            pws = self.pre.last_ws
            qkv_all = pws["qkv"].view(Lyr, B, S, 3, D)
            ws["kc"][:, :, :S].copy_(qkv_all[:, :, :, 1])
            ws["vc"][:, :, :S].copy_(qkv_all[:, :, :, 2])
            if masked:
                ws["kbias"].zero_()
                ws["kbias"][:, :S].copy_(pws["kbias"])
                ws["pos0"].copy_(attention_mask.sum(1).to(torch.int32))     # position of new token t = n_valid + t
            else:
                ws["pos0"].fill_(S)
            logits_all = torch.empty((max_new_tokens, B, V), dtype=torch.float32, device=self.device) if return_logits else None
            toks = torch.empty((B, max_new_tokens), dtype=torch.int64, device=self.device)
            L.check(self.lib.hk_argmax_f32(out["logits"].data_ptr(), ws["tok"].data_ptr(), B, V, V, stream), "hk_argmax_f32")
            if return_logits:
                logits_all[0].copy_(out["logits"])
            finished = [False] * B
            n_out, checked = 0, 0
            for t in range(max_new_tokens):
                if forced_tokens is not None:
                    ws["tok"].copy_(forced_tokens[:, t])
                elif next_token_fn is not None:
                    nxt = next_token_fn(out["logits"] if t == 0 else ws["logits"], toks[:, :t])
                    if nxt.dtype != torch.int64 or tuple(nxt.shape) != (B,) or not nxt.is_cuda:
                        raise RuntimeError(f"generate: next_token_fn must return int64 [{B}] on the device")
                    ws["tok"].copy_(nxt)
                toks[:, t].copy_(ws["tok"])
                n_out = t + 1
                if (stops or eos_token_id is not None) and (n_out - checked >= sync_every or n_out == max_new_tokens):
                    host = toks[:, :n_out].cpu()                          # synchronises with the stream
                    cut = self._first_stop(host, checked, stops, eos_token_id, finished)
                    checked = n_out
                    if cut is not None:
                        n_out = cut
                        break
                if t + 1 == max_new_tokens:
                    break
                self._step(plan, t, S, stream)
                if return_logits:
                    logits_all[t + 1].copy_(ws["logits"])
            tokens = toks[:, :n_out].clone()
            if eos_token_id is not None:
                # rows that finished earlier carry pad_token_id after their eos (generate :  next_tokens * unfinished + pad * (1 - unfinished))
                host = tokens.cpu()
                for b in range(B):
                    hit = (host[b] == eos_token_id).nonzero()
                    if len(hit):
                        host[b, int(hit[0]) + 1:] = pad_token_id
                tokens = host.to(self.device)
        self.k_cache, self.v_cache = ws["kc"], ws["vc"]
        res = {"tokens": tokens}
        if return_logits:
            res["logits"] = logits_all[:n_out]
        return res

    @staticmethod
    def _first_stop(host: torch.Tensor, checked: int, stops: List[List[int]], eos: Optional[int], finished: List[bool]):
        """Length at which generation ends, looking at tokens [checked, n) in order; None = go on."""
        B, n = host.shape
        for length in range(checked + 1, n + 1):
            if eos is not None:
                for b in range(B):
                    if not finished[b] and int(host[b, length - 1]) == eos:
                        finished[b] = True
                if all(finished):
                    return length
            row0 = host[0, :length].tolist()
            for s in stops:
                if length >= len(s) and row0[-len(s):] == s:
                    return length
        return None


def decode_bytes_per_token(cfg, B: int, n_keys: int) -> float:
    """Algorithmic HBM bytes of one decode step: every projection weight and lm_head once (bf16), the K/V cache rows read
    (bf16, B sequences), the embedding rows; activations are negligible."""
    D, I, Lyr, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.vocab_size
    weights = Lyr * (4 * D * D + 3 * D * I) * 2 + V * D * 2
    cache = Lyr * B * n_keys * 2 * D * 2
    return float(weights + cache + B * D * 2)
