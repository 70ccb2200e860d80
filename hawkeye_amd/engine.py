"""Execution engine of the MI355X encode_img path.

`HipEncoderEngine` owns (a) the packed device weights -- bf16 GEMM operands in
nn.Linear layout, fp32 biases / LayerNorm terms, the fused QKV and 6-layer
cross-K/V matrices -- and (b) one *plan* per input shape (B, T, Lt): a persistent
workspace in HBM plus the flat list of C-ABI calls (descriptors prebuilt) that
`HawkEye_it.encode_img` (models/hawkeye_it.py:219-270) lowers to.  Running a plan
is a loop of ctypes calls on the current HIP stream: no allocation, no host
synchronisation, no PyTorch arithmetic.

Data layout in HBM (row-major, M = B*N image tokens, N = T*L):
    x      bf16|fp32 [M, D] residual stream of the ViT (updated in place by the proj / fc2 epilogues)
    xn     bf16 [M, D]      LayerNorm output = GEMM operand
    qkv    bf16 [M, 3D]     columns [q | k | v], head h at h*64 inside each third (vit.py:88)
    att    bf16 [M, D]      attention output, already in the [B,N,H*64] layout proj consumes (vit.py:97)
    hid    bf16 [M, 4D]     GELU(fc1)
    kvc    bf16 [M, 2*d*n_cross]  cross-attention K|V of all cross layers from ONE GEMM
    Q-Former hidden states: fp32 + bf16 copies [B, S, d], S = n_query + Lt
"""
import ctypes as C
import math
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L
from . import ops
from .config import EncoderConfig

LOG2E = 1.4426950408889634


class _Pool:
    """Grow-only device buffers shared by every plan of an engine.

    A plan's workspace tensors are views of these buffers, so the memory an engine holds is the largest size each
    buffer was ever asked for -- not the sum over every (B, T, Lt) it has seen (instruction length varies per batch
    with padding='longest', hawkeye_it.py:243-249: a real eval or training run meets tens of shapes).  Plans run one
    after another on a stream, so sharing is safe under the same rule as before: one call in flight per engine.
    When a buffer has to grow, the plans built earlier hold descriptors into the old allocation: `epoch` moves and
    the owner drops them."""

    def __init__(self, device):
        self.device = device
        self.bufs: Dict[str, torch.Tensor] = {}
        self.epoch = 0

    def view(self, name, shape, dtype) -> torch.Tensor:
        n = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
        t = self.bufs.get(name)
        if t is None or t.numel() < n:
            if t is not None:
                self.epoch += 1
                self.bufs[name] = t = None      # release before the larger allocation
            t = torch.empty((n + 255) // 256 * 256, dtype=torch.uint8, device=self.device)
            self.bufs[name] = t
        return t[:n].view(dtype).view(shape)

    def nbytes(self) -> int:
        return sum(t.numel() for t in self.bufs.values())


class _Plan:
    def __init__(self, pool: Optional[_Pool] = None):
        self.pool = pool
        self.pool_epoch = pool.epoch if pool is not None else 0
        self.ws: Dict[str, torch.Tensor] = {}
        self.steps: List[Tuple] = []       # (fn, args) ; stream appended at run time
        self.keep: List = []               # descriptors / ctypes objects kept alive
        self.patch_desc: Optional[L.HkGemmDesc] = None
        self.vit_end = 0                   # steps[:vit_end] = the ViT up to vision_layernorm (what training reuses)
        # captured calls, keyed by (input kind, table address, norm) -> (graph, the table it reads: kept alive with it)
        self.graphs: "OrderedDict[Tuple, Tuple]" = OrderedDict()
        # Q-Former text-branch FFN of each layer, free to run beside the query branch on a second stream:
        # (fork, start, end) step indices -- the fork point is right after the self-attention LayerNorm, steps
        # [start, end) are the branch, the main stream joins before step `end`
        self.side: List[Tuple[int, int, int]] = []
        self.side_events: List = []

    def buf(self, name, shape, dtype, device):
        t = self.pool.view(name, shape, dtype) if self.pool is not None else torch.empty(shape, dtype=dtype, device=device)
        self.ws[name] = t
        return t


MAX_PLANS = 64      # host-side descriptor sets kept per engine (LRU); device memory is pooled, not per plan


class HipEncoderEngine:
    def __init__(self, cfg: EncoderConfig, device="cuda", residual_dtype=torch.bfloat16, halves=("vit", "qformer")):
        """residual_dtype: storage type of the ViT residual stream `x` (bf16 halves the proj / fc2
        epilogue and LayerNorm traffic; the reference's CUDA autocast path keeps it in fp16).
        Arithmetic is fp32 either way: each update is acc + bias + float(x), rounded once.
        halves: which part of the path this engine runs -- both (default), or one of them when the model is placed on
        two devices (set_device_ids, hawkeye_it.py:197-211): the "vit" engine ends with image_embeds (plan.ws["xn"]),
        the "qformer" engine starts from them."""
        self.halves = tuple(halves)
        if not self.halves or any(h not in ("vit", "qformer") for h in self.halves):
            raise RuntimeError(f"halves must be a non-empty subset of ('vit', 'qformer'), got {halves!r}")
        if residual_dtype not in (torch.bfloat16, torch.float32):
            raise RuntimeError("residual_dtype must be torch.bfloat16 or torch.float32")
        self.residual_dtype = residual_dtype
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipEncoderEngine runs on an MI355X only (device must be cuda:N)")
        self.lib = L.load()
        if cfg.encoder_embed_dim % (64 * cfg.encoder_num_heads) != 0 or \
                cfg.encoder_embed_dim // cfg.encoder_num_heads != 64 or cfg.qf_hidden // cfg.qf_heads != 64:
            raise RuntimeError("the HIP attention kernel is specialised for head size 64")
        self.w: Dict[str, torch.Tensor] = {}
        self.plans: "OrderedDict[Tuple[int, int, int], _Plan]" = OrderedDict()
        self.pool = _Pool(self.device)
        self.w_tiled = os.environ.get("HK_VIT_TILED", "0") == "1"
        self._w64: Dict[Tuple, torch.Tensor] = {}
        self.calls = 0          # every call that writes the pooled workspace counts (the training path's in-flight guard)
        self.n_cross = len([i for i in range(cfg.qf_layers) if i % cfg.cross_attention_freq == 0])
        self._tag = ""          # label of the plan steps being appended (profile() groups by it)
        # Optional hipGraph replay of a whole call for small sizes (HK_GRAPH=1, limit HK_GRAPH_MAX_FRAMES).  Off by
        # default: measured on MI355X it changes nothing (B=1,T=12: 6.22 vs 6.24 ms) -- the floor of a small call
        # is the ~300-600 DEPENDENT kernels' own dispatch-to-completion latency, which a graph does not shorten;
        # the host-side enqueue it removes was already hidden behind the GPU.  Kept for callers that want the
        # host thread free.
        self.use_graphs = os.environ.get("HK_GRAPH", "0") == "1"
        # HK_SIDE_STREAM=0: the Q-Former text-branch FFN runs in line on the caller's stream (A/B aid)
        self.side_stream = os.environ.get("HK_SIDE_STREAM", "1") != "0"
        self._side: Optional["torch.cuda.Stream"] = None
        # norm1 / norm2 of the ViT blocks folded into the qkv / fc1 GEMMs (statistics pass + row terms in the GEMM
        # epilogue instead of a LayerNorm pass that writes its output); needs the bf16 residual stream the statistics
        # kernel reads.  HK_FOLD_LN=0 keeps the standalone LayerNorm (A/B aid).
        self.fold_ln = residual_dtype == torch.bfloat16 and os.environ.get("HK_FOLD_LN", "1") != "0"
        # ... and the statistics themselves come out of the proj / fc2 epilogues that write the residual stream
        # (hk_gemm_desc.row_part + hk_row_stats_finalize) instead of a pass over it.  HK_EPI_STATS=0: A/B aid.
        self.epi_stats = os.environ.get("HK_EPI_STATS", "1") != "0"
        self.graph_max_frames = int(os.environ.get("HK_GRAPH_MAX_FRAMES", "64"))

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor], parts=("vit", "qformer")):
        """Pack a reference-keyed fp32 state dict into device operands.
        parts: which half to (re)pack -- "vit" (patch-embed, blocks, vision_layernorm) and/or "qformer"
        (Q-Former, query tokens, llama_proj); the training path repacks only what changed.
        Operands that already exist with the same shape are overwritten in place: cached plans hold their
        addresses."""
        cfg, dev = self.cfg, self.device
        parts = tuple(h for h in parts if h in self.halves)     # an engine packs only the half it runs
        w: Dict[str, torch.Tensor] = {}

        def f32(key):
            return sd[key].detach().to(device=dev, dtype=torch.float32).contiguous()

        def bf(t):
            return ops.cast_bf16(t.to(device=dev, dtype=torch.float32).contiguous())

        with torch.cuda.device(dev):
            D = cfg.encoder_embed_dim
            p = "vision_encoder.encoder."
            if "vit" in parts:
                w["patch.w"] = bf(sd[p + "patch_embed.proj.weight"].detach().reshape(D, -1))
                w["patch.b"] = f32(p + "patch_embed.proj.bias")
            for i in range(cfg.run_depth if "vit" in parts else 0):
                b = f"{p}blocks.{i}."
                o = f"vit.{i}."
                w[o + "ln1.g"], w[o + "ln1.b"] = f32(b + "norm1.weight"), f32(b + "norm1.bias")
                qb, vb = f32(b + "attn.q_bias"), f32(b + "attn.v_bias")
                w[o + "qkv.b"] = torch.cat((qb, torch.zeros_like(vb), vb)).contiguous()   # vit.py:85
                w[o + "proj.w"], w[o + "proj.b"] = bf(sd[b + "attn.proj.weight"].detach()), f32(b + "attn.proj.bias")
                w[o + "ln2.g"], w[o + "ln2.b"] = f32(b + "norm2.weight"), f32(b + "norm2.bias")
                w[o + "fc1.b"] = f32(b + "mlp.fc1.bias")
                if self.fold_ln:
                    # gamma into the weight, beta into the bias, column sums for the mean term (ops.fold_layernorm)
                    for nm, wk, g, bt in (("qkv", b + "attn.qkv.weight", "ln1.g", "ln1.b"),
                                          ("fc1", b + "mlp.fc1.weight", "ln2.g", "ln2.b")):
                        w[o + nm + ".w"], w[o + nm + ".c"], w[o + nm + ".b"] = ops.fold_layernorm(
                            f32(wk), w[o + nm + ".b"], w[o + g], w[o + bt])
                else:
                    w[o + "qkv.w"] = bf(sd[b + "attn.qkv.weight"].detach())
                    w[o + "fc1.w"] = bf(sd[b + "mlp.fc1.weight"].detach())
                w[o + "fc2.w"], w[o + "fc2.b"] = bf(sd[b + "mlp.fc2.weight"].detach()), f32(b + "mlp.fc2.bias")
            if "vit" in parts:
                w["vln.g"], w["vln.b"] = f32("vision_layernorm.weight"), f32("vision_layernorm.bias")
            if "qformer" in parts:
                self._pack_qformer(sd, w, f32, bf)
            stale = False
            for k, t in w.items():
                old = self.w.get(k)
                if old is not None and old.shape == t.shape and old.dtype == t.dtype:
                    old.copy_(t)
                else:
                    stale = stale or old is not None
                    self.w[k] = t
            if stale:
                self.plans.clear()      # an operand moved: descriptors of cached plans point at the old one
        return self

    def _pack_qformer(self, sd, w, f32, bf):
        cfg, dev = self.cfg, self.device
        with torch.cuda.device(dev):
            q = "qformer.bert."
            w["emb.word"] = f32(q + "embeddings.word_embeddings.weight")
            w["emb.pos"] = f32(q + "embeddings.position_embeddings.weight")
            w["emb.ln.g"], w["emb.ln.b"] = f32(q + "embeddings.LayerNorm.weight"), f32(q + "embeddings.LayerNorm.bias")
            qt = sd["query_tokens"].detach()
            if cfg.extra_num_query_token > 0:
                qt = torch.cat([qt, sd["extra_query_tokens"].detach().to(qt.device)], dim=1)   # hawkeye_it.py:238
            w["query"] = qt.reshape(-1, cfg.qf_hidden).to(device=dev, dtype=torch.float32).contiguous()
            kv_w, kv_b = [], []
            for i in range(cfg.qf_layers):
                l = f"{q}encoder.layer.{i}."
                o = f"qf.{i}."
                s = l + "attention.self."
                w[o + "qkv.w"] = bf(torch.cat([sd[s + n + ".weight"].detach() for n in ("query", "key", "value")]))
                w[o + "qkv.b"] = torch.cat([f32(s + n + ".bias") for n in ("query", "key", "value")]).contiguous()
                w[o + "ao.w"], w[o + "ao.b"] = bf(sd[l + "attention.output.dense.weight"].detach()), f32(l + "attention.output.dense.bias")
                w[o + "ao.g"], w[o + "ao.bt"] = f32(l + "attention.output.LayerNorm.weight"), f32(l + "attention.output.LayerNorm.bias")
                if i % cfg.cross_attention_freq == 0:
                    c = l + "crossattention."
                    w[o + "cq.w"], w[o + "cq.b"] = bf(sd[c + "self.query.weight"].detach()), f32(c + "self.query.bias")
                    kv_w += [sd[c + "self.key.weight"].detach(), sd[c + "self.value.weight"].detach()]
                    kv_b += [f32(c + "self.key.bias"), f32(c + "self.value.bias")]
                    w[o + "co.w"], w[o + "co.b"] = bf(sd[c + "output.dense.weight"].detach()), f32(c + "output.dense.bias")
                    w[o + "co.g"], w[o + "co.bt"] = f32(c + "output.LayerNorm.weight"), f32(c + "output.LayerNorm.bias")
                for tag, a, bb in (("t", "intermediate.", "output."), ("q", "intermediate_query.", "output_query.")):
                    w[o + tag + "i.w"], w[o + tag + "i.b"] = bf(sd[l + a + "dense.weight"].detach()), f32(l + a + "dense.bias")
                    w[o + tag + "o.w"], w[o + tag + "o.b"] = bf(sd[l + bb + "dense.weight"].detach()), f32(l + bb + "dense.bias")
                    w[o + tag + "o.g"], w[o + tag + "o.bt"] = f32(l + bb + "LayerNorm.weight"), f32(l + bb + "LayerNorm.bias")
            # all cross layers read the same image_embeds (Qformer.py:198-201): one [2*d*n_cross, D] weight
            w["ckv.w"] = bf(torch.cat([t.to("cpu") for t in kv_w]))
            w["ckv.b"] = torch.cat(kv_b).contiguous()
            w["proj.w"], w["proj.b"] = bf(sd["llama_proj.weight"].detach()), f32("llama_proj.bias")

    # ------------------------------------------------------------------ plan
    def _gemm(self, plan, A, W, Cout, M, N, K, **kw):
        # HK_VIT_TILED=1 (experiment, default off): stream a 64 x 64-block copy of the weight (hk_gemm_desc.w_tiled)
        if self.w_tiled and "ldw" not in kw and W.dim() == 2 and W.is_contiguous() and N % 64 == 0 and K % 64 == 0 \
                and tuple(W.shape) == (N, K):
            key = (W.data_ptr(), N, K)
            t = self._w64.get(key)
            if t is None:
                t = self._w64[key] = ops.tile_gemm_weight(W)
            W, kw = t, dict(kw, w_tiled=True)
        d = ops.gemm_desc(A, W.data_ptr(), Cout, M, N, K, **kw)
        plan.keep.append(d)
        plan.steps.append((self.lib.hk_gemm_bf16, (C.byref(d),), "hk_gemm_bf16", self._tag))
        return d

    @staticmethod
    def key_splits_for(B: int, H: int, Nq: int, Nk: int) -> int:
        """Key-split factor of a cross-attention launch (hk_attn_desc.key_splits): 1 unless the launch is a handful of
        workgroups each walking every key in turn (B = 1: 12 workgroups x 37 tiles).  Isolated, B = 1 (tools/
        attn_split_bench.py): T = 12 28.2 -> 17.3 us, T = 16 36.4 -> 17.4 us at 4 slices (two launches: the floor);
        the ViT's self-attention (304 workgroups at B = 1) gains nothing (38.9 -> 35.9 us at 2 slices, T = 16 slower)
        and stays unsplit."""
        if os.environ.get("HK_ATTN_SPLIT", "1") == "0":
            return 1
        wgs = B * H * -(-Nq // 128)
        s = min(4, 128 // max(wgs, 1), Nk // 256)
        return s if s >= 2 else 1

    def _attn(self, plan, *a, **kw):
        d = ops.attn_desc(*a, **kw)
        plan.keep.append(d)
        plan.steps.append((self.lib.hk_flash_attn_bf16, (C.byref(d),), "hk_flash_attn_bf16", self._tag))

    def _ln(self, plan, x, in_f32, gamma, beta, ob, of, rows, D, eps, out_rows=None):
        orow = None
        if out_rows is not None:
            r = L.HkRows(*out_rows)
            plan.keep.append(r)
            orow = C.byref(r)
        plan.steps.append((self.lib.hk_layernorm,
                           (x, L.HK_DT_F32 if in_f32 else L.HK_DT_BF16, None, gamma.data_ptr(), beta.data_ptr(),
                            ob, of, orow, rows, D, eps), "hk_layernorm", self._tag))

    def _ln_parts(self, plan, parts, n_parts, stride, gamma, beta, ob, of, rows, D, eps, out_rows=None):
        orow = None
        if out_rows is not None:
            r = L.HkRows(*out_rows)
            plan.keep.append(r)
            orow = C.byref(r)
        plan.steps.append((self.lib.hk_layernorm_parts,
                           (parts, L.HK_DT_F32, parts + 4 * stride, n_parts - 1, stride, gamma.data_ptr(), beta.data_ptr(),
                            ob, of, orow, rows, D, eps), "hk_layernorm_parts", self._tag))

    @staticmethod
    def split_k_for(rows: int, N: int, K: int) -> int:
        """Split-K factor of a dense layer: 1 unless the launch has few output tiles and a long contraction (the
        Q-Former's 3072 -> 768 layers at small batch: 12 workgroups x 48 K-tiles in a row).  A divisor of K / 64 that
        brings the launch to about one workgroup per CU, slices of at least 6 K-tiles."""
        if os.environ.get("HK_SPLIT_K", "1") == "0":
            return 1
        tiles = -(-rows // 128) * -(-N // 64)
        kt = K // 64
        # HK_SPLIT_K_SHORT=0 (A/B aid): only the long contractions (K >= 1536), as before the 768 -> 768 layers joined
        short = os.environ.get("HK_SPLIT_K_SHORT", "1") != "0"
        if tiles > 128 or kt < (12 if short else 24):
            return 1
        best = 1
        for s in range(2, 9):
            if kt % s == 0 and kt // s >= (4 if kt < 24 else 6) and tiles * s <= 288:
                best = s
        return best

    def split_k_residual(self, M: int, N: int, K: int) -> int:
        """Split-K factor of a proj / fc2 update of the residual stream on a SMALL call (B = 1 eval calls: fc2 at
        M = 1568 is 104 workgroups walking 64 K-tiles one after the other, a latency chain).  Only while the slices
        make at most one workgroup per CU and are at least 16 K-tiles long.  With 128x128 tiles that covered T <= 8 only
        (measured on one box, B = 1: T = 8 5.42 -> 5.07 ms; T = 12 / 16, 152 / 200 tiles, 5.87 -> 6.01 / 6.05 -> 6.74 ms:
        two workgroups per CU on the 2-stage kernel, and the fp32 slices cost more than the shorter chain saves); with
        256x128 tiles the slices of T = 6 ... 16 fit one round too (below)."""
        if os.environ.get("HK_SPLIT_K", "1") == "0" or self.residual_dtype != torch.bfloat16 or not self.fold_ln:
            return 1
        if -(-M // 256) * -(-N // 256) >= 128:       # the persistent kernel takes it
            return 1
        kt = K // 64
        cap = int(os.environ.get("HK_SPLIT_K_VIT", "4"))   # the switch: A/B aid
        # one round of 256x128 tiles when the slices make 160-256 of them (the dispatcher's rule for these launches):
        # tools/small_gemm.py, fc2 + the consumer's sum against unsplit + row_stats_finalize: M = 2352 (T = 12) 3 slices
        # 37.4 vs 50.3 us, M = 3136 (T = 16) 2 slices 45.0 vs 51.6, M = 1568 (T = 8) 4 slices 30.1 vs 36.8; proj (K = 1024:
        # slices under 16 K-tiles) gains nothing and stays unsplit.  In whole calls (tools/b1_bench.py, same box, both rules
        # of HK_GEMM_MID on / off): B = 1 T = 8 4.97 -> 4.45 ms, T = 10 5.20 -> 4.70, T = 12 5.72 -> 5.31, B = 2 T = 4 4.74 -> 4.20;
        # the 2-slice case (M = 3136: B = 1 T = 16, B = 2 T = 8) measured +1 % in the call and is left unsplit
        if os.environ.get("HK_GEMM_MID", "1") != "0":
            tmid = -(-M // 256) * -(-N // 128)
            s = min(cap, 256 // max(tmid, 1), kt // 16)
            # (two slices too since the 256x128 tile runs three stages: HK_SPLIT_K_VIT_MIN=3 / 2 interleaved on one box, B = 1 T = 14
            # 5.44 -> 5.35 ms, T = 16 5.76 -> 5.68, B = 2 T = 8 5.33 -> 5.24; with the two-stage tile the 2-slice case had measured +1 %)
            if s >= int(os.environ.get("HK_SPLIT_K_VIT_MIN", "2")) and tmid * s >= 160:
                return s
        t128 = -(-M // 128) * -(-N // 128)
        s = min(cap, 256 // max(t128, 1), kt // 16)
        return s if s >= 2 else 1

    def _build_plan(self, B: int, T: int, Lt: int) -> _Plan:
        cfg, dev, w = self.cfg, self.device, self.w
        plan = _Plan(self.pool)
        D, Hv = cfg.encoder_embed_dim, cfg.encoder_num_heads
        Lp = cfg.patches_per_frame
        N = T * Lp
        M = B * N
        Kp = cfg.patch_k
        d, Hq, inter = cfg.qf_hidden, cfg.qf_heads, cfg.qf_intermediate
        nq = cfg.n_query
        S = nq + Lt
        KVW = 2 * d * self.n_cross
        bf16, f32 = torch.bfloat16, torch.float32
        with torch.cuda.device(dev):
            run_vit, run_qf = "vit" in self.halves, "qformer" in self.halves
            xn = plan.buf("xn", (M, D), bf16, dev)          # image_embeds: the hand-over point between the two halves
            ids = plan.buf("ids", (B, max(Lt, 1)), torch.int64, dev)
            kbias = plan.buf("kbias", (B, S), f32, dev)
            plan.buf("mask", (B, max(Lt, 1)), torch.int64, dev)
            sk2 = sk_q = sk_t = 1
            if run_vit:
                pa = plan.buf("patches", (M, Kp), bf16, dev)
                x = plan.buf("x", (M, D), self.residual_dtype, dev)
                rstat = plan.buf("rstat", ((M + 255) // 256 * 256, 2), f32, dev) if self.fold_ln else None   # whole 256-row tiles are read
                rpart = plan.buf("rpart", (M, D // 64, 2), f32, dev) if self.fold_ln and self.epi_stats else None
                qkv = plan.buf("qkv", (M, 3 * D), bf16, dev)
                att = plan.buf("att", (M, D), bf16, dev)
                hid = plan.buf("hid", (M, D * cfg.mlp_ratio), bf16, dev)
                sk2 = self.split_k_residual(M, D, D * cfg.mlp_ratio)
                parts_x = plan.buf("parts_x", (sk2, M, D), f32, dev) if sk2 > 1 else None
            if run_qf:
                kvc = plan.buf("kvc", (M, KVW), bf16, dev)
                hf = plan.buf("hf", (B * S, d), f32, dev)
                hb = plan.buf("hb", (B * S, d), bf16, dev)
                qqkv = plan.buf("qqkv", (B * S, 3 * d), bf16, dev)
                ctx = plan.buf("ctx", (B * S, d), bf16, dev)
                tf = plan.buf("tf", (B * S, d), f32, dev)
                af = plan.buf("af", (B * S, d), f32, dev)
                ab = plan.buf("ab", (B * S, d), bf16, dev)
                cq = plan.buf("cq", (B * nq, d), bf16, dev)
                cctx = plan.buf("cctx", (B * nq, d), bf16, dev)
                cf = plan.buf("cf", (B * nq, d), f32, dev)
                cb = plan.buf("cb", (B * nq, d), bf16, dev)
                fh = plan.buf("fh", (B * nq, inter), bf16, dev)
                fht = plan.buf("fh_t", (B * max(Lt, 1), inter), bf16, dev)   # text-branch temporaries of their own: the branch
                tft = plan.buf("tf_t", (B * max(Lt, 1), d), f32, dev)         # may run beside the query branch (plan.side)
                out = plan.buf("out", (B * nq, cfg.llama_hidden), f32, dev)
                # split-K partial products of the 3072 -> 768 dense layers at small batch (query / text branch: the
                # text branch may run beside the query branch, so it has slices of its own)
                sk_q = self.split_k_for(B * nq, d, inter)
                sk_t = self.split_k_for(B * Lt, d, inter) if Lt > 0 else 1
                parts_q = plan.buf("parts_q", (sk_q, B * nq, d), f32, dev) if sk_q > 1 else None
                parts_t = plan.buf("parts_t", (sk_t, B * Lt, d), f32, dev) if sk_t > 1 else None
                # the attention output layers (768 -> 768, 12 K-tiles) on a handful of workgroups: same treatment
                sk_a = self.split_k_for(B * S, d, d)
                sk_c = self.split_k_for(B * nq, d, d)
                parts_a = plan.buf("parts_a", (sk_a, B * S, d), f32, dev) if sk_a > 1 else None
                parts_c = plan.buf("parts_c", (sk_c, B * nq, d), f32, dev) if sk_c > 1 else None

        P = lambda t: t.data_ptr()  # noqa: E731
        plain = lambda ld: (ld, 0, 0)  # noqa: E731
        qscale = (64 ** -0.5) * LOG2E

        if run_vit:
            # (the ViT's self-attention is never key-split: 304 workgroups at B = 1 already, see key_splits_for)
            # ---- ViT: patch-embed GEMM (+bias, + position table broadcast over the batch) ----
            # step 0 (patchify) and the table pointer are bound at run time
            xf32 = self.residual_dtype == torch.float32
            rkw = dict(out_f32=xf32, res_bf16=not xf32)   # in-place residual-stream update
            self._tag = "vit.patch_embed"
            plan.patch_desc = self._gemm(plan, P(pa), w["patch.w"], P(x), M, D, Kp, bias=P(w["patch.b"]), residual=0,
                                         a_rows=plain(Kp), c_rows=plain(D), r_rows=(D, N, 0), out_f32=xf32)
            for i in range(cfg.run_depth):
                o = f"vit.{i}."
                fold = {}
                if self.fold_ln:
                    # vit.py:125 norm1 folded into attn.qkv: statistics of the residual stream, then the GEMM on x itself
                    self._tag = "vit.ln_stats"
                    if sk2 > 1 and i > 0:
                        pass                             # the previous block's split-K tail (hk_residual_add_parts) left (mean, rstd)
                    elif rpart is not None and i > 0:   # the previous block's fc2 epilogue left the row sums
                        plan.steps.append((self.lib.hk_row_stats_finalize, (P(rpart), P(rstat), M, D // 64, cfg.vit_ln_eps),
                                           "hk_row_stats_finalize", self._tag))
                    else:
                        plan.steps.append((self.lib.hk_row_stats, (P(x), P(rstat), M, D, cfg.vit_ln_eps), "hk_row_stats", self._tag))
                    fold = dict(row_stats=P(rstat), col_c=P(w[o + "qkv.c"]))
                else:
                    self._tag = "vit.ln"
                    self._ln(plan, P(x), xf32, w[o + "ln1.g"], w[o + "ln1.b"], P(xn), None, M, D, cfg.vit_ln_eps)
                self._tag = "vit.qkv"
                self._gemm(plan, P(x) if self.fold_ln else P(xn), w[o + "qkv.w"], P(qkv), M, 3 * D, D, bias=P(w[o + "qkv.b"]),
                           a_rows=plain(D), c_rows=plain(3 * D), scale_cols=D, scale=qscale, **fold)
                self._tag = "vit.attn"
                self._attn(plan, P(qkv), P(qkv) + 2 * D, P(qkv) + 4 * D, P(att), B, Hv, N, N,
                           3 * D, 3 * D, 3 * D, D, N * 3 * D, N * 3 * D, N * 3 * D, N * D)
                self._tag = "vit.proj"
                pkw = dict(row_part=P(rpart)) if rpart is not None else {}
                self._gemm(plan, P(att), w[o + "proj.w"], P(x), M, D, D, bias=P(w[o + "proj.b"]), residual=P(x),
                           a_rows=plain(D), c_rows=plain(D), r_rows=plain(D), **rkw, **pkw)
                if self.fold_ln:
                    self._tag = "vit.ln_stats"
                    if rpart is not None:
                        plan.steps.append((self.lib.hk_row_stats_finalize, (P(rpart), P(rstat), M, D // 64, cfg.vit_ln_eps),
                                           "hk_row_stats_finalize", self._tag))
                    else:
                        plan.steps.append((self.lib.hk_row_stats, (P(x), P(rstat), M, D, cfg.vit_ln_eps), "hk_row_stats", self._tag))
                    fold = dict(row_stats=P(rstat), col_c=P(w[o + "fc1.c"]))
                else:
                    self._tag = "vit.ln"
                    self._ln(plan, P(x), xf32, w[o + "ln2.g"], w[o + "ln2.b"], P(xn), None, M, D, cfg.vit_ln_eps)
                self._tag = "vit.fc1_gelu"
                self._gemm(plan, P(x) if self.fold_ln else P(xn), w[o + "fc1.w"], P(hid), M, D * cfg.mlp_ratio, D,
                           bias=P(w[o + "fc1.b"]), a_rows=plain(D), c_rows=plain(D * cfg.mlp_ratio),
                           act=L.HK_ACT_GELU_ERF, **fold)
                self._tag = "vit.fc2"
                if i == cfg.run_depth - 1:
                    pkw = {}                            # followed by vision_layernorm, which is not folded
                if sk2 > 1:
                    # small call: K split over workgroups, fp32 slices, then one pass that adds them into the residual
                    # stream and leaves the next block's norm1 statistics (in place of hk_row_stats_finalize)
                    self._gemm(plan, P(hid), w[o + "fc2.w"], P(parts_x), M, D, D * cfg.mlp_ratio, bias=P(w[o + "fc2.b"]),
                               a_rows=plain(D * cfg.mlp_ratio), c_rows=plain(D), out_f32=True, split_k=sk2, split_stride=M * D)
                    self._tag = "vit.fc2_tail"
                    plan.steps.append((self.lib.hk_residual_add_parts,
                                       (P(x), P(parts_x), sk2, M * D, P(rstat) if i < cfg.run_depth - 1 else None, M, D,
                                        cfg.vit_ln_eps), "hk_residual_add_parts", self._tag))
                else:
                    self._gemm(plan, P(hid), w[o + "fc2.w"], P(x), M, D, D * cfg.mlp_ratio, bias=P(w[o + "fc2.b"]),
                               residual=P(x), a_rows=plain(D * cfg.mlp_ratio), c_rows=plain(D), r_rows=plain(D), **rkw, **pkw)
            # vision_layernorm (hawkeye_it.py:234) then the K/V projections of every cross layer at once
            self._tag = "vision_layernorm"
            self._ln(plan, P(x), xf32, w["vln.g"], w["vln.b"], P(xn), None, M, D, cfg.vision_ln_eps)
        plan.vit_end = len(plan.steps)
        if run_qf:
            ksc = self.key_splits_for(B, Hq, nq, N)
            plan.cross_key_splits = ksc
            ckw = {}
            if ksc > 1:
                ckw = dict(key_splits=ksc, o_part=P(plan.buf("cattn_part", (B * ksc, nq, d), bf16, dev)),
                           lse_part=P(plan.buf("cattn_lse", (B * ksc, Hq, nq), f32, dev)))
            self._tag = "qformer.cross_kv"
            self._gemm(plan, P(xn), w["ckv.w"], P(kvc), M, KVW, D, bias=P(w["ckv.b"]), a_rows=plain(D), c_rows=plain(KVW))

            # ---- Q-Former ----
            self._tag = "qformer"
            plan.steps.append((self.lib.hk_qformer_embed_ln,
                               (P(w["query"]), P(w["emb.word"]), P(w["emb.pos"]), P(ids) if Lt > 0 else None,
                                P(w["emb.ln.g"]), P(w["emb.ln.b"]), P(hf), P(hb), B, nq, Lt, d, w["emb.word"].shape[0],
                                cfg.qf_ln_eps), "hk_qformer_embed_ln", "qformer"))
            qrows = (d, nq, S * d)
            trows = (d, Lt, S * d)
            eb = 2  # bytes per bf16
            ci = 0
            for i in range(cfg.qf_layers):
                o = f"qf.{i}."
                last = i == cfg.qf_layers - 1
                self._gemm(plan, P(hb), w[o + "qkv.w"], P(qqkv), B * S, 3 * d, d, bias=P(w[o + "qkv.b"]),
                           a_rows=plain(d), c_rows=plain(3 * d), scale_cols=d, scale=qscale)
                self._attn(plan, P(qqkv), P(qqkv) + d * eb, P(qqkv) + 2 * d * eb, P(ctx), B, Hq, S, S,
                           3 * d, 3 * d, 3 * d, d, S * 3 * d, S * 3 * d, S * 3 * d, S * d, key_bias=P(kbias))
                if sk_a > 1:
                    self._gemm(plan, P(ctx), w[o + "ao.w"], P(parts_a), B * S, d, d, bias=P(w[o + "ao.b"]), residual=P(hf),
                               a_rows=plain(d), c_rows=plain(d), r_rows=plain(d), out_f32=True, split_k=sk_a,
                               split_stride=B * S * d)
                    self._ln_parts(plan, P(parts_a), sk_a, B * S * d, w[o + "ao.g"], w[o + "ao.bt"], P(ab), P(af), B * S, d,
                                   cfg.qf_ln_eps)
                else:
                    self._gemm(plan, P(ctx), w[o + "ao.w"], P(tf), B * S, d, d, bias=P(w[o + "ao.b"]), residual=P(hf),
                               a_rows=plain(d), c_rows=plain(d), r_rows=plain(d), out_f32=True)
                    self._ln(plan, P(tf), True, w[o + "ao.g"], w[o + "ao.bt"], P(ab), P(af), B * S, d, cfg.qf_ln_eps)
                fork = len(plan.steps)
                if i % cfg.cross_attention_freq == 0:
                    self._gemm(plan, P(ab), w[o + "cq.w"], P(cq), B * nq, d, d, bias=P(w[o + "cq.b"]),
                               a_rows=qrows, c_rows=plain(d), scale_cols=d, scale=qscale)
                    self._attn(plan, P(cq), P(kvc) + ci * 2 * d * eb, P(kvc) + (ci * 2 + 1) * d * eb, P(cctx), B, Hq, nq, N,
                               d, KVW, KVW, d, nq * d, N * KVW, N * KVW, nq * d, **ckw)
                    if sk_c > 1:
                        self._gemm(plan, P(cctx), w[o + "co.w"], P(parts_c), B * nq, d, d, bias=P(w[o + "co.b"]), residual=P(af),
                                   a_rows=plain(d), c_rows=plain(d), r_rows=qrows, out_f32=True, split_k=sk_c,
                                   split_stride=B * nq * d)
                        self._ln_parts(plan, P(parts_c), sk_c, B * nq * d, w[o + "co.g"], w[o + "co.bt"], P(cb), P(cf), B * nq, d,
                                       cfg.qf_ln_eps)
                    else:
                        self._gemm(plan, P(cctx), w[o + "co.w"], P(tf), B * nq, d, d, bias=P(w[o + "co.b"]), residual=P(af),
                                   a_rows=plain(d), c_rows=plain(d), r_rows=qrows, out_f32=True)
                        self._ln(plan, P(tf), True, w[o + "co.g"], w[o + "co.bt"], P(cb), P(cf), B * nq, d, cfg.qf_ln_eps)
                    qin_b, qin_f, qin_rows = P(cb), P(cf), plain(d)
                    ci += 1
                else:
                    qin_b, qin_f, qin_rows = P(ab), P(af), qrows
                # query-branch FFN (Qformer.py:514-517) -> rows [:, :nq] of the next hidden state
                self._gemm(plan, qin_b, w[o + "qi.w"], P(fh), B * nq, inter, d, bias=P(w[o + "qi.b"]),
                           a_rows=qin_rows, c_rows=plain(inter), act=L.HK_ACT_GELU_ERF)
                if sk_q > 1:
                    self._gemm(plan, P(fh), w[o + "qo.w"], P(parts_q), B * nq, d, inter, bias=P(w[o + "qo.b"]), residual=qin_f,
                               a_rows=plain(inter), c_rows=plain(d), r_rows=qin_rows, out_f32=True, split_k=sk_q,
                               split_stride=B * nq * d)
                    self._ln_parts(plan, P(parts_q), sk_q, B * nq * d, w[o + "qo.g"], w[o + "qo.bt"], P(hb), P(hf), B * nq, d,
                                   cfg.qf_ln_eps, out_rows=qrows)
                else:
                    self._gemm(plan, P(fh), w[o + "qo.w"], P(tf), B * nq, d, inter, bias=P(w[o + "qo.b"]), residual=qin_f,
                               a_rows=plain(inter), c_rows=plain(d), r_rows=qin_rows, out_f32=True)
                    self._ln(plan, P(tf), True, w[o + "qo.g"], w[o + "qo.bt"], P(hb), P(hf), B * nq, d, cfg.qf_ln_eps,
                             out_rows=qrows)
                # text-branch FFN (Qformer.py:509-512); its last-layer result is never read (hawkeye_it.py:269)
                if Lt > 0 and not last:
                    toff = nq * d
                    start = len(plan.steps)
                    self._gemm(plan, P(ab) + toff * eb, w[o + "ti.w"], P(fht), B * Lt, inter, d, bias=P(w[o + "ti.b"]),
                               a_rows=trows, c_rows=plain(inter), act=L.HK_ACT_GELU_ERF)
                    if sk_t > 1:
                        self._gemm(plan, P(fht), w[o + "to.w"], P(parts_t), B * Lt, d, inter, bias=P(w[o + "to.b"]),
                                   residual=P(af) + toff * 4, a_rows=plain(inter), c_rows=plain(d), r_rows=trows, out_f32=True,
                                   split_k=sk_t, split_stride=B * Lt * d)
                        self._ln_parts(plan, P(parts_t), sk_t, B * Lt * d, w[o + "to.g"], w[o + "to.bt"], P(hb) + toff * eb,
                                       P(hf) + toff * 4, B * Lt, d, cfg.qf_ln_eps, out_rows=trows)
                    else:
                        self._gemm(plan, P(fht), w[o + "to.w"], P(tft), B * Lt, d, inter, bias=P(w[o + "to.b"]),
                                   residual=P(af) + toff * 4, a_rows=plain(inter), c_rows=plain(d), r_rows=trows, out_f32=True)
                        self._ln(plan, P(tft), True, w[o + "to.g"], w[o + "to.bt"], P(hb) + toff * eb, P(hf) + toff * 4,
                                 B * Lt, d, cfg.qf_ln_eps, out_rows=trows)
                    # reads rows [nq:] of ab / af (complete at `fork`), writes rows [nq:] of hb / hf (read by the next
                    # layer's qkv GEMM, the step after `end`); the query branch touches rows [:nq] only
                    plan.side.append((fork, start, len(plan.steps)))
            # llama_proj on the query rows (hawkeye_it.py:269)
            self._tag = "llama_proj"
            self._gemm(plan, P(hb), w["proj.w"], P(out), B * nq, cfg.llama_hidden, d, bias=P(w["proj.b"]),
                       a_rows=qrows, c_rows=plain(cfg.llama_hidden), out_f32=True)
        return plan

    # ------------------------------------------------------------------ run
    @torch.no_grad()
    def forward(self, image: torch.Tensor, input_ids: Optional[torch.Tensor], text_mask: Optional[torch.Tensor],
                pos_embed: torch.Tensor, profile: Optional[list] = None, norm=None) -> torch.Tensor:
        """image [B,T,3,H,W] fp32 on the engine's GPU, or decoded frames [B,T,H,W,3] uint8 (normalised on the
        device with norm = (mean3, std3), default the CLIP statistics of the eval scripts); input_ids/text_mask [B,Lt] int64 (or None);
        pos_embed fp32 [N,D] on the GPU (the table the caller's encoder holds for this T).
        Returns fp32 [B, n_query, llama_hidden] -- a view of the plan's output buffer."""
        if self.halves != ("vit", "qformer"):
            raise RuntimeError("forward() needs an engine that runs both halves; a two-device placement goes through "
                               "forward_vit() on one engine and forward_qformer() on the other")
        plan, image, u8, B, T, Lt, H, W = self._prepare(image, input_ids, text_mask, pos_embed)
        cfg, ws, nq = self.cfg, plan.ws, self.cfg.n_query
        with torch.cuda.device(self.device):
            use_graph = profile is None and self.use_graphs and B * T <= self.graph_max_frames
            if use_graph:
                # Launch-bound sizes (a B=1 call is ~600 launches of a few microseconds each): the whole call is
                # one hipGraph replay reading a fixed input buffer.  First call of a (plan, table, input kind):
                # one eager pass (per-kernel static set-up happens there), then capture.
                name = "in_u8" if u8 else "in_f32"
                if name not in ws:
                    plan.buf(name, tuple(image.shape), image.dtype, self.device)
                ws[name].copy_(image, non_blocking=True)
                gkey = (name, pos_embed.data_ptr(), norm)
                g = plan.graphs.get(gkey, (None, None))[0]
                if g is None:
                    self._enqueue(plan, ws[name], u8, norm, pos_embed, B, T, Lt, H, W, None)
                    torch.cuda.current_stream().synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._enqueue(plan, ws[name], u8, norm, pos_embed, B, T, Lt, H, W, None)
                    while len(plan.graphs) >= 4:   # e.g. a caller that assigns a fresh pos_embed tensor per call
                        plan.graphs.popitem(last=False)
                    plan.graphs[gkey] = (g, pos_embed)   # the captured launches read this table's address
                g.replay()
            else:
                self._enqueue(plan, image, u8, norm, pos_embed, B, T, Lt, H, W, profile)
        return ws["out"].view(B, nq, cfg.llama_hidden)

    @torch.no_grad()
    def forward_vit(self, image, input_ids, text_mask, pos_embed, plan=None, norm=None):
        """The frozen half of a training call (row f2): patch gather, ViT blocks, vision_layernorm and the text
        key bias, i.e. the plan's launches up to `vit_end`; leaves image_embeds (bf16) in plan.ws["xn"]."""
        plan, image, u8, B, T, Lt, H, W = self._prepare(image, input_ids, text_mask, pos_embed)
        with torch.cuda.device(self.device):
            self._enqueue(plan, image, u8, norm, pos_embed, B, T, Lt, H, W, None, upto=plan.vit_end)
        return plan

    @torch.no_grad()
    def forward_qformer(self, image_embeds: torch.Tensor, input_ids, text_mask) -> torch.Tensor:
        """The second half on its own (two-device placement, hawkeye_it.py:230-234: the ViT may sit on another GPU and
        hands over image_embeds): image_embeds bf16 [B, N, D] = vision_layernorm's output, on any GPU -- copied into
        this engine's plan -- then cross K/V, the Q-Former and llama_proj.  Returns fp32 [B, n_query, llama_hidden]."""
        plan, B, T, Lt = self.load_image_embeds(image_embeds, input_ids, text_mask, need_weights=True)
        with torch.cuda.device(self.device):
            self._enqueue(plan, None, False, None, None, B, T, Lt, 0, 0, None, lo=plan.vit_end, no_input=True, key_bias=False)
        return plan.ws["out"].view(B, self.cfg.n_query, self.cfg.llama_hidden)

    @torch.no_grad()
    def load_image_embeds(self, image_embeds: torch.Tensor, input_ids, text_mask, need_weights: bool = False):
        """The hand-over point of a two-device placement (hawkeye_it.py:234 `.to(self.device)`): image_embeds bf16
        [B, N, D] from any GPU -> this engine's plan (`xn`), token ids / mask -> the plan, key bias launched.  Used by
        forward_qformer and by the training engine, whose Q-Former forward starts from exactly these buffers.
        Returns (plan, B, T, Lt)."""
        cfg = self.cfg
        if "qformer" not in self.halves or (need_weights and not self.w):
            raise RuntimeError("this engine does not hold the Q-Former half (or has no weights yet)")
        B, N, D = image_embeds.shape
        Lp = cfg.patches_per_frame
        if N % Lp != 0 or D != cfg.encoder_embed_dim or image_embeds.dtype != torch.bfloat16:
            raise RuntimeError(f"image_embeds must be bf16 [B, T*{Lp}, {cfg.encoder_embed_dim}], got {tuple(image_embeds.shape)} {image_embeds.dtype}")
        Lt = 0 if input_ids is None else int(input_ids.shape[1])
        plan = self.plan_for(B, N // Lp, Lt)
        self.calls += 1
        with torch.cuda.device(self.device):
            plan.ws["xn"].view(B, N, D).copy_(image_embeds, non_blocking=True)     # the device hop of hawkeye_it.py:234
            if Lt > 0:
                plan.ws["ids"].copy_(input_ids, non_blocking=True)
                plan.ws["mask"].copy_(text_mask, non_blocking=True)
            L.check(self.lib.hk_qformer_key_bias(plan.ws["mask"].data_ptr() if Lt > 0 else None, plan.ws["kbias"].data_ptr(), B,
                                                 cfg.n_query, Lt, torch.cuda.current_stream().cuda_stream), "hk_qformer_key_bias")
        return plan, B, N // Lp, Lt

    def _prepare(self, image, input_ids, text_mask, pos_embed):
        """Input validation, plan lookup, copy of the token ids / mask into the plan."""
        cfg = self.cfg
        if not self.w:
            raise RuntimeError("HipEncoderEngine: load_state_dict() has not been called")
        u8 = image.dtype == torch.uint8
        if u8:
            # row f3: decoded frames [B,T,H,W,3] uint8; /255, mean/std fused into the patch gather
            if image.dim() != 5 or image.shape[4] != 3:
                raise RuntimeError(f"uint8 frames must be [B,T,H,W,3], got {tuple(image.shape)}")
            B, T, H, W, _ = image.shape
        else:
            if image.dim() != 5 or image.shape[2] != 3:
                raise RuntimeError(f"image must be [B,T,3,H,W], got {tuple(image.shape)}")
            B, T, _, H, W = image.shape
        if H != cfg.img_size or W != cfg.img_size:
            raise RuntimeError(f"image must be {cfg.img_size}x{cfg.img_size}, got {H}x{W}")
        if image.device != self.device:
            raise RuntimeError(f"image is on {image.device}, engine on {self.device}")
        if u8:
            image = image.contiguous()
        elif image.dtype != torch.float32 or not image.is_contiguous():
            image = image.float().contiguous()
        Lt = 0 if input_ids is None else int(input_ids.shape[1])
        N = T * cfg.patches_per_frame
        if pos_embed.shape[-2:] != (N, cfg.encoder_embed_dim) or pos_embed.dtype != torch.float32 or \
                pos_embed.device != self.device or not pos_embed.is_contiguous():
            raise RuntimeError(f"pos_embed must be a contiguous fp32 [{N},{cfg.encoder_embed_dim}] tensor on {self.device}")
        plan = self.plan_for(B, T, Lt)
        self.calls += 1
        with torch.cuda.device(self.device):
            if Lt > 0:
                plan.ws["ids"].copy_(input_ids, non_blocking=True)
                plan.ws["mask"].copy_(text_mask, non_blocking=True)
        return plan, image, u8, B, T, Lt, H, W

    def plan_for(self, B: int, T: int, Lt: int) -> _Plan:
        """The plan of a shape: cached (LRU, MAX_PLANS descriptor sets), rebuilt when the pooled workspace moved."""
        key = (B, T, Lt)
        plan = self.plans.get(key)
        if plan is not None and plan.pool_epoch == self.pool.epoch:
            self.plans.move_to_end(key)
            return plan
        if plan is not None:
            del self.plans[key]
        epoch = self.pool.epoch
        plan = self._build_plan(B, T, Lt)
        if self.pool.epoch != epoch:
            self.plans.clear()              # a buffer grew: every older plan points into the released allocation
        plan.pool_epoch = self.pool.epoch
        self.plans[key] = plan
        while len(self.plans) > MAX_PLANS:
            self.plans.popitem(last=False)
        return plan

    def _enqueue(self, plan, image, u8, norm, pos_embed, B, T, Lt, H, W, profile, upto=None, lo=0, no_input=False, key_bias=True):
        """Every launch of one call, in order, on torch's current stream (also the body of the captured graph).
        lo / no_input: the launches from plan.steps[lo] on, without the patch gather (the Q-Former half of a two-device
        placement starts from image_embeds already in plan.ws["xn"])."""
        cfg, ws = self.cfg, plan.ws
        stream = torch.cuda.current_stream().cuda_stream
        if key_bias:
            L.check(self.lib.hk_qformer_key_bias(ws["mask"].data_ptr() if Lt > 0 else None, ws["kbias"].data_ptr(), B,
                                                 cfg.n_query, Lt, stream), "hk_qformer_key_bias")
        if no_input:
            pass
        elif u8:
            mean, std = norm if norm is not None else (ops.CLIP_MEAN, ops.CLIP_STD)
            m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
            L.check(self.lib.hk_patchify_u8(image.data_ptr(), ws["patches"].data_ptr(), B, T, H, W, cfg.patch_size,
                                            m3, s3, stream), "hk_patchify_u8")
        else:
            L.check(self.lib.hk_patchify_bf16(image.data_ptr(), ws["patches"].data_ptr(), B, T, H, W,
                                              cfg.patch_size, stream), "hk_patchify_bf16")
        if not no_input:
            plan.patch_desc.residual = pos_embed.data_ptr()
        if profile is None and upto is None and self.side_stream and plan.side:
            # text-branch segments on the side stream: fork event recorded on the main stream at the fork point, the
            # segment waits for it, the main stream waits for the segment's end event before the step that follows it
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            side, side_h = self._side, self._side.cuda_stream
            while len(plan.side_events) < 2 * len(plan.side):
                plan.side_events.append(torch.cuda.Event())
            forks = {f: k for k, (f, _, _) in enumerate(plan.side)}
            starts = {a: k for k, (_, a, _) in enumerate(plan.side)}
            ends = {e: k for k, (_, _, e) in enumerate(plan.side)}
            cur = None
            for idx, (fn, args, name, _) in enumerate(plan.steps):
                if idx < lo:
                    continue
                if idx in ends:
                    k = ends[idx]
                    plan.side_events[2 * k + 1].record(side)
                    main.wait_event(plan.side_events[2 * k + 1])
                    cur = None
                if idx in forks:
                    plan.side_events[2 * forks[idx]].record(main)
                if idx in starts:
                    cur = starts[idx]
                    side.wait_event(plan.side_events[2 * cur])
                rc = fn(*args, side_h if cur is not None else stream)
                if rc != 0:
                    L.check(rc, name)
        elif profile is None:
            for fn, args, name, _ in plan.steps[lo:upto]:
                rc = fn(*args, stream)
                if rc != 0:
                    L.check(rc, name)
        else:
            # one HIP event pair per launch on the launch stream; resolved by profile()
            for fn, args, name, tag in plan.steps:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(*args, stream)
                e1.record()
                if rc != 0:
                    L.check(rc, name)
                profile.append((tag, e0, e1))

    def profile(self, image, input_ids, text_mask, pos_embed, reps: int = 3) -> Dict[str, Tuple[int, float]]:
        """In-situ launch durations: {tag: (launches per call, mean us per launch)} from HIP events
        recorded around every launch of `reps` forward calls (same stream, same order as forward)."""
        acc: Dict[str, List[float]] = {}
        for _ in range(reps):
            rec: List = []
            self.forward(image, input_ids, text_mask, pos_embed, profile=rec)
            torch.cuda.synchronize(self.device)
            for tag, e0, e1 in rec:
                acc.setdefault(tag, []).append(e0.elapsed_time(e1) * 1e3)
        return {t: (len(v) // reps, sum(v) / len(v)) for t, v in acc.items()}

    def workspace_bytes(self, B: int, T: int, Lt: int) -> int:
        """Bytes of the pooled workspace the plan of this shape uses (0 if it was never built)."""
        plan = self.plans.get((B, T, Lt))
        return 0 if plan is None else sum(t.numel() * t.element_size() for t in plan.ws.values())

    def pooled_bytes(self) -> int:
        """Device memory the engine holds for workspace, whatever the number of shapes it has seen."""
        return self.pool.nbytes()


def flops_per_clip(cfg: EncoderConfig, T: int, Lt: int) -> float:
    """Algorithmic FLOPs of one clip (2 flops per MAC, softmax/LN/GELU excluded)."""
    N = T * cfg.patches_per_frame
    D, d = cfg.encoder_embed_dim, cfg.qf_hidden
    S = cfg.n_query + Lt
    nq = cfg.n_query
    inter = cfg.qf_intermediate
    Hm = D * cfg.mlp_ratio
    f_vit = 2 * N * cfg.patch_k * D + cfg.run_depth * (2 * N * D * 3 * D + 4 * N * N * D + 2 * N * D * D + 4 * N * D * Hm)
    n_cross = len([i for i in range(cfg.qf_layers) if i % cfg.cross_attention_freq == 0])
    f_qf = cfg.qf_layers * (8 * S * d * d + 4 * S * S * d + 4 * nq * d * inter + 4 * Lt * d * inter) + \
        n_cross * (4 * nq * d * d + 4 * N * D * d + 4 * nq * N * d)
    f_proj = 2 * nq * d * cfg.llama_hidden
    return float(f_vit + f_qf + f_proj)
