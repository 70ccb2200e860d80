"""Row f2: the stage-3 training step of the encode_img path (frozen ViT, trainable Q-Former / query tokens /
llama_proj: models/hawkeye_it.py:82-99, tasks/train_it.py:63-80) on MI355X.

`QFormerTrainEngine` sits beside `HipEncoderEngine`: the ViT half of a call is the inference plan unchanged
(frozen, eval mode, no gradient), the Q-Former half is replaced by a forward that applies dropout / drop-path
(Qformer.py:120,271,315-316,403-404) and keeps what the backward needs, and by the backward itself.  Everything is a
flat list of C-ABI calls per (B, T, Lt) plan, enqueued on the current HIP stream; no PyTorch arithmetic.

Memory design (sized for 288 GB, not for reuse):
  * trainable parameters live in ONE flat fp32 arena (`ParamArena`): every nn.Parameter of the Q-Former, the query
    tokens and llama_proj is a view into it, ordered so that tensors the kernels fuse are adjacent (q|k|v weights and
    biases of a layer, LayerNorm weight|bias, query|extra query tokens, the key|value projections of all cross
    layers).  Gradients live in a second arena with the same layout, in backward-completion order, so a gradient
    all-reduce can start on finished ranges while earlier layers are still in backward (hawkeye_amd/dist.py).
  * one launch casts the whole parameter arena to a bf16 mirror (the forward GEMM operands are views of it);
    the transposed operands of the dgrad GEMMs are made per matrix by hk_transpose_bf16.
  * every activation the backward needs is kept per layer (a few MB per layer at the shipped batch sizes); dropout
    masks are never stored: the backward regenerates them from (seed, site, index).
"""
import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L
from . import ops
from .config import EncoderConfig
from .engine import HipEncoderEngine, LOG2E, _Pool

# dropout-site kinds inside one BertLayer (the oracle and the golden generator use the same numbering)
SELF_PROB, SELF_HID, SELF_PATH, CROSS_PROB, CROSS_HID, CROSS_PATH, QFFN_HID, QFFN_PATH, TFFN_HID, TFFN_PATH = range(10)
EMBED_SITE = 4095


def site(layer: int, kind: int) -> int:
    return layer * 16 + kind


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def trainable_names(cfg: EncoderConfig) -> List[str]:
    """Arena order: reverse of the forward (= order in which the backward finishes gradients), fused groups adjacent."""
    names = ["llama_proj.weight", "llama_proj.bias"]
    q = "qformer.bert."
    cross = [i for i in range(cfg.qf_layers) if i % cfg.cross_attention_freq == 0]
    for i in reversed(range(cfg.qf_layers)):
        l = f"{q}encoder.layer.{i}."
        for blk in ("output_query", "output"):
            inter = "intermediate_query" if blk == "output_query" else "intermediate"
            names += [l + blk + ".dense.weight", l + blk + ".dense.bias", l + blk + ".LayerNorm.weight",
                      l + blk + ".LayerNorm.bias", l + inter + ".dense.weight", l + inter + ".dense.bias"]
        if i in cross:
            c = l + "crossattention."
            names += [c + "output.dense.weight", c + "output.dense.bias", c + "output.LayerNorm.weight",
                      c + "output.LayerNorm.bias", c + "self.query.weight", c + "self.query.bias"]
        a = l + "attention."
        names += [a + "output.dense.weight", a + "output.dense.bias", a + "output.LayerNorm.weight",
                  a + "output.LayerNorm.bias", a + "self.query.weight", a + "self.key.weight", a + "self.value.weight",
                  a + "self.query.bias", a + "self.key.bias", a + "self.value.bias"]
    e = q + "embeddings."
    names += [e + "LayerNorm.weight", e + "LayerNorm.bias", e + "position_embeddings.weight", e + "word_embeddings.weight",
              "query_tokens"]
    if cfg.extra_num_query_token > 0:
        names.append("extra_query_tokens")
    for i in cross:
        c = f"{q}encoder.layer.{i}.crossattention.self."
        names += [c + "key.weight", c + "value.weight"]
    for i in cross:
        c = f"{q}encoder.layer.{i}.crossattention.self."
        names += [c + "key.bias", c + "value.bias"]
    return names


class ParamArena:
    """Flat fp32 storage of the trainable parameters (+ same-layout gradient arena and bf16 mirror)."""

    def __init__(self, params: Dict[str, torch.nn.Parameter], names: List[str], device):
        self.names = names
        self.params = [params[n] for n in names]
        self.offset: Dict[str, int] = {}
        off = 0
        for n, p in zip(names, self.params):
            if p.numel() % 8 != 0:
                raise RuntimeError(f"{n}: {p.numel()} elements; the arena needs multiples of 8")
            self.offset[n] = off
            off += p.numel()
        self.total = off
        self.flat = torch.empty(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.bf16 = torch.empty(off, dtype=torch.bfloat16, device=device)
        self.adopt()

    def adopt(self):
        """Move every parameter's storage into the arena (values kept).  `.grad` is not touched here: gradients reach
        the parameters as views of the gradient arena through the autograd node (model._EncodeImgFn) or through
        bind_grads()."""
        with torch.no_grad():
            for n, p in zip(self.names, self.params):
                o = self.offset[n]
                view = self.flat[o:o + p.numel()].view(p.shape)
                if p.data_ptr() != view.data_ptr():
                    view.copy_(p.data.to(device=self.flat.device, dtype=torch.float32))
                    p.data = view

    def attached(self) -> bool:
        return all(p.data_ptr() == self.flat.data_ptr() + 4 * self.offset[n] for n, p in zip(self.names, self.params))

    def version(self) -> int:
        return sum(p._version for p in self.params)

    def grad_view(self, n: str, p=None) -> torch.Tensor:
        p = p if p is not None else self.params[self.names.index(n)]
        o = self.offset[n]
        return self.grad[o:o + p.numel()].view(p.shape)

    def grad_aliases(self, n: str, p) -> bool:
        """True if p.grad is the arena's own memory for this parameter (autograd adopted the returned view)."""
        return p.grad is not None and p.grad.data_ptr() == self.g32(n) and p.grad.is_contiguous()

    def bind_grads(self, names=None):
        """Make `p.grad` the arena view of every parameter in `names` (default: every parameter that requires grad).
        A gradient that lives elsewhere (autograd cloned it, or accumulated into its own tensor) is copied in first,
        so afterwards the flat arena IS what the optimizer reads: the precondition of a flat all-reduce."""
        with torch.no_grad():
            for n, p in zip(self.names, self.params):
                if not p.requires_grad or (names is not None and n not in names):
                    continue
                if self.grad_aliases(n, p):
                    continue
                view = self.grad_view(n, p)
                if p.grad is None:
                    view.zero_()
                else:
                    view.copy_(p.grad)
                p.grad = view

    def p32(self, n: str) -> int:
        return self.flat.data_ptr() + 4 * self.offset[n]

    def p16(self, n: str) -> int:
        return self.bf16.data_ptr() + 2 * self.offset[n]

    def g32(self, n: str) -> int:
        return self.grad.data_ptr() + 4 * self.offset[n]


class _TrainPlan:
    def __init__(self, pool: Optional[_Pool] = None):
        self.pool = pool
        self.pool_epoch = pool.epoch if pool is not None else 0
        self.ws: Dict[str, torch.Tensor] = {}
        self.fwd: List[Tuple] = []
        self.bwd: List[Tuple] = []
        self.keep: List = []
        self.drops: List[Tuple] = []      # (HkDropout, p, p_path)
        self.attns: List[Tuple] = []      # (HkAttnTrainDesc, p)
        self.marks: List[Tuple[int, int]] = []   # (index into bwd after which ..., gradient arena offset finished)
        self.stamp: Optional[Tuple[int, int]] = None   # (inference-engine calls, training forwards) after this plan's forward
        self.unused: frozenset = frozenset()     # trainable parameters this shape never touches (gradient None)
        # Weight gradients beside the dgrad chain (a second stream): bwd index of a wgrad step -> name of the dY buffer it reads;
        # bwd index of a step that OVERWRITES such a buffer -> its name (that step first waits for the wgrad still reading it)
        self.bwd_side: Dict[int, str] = {}
        self.bwd_writes: Dict[int, str] = {}
        self.bwd_events: Dict[int, Tuple] = {}   # wgrad step -> (fork event, done event), made on first use

    def buf(self, name, shape, dtype, device):
        t = self.pool.view(name, shape, dtype) if self.pool is not None else torch.empty(shape, dtype=dtype, device=device)
        self.ws[name] = t
        return t


class QFormerTrainEngine:
    def __init__(self, enc: HipEncoderEngine, params: Dict[str, torch.nn.Parameter], p_hidden: float = 0.1,
                 p_attn: float = 0.1, drop_path_rate: float = 0.2):
        """enc: the inference engine of the same model (its ViT plan and packed ViT weights are reused as they are).
        params: the model's trainable nn.Parameters by state-dict name.  Dropout probabilities as blip2.py:56-64."""
        self.enc, self.cfg, self.device, self.lib = enc, enc.cfg, enc.device, enc.lib
        cfg = self.cfg
        if cfg.qf_hidden > 1024:
            raise RuntimeError("the training LayerNorm kernels hold rows of at most 1024 elements")
        self.p_hidden, self.p_attn = float(p_hidden), float(p_attn)
        # blip2.py:63: drop_path_list = [x.item() for x in torch.linspace(0, rate, num_hidden_layers)]
        self.path_rates = [x.item() for x in torch.linspace(0, drop_path_rate, cfg.qf_layers)]
        self.cross = [i for i in range(cfg.qf_layers) if i % cfg.cross_attention_freq == 0]
        with torch.cuda.device(self.device):
            self.arena = ParamArena(params, trainable_names(cfg), self.device)
        self.wt: Dict[str, torch.Tensor] = {}          # transposed bf16 operands of the dgrad GEMMs
        self._pack_steps: List[Tuple] = []
        self._packed_version = None
        self.plans: Dict[Tuple[int, int, int], _TrainPlan] = {}
        # saved activations and backward scratch of every shape share one grow-only pool, like the inference workspace
        # (engine._Pool): one training call is in flight per engine, whatever its shape
        self.pool = _Pool(self.device)
        self.forwards = 0
        # HK_WGRAD_SIDE=1: weight gradients on a second stream beside the dgrad chain (fork / join by events, same bits:
        # tests/test_gpu_train.py).  MEASURED SLOWER at the stage-3 shape and therefore off: B = 2, T = 12 step 12.28 -> 12.58 ms,
        # backward 3.58 -> 3.63 ms (device span unchanged, host enqueue 1.0 -> 3.1 ms for the 168 event calls): the wgrad
        # launches are 7-9 us on 36-144 workgroups and the cross-queue event hand-offs cost the chain as much as they free.
        self.wgrad_side = os.environ.get("HK_WGRAD_SIDE", "0") == "1"
        self._side: Optional["torch.cuda.Stream"] = None
        self._build_pack()

    # ------------------------------------------------------------------ weights
    def _lin_names(self):
        """(key, [weight names fused along the output dim], in_features) of every trainable Linear."""
        cfg, q = self.cfg, "qformer.bert."
        out = [("proj", ["llama_proj.weight"], cfg.qf_hidden)]
        for i in range(cfg.qf_layers):
            l = f"{q}encoder.layer.{i}."
            a = l + "attention."
            out.append((f"{i}.qkv", [a + "self.query.weight", a + "self.key.weight", a + "self.value.weight"], cfg.qf_hidden))
            out.append((f"{i}.ao", [a + "output.dense.weight"], cfg.qf_hidden))
            if i in self.cross:
                c = l + "crossattention."
                out.append((f"{i}.cq", [c + "self.query.weight"], cfg.qf_hidden))
                out.append((f"{i}.co", [c + "output.dense.weight"], cfg.qf_hidden))
            out.append((f"{i}.qi", [l + "intermediate_query.dense.weight"], cfg.qf_hidden))
            out.append((f"{i}.qo", [l + "output_query.dense.weight"], cfg.qf_intermediate))
            out.append((f"{i}.ti", [l + "intermediate.dense.weight"], cfg.qf_hidden))
            out.append((f"{i}.to", [l + "output.dense.weight"], cfg.qf_intermediate))
        return out

    def _build_pack(self):
        """Per optimizer step: one cast of the whole arena to bf16, one transpose per Linear (W^T for dgrad)."""
        ar, dev = self.arena, self.device
        steps = [(self.lib.hk_cast_f32_bf16, (ar.flat.data_ptr(), ar.bf16.data_ptr(), ar.total), "hk_cast_f32_bf16")]
        # HK_PACK_BATCH=0 (A/B aid): one hk_transpose_bf16 launch per weight, as in round 2
        batch = os.environ.get("HK_PACK_BATCH", "1") != "0"
        jobs, tile0 = [], 0
        with torch.cuda.device(dev):
            for key, wnames, k_in in self._lin_names():
                n_out = sum(ar.params[ar.names.index(n)].shape[0] for n in wnames)
                if n_out % 64 != 0 or k_in % 16 != 0:
                    raise RuntimeError(f"Linear {key}: out={n_out} must be a multiple of 64 and in={k_in} of 16")
                t = torch.zeros((k_in, n_out), dtype=torch.bfloat16, device=dev)
                self.wt[key] = t
                col = 0
                for n in wnames:
                    rows = ar.params[ar.names.index(n)].shape[0]
                    if batch:
                        jobs.append(L.HkTransposeJob(ar.p32(n), t.data_ptr() + 2 * col, k_in, n_out, rows, k_in, tile0, 0))
                        tile0 += -(-rows // 64) * -(-k_in // 64)
                    else:
                        steps.append((self.lib.hk_transpose_bf16,
                                      (ar.p32(n), L.HK_DT_F32, None, rows, k_in, t.data_ptr() + 2 * col, n_out), "hk_transpose_bf16"))
                    col += rows
            if batch:
                # the job table lives on the device (filled once: the arena and the W^T buffers never move)
                raw = (L.HkTransposeJob * len(jobs))(*jobs)
                host = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8)
                self._pack_jobs = host.to(dev)
                steps.append((self.lib.hk_transpose_f32_bf16_batch, (self._pack_jobs.data_ptr(), len(jobs), tile0),
                              "hk_transpose_f32_bf16_batch"))
        self._pack_steps = steps

    def pack(self, force: bool = False):
        """Refresh the bf16 operands if a parameter changed (optimizer step, load_state_dict)."""
        ar = self.arena
        if not ar.attached():
            raise RuntimeError("a trainable parameter no longer lives in the engine's arena (was the model moved with "
                               ".to()/.float() after the first training call?); call model.reset_train_engine()")
        v = ar.version()
        if not force and v == self._packed_version:
            return
        stream = torch.cuda.current_stream().cuda_stream
        for fn, args, name in self._pack_steps:
            rc = fn(*args, stream)
            if rc != 0:
                L.check(rc, name)
        self._packed_version = v

    # ------------------------------------------------------------------ plan helpers
    def _rows(self, plan, r):
        if r is None:
            return None
        s = L.HkRows(*r)
        plan.keep.append(s)
        return C.byref(s)

    def _drop(self, plan, p, st, p_path=0.0, path_site=0, rps=1):
        d = L.HkDropout()
        d.p, d.p_path, d.seed, d.site, d.path_site, d.rows_per_sample = 0.0, 0.0, 0, st, path_site, rps
        plan.drops.append((d, p, p_path))
        return C.byref(d)

    def _gemm(self, steps, plan, A, W, Cout, M, N, K, **kw):
        d = ops.gemm_desc(A, W, Cout, M, N, K, **kw)
        plan.keep.append(d)
        steps.append((self.lib.hk_gemm_bf16, (C.byref(d),), "hk_gemm_bf16"))

    def _build_plan(self, B: int, T: int, Lt: int) -> _TrainPlan:
        cfg, dev, ar, lib = self.cfg, self.device, self.arena, self.lib
        plan = _TrainPlan(self.pool)
        enc_plan = self.enc.plan_for(B, T, Lt)
        plan.enc_plan = enc_plan
        d, H, I, nq, D = cfg.qf_hidden, cfg.qf_heads, cfg.qf_intermediate, cfg.n_query, cfg.encoder_embed_dim
        Lyr = cfg.qf_layers
        N = T * cfg.patches_per_frame
        M, S = B * N, nq + Lt
        R, Rq, Rt = B * S, B * nq, B * Lt
        KVW = 2 * d * len(self.cross)
        LH = cfg.llama_hidden
        bf16, f32 = torch.bfloat16, torch.float32
        P = lambda t: t.data_ptr()  # noqa: E731
        plain = lambda ld: (ld, 0, 0)  # noqa: E731
        qrows, trows = (d, nq, S * d), (d, Lt, S * d)
        toff = nq * d
        qscale = (64 ** -0.5) * LOG2E
        sm_scale = 64 ** -0.5
        eps = cfg.qf_ln_eps
        q = "qformer.bert."
        with torch.cuda.device(dev):
            xn = enc_plan.ws["xn"]
            kvc = enc_plan.ws["kvc"]
            kbias = enc_plan.ws["kbias"]
            ids = enc_plan.ws["ids"]
            zemb = plan.buf("zemb", (R, d), f32, dev)
            hf = [plan.buf(f"hf{i}", (R, d), f32, dev) for i in range(Lyr + 1)]
            hb = [plan.buf(f"hb{i}", (R, d), bf16, dev) for i in range(Lyr + 1)]
            tf = plan.buf("tf", (R, d), f32, dev)
            out = plan.buf("out", (Rq, LH), f32, dev)
            sv = {}
            for i in range(Lyr):
                sv[i] = dict(
                    qkv=plan.buf(f"qkv{i}", (R, 3 * d), bf16, dev), ctx=plan.buf(f"ctx{i}", (R, d), bf16, dev),
                    ctx32=plan.buf(f"ctx32{i}", (R, d), f32, dev),
                    lse=plan.buf(f"lse{i}", (B, H, S), f32, dev), z1=plan.buf(f"z1{i}", (R, d), f32, dev),
                    af=plan.buf(f"af{i}", (R, d), f32, dev), ab=plan.buf(f"ab{i}", (R, d), bf16, dev),
                    uq=plan.buf(f"uq{i}", (Rq, I), bf16, dev), gq=plan.buf(f"gq{i}", (Rq, I), bf16, dev),
                    zq=plan.buf(f"zq{i}", (Rq, d), f32, dev))
                if i in self.cross:
                    sv[i].update(cq=plan.buf(f"cq{i}", (Rq, d), bf16, dev), cctx=plan.buf(f"cctx{i}", (Rq, d), bf16, dev),
                                 cctx32=plan.buf(f"cctx32{i}", (Rq, d), f32, dev),
                                 lsec=plan.buf(f"lsec{i}", (B, H, nq), f32, dev), zc=plan.buf(f"zc{i}", (Rq, d), f32, dev),
                                 cf=plan.buf(f"cf{i}", (Rq, d), f32, dev), cb=plan.buf(f"cb{i}", (Rq, d), bf16, dev))
                if Lt > 0 and i < Lyr - 1:
                    sv[i].update(ut=plan.buf(f"ut{i}", (Rt, I), bf16, dev), gt=plan.buf(f"gt{i}", (Rt, I), bf16, dev),
                                 zt=plan.buf(f"zt{i}", (Rt, d), f32, dev))
            # backward scratch
            dout_b = plan.buf("dout_b", (Rq, LH), bf16, dev)
            dh = plan.buf("dh", (R, d), f32, dev)
            da = plan.buf("da", (R, d), f32, dev)
            dc = plan.buf("dc", (Rq, d), f32, dev)
            dzs = plan.buf("dzs", (R, d), f32, dev)          # dz of the block being processed (residual gradient)
            do_b = plan.buf("do_b", (R, d), bf16, dev)
            dgi = plan.buf("dgi", (max(Rq, Rt, 1), I), bf16, dev)
            dui = plan.buf("dui", (max(Rq, Rt, 1), I), bf16, dev)
            dctx = plan.buf("dctx", (R, d), bf16, dev)
            dqkv = plan.buf("dqkv", (R, 3 * d), bf16, dev)
            dcq = plan.buf("dcq", (Rq, d), bf16, dev)
            dkvc = plan.buf("dkvc", (M, KVW), bf16, dev)
            dsum = plan.buf("dsum", (B, H, max(S, nq)), f32, dev)
            # HK_WGRAD_FUSED=0 (A/B aid): round 2's weight gradients -- transposed copies of dY and X, the inference GEMM on
            # them and a column-sum launch for the bias -- instead of the one-launch hk_wgrad_bf16
            fused_wgrad = os.environ.get("HK_WGRAD_FUSED", "1") != "0" and all(v % 128 == 0 for v in (d, I, LH, KVW, D))
            if not fused_wgrad:
                rmax = pad64(max(R, M))
                tA = plan.buf("tA", (max(3 * d, I, LH, KVW) * rmax,), bf16, dev)
                tB = plan.buf("tB", (max(d, I, D) * rmax,), bf16, dev)
                csws = plan.buf("csws", (L.HK_COLSUM_MAX_SPLIT * max(3 * d, I, LH, KVW),), f32, dev)
            lnws = plan.buf("lnws", (L.HK_LNBWD_MAX_BLOCKS * 2 * d,), f32, dev)

        def lname(i, s):
            return f"{q}encoder.layer.{i}.{s}"

        # ------------------------------------------------------------ forward (Q-Former half)
        F = plan.fwd
        e = q + "embeddings."
        self._gemm(F, plan, P(xn), ar.p16(lname(self.cross[0], "crossattention.self.key.weight")), P(kvc), M, KVW, D,
                   bias=ar.p32(lname(self.cross[0], "crossattention.self.key.bias")), a_rows=plain(D), c_rows=plain(KVW))
        F.append((lib.hk_qformer_embed_gather,
                  (ar.p32("query_tokens"), ar.p32(e + "word_embeddings.weight"), ar.p32(e + "position_embeddings.weight"),
                   P(ids) if Lt > 0 else None, P(zemb), B, nq, Lt, d, cfg.qf_vocab), "hk_qformer_embed_gather"))
        emb_post = self._drop(plan, self.p_hidden, EMBED_SITE)
        F.append((lib.hk_drop_add_ln, (None, P(zemb), None, None, emb_post, ar.p32(e + "LayerNorm.weight"),
                                       ar.p32(e + "LayerNorm.bias"), None, P(hf[0]), P(hb[0]), None, R, d, eps),
                  "hk_drop_add_ln"))

        def attn(steps, fn, **kw):
            dsc = L.HkAttnTrainDesc()
            for k_, v_ in kw.items():
                setattr(dsc, k_, v_)
            dsc.softmax_scale = sm_scale
            dsc.p_drop, dsc.seed = 0.0, 0
            plan.attns.append((dsc, self.p_attn))
            steps.append((fn, (C.byref(dsc),), "hk_attn_train"))

        def self_attn_args(i):
            s = sv[i]
            return dict(Q=P(s["qkv"]), K=P(s["qkv"]) + 2 * d, V=P(s["qkv"]) + 4 * d, O=P(s["ctx"]), lse=P(s["lse"]),
                        key_bias=P(kbias), B=B, H=H, Nq=S, Nk=S, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d,
                        bsq=S * 3 * d, bsk=S * 3 * d, bsv=S * 3 * d, bso=S * d, site=site(i, SELF_PROB), o_f32=P(s["ctx32"]))

        def cross_attn_args(i):
            s, ci = sv[i], self.cross.index(i)
            return dict(Q=P(s["cq"]), K=P(kvc) + ci * 2 * d * 2, V=P(kvc) + (ci * 2 + 1) * d * 2, O=P(s["cctx"]),
                        lse=P(s["lsec"]), key_bias=None, B=B, H=H, Nq=nq, Nk=N, ldq=d, ldk=KVW, ldv=KVW, ldo=d,
                        bsq=nq * d, bsk=N * KVW, bsv=N * KVW, bso=nq * d, site=site(i, CROSS_PROB), o_f32=P(s["cctx32"]))

        def dal(steps, t_ptr, res_ptr, res_rows, drop, gname, z_ptr, yf_ptr, yb_ptr, out_rows, rows):
            steps.append((lib.hk_drop_add_ln, (t_ptr, res_ptr, self._rows(plan, res_rows), drop, None, ar.p32(gname + ".weight"),
                                               ar.p32(gname + ".bias"), z_ptr, yf_ptr, yb_ptr, self._rows(plan, out_rows),
                                               rows, d, eps), "hk_drop_add_ln"))

        for i in range(Lyr):
            s = sv[i]
            a = lname(i, "attention.")
            pr = self.path_rates[i]
            self._gemm(F, plan, P(hb[i]), ar.p16(a + "self.query.weight"), P(s["qkv"]), R, 3 * d, d,
                       bias=ar.p32(a + "self.query.bias"), a_rows=plain(d), c_rows=plain(3 * d), scale_cols=d, scale=qscale)
            attn(F, lib.hk_attn_train_fwd, **self_attn_args(i))
            self._gemm(F, plan, P(s["ctx"]), ar.p16(a + "output.dense.weight"), P(tf), R, d, d,
                       bias=ar.p32(a + "output.dense.bias"), a_rows=plain(d), c_rows=plain(d), out_f32=True)
            dal(F, P(tf), P(hf[i]), None, self._drop(plan, self.p_hidden, site(i, SELF_HID), pr, site(i, SELF_PATH), S),
                a + "output.LayerNorm", P(s["z1"]), P(s["af"]), P(s["ab"]), None, R)
            if i in self.cross:
                c = lname(i, "crossattention.")
                self._gemm(F, plan, P(s["ab"]), ar.p16(c + "self.query.weight"), P(s["cq"]), Rq, d, d,
                           bias=ar.p32(c + "self.query.bias"), a_rows=qrows, c_rows=plain(d), scale_cols=d, scale=qscale)
                attn(F, lib.hk_attn_train_fwd, **cross_attn_args(i))
                self._gemm(F, plan, P(s["cctx"]), ar.p16(c + "output.dense.weight"), P(tf), Rq, d, d,
                           bias=ar.p32(c + "output.dense.bias"), a_rows=plain(d), c_rows=plain(d), out_f32=True)
                dal(F, P(tf), P(s["af"]), qrows, self._drop(plan, self.p_hidden, site(i, CROSS_HID), pr, site(i, CROSS_PATH), nq),
                    c + "output.LayerNorm", P(s["zc"]), P(s["cf"]), P(s["cb"]), None, Rq)
                qin_b, qin_f, qin_rows = P(s["cb"]), P(s["cf"]), plain(d)
            else:
                qin_b, qin_f, qin_rows = P(s["ab"]), P(s["af"]), qrows
            s["qin"] = (qin_b, qin_rows)
            self._gemm(F, plan, qin_b, ar.p16(lname(i, "intermediate_query.dense.weight")), P(s["uq"]), Rq, I, d,
                       bias=ar.p32(lname(i, "intermediate_query.dense.bias")), a_rows=qin_rows, c_rows=plain(I))
            F.append((lib.hk_gelu_bf16, (P(s["uq"]), P(s["gq"]), Rq * I), "hk_gelu_bf16"))
            self._gemm(F, plan, P(s["gq"]), ar.p16(lname(i, "output_query.dense.weight")), P(tf), Rq, d, I,
                       bias=ar.p32(lname(i, "output_query.dense.bias")), a_rows=plain(I), c_rows=plain(d), out_f32=True)
            dal(F, P(tf), qin_f, qin_rows, self._drop(plan, self.p_hidden, site(i, QFFN_HID), pr, site(i, QFFN_PATH), nq),
                lname(i, "output_query.LayerNorm"), P(s["zq"]), P(hf[i + 1]), P(hb[i + 1]), qrows, Rq)
            if "ut" in s:
                self._gemm(F, plan, P(s["ab"]) + toff * 2, ar.p16(lname(i, "intermediate.dense.weight")), P(s["ut"]), Rt, I, d,
                           bias=ar.p32(lname(i, "intermediate.dense.bias")), a_rows=trows, c_rows=plain(I))
                F.append((lib.hk_gelu_bf16, (P(s["ut"]), P(s["gt"]), Rt * I), "hk_gelu_bf16"))
                self._gemm(F, plan, P(s["gt"]), ar.p16(lname(i, "output.dense.weight")), P(tf), Rt, d, I,
                           bias=ar.p32(lname(i, "output.dense.bias")), a_rows=plain(I), c_rows=plain(d), out_f32=True)
                dal(F, P(tf), P(s["af"]) + toff * 4, trows,
                    self._drop(plan, self.p_hidden, site(i, TFFN_HID), pr, site(i, TFFN_PATH), Lt),
                    lname(i, "output.LayerNorm"), P(s["zt"]), P(hf[i + 1]) + toff * 4, P(hb[i + 1]) + toff * 2, trows, Rt)
        self._gemm(F, plan, P(hb[Lyr]), ar.p16("llama_proj.weight"), P(out), Rq, LH, d, bias=ar.p32("llama_proj.bias"),
                   a_rows=qrows, c_rows=plain(LH), out_f32=True)

        # ------------------------------------------------------------ backward
        Bw = plan.bwd
        # the bf16 gradient buffers the weight gradients read (each is rewritten block after block by the dgrad chain)
        dyname = {P(do_b): "do_b", P(dui): "dui", P(dcq): "dcq", P(dqkv): "dqkv", P(dout_b): "dout_b", P(dkvc): "dkvc"}

        def transpose(x_ptr, x_rows, rows, cols, out_ptr):
            Bw.append((lib.hk_transpose_bf16, (x_ptr, L.HK_DT_BF16, self._rows(plan, x_rows), rows, cols, out_ptr, pad64(rows)),
                       "hk_transpose_bf16"))

        def wgrad(dy_ptr, rows, n_out, x_ptr, x_rows, k_in, wname, bname):
            """dW[n_out, k_in] = dY^T X (contraction over `rows`), db = column sums of dY."""
            if fused_wgrad:
                # off the critical path: nothing in the backward reads a weight gradient, so the launch may run beside the
                # dgrad chain (plan.bwd_side; backward() forks / joins by events).  x operands are saved activations.
                plan.bwd_side[len(Bw)] = dyname[dy_ptr]
                Bw.append((lib.hk_wgrad_bf16, (dy_ptr, n_out, x_ptr, self._rows(plan, x_rows), ar.g32(wname), k_in, ar.g32(bname),
                                               rows, n_out, k_in), "hk_wgrad_bf16"))
                return
            transpose(dy_ptr, None, rows, n_out, P(tA))
            transpose(x_ptr, x_rows, rows, k_in, P(tB))
            rp = pad64(rows)
            self._gemm(Bw, plan, P(tA), P(tB), ar.g32(wname), n_out, k_in, rp, a_rows=plain(rp), c_rows=plain(k_in), out_f32=True)
            Bw.append((lib.hk_colsum, (dy_ptr, L.HK_DT_BF16, n_out, rows, n_out, ar.g32(bname), P(csws)), "hk_colsum"))

        def dgrad(dy_ptr, rows, n_out, key, k_in, c_ptr, c_rows, residual=None, r_rows=(0, 0, 0), out_f32=True):
            """dX[rows, k_in] = dY W (+ residual), through the transposed operand W^T [k_in, n_out]."""
            self._gemm(Bw, plan, dy_ptr, P(self.wt[key]), c_ptr, rows, k_in, n_out, a_rows=plain(n_out), c_rows=c_rows,
                       residual=residual, r_rows=r_rows, out_f32=out_f32)

        def lnbwd(dy_ptr, dy_rows, z_ptr, gname, drop, post, dz_ptr, dt_ptr, rows):
            if dt_ptr is not None:
                plan.bwd_writes[len(Bw)] = dyname[dt_ptr]
            Bw.append((lib.hk_layernorm_bwd, (dy_ptr, self._rows(plan, dy_rows), z_ptr, ar.p32(gname + ".weight"), drop, post,
                                              dz_ptr, dt_ptr, ar.g32(gname + ".weight"), ar.g32(gname + ".bias"), P(lnws),
                                              rows, d, eps), "hk_layernorm_bwd"))

        def ffn_bwd(i, rows, rps, dy_ptr, dy_rows, z, u, g, x_ptr, x_rows, blk, inter, kinds, key_o, key_i, dst_ptr, dst_rows):
            pr = self.path_rates[i]
            lnbwd(dy_ptr, dy_rows, P(z), lname(i, blk + ".LayerNorm"),
                  self._drop(plan, self.p_hidden, site(i, kinds[0]), pr, site(i, kinds[1]), rps), None, P(dzs), P(do_b), rows)
            wgrad(P(do_b), rows, d, P(g), None, I, lname(i, blk + ".dense.weight"), lname(i, blk + ".dense.bias"))
            dgrad(P(do_b), rows, d, key_o, I, P(dgi), plain(I), out_f32=False)
            plan.bwd_writes[len(Bw)] = "dui"
            Bw.append((lib.hk_gelu_bwd_bf16, (P(dgi), P(u), P(dui), rows * I), "hk_gelu_bwd_bf16"))
            wgrad(P(dui), rows, I, x_ptr, x_rows, d, lname(i, inter + ".dense.weight"), lname(i, inter + ".dense.bias"))
            dgrad(P(dui), rows, I, key_i, d, dst_ptr, dst_rows, residual=P(dzs), r_rows=plain(d))

        Bw.append((lib.hk_cast_f32_bf16, (None, P(dout_b), Rq * LH), "hk_cast_f32_bf16"))     # arg 0 bound at run time
        wgrad(P(dout_b), Rq, LH, P(hb[Lyr]), qrows, d, "llama_proj.weight", "llama_proj.bias")
        dgrad(P(dout_b), Rq, LH, "proj", d, P(dh), qrows)
        plan.marks.append((len(Bw), ar.offset[lname(Lyr - 1, "output_query.dense.weight")]))
        for i in reversed(range(Lyr)):
            s = sv[i]
            a = lname(i, "attention.")
            pr = self.path_rates[i]
            has_cross = i in self.cross
            qin_b, qin_rows = s["qin"]
            # query-branch FFN: gradient lands in dc (cross layers) or in the query rows of da
            ffn_bwd(i, Rq, nq, P(dh), qrows, s["zq"], s["uq"], s["gq"], qin_b, qin_rows, "output_query", "intermediate_query",
                    (QFFN_HID, QFFN_PATH), f"{i}.qo", f"{i}.qi", P(dc) if has_cross else P(da), plain(d) if has_cross else qrows)
            if "ut" in s:
                ffn_bwd(i, Rt, Lt, P(dh) + toff * 4, trows, s["zt"], s["ut"], s["gt"], P(s["ab"]) + toff * 2, trows, "output",
                        "intermediate", (TFFN_HID, TFFN_PATH), f"{i}.to", f"{i}.ti", P(da) + toff * 4, trows)
            if has_cross:
                c = lname(i, "crossattention.")
                lnbwd(P(dc), None, P(s["zc"]), c + "output.LayerNorm",
                      self._drop(plan, self.p_hidden, site(i, CROSS_HID), pr, site(i, CROSS_PATH), nq), None, P(dzs), P(do_b), Rq)
                wgrad(P(do_b), Rq, d, P(s["cctx"]), None, d, c + "output.dense.weight", c + "output.dense.bias")
                dgrad(P(do_b), Rq, d, f"{i}.co", d, P(dctx), plain(d), out_f32=False)
                ci = self.cross.index(i)
                plan.bwd_writes[len(Bw)] = "dcq"
                attn(Bw, lib.hk_attn_train_bwd, dO=P(dctx), dQ=P(dcq), dK=P(dkvc) + ci * 2 * d * 2,
                     dV=P(dkvc) + (ci * 2 + 1) * d * 2, dsum=P(dsum), lddq=d, lddk=KVW, lddv=KVW, bsdq=nq * d, bsdk=N * KVW,
                     bsdv=N * KVW, **cross_attn_args(i))
                wgrad(P(dcq), Rq, d, P(s["ab"]), qrows, d, c + "self.query.weight", c + "self.query.bias")
                dgrad(P(dcq), Rq, d, f"{i}.cq", d, P(da), qrows, residual=P(dzs), r_rows=plain(d))
            # self-attention output block and the attention itself
            lnbwd(P(da), None, P(s["z1"]), a + "output.LayerNorm",
                  self._drop(plan, self.p_hidden, site(i, SELF_HID), pr, site(i, SELF_PATH), S), None, P(dzs), P(do_b), R)
            wgrad(P(do_b), R, d, P(s["ctx"]), None, d, a + "output.dense.weight", a + "output.dense.bias")
            dgrad(P(do_b), R, d, f"{i}.ao", d, P(dctx), plain(d), out_f32=False)
            plan.bwd_writes[len(Bw)] = "dqkv"
            attn(Bw, lib.hk_attn_train_bwd, dO=P(dctx), dQ=P(dqkv), dK=P(dqkv) + 2 * d, dV=P(dqkv) + 4 * d, dsum=P(dsum),
                 lddq=3 * d, lddk=3 * d, lddv=3 * d, bsdq=S * 3 * d, bsdk=S * 3 * d, bsdv=S * 3 * d, **self_attn_args(i))
            wgrad(P(dqkv), R, 3 * d, P(hb[i]), None, d, a + "self.query.weight", a + "self.query.bias")
            dgrad(P(dqkv), R, 3 * d, f"{i}.qkv", d, P(dh), plain(d), residual=P(dzs), r_rows=plain(d))
            nxt = lname(i - 1, "output_query.dense.weight") if i > 0 else e + "LayerNorm.weight"
            plan.marks.append((len(Bw), ar.offset[nxt]))
        # embeddings: LayerNorm -> dropout (post), then the gather
        lnbwd(P(dh), None, P(zemb), e + "LayerNorm", None, self._drop(plan, self.p_hidden, EMBED_SITE), P(dzs), None, R)
        Bw.append((lib.hk_qformer_embed_bwd, (P(dzs), P(ids) if Lt > 0 else None, ar.g32("query_tokens"),
                                              ar.g32(e + "word_embeddings.weight"), ar.g32(e + "position_embeddings.weight"),
                                              B, nq, Lt, d, cfg.qf_vocab), "hk_qformer_embed_bwd"))
        # key / value projections of all cross layers in one wgrad (they all read the same image_embeds)
        kw0 = lname(self.cross[0], "crossattention.self.key.weight")
        kb0 = lname(self.cross[0], "crossattention.self.key.bias")
        plan.marks.append((len(Bw), ar.offset[kw0]))
        wgrad(P(dkvc), M, KVW, P(xn), None, D, kw0, kb0)
        plan.marks.append((len(Bw), ar.total))
        plan.sv = sv
        if Lt == 0:
            # qformer_text_input=False (hawkeye_it.py:261-267): the text FFNs and the word / position tables never enter
            # the graph, their gradients are None in the reference.  (With Lt > 0 every trainable parameter gets a
            # gradient there, the last layer's text FFN included -- zeros, since only [:, :96] is read:
            # tests/golden/train_small.npz `unused` is empty.)
            un = [e + "position_embeddings.weight", e + "word_embeddings.weight"]
            for i in range(Lyr):
                for blk, tail in (("output", ("dense.weight", "dense.bias", "LayerNorm.weight", "LayerNorm.bias")),
                                  ("intermediate", ("dense.weight", "dense.bias"))):
                    un += [lname(i, f"{blk}.{t_}") for t_ in tail]
            plan.unused = frozenset(un)
        return plan

    # ------------------------------------------------------------------ run
    def _arm(self, plan: _TrainPlan, train: bool, seed: int):
        for dsc, p, pp in plan.drops:
            dsc.p, dsc.p_path, dsc.seed = (p, pp, seed) if train else (0.0, 0.0, 0)
        for dsc, p in plan.attns:
            dsc.p_drop, dsc.seed = (p, seed) if train else (0.0, 0)

    def _run(self, steps, stream, lo=0, hi=None):
        for fn, args, name in steps[lo:hi]:
            rc = fn(*args, stream)
            if rc != 0:
                L.check(rc, name)

    @torch.no_grad()
    def forward(self, image, input_ids, text_mask, pos_embed, *, train: bool, seed: int = 0, norm=None, image_embeds=None):
        """Same inputs as HipEncoderEngine.forward.  train=True applies dropout / drop-path with `seed`.
        image_embeds (bf16 [B, N, D], any GPU): two-device placement (hawkeye_it.py:197-211, tasks/shared_utils.py:48-51) --
        the frozen ViT ran on another engine / device; this engine (whose `enc` then holds the Q-Former half only) starts
        from the handed-over vision_layernorm output and `image` / `pos_embed` are not read.
        Returns (out fp32 [B, nq, llama_hidden] -- a view of the plan's buffer --, plan)."""
        cfg = self.cfg
        if image_embeds is not None:
            B, T = int(image_embeds.shape[0]), int(image_embeds.shape[1]) // cfg.patches_per_frame
        else:
            B, T = int(image.shape[0]), int(image.shape[1])
        Lt = 0 if input_ids is None else int(input_ids.shape[1])
        key = (B, T, Lt)
        enc_plan = self.enc.plan_for(B, T, Lt)
        plan = self.plans.get(key)
        if plan is None or plan.enc_plan is not enc_plan or plan.pool_epoch != self.pool.epoch:
            # new shape, or the pooled buffers it points into moved (here or in the inference engine)
            epoch = self.pool.epoch
            plan = self._build_plan(B, T, Lt)
            if self.pool.epoch != epoch:
                self.plans.clear()
            plan.pool_epoch = self.pool.epoch
            self.plans[key] = plan
            while len(self.plans) > 64:
                self.plans.pop(next(iter(self.plans)))
        self.pack()
        with torch.cuda.device(self.device):
            # the frozen ViT: the inference plan's launches up to vision_layernorm (validation of the inputs included)
            got = (self.enc.load_image_embeds(image_embeds, input_ids, text_mask)[0] if image_embeds is not None else
                   self.enc.forward_vit(image, input_ids, text_mask, pos_embed, norm=norm))
            if got is not plan.enc_plan:
                raise RuntimeError("internal: the inference plan changed between lookup and launch")
            self._arm(plan, train, seed)
            plan.train, plan.seed = train, seed
            self._run(plan.fwd, torch.cuda.current_stream().cuda_stream)
        self.forwards += 1
        plan.stamp = (self.enc.calls, self.forwards)
        return plan.ws["out"].view(B, cfg.n_query, cfg.llama_hidden), plan

    def check_in_flight(self, plan: _TrainPlan, stamp=None):
        """The backward reads activations the forward left in pooled buffers (its own and the inference engine's
        image_embeds / cross K,V / key bias / ids): any encode_img call on this model since that forward -- training
        or inference, any shape -- has overwritten them."""
        stamp = plan.stamp if stamp is None else stamp      # a caller that kept the stamp of ITS forward passes it
        if stamp != (self.enc.calls, self.forwards) or plan.pool_epoch != self.pool.epoch:
            raise RuntimeError("encode_img was called again on this model between a training forward and its backward(): "
                               "the saved activations of that forward are gone (one in-flight training call per model; "
                               "run validation probes after backward, or on a second model instance)")

    @torch.no_grad()
    def backward(self, plan: _TrainPlan, dout: torch.Tensor, reducer=None):
        """Gradients of sum(out * dout) w.r.t. every trainable parameter -> the gradient arena (overwritten, not
        accumulated).  reducer: optional hawkeye_amd.dist.BucketReducer notified as arena ranges complete."""
        ar = self.arena
        self.check_in_flight(plan)
        if dout.device != self.device or dout.numel() != plan.ws["out"].numel():
            raise RuntimeError("backward: dout must be a GPU tensor shaped like encode_img's output")
        dout = dout.reshape(plan.ws["out"].shape).to(torch.float32).contiguous()
        with torch.cuda.device(self.device):
            self._arm(plan, plan.train, plan.seed)
            stream = torch.cuda.current_stream().cuda_stream
            ar.grad.zero_()
            plan.ws["dh"].zero_()
            plan.ws["da"].zero_()
            fn, args, name = plan.bwd[0]
            plan.bwd[0] = (fn, (dout.data_ptr(),) + tuple(args[1:]), name)
            side_on = self.wgrad_side and bool(plan.bwd_side)
            lo = 0
            if not side_on:
                for hi, done in plan.marks:
                    self._run(plan.bwd, stream, lo, hi)
                    lo = hi
                    if reducer is not None:
                        reducer.ready(done)
            else:
                # The dgrad chain (LayerNorm / GELU / attention backward, dX GEMMs) is a string of small dependent launches;
                # the weight gradients hang off it and nothing downstream reads them.  They go to a second stream: fork =
                # an event on the caller's stream where the wgrad stood (its dY is complete there), join = the step that
                # next overwrites that dY buffer waits for the wgrad's done-event; every gradient range handed to the
                # reducer, and the end of the call, wait for the side stream's last launch (it runs in order).
                main = torch.cuda.current_stream()
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                side, side_h = self._side, self._side.cuda_stream
                pending: Dict[str, "torch.cuda.Event"] = {}
                last = None
                for hi, done in plan.marks:
                    for idx in range(lo, hi):
                        fn, args, name = plan.bwd[idx]
                        key = plan.bwd_side.get(idx)
                        if key is not None:
                            ev = plan.bwd_events.get(idx)
                            if ev is None:
                                ev = plan.bwd_events[idx] = (torch.cuda.Event(), torch.cuda.Event())
                            ev[0].record(main)
                            side.wait_event(ev[0])
                            rc = fn(*args, side_h)
                            ev[1].record(side)
                            pending[key] = last = ev[1]
                        else:
                            w = plan.bwd_writes.get(idx)
                            if w is not None and w in pending:
                                main.wait_event(pending.pop(w))
                            rc = fn(*args, stream)
                        if rc != 0:
                            L.check(rc, name)
                    lo = hi
                    if reducer is not None:
                        if last is not None:
                            main.wait_event(last)
                            pending.clear()
                        reducer.ready(done)
                if last is not None:
                    main.wait_event(last)
            if reducer is not None:
                reducer.finish()
        return ar.grad
