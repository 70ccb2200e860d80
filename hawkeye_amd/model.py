"""Drop-in host-side mirror of the reference's encode_img surface.

`HawkEyeEncoder` exposes exactly what tasks/test.py, tasks/test_recursive_grounding.py,
tasks/train_it.py and demo.ipynb touch on `HawkEye_it` for this path:

  * the attribute tree and state-dict keys of models/hawkeye_it.py:57-80,183-185
    (`vision_encoder.encoder.{patch_embed.proj,blocks.N.*}`, `vision_layernorm`,
    `qformer.bert.*`, `query_tokens`, `extra_query_tokens`, `llama_proj`), so
    `load_state_dict(torch.load("hawkeye.pth"), strict=False)` works unchanged and
    silently drops the checkpoint's 24th ViT block like the reference does;
  * `vision_encoder.encoder.pos_embed` / `.img_pos_embed` as plain assignable
    tensors read at call time (demo.ipynb cell 6; vit.py:304-307);
  * `encode_img(image, instruction) -> (inputs_llama [B,96,4096], use_image)`
    (hawkeye_it.py:219-270), `query_tokens` registered first so `.device`
    (blip2.py:32-34) follows it, `set_device_ids`, `.tokenizer`.

The modules below are parameter containers only: they have no forward of their
own.  All arithmetic runs in libhawkeye_hip.so through HipEncoderEngine; there
is no PyTorch or CPU fallback, and encode_img raises if the image is not on an
MI355X.
"""
from types import SimpleNamespace
from typing import Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from . import ops, posembed
from .config import EncoderConfig
from .engine import HipEncoderEngine
from .train import QFormerTrainEngine, trainable_names


# Bumped whenever a module or parameter is (re)assigned on one of this file's container modules; HawkEyeEncoder keys its
# cached list of parameter slots on it (see HawkEyeEncoder._weights_key).
_STRUCTURE_VERSION = [0]


class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(f"{type(self).__name__} is a parameter container; run HawkEyeEncoder.encode_img")

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)):
            _STRUCTURE_VERSION[0] += 1
        super().__setattr__(name, value)


class _Attention(_NoForward):          # vit.py:57-79
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.proj = nn.Linear(dim, dim)


class _Mlp(_NoForward):                # vit.py:38-46
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(_NoForward):              # vit.py:103-122 (init_values=0 => no gamma_1/gamma_2)
    def __init__(self, dim, hidden, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attention(dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, hidden)


class _PatchEmbed(_NoForward):         # vit.py:134-150
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        p = cfg.patch_size
        self.proj = nn.Conv3d(3, cfg.encoder_embed_dim, kernel_size=(1, p, p), stride=(1, p, p))


class _VitEncoder(_NoForward):         # vit.py:244-292
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        D = cfg.encoder_embed_dim
        self.patch_embed = _PatchEmbed(cfg)
        self.depth = cfg.run_depth
        # plain tensor attributes, not buffers/parameters (vit.py:278-279)
        self.pos_embed = posembed.video_pos_embed(cfg.num_frames, cfg.patches_per_frame, D, cfg.ckpt_num_frame)
        self.img_pos_embed = posembed.image_pos_embed(cfg.patches_per_frame, D)
        self.blocks = nn.ModuleList([_Block(D, D * cfg.mlp_ratio, cfg.vit_ln_eps) for _ in range(self.depth)])
        self.norm = nn.Identity()


class _VisionTransformer(_NoForward):  # vit.py:327-384
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.encoder = _VitEncoder(cfg)


class _BertEmbeddings(_NoForward):     # Qformer.py:52-78
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg.qf_vocab, cfg.qf_hidden, padding_idx=0)
        self.position_embeddings = nn.Embedding(cfg.qf_max_pos, cfg.qf_hidden)
        self.LayerNorm = nn.LayerNorm(cfg.qf_hidden, eps=cfg.qf_ln_eps)
        self.register_buffer("position_ids", torch.arange(cfg.qf_max_pos).expand((1, -1)))


class _BertSelfAttention(_NoForward):  # Qformer.py:124-146
    def __init__(self, d, kv_width):
        super().__init__()
        self.query = nn.Linear(d, d)
        self.key = nn.Linear(kv_width, d)
        self.value = nn.Linear(kv_width, d)


class _BertDenseLN(_NoForward):        # Qformer.py:305-311 / 393-399
    def __init__(self, d_in, d, eps):
        super().__init__()
        self.dense = nn.Linear(d_in, d)
        self.LayerNorm = nn.LayerNorm(d, eps=eps)


class _BertAttention(_NoForward):      # Qformer.py:321-326
    def __init__(self, d, kv_width, eps):
        super().__init__()
        self.self = _BertSelfAttention(d, kv_width)
        self.output = _BertDenseLN(d, d, eps)


class _BertIntermediate(_NoForward):   # Qformer.py:378-385
    def __init__(self, d, inter):
        super().__init__()
        self.dense = nn.Linear(d, inter)


class _BertLayer(_NoForward):          # Qformer.py:409-433
    def __init__(self, cfg: EncoderConfig, layer_num: int):
        super().__init__()
        d, eps = cfg.qf_hidden, cfg.qf_ln_eps
        self.attention = _BertAttention(d, d, eps)
        self.has_cross_attention = layer_num % cfg.cross_attention_freq == 0
        if self.has_cross_attention:
            self.crossattention = _BertAttention(d, cfg.encoder_embed_dim, eps)
        self.intermediate = _BertIntermediate(d, cfg.qf_intermediate)
        self.output = _BertDenseLN(cfg.qf_intermediate, d, eps)
        self.intermediate_query = _BertIntermediate(d, cfg.qf_intermediate)
        self.output_query = _BertDenseLN(cfg.qf_intermediate, d, eps)


class _BertEncoder(_NoForward):        # Qformer.py:520-526
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.layer = nn.ModuleList([_BertLayer(cfg, i) for i in range(cfg.qf_layers)])


class _BertModel(_NoForward):          # Qformer.py:710-730
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.embeddings = _BertEmbeddings(cfg)
        self.encoder = _BertEncoder(cfg)


class _QFormer(_NoForward):            # Qformer.py:1001 BertLMHeadModel with cls=None (hawkeye_it.py:73)
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.config = SimpleNamespace(hidden_size=cfg.qf_hidden, num_hidden_layers=cfg.qf_layers,
                                      num_attention_heads=cfg.qf_heads, intermediate_size=cfg.qf_intermediate,
                                      vocab_size=cfg.qf_vocab, encoder_width=cfg.encoder_embed_dim,
                                      query_length=cfg.num_query_token, cross_attention_freq=cfg.cross_attention_freq)
        self.bert = _BertModel(cfg)
        self.cls = None

    def resize_token_embeddings(self, n: int):
        """hawkeye_it.py:72 -- keep existing rows, N(0, 0.02) for new ones."""
        old = self.bert.embeddings.word_embeddings
        if n == old.num_embeddings:
            return old
        new = nn.Embedding(n, old.embedding_dim, padding_idx=0).to(old.weight.device, old.weight.dtype)
        new.weight.data.normal_(mean=0.0, std=0.02)
        k = min(n, old.num_embeddings)
        new.weight.data[:k] = old.weight.data[:k]
        self.bert.embeddings.word_embeddings = new
        self.config.vocab_size = n
        return new


def disabled_train(self, mode=True):   # blip2.py:90-93
    return self


class _Embeds:
    """image_embeds handed over from the ViT's device in a two-device placement (what _EncodeImgFn gets instead of the clip)."""

    def __init__(self, t: torch.Tensor):
        self.t = t


class _EncodeImgFn(torch.autograd.Function):
    """encode_img as one autograd node (row f2): forward = the HIP training forward, backward = the HIP backward.
    The trainable parameters are inputs of the node, so autograd (and DistributedDataParallel's hooks,
    tasks/shared_utils.py:54-59) see real gradients for real nn.Parameters.

    The gradients the node returns are VIEWS of the engine's flat gradient arena (no 741 MB copy per step).  Autograd
    either adopts such a view as `p.grad` or adds it into an existing `p.grad`; the arena is overwritten by the next
    backward, so a `p.grad` that still aliases it when the next backward starts (gradient accumulation,
    zero_grad(set_to_none=False)) is moved to a tensor of its own first."""

    @staticmethod
    def forward(ctx, model, image, ids, mask, table, norm, train, seed, *params):
        eng = model._train_engine
        if isinstance(image, _Embeds):
            out, plan = eng.forward(None, ids, mask, None, train=train, seed=seed, image_embeds=image.t)
        else:
            out, plan = eng.forward(image, ids, mask, table, train=train, seed=seed, norm=norm)
        ctx.model, ctx.plan, ctx.engine, ctx.stamp = model, plan, eng, plan.stamp
        return out.clone()

    @staticmethod
    def backward(ctx, dout):
        eng, plan, model = ctx.engine, ctx.plan, ctx.model
        if eng is not model._train_engine:
            raise RuntimeError("the training engine was rebuilt between forward and backward()")
        eng.check_in_flight(plan, ctx.stamp)
        ar = eng.arena
        with torch.no_grad():
            for n, p in zip(ar.names, ar.params):
                if ar.grad_aliases(n, p):
                    p.grad = p.grad.clone()
        reducer = model._grad_reducer()
        eng.backward(plan, dout, reducer)
        used = [p.requires_grad and n not in plan.unused for n, p in zip(ar.names, ar.params)]
        if reducer is not None and reducer.active:
            from . import dist as hdist
            used = [u and p.requires_grad for u, p in zip(hdist.used_anywhere(used, ar.grad.device), ar.params)]
        grads = tuple(ar.grad_view(n, p) if u else None for (n, p), u in zip(zip(ar.names, ar.params), used))
        return (None,) * 8 + grads


class HawkEyeEncoder(nn.Module):
    """The hot-path half of HawkEye_it (models/hawkeye_it.py:15-270) on MI355X."""

    def __init__(self, cfg: EncoderConfig = EncoderConfig(), tokenizer=None, max_txt_len: int = 512,
                 out_dtype: torch.dtype = torch.float16, residual_dtype: torch.dtype = torch.bfloat16,
                 freeze_vit: bool = True, freeze_qformer: bool = True):
        """freeze_vit / freeze_qformer: hawkeye_it.py:26-27,82-99 (both default True there; stage 3 trains with
        freeze_vit=True, freeze_qformer=False, scripts/train/config_7b_stage3.py:40-41; the eval config the two test
        scripts load, configs/config.json:6-7, sets both False and then runs under model.eval() + torch.no_grad()).
        freeze_vit=False constructs like the reference: the ViT parameters keep requires_grad=True and the ViT is
        not pinned to eval mode.  Inference is the same launch sequence either way (the ViT has no dropout and
        drop-path 0, vit.py:118-119, config.json:18); only a call that would have to produce a ViT GRADIENT (grad
        mode on and a ViT parameter requiring grad) is refused: the path has no ViT backward."""
        super().__init__()
        self.cfg = cfg
        # registered first: Blip2Base.device is "device of the first parameter" (blip2.py:32-34)
        self.query_tokens = nn.Parameter(torch.zeros(1, cfg.num_query_token, cfg.qf_hidden))
        self.query_tokens.data.normal_(mean=0.0, std=0.02)
        self.vision_encoder = _VisionTransformer(cfg)
        self.vision_layernorm = nn.LayerNorm(cfg.encoder_embed_dim, eps=cfg.vision_ln_eps)
        self.qformer = _QFormer(cfg)
        self.extra_num_query_token = cfg.extra_num_query_token
        if cfg.extra_num_query_token > 0:
            self.extra_query_tokens = nn.Parameter(torch.zeros(1, cfg.extra_num_query_token, cfg.qf_hidden))
        self.llama_proj = nn.Linear(cfg.qf_hidden, cfg.llama_hidden)
        self.tokenizer = tokenizer
        self.max_txt_len = max_txt_len
        self.qformer_text_input = True
        self.low_resource = False
        self.residual_dtype = residual_dtype   # ViT residual-stream storage (see HipEncoderEngine)
        self.out_dtype = out_dtype        # the reference returns fp16 under its autocast (blip2.py:42)
        self.devices = [torch.device("cuda", 0), torch.device("cuda", 0)]
        self._engine: Optional[HipEncoderEngine] = None
        self._vit_engine: Optional[HipEncoderEngine] = None    # two-device placement only (set_device_ids([a, b]))
        self._force_split = False       # tests: run the two-engine path with both engines on ONE device
        self._engine_key = {"vit": None, "qformer": None}
        self._train_engine: Optional[QFormerTrainEngine] = None
        self._overlap = None      # (average, bucket_mb) once enable_overlapped_grad_reduce() was called
        self._reducer = None
        # Q-Former dropout (blip2.py:56-64; config.json:29-31).  Active iff self.qformer.training, like nn.Dropout.
        self.qformer_hidden_dropout_prob = 0.1
        self.qformer_attention_probs_dropout_prob = 0.1
        self.qformer_drop_path_rate = 0.2
        self._drop_seed, self._drop_calls = 0, 0
        self._table_cache = {}
        # hawkeye_it.py:82-99
        self.freeze_vit, self.freeze_qformer = bool(freeze_vit), bool(freeze_qformer)
        if freeze_vit:
            for prm in list(self.vision_encoder.parameters()) + list(self.vision_layernorm.parameters()):
                prm.requires_grad = False
            self.vision_encoder = self.vision_encoder.eval()
            self.vision_encoder.train = disabled_train.__get__(self.vision_encoder)
            self.vision_layernorm = self.vision_layernorm.eval()
            self.vision_layernorm.train = disabled_train.__get__(self.vision_layernorm)
        if freeze_qformer:
            for prm in self.qformer.parameters():
                prm.requires_grad = False
            self.qformer = self.qformer.eval()
            self.qformer.train = disabled_train.__get__(self.qformer)
            self.query_tokens.requires_grad = False
        # row f4: True = ignore the pos_embed attributes and generate the sinusoid table for the clip's frame
        # count on the device (cached per T); False (default) = the reference contract, read the attribute
        self.device_pos_embed = False

    # ---- reference module protocol -------------------------------------------------
    @property
    def device(self):
        # blip2.py:43-44: the device of the module's FIRST parameter (query_tokens), without listing all 700 of them
        return next(self.parameters()).device

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)):
            _STRUCTURE_VERSION[0] += 1
        super().__setattr__(name, value)

    def invalidate_parameter_cache(self):
        """Forget the cached parameter slots (see _weights_key).  Needed only after replacing an ITEM of a ModuleList
        (`blocks[i] = ...`, `layer[i] = ...`); every other replacement or in-place change is seen without it."""
        _STRUCTURE_VERSION[0] += 1

    def set_device_ids(self, device_ids):           # hawkeye_it.py:197-211
        """One id: everything on that GPU.  Two ids: the reference's optional split -- vision_encoder +
        vision_layernorm on devices[1], Q-Former / query tokens / llama_proj on devices[0]; encode_img then runs the
        ViT half on devices[1] and hands image_embeds (bf16 [B, N, 1024]) over to devices[0] (hawkeye_it.py:230-234).
        A 288 GB MI355X never needs it; it is there because the reference's trainer exposes it (model_parallel,
        tasks/shared_utils.py:48-51).  Training works the same way (round 3): the frozen ViT runs on devices[1], the
        training engine on devices[0] starts from the handed-over image_embeds."""
        assert isinstance(device_ids, list) and 1 <= len(device_ids) <= 2
        if len(device_ids) == 1:
            device_ids = [device_ids[0], device_ids[0]]
        self.devices = [torch.device("cuda", i) if isinstance(i, int) else torch.device(i) for i in device_ids]
        self.to(self.devices[0])
        if self.devices[1] != self.devices[0]:
            self.vision_encoder.to(self.devices[1])
            self.vision_layernorm.to(self.devices[1])
        self._engine = self._vit_engine = None
        self._engine_key = {"vit": None, "qformer": None}
        self._train_engine = None

    def _split_placement(self) -> bool:
        return self._force_split or next(self.vision_layernorm.parameters()).device != self.device

    def _sync_split_engines(self, train: bool = False):
        """Two engines, one per half, each on the device its parameters live on.  train=True: the Q-Former half's operands
        are the training engine's arena, so only the ViT half is (re)packed here."""
        dq = self.device
        dv = next(self.vision_layernorm.parameters()).device
        if dq.type != "cuda" or dv.type != "cuda":
            raise RuntimeError("HawkEyeEncoder.encode_img needs both halves on MI355X devices; there is no CPU path "
                               "(low_resource / vit_to_cpu of the reference are not offered)")
        if (self._engine is None or self._engine.device != dq or self._engine.halves != ("qformer",)
                or self._vit_engine is None or self._vit_engine.device != dv):
            self._engine = HipEncoderEngine(self.cfg, dq, self.residual_dtype, halves=("qformer",))
            self._vit_engine = HipEncoderEngine(self.cfg, dv, self.residual_dtype, halves=("vit",))
            self._engine_key = {"vit": None, "qformer": None}
        for part, eng in (("vit", self._vit_engine), ("qformer", self._engine)):
            if train and part == "qformer" and self._engine_key[part] is not None:
                continue        # packed once (its plans' descriptors need operands to point at); never refreshed while training
            key = (str(eng.device), self._weights_key(part))
            if self._engine_key[part] != key:
                eng.load_state_dict(self.state_dict(), parts=(part,))
                self._engine_key[part] = key
        return self._vit_engine, self._engine

    @classmethod
    def from_reference_config(cls, config, **kw):
        """Build from the reference's model config dict: the `model` entry of configs/config.json (what tasks/test.py
        :649-651 and tasks/test_recursive_grounding.py:497 load; freeze_vit = freeze_qformer = false, max_txt_len 512,
        num_frames 32) or of scripts/train/config_7b_stage3.py:35-85 after its ${...} references are resolved
        (freeze_vit True, freeze_qformer False, num_frames 12).  Keys are read with the defaults of
        hawkeye_it.py:22-53 / vit.py:408-422; keys of the LLM half (llama_model_path, lora_*, prompt tokens) are
        ignored."""
        ve = config["vision_encoder"]
        cfg = EncoderConfig(
            img_size=ve["img_size"], patch_size=ve["patch_size"], encoder_embed_dim=ve["encoder_embed_dim"],
            encoder_depth=ve["encoder_depth"], return_index=ve.get("return_index", -1),
            encoder_num_heads=ve["encoder_num_heads"], num_frames=int(ve["num_frames"]),
            num_query_token=config["num_query_token"], extra_num_query_token=config.get("extra_num_query_token", 32))
        if config.get("low_resource", False):
            raise NotImplementedError("low_resource=True (8-bit LLM + ViT on the CPU, hawkeye_it.py:29,213-223) is not "
                                      "offered: the path has no CPU arithmetic")
        kw.setdefault("freeze_vit", config.get("freeze_vit", True))
        kw.setdefault("freeze_qformer", config.get("freeze_qformer", True))
        m = cls(cfg, max_txt_len=int(config.get("max_txt_len", 32)), **kw)
        m.qformer_text_input = bool(config.get("qformer_text_input", True))                    # hawkeye_it.py:36
        m.qformer_hidden_dropout_prob = config.get("qformer_hidden_dropout_prob", 0.1)          # hawkeye_it.py:32-34
        m.qformer_attention_probs_dropout_prob = config.get("qformer_attention_probs_dropout_prob", 0.1)
        m.qformer_drop_path_rate = config.get("qformer_drop_path_rate", 0.1)
        return m

    # ---- engine plumbing -----------------------------------------------------------
    def _weights_key(self, part: str):
        """(data_ptr, version) of every parameter of one half: changes when a parameter is replaced, moved or modified
        in place.  Walking named_parameters() over ~230 modules costs 0.5-0.8 ms per half -- host time that is NOT
        hidden behind the device when the caller syncs on every result, as the eval scripts do -- so the walk is done
        once per module structure and what is re-read per call are the (module, name) slots it found: a parameter
        replaced in its slot, `.data` swapped by .to() / .float(), an optimizer step or a load_state_dict all show up;
        modules or parameters assigned on this file's containers bump the structure version."""
        key = []
        for params, pname in self._slots(part):
            p = params.get(pname)
            key.append((p.data_ptr(), p._version) if p is not None else (0, 0))   # a slot emptied since the walk
        return tuple(key)

    def _slots(self, part: str):
        """[(module._parameters, name)] of one half ("vit": vision_encoder + vision_layernorm; "qformer": the rest), in
        named_parameters() order; rebuilt when the structure version moved."""
        cache = self.__dict__.get("_slot_cache")
        if cache is None or cache[0] != _STRUCTURE_VERSION[0]:
            slots = {"vit": [], "qformer": []}
            for mname, mod in self.named_modules():
                half = "vit" if (mname + ".").startswith(("vision_encoder.", "vision_layernorm.")) else "qformer"
                for pname in mod._parameters:
                    if mod._parameters[pname] is not None:
                        slots[half].append((mod._parameters, pname))
            cache = (_STRUCTURE_VERSION[0], slots)
            self.__dict__["_slot_cache"] = cache
        return cache[1][part]

    def _any_requires_grad(self, part: str) -> bool:
        return any(p is not None and p.requires_grad for p in (params.get(pname) for params, pname in self._slots(part)))

    def sync_engine(self, force: bool = False, parts=("vit", "qformer")) -> HipEncoderEngine:
        """(Re)pack device weights when a parameter was replaced or modified in place.  The two halves are tracked
        separately: a training step changes Q-Former parameters only and never repacks the frozen ViT."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("HawkEyeEncoder.encode_img needs the model on an MI355X (model.to('cuda')); "
                               "there is no CPU path")
        if self._engine is None or self._engine.device != dev or self._engine.halves != ("vit", "qformer"):
            self._engine = HipEncoderEngine(self.cfg, dev, self.residual_dtype)
            self._engine_key = {"vit": None, "qformer": None}
            self._train_engine = None
        if self._engine_key["qformer"] is None and "qformer" not in parts:
            parts = tuple(parts) + ("qformer",)      # plans are built over both halves: pack each at least once
        todo = []
        for part in parts:
            key = (str(dev), self._weights_key(part))
            if force or self._engine_key[part] != key:
                todo.append(part)
                self._engine_key[part] = key
        if todo:
            self._engine.load_state_dict(self.state_dict(), parts=tuple(todo))
        return self._engine

    # ---- row f2: training ------------------------------------------------------------
    def set_dropout_seed(self, seed: int):
        """Seed of the counter-based dropout / drop-path masks (include/hawkeye_hip.h, hk_dropout).  Call n after this
        uses seed + n * 0x9E3779B9.  Under data parallelism give every rank its own seed."""
        self._drop_seed, self._drop_calls = int(seed) & 0xFFFFFFFF, 0

    def trainable_parameters(self):
        """{state-dict name: nn.Parameter} of what stage 3 trains on this path (hawkeye_it.py:93-99, 183-185)."""
        named = dict(self.named_parameters())
        return {n: named[n] for n in trainable_names(self.cfg)}

    def sync_train_engine(self, eng: Optional[HipEncoderEngine] = None) -> QFormerTrainEngine:
        """eng: the inference engine the training engine hangs on -- the single engine (default), or the Q-Former-half engine
        of a two-device placement."""
        if eng is None:
            eng = self.sync_engine(parts=("vit",))
        te = self._train_engine
        if te is None or te.enc is not eng or not te.arena.attached():
            te = QFormerTrainEngine(eng, self.trainable_parameters(), self.qformer_hidden_dropout_prob,
                                    self.qformer_attention_probs_dropout_prob, self.qformer_drop_path_rate)
            self._train_engine = te
        return te

    def reset_train_engine(self):
        self._train_engine = None
        self._reducer = None

    # Data parallelism of the stage-3 step (tasks/shared_utils.py:54-59 wraps the model in DistributedDataParallel).
    # Three routes, all ending with `p.grad` of every replica holding the same averaged gradient:
    #   1. DistributedDataParallel(model, find_unused_parameters=True) as the reference does: the node's gradients reach
    #      DDP's hooks like any other; nothing to call.
    #   2. out.backward(); model.all_reduce_gradients(): blocking flat all-reduce of the gradient arena.
    #   3. model.enable_overlapped_grad_reduce() once, then just out.backward(): the arena is all-reduced in buckets on a
    #      side stream WHILE the backward is still producing earlier layers' gradients (hawkeye_amd/dist.py BucketReducer).
    # Do not combine 1 with 2 / 3 (the gradients would be averaged twice).
    def enable_overlapped_grad_reduce(self, average: bool = True, bucket_mb: int = 64, force: bool = False):
        self._overlap = (bool(average), int(bucket_mb), bool(force))
        self._reducer = None

    def disable_overlapped_grad_reduce(self):
        self._overlap = None
        self._reducer = None

    def _grad_reducer(self):
        if self._overlap is None:
            return None
        te = self._train_engine
        if self._reducer is None or self._reducer.grad is not te.arena.grad:
            from . import dist as hdist
            self._reducer = hdist.BucketReducer(te.arena.grad, average=self._overlap[0], bucket_bytes=self._overlap[1] << 20,
                                                  force=self._overlap[2])
        return self._reducer

    def all_reduce_gradients(self, average: bool = True, bucket_mb: int = 64):
        """Route 2: data-parallel gradient exchange of the trainable parameters on the flat gradient arena
        (hawkeye_amd/dist.py reduce_param_grads); the alternative to wrapping the model in DistributedDataParallel.
        Whatever autograd did with the node's gradients (adopted the arena views, cloned them, accumulated several
        micro-batches), they are first made views of the arena again, so the reduced values ARE the `p.grad` the
        optimizer reads."""
        from . import dist as hdist
        te = self._train_engine
        if te is None:
            raise RuntimeError("all_reduce_gradients: no training call has run yet")
        hdist.reduce_param_grads(te.arena, average=average, bucket_bytes=bucket_mb << 20)

    def _table(self, use_image: bool, n_tokens: int, dev) -> torch.Tensor:
        enc = self.vision_encoder.encoder
        if self.device_pos_embed:
            key = ("dev", use_image, n_tokens, str(dev))
            hit = self._table_cache.get(key)
            if hit is None:
                Lp = self.cfg.patches_per_frame
                hit = ops.pos_embed(n_tokens // Lp, Lp, self.cfg.encoder_embed_dim,
                                    -1 if use_image else self.cfg.ckpt_num_frame, dev)
                self._table_cache[key] = hit
            return hit
        t = enc.img_pos_embed if use_image else enc.pos_embed     # read at call time (vit.py:304-307)
        if t.shape[-2] != n_tokens:
            raise RuntimeError(
                f"pos_embed has {t.shape[-2]} positions but the clip has {n_tokens} tokens "
                "(assign vision_encoder.encoder.pos_embed for this frame count, as demo.ipynb does)")
        key = (use_image, t.data_ptr(), t._version, tuple(t.shape), str(dev))
        hit = self._table_cache.get(use_image)
        if hit is None or hit[0] != key:
            dt = t.detach().reshape(n_tokens, -1).to(device=dev, dtype=torch.float32).contiguous()
            self._table_cache[use_image] = (key, dt, t)   # keep `t` alive so data_ptr cannot be recycled
            return dt
        return hit[1]

    def _tokenize(self, instruction, B: int, dev) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        if isinstance(instruction, (tuple, list)) and len(instruction) == 2 and torch.is_tensor(instruction[0]):
            ids, mask = instruction                  # pre-tokenised (input_ids, attention_mask)
        else:
            if self.tokenizer is None:
                raise RuntimeError("encode_img got text but model.tokenizer is None; assign a BertTokenizer-compatible "
                                   "callable or pass (input_ids, attention_mask) tensors")
            tok = self.tokenizer(instruction, padding="longest", truncation=True, max_length=self.max_txt_len,
                                 return_tensors="pt")     # hawkeye_it.py:243-249
            ids, mask = tok.input_ids, tok.attention_mask
        if ids.shape[0] == 1 and B > 1:
            ids, mask = ids.expand(B, -1), mask.expand(B, -1)
        if ids.shape[0] != B:
            raise RuntimeError(f"instruction batch {ids.shape[0]} does not match image batch {B}")
        return ids.to(dev, torch.int64).contiguous(), mask.to(dev, torch.int64).contiguous()

    # ---- the hot path --------------------------------------------------------------
    def encode_img(self, image: torch.Tensor, instruction: Union[str, Sequence[str], Tuple[torch.Tensor, torch.Tensor]]):
        """hawkeye_it.py:219-270.  image [B,T,3,H,W] float on the model's GPU.

        Like the reference module it follows the module's mode and autograd's: dropout / drop-path act iff
        `self.qformer.training` (nn.Dropout semantics; the eval scripts call model.eval()), and the result carries
        gradients to the Q-Former / query tokens / llama_proj iff grad mode is on and one of them requires grad
        (stage 3, train_it.py:63-80).  The ViT is frozen (hawkeye_it.py:82-91) and never receives a gradient."""
        return self._encode(image, instruction, None)

    def encode_frames(self, frames: torch.Tensor, instruction, mean=ops.CLIP_MEAN, std=ops.CLIP_STD):
        """Row f3: encode_img on decoded frames.  frames [B,T,H,W,3] uint8 (resized + centre-cropped, as
        decord/PIL produce them, tasks/test_recursive_grounding.py:364-385) on the model's GPU; the /255 and
        mean/std steps of the eval transform (:333-343) run inside the patch gather.  Same return as encode_img."""
        if frames.dtype != torch.uint8:
            raise RuntimeError(f"encode_frames expects uint8 frames, got {frames.dtype}")
        return self._encode(frames, instruction, (tuple(mean), tuple(std)))

    def _encode_split(self, image, instruction, norm):
        """hawkeye_it.py:226-269 with the ViT on devices[1]: image -> devices[1], ViT + vision_layernorm there,
        image_embeds -> devices[0] (:234), Q-Former + llama_proj there, result back on image.device."""
        device = image.device
        ev, eq = self._sync_split_engines()
        B, T = image.shape[0], image.shape[1]
        use_image = True if T == 1 else False
        table = self._table(use_image, T * self.cfg.patches_per_frame, ev.device)
        ids = mask = None
        if self.qformer_text_input:
            ids, mask = self._tokenize(instruction, B, eq.device)
        with torch.no_grad():
            plan = ev.forward_vit(image.to(ev.device), None, None, table, norm=norm)        # :230-231
            N = T * self.cfg.patches_per_frame
            embeds = plan.ws["xn"].view(B, N, self.cfg.encoder_embed_dim)
            if eq.device != ev.device:
                # the consumer's stream must not start the copy before the producer's stream has written xn
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(ev.device))
                torch.cuda.current_stream(eq.device).wait_event(done)
            out = eq.forward_qformer(embeds, ids, mask)
            out = out.to(self.out_dtype) if self.out_dtype != torch.float32 else out.clone()
        return out.to(device), use_image

    def _encode_split_train(self, image, instruction, norm, dropout, need_grad):
        """The stage-3 call with the reference's optional two-device placement (hawkeye_it.py:197-211 under
        tasks/shared_utils.py:48-51): frozen ViT + vision_layernorm on devices[1] (inference launches, no_grad), image_embeds
        handed to devices[0] (:234), training forward / backward of the Q-Former half there."""
        device = image.device
        ev, eq = self._sync_split_engines(train=True)
        B, T = image.shape[0], image.shape[1]
        use_image = True if T == 1 else False
        table = self._table(use_image, T * self.cfg.patches_per_frame, ev.device)
        ids = mask = None
        if self.qformer_text_input:
            ids, mask = self._tokenize(instruction, B, eq.device)
        with torch.no_grad():
            plan = ev.forward_vit(image.to(ev.device), None, None, table, norm=norm)
            embeds = _Embeds(plan.ws["xn"].view(B, T * self.cfg.patches_per_frame, self.cfg.encoder_embed_dim))
            if eq.device != ev.device:
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(ev.device))
                torch.cuda.current_stream(eq.device).wait_event(done)
        te = self.sync_train_engine(eq)
        seed = (self._drop_seed + self._drop_calls * 0x9E3779B9) & 0xFFFFFFFF
        self._drop_calls += 1
        if need_grad:
            out = _EncodeImgFn.apply(self, embeds, ids, mask, None, None, dropout, seed, *te.arena.params)
        else:
            with torch.no_grad():
                out = te.forward(None, ids, mask, None, train=dropout, seed=seed, image_embeds=embeds.t)[0].clone()
        out = out.to(self.out_dtype) if self.out_dtype != torch.float32 else out
        return out.to(device), use_image

    def _encode(self, image, instruction, norm):
        device = image.device
        dropout = self.qformer.training
        need_grad = torch.is_grad_enabled() and self._any_requires_grad("qformer")
        train_path = dropout or need_grad
        if not self.freeze_vit and torch.is_grad_enabled() and self._any_requires_grad("vit"):
            raise NotImplementedError(
                "this call would need a gradient for the ViT (grad mode is on and a vision_encoder / vision_layernorm "
                "parameter requires grad): the HIP path has no ViT backward.  Stage 3 freezes the ViT "
                "(scripts/train/config_7b_stage3.py:40); for inference call under model.eval() + torch.no_grad() as "
                "tasks/test.py and tasks/test_recursive_grounding.py do, or set requires_grad=False on those parameters")
        if self.low_resource:
            raise NotImplementedError("low_resource (ViT in fp32 on the CPU, hawkeye_it.py:213-223) is not offered: the "
                                      "path has no CPU arithmetic; a 288 GB MI355X holds the 0.6 GB of ViT operands")
        if self._split_placement():
            if train_path:
                return self._encode_split_train(image, instruction, norm, dropout, need_grad)
            return self._encode_split(image, instruction, norm)
        eng = self.sync_engine(parts=("vit",) if train_path else ("vit", "qformer"))
        if device != eng.device:
            image = image.to(eng.device)
        B, T = image.shape[0], image.shape[1]
        use_image = True if T == 1 else False                       # :227
        table = self._table(use_image, T * self.cfg.patches_per_frame, eng.device)
        ids = mask = None
        if self.qformer_text_input:
            ids, mask = self._tokenize(instruction, B, eng.device)
        if train_path:
            te = self.sync_train_engine()
            seed = (self._drop_seed + self._drop_calls * 0x9E3779B9) & 0xFFFFFFFF
            self._drop_calls += 1
            if need_grad:
                out = _EncodeImgFn.apply(self, image, ids, mask, table, norm, dropout, seed, *te.arena.params)
            else:
                with torch.no_grad():
                    out = te.forward(image, ids, mask, table, train=dropout, seed=seed, norm=norm)[0].clone()
            out = out.to(self.out_dtype) if self.out_dtype != torch.float32 else out
            return out.to(device), use_image
        with torch.no_grad():
            out = eng.forward(image, ids, mask, table, norm=norm)
            out = out.to(self.out_dtype) if self.out_dtype != torch.float32 else out.clone()
        return out.to(device), use_image
