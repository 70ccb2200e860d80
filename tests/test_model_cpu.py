"""Host logic of HawkEyeEncoder that runs before the first launch of every call, on CPU: the weight-change key taken
over cached parameter slots must see everything the walk over named_parameters() it replaced saw (the
module protocol -- .to(), load_state_dict, optimizer steps, resize_token_embeddings, parameter assignment)."""
import torch
from torch import nn

from hawkeye_amd.config import SMALL
from hawkeye_amd.model import HawkEyeEncoder


def _walk_key(m, part):
    vit = part == "vit"
    return tuple((p.data_ptr(), p._version) for n, p in m.named_parameters()
                 if n.startswith(("vision_encoder.", "vision_layernorm.")) == vit)


def test_weight_key_matches_a_walk_over_named_parameters_and_sees_every_kind_of_change():
    m = HawkEyeEncoder(SMALL)
    for part in ("vit", "qformer"):
        assert m._weights_key(part) == _walk_key(m, part)
    kq, kv = m._weights_key("qformer"), m._weights_key("vit")
    assert m._weights_key("qformer") == kq and m._weights_key("vit") == kv          # stable without a change

    def changed(part="qformer"):
        nonlocal kq, kv
        new = m._weights_key(part)
        assert new == _walk_key(m, part)
        was = kq if part == "qformer" else kv
        if part == "qformer":
            kq = new
        else:
            kv = new
        return new != was

    q = m.qformer.bert.encoder.layer[0].attention.self.query
    with torch.no_grad():
        q.weight.add_(1.0)                                                          # an optimizer step
    assert changed() and not changed("vit")
    q.weight = nn.Parameter(torch.zeros_like(q.weight))                             # parameter replaced in a leaf module
    assert changed()
    m.qformer.resize_token_embeddings(SMALL.qf_vocab + 1)                           # module replaced (hawkeye_it.py:72)
    assert changed()
    m.query_tokens = nn.Parameter(torch.zeros_like(m.query_tokens))                 # parameter replaced on the root
    assert changed()
    m.load_state_dict(m.state_dict())                                               # copy_ into every parameter
    assert changed() and changed("vit")
    m.float()                                                                       # already fp32: nothing moves
    assert not changed() and not changed("vit")
    m.double()                                                                      # .data swapped by _apply
    assert changed() and changed("vit")
    m.vision_encoder.encoder.blocks[0] = type(m.vision_encoder.encoder.blocks[0])(SMALL.encoder_embed_dim,
                                                                                   SMALL.encoder_embed_dim * SMALL.mlp_ratio,
                                                                                   SMALL.vit_ln_eps).double()
    m.invalidate_parameter_cache()                                                  # the one documented blind spot
    assert changed("vit")


def test_device_is_the_first_parameters_and_requires_grad_follows_the_flags():
    m = HawkEyeEncoder(SMALL)
    assert m.device == next(m.parameters()).device == m.query_tokens.device        # blip2.py:43-44
    assert not m._any_requires_grad("vit")                                          # frozen ViT (hawkeye_it.py:82-91)
    want = any(p.requires_grad for n, p in m.named_parameters()
               if not n.startswith(("vision_encoder.", "vision_layernorm.")))
    assert m._any_requires_grad("qformer") == want
    for p in m.parameters():
        p.requires_grad = False
    assert not m._any_requires_grad("qformer")
    m.llama_proj.bias.requires_grad = True
    assert m._any_requires_grad("qformer")


# The `model` entries of the reference's two shipped configs, as dict literals (values of /root/reference/configs/
# config.json:2-44 and scripts/train/config_7b_stage3.py:35-85 with its ${num_frames} / ${max_txt_l} references resolved
# to 12 / 512, :13,16).  The reference tree does not travel to the GPU box, so nothing is read from it here.
_VE = dict(name="vit_l14", img_size=224, patch_size=16, d_model=1024, encoder_embed_dim=1024, encoder_depth=24,
           encoder_num_heads=16, drop_path_rate=0.0, tubelet_size=1, use_checkpoint=False, checkpoint_num=0,
           pretrained="", return_index=-2, vit_add_ln=True, ckpt_num_frame=4)
_COMMON = dict(model_cls="HawkEye_it", llama_model_path="model/vicuna-7b-v0", low_resource=False, num_query_token=32,
               qformer_hidden_dropout_prob=0.1, qformer_attention_probs_dropout_prob=0.1, qformer_drop_path_rate=0.2,
               extra_num_query_token=64, qformer_text_input=True, system="", start_token="<Video>",
               end_token="</Video>", img_start_token="<Image>", img_end_token="</Image>", random_shuffle=True,
               use_lora=True, lora_r=16, lora_alpha=32, lora_dropout=0.1)
EVAL_CONFIG = dict(_COMMON, bert_path="bert-base-uncased", freeze_vit=False, freeze_qformer=False, max_txt_len=512,
                   vision_encoder=dict(_VE, num_frames=32))
STAGE3_CONFIG = dict(_COMMON, vit_blip_model_path="model/VideoChat2/umt_l16_qformer.pth",
                     videochat2_model_path="model/VideoChat2/videochat2_7b_stage2.pth", freeze_vit=True,
                     freeze_qformer=False, max_txt_len=512, add_temp_embed=False, add_second_msg=True,
                     use_flash_attention=False, vision_encoder=dict(_VE, num_frames=12))


def _is_vit(n):
    return n.startswith(("vision_encoder.", "vision_layernorm."))


def test_from_reference_config_builds_from_both_shipped_configs():
    """the documented drop-in line must construct from the reference's own
    configs.  hawkeye_it.py:82-99 decides requires_grad and the pinned eval modes from the two freeze flags."""
    for conf, frames, vit_frozen in ((EVAL_CONFIG, 32, False), (STAGE3_CONFIG, 12, True)):
        m = HawkEyeEncoder.from_reference_config(conf)
        assert m.cfg.run_depth == 23 and len(m.vision_encoder.encoder.blocks) == 23        # vit.py:261
        assert m.cfg.n_query == 96 and m.extra_query_tokens.shape == (1, 64, 768)
        assert m.max_txt_len == 512 and m.qformer_text_input is True
        assert (m.qformer_hidden_dropout_prob, m.qformer_attention_probs_dropout_prob,
                m.qformer_drop_path_rate) == (0.1, 0.1, 0.2)
        assert m.vision_encoder.encoder.pos_embed.shape == (1, frames * 196, 1024)
        for n, p in m.named_parameters():
            if _is_vit(n):
                assert p.requires_grad == (not vit_frozen), n
            else:
                assert p.requires_grad, n             # freeze_qformer=False in both; extra queries / proj always train
        assert m._any_requires_grad("vit") == (not vit_frozen)
        # frozen halves are pinned to eval mode (disabled_train), unfrozen ones follow .train() / .eval()
        m.train()
        assert m.vision_encoder.training == (not vit_frozen) and m.vision_layernorm.training == (not vit_frozen)
        assert m.qformer.training
        m.eval()
        assert not m.vision_encoder.training and not m.qformer.training
        assert next(iter(m.state_dict())) == "query_tokens"


def test_from_reference_config_defaults_and_refusals():
    import pytest
    conf = {k: v for k, v in EVAL_CONFIG.items() if k not in ("freeze_vit", "freeze_qformer", "max_txt_len",
                                                              "qformer_drop_path_rate", "extra_num_query_token")}
    m = HawkEyeEncoder.from_reference_config(conf)        # hawkeye_it.py:26-27,34-35,38 defaults
    assert m.freeze_vit and m.freeze_qformer and m.max_txt_len == 32 and m.qformer_drop_path_rate == 0.1
    assert m.cfg.extra_num_query_token == 32
    assert not any(p.requires_grad for n, p in m.named_parameters() if n.startswith(("qformer.", "vision_")))
    assert not m.query_tokens.requires_grad and m.extra_query_tokens.requires_grad and m.llama_proj.weight.requires_grad
    with pytest.raises(NotImplementedError):
        HawkEyeEncoder.from_reference_config(dict(EVAL_CONFIG, low_resource=True))


def test_integration_md_ctypes_stub_matches_the_descriptor():
    """The maintainer's stub in INTEGRATION.md §2 must lay hk_gemm_desc out exactly as the library reads it (r02: it
    stopped 16 bytes short of split_k / split_stride)."""
    import ctypes as C
    import os
    import re
    from hawkeye_amd import lib as L
    text = open(os.path.join(os.path.dirname(__file__), "..", "INTEGRATION.md")).read()
    block = text[text.index("class Rows(C.Structure)"):text.index("lib.hk_gemm_bf16.argtypes")]
    ns = {"C": C}
    exec(block, ns)                                      # the stub's own two class statements, nothing else
    Rows, Gemm = ns["Rows"], ns["Gemm"]
    assert C.sizeof(Rows) == C.sizeof(L.HkRows)
    assert C.sizeof(Gemm) == C.sizeof(L.HkGemmDesc)
    assert [(n, getattr(Gemm, n).offset) for n, _ in Gemm._fields_] == \
           [(n, getattr(L.HkGemmDesc, n).offset) for n, _ in L.HkGemmDesc._fields_]
    assert re.search(r"split_k.*split_stride", block)


def test_prefill_split_k_rule_for_skinny_prompts():
    """LlamaPrefill.split_k_for (host logic, no GPU): o_proj / down_proj of a B = 1 / 2 prompt are split over K so that the
    slices make at most one round of 256x128 tiles; the Cfg4 batch (B = 8) and anything wider stay unsplit."""
    from hawkeye_amd.llm import LlamaPrefill
    f = LlamaPrefill.split_k_for
    assert f(352, 4096, 4096) == 4 and f(352, 4096, 11008) == 4          # 2 x 32 tiles x 4 slices = 256 workgroups
    assert f(704, 4096, 4096) == 2 and f(704, 4096, 11008) == 2
    assert f(2816, 4096, 4096) == 1 and f(2816, 4096, 11008) == 1
    assert f(352, 4096, 64 * 20) == 1                                     # slices shorter than 16 K-tiles are not worth it
    assert f(352, 4096, 64 * 34) == 2                                     # 34 K-tiles: 4 does not divide, 2 does
