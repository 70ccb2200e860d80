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
