"""Static shape description of the encode_img path.

Field names follow the reference's config keys (configs/config.json:15-33,
models/blip2/vit.py:408-422, models/blip2/blip2.py:47-72).
"""
from dataclasses import dataclass, asdict, replace


@dataclass(frozen=True)
class EncoderConfig:
    # --- UMT ViT (models/blip2/vit.py:244-324) ---
    img_size: int = 224
    patch_size: int = 16
    encoder_embed_dim: int = 1024
    encoder_depth: int = 24          # blocks in the checkpoint
    return_index: int = -2           # blocks actually run = depth + return_index + 1 (vit.py:261)
    encoder_num_heads: int = 16
    mlp_ratio: int = 4
    num_frames: int = 16             # T the video pos_embed is built for
    ckpt_num_frame: int = 4          # never forwarded by build_vit => always 4 (vit.py:350)
    vit_ln_eps: float = 1e-6         # vit.py:343
    vision_ln_eps: float = 1e-12     # blip2.py:83
    # --- Q-Former (models/blip2/Qformer.py, bert-base-uncased shape) ---
    qf_hidden: int = 768
    qf_layers: int = 12
    qf_heads: int = 12
    qf_intermediate: int = 3072
    qf_vocab: int = 30523            # 30522 + [DEC] (hawkeye_it.py:72, blip2.py:29)
    qf_max_pos: int = 512
    qf_ln_eps: float = 1e-12
    cross_attention_freq: int = 2    # blip2.py:60
    num_query_token: int = 32        # config.json:28
    extra_num_query_token: int = 64  # config.json:32
    # --- projection (hawkeye_it.py:183-185) ---
    llama_hidden: int = 4096

    @property
    def run_depth(self) -> int:
        return self.encoder_depth + self.return_index + 1

    @property
    def patches_per_frame(self) -> int:
        return (self.img_size // self.patch_size) ** 2

    @property
    def n_query(self) -> int:
        return self.num_query_token + self.extra_num_query_token

    @property
    def patch_k(self) -> int:
        return 3 * self.patch_size * self.patch_size

    def to_dict(self):
        return asdict(self)


FULL = EncoderConfig()

# Reduced shape used by the committed full-tensor fixture (tests/golden/small_*.npz).
# Head size stays 64 because the HIP attention kernel is specialised for it.
SMALL = EncoderConfig(
    img_size=64, patch_size=16, encoder_embed_dim=256, encoder_depth=3, return_index=-2,
    encoder_num_heads=4, num_frames=2, qf_hidden=128, qf_layers=2, qf_heads=2,
    qf_intermediate=512, qf_vocab=1000, qf_max_pos=64, num_query_token=8,
    extra_num_query_token=8, llama_hidden=256,
)
# Reduced shape of the training fixture (tests/golden/train_small.npz): four Q-Former layers so that cross-attention
# (layers 0, 2) and non-zero drop-path rates (layers 1..3) both occur more than once.
SMALL_TRAIN = replace(SMALL, qf_layers=4)


@dataclass(frozen=True)
class LlmConfig:
    """Shape of the LLaMA decoder that consumes encode_img's output (row f1): Vicuna-7B-v0's config.json values,
    i.e. LlamaConfig defaults (models/blip2/modeling_llama.py:16, hawkeye_it.py:112-133)."""
    hidden_size: int = 4096
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    intermediate_size: int = 11008
    vocab_size: int = 32000
    rms_norm_eps: float = 1e-6
    rope_base: float = 10000.0
    max_position_embeddings: int = 2048

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def to_dict(self):
        return asdict(self)


VICUNA_7B = LlmConfig()
# reduced shape of the committed fixture (tests/golden/llm_small.npz); head size stays 128
SMALL_LLM = LlmConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=320,
                      max_position_embeddings=256)
