"""ORACLE -- test infrastructure, not product code.

numpy restatement of the counter-based dropout hash of libhawkeye_hip.so (include/hawkeye_hip.h, `hk_dropout`).
The reference draws its dropout / drop-path masks from torch's stateful generator (nn.Dropout, Qformer.py:70,148,
311,399; timm drop_path, Qformer.py:24,299), whose stream no other implementation can reproduce, so train-mode
parity is checked by feeding THESE masks to the oracle (and, in oracle/make_goldens.py, to the reference's own
modules) and comparing against the HIP path that regenerates them on the device.
Pinned against the library by tests/test_train_cpu.py (hk_dropout_keep_u32 is a host function).

Site numbering of the Q-Former (shared with hawkeye_amd/train.py): see `site()`.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def fmix32(h):
    """MurmurHash3 32-bit finaliser on uint32 arrays (computed in uint64 to avoid overflow warnings)."""
    h = np.asarray(h, dtype=np.uint64) & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def site_key(seed: int, site: int) -> np.uint64:
    inner = fmix32(np.uint64((site * 0x9E3779B9 + 0x7F4A7C15) & 0xFFFFFFFF))
    return fmix32(np.uint64(seed & 0xFFFFFFFF) ^ inner)


def hash_u32(seed: int, site: int, idx) -> np.ndarray:
    return fmix32((np.asarray(idx, dtype=np.uint64) & M32) ^ site_key(seed, site))


def threshold(p: float) -> int:
    t = float(np.float32(p)) * 4294967296.0
    return 0 if t <= 0 else min(int(t), 4294967295)


def keep_mask(seed: int, site: int, shape, p: float) -> np.ndarray:
    """Boolean keep mask of a site's tensor (flat row-major element index)."""
    n = int(np.prod(shape))
    if p <= 0:
        return np.ones(shape, dtype=bool)
    return (hash_u32(seed, site, np.arange(n, dtype=np.uint64)) >= np.uint64(threshold(p))).reshape(shape)


def scale(p: float) -> np.float32:
    """1 / (1 - p) as the kernels compute it (fp32)."""
    return np.float32(1.0) / (np.float32(1.0) - np.float32(p)) if p > 0 else np.float32(1.0)


# kinds of dropout site inside one BertLayer (Qformer.py:435-507)
SELF_PROB, SELF_HID, SELF_PATH, CROSS_PROB, CROSS_HID, CROSS_PATH, QFFN_HID, QFFN_PATH, TFFN_HID, TFFN_PATH = range(10)
EMBED_SITE = 4095


def site(layer: int, kind: int) -> int:
    return layer * 16 + kind


class MaskDropper:
    """The `drop` object oracle/encode_img_oracle.py takes: dropout / drop-path with the masks of the device hash."""

    def __init__(self, seed: int, p_hidden: float, p_attn: float, path_rates):
        self.seed, self.p_hidden, self.p_attn, self.path_rates = seed, p_hidden, p_attn, list(path_rates)

    def elem(self, site_id, x, p):
        import torch
        if p <= 0:
            return x
        m = torch.from_numpy(keep_mask(self.seed, site_id, tuple(x.shape), p))
        return x * m * float(scale(p))

    def path(self, site_id, x, p):
        """timm drop_path: one Bernoulli draw per sample, kept samples scaled by 1 / keep_prob."""
        import torch
        if p <= 0:
            return x
        m = torch.from_numpy(keep_mask(self.seed, site_id, (x.shape[0],), p))
        return x * m.view(-1, *([1] * (x.dim() - 1))) * float(scale(p))


def path_rates(n_layers: int, rate: float = 0.2):
    """blip2.py:63 -- drop_path_list = linspace(0, qformer_drop_path_rate, num_hidden_layers)."""
    import torch
    return [x.item() for x in torch.linspace(0, rate, n_layers)]
