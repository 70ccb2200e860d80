"""ctypes binding of libhawkeye_hip.so (include/hawkeye_hip.h).

The library is built in-tree by `make -C hawkeye_amd/csrc` (or
__graft_entry__.build()).  There is no fallback: if the shared object is
missing, load() raises, and every op in hawkeye_amd.ops goes through load().
"""
import ctypes as C
import os

# torch first: PyTorch-ROCm ships its own libamdhip64.so.7 and must be the HIP runtime of the
# process.  If this library were loaded before torch, its DT_NEEDED would pull /opt/rocm's copy
# in first and the process would end up with a mixed runtime ("no ROCm-capable device").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# HK_LIB_PATH: development aid (tools/ab_lib.py times two builds of the library on the same device); never a fallback
LIB_PATH = os.environ.get("HK_LIB_PATH") or os.path.join(_HERE, "libhawkeye_hip.so")

HK_ACT_NONE, HK_ACT_GELU_ERF, HK_ACT_SWIGLU = 0, 1, 2
HK_DT_BF16, HK_DT_F32 = 0, 1


class HkRows(C.Structure):
    _fields_ = [("ld", C.c_int64), ("rows_per_batch", C.c_int64), ("batch_stride", C.c_int64)]


class HkGemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("a_rows", HkRows), ("ldw", C.c_int64), ("c_rows", HkRows), ("r_rows", HkRows),
        ("act", C.c_int32), ("out_dtype", C.c_int32), ("res_dtype", C.c_int32), ("scale_cols", C.c_int32),
        ("scale", C.c_float),
        ("row_stats", C.c_void_p), ("col_c", C.c_void_p), ("row_part", C.c_void_p),
        ("split_k", C.c_int32), ("w_tiled", C.c_int32), ("split_stride", C.c_int64),
    ]


class HkGemvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("W", C.c_void_p), ("out", C.c_void_p), ("residual", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("out_f32", C.c_int32),
        ("ldx", C.c_int64), ("ldo", C.c_int64), ("act", C.c_int32), ("norm_eps", C.c_float),
        ("norm_weight", C.c_void_p), ("w_tiled", C.c_int32), ("reserved", C.c_int32),
    ]


class HkAttnDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p), ("key_bias", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32),
        ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldo", C.c_int64),
        ("bsq", C.c_int64), ("bsk", C.c_int64), ("bsv", C.c_int64), ("bso", C.c_int64),
        ("head_dim", C.c_int32), ("causal", C.c_int32),
        ("key_splits", C.c_int32), ("o_part", C.c_void_p), ("lse_part", C.c_void_p),
    ]


class HkTransposeJob(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("ld", C.c_int64), ("ldo", C.c_int64),
                ("R", C.c_int32), ("C", C.c_int32), ("tile0", C.c_int32), ("pad_", C.c_int32)]


class HkDropout(C.Structure):
    _fields_ = [("p", C.c_float), ("p_path", C.c_float), ("seed", C.c_uint32), ("site", C.c_uint32),
                ("path_site", C.c_uint32), ("rows_per_sample", C.c_int32)]


class HkAttnTrainDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p), ("lse", C.c_void_p),
        ("key_bias", C.c_void_p), ("dO", C.c_void_p), ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p),
        ("dsum", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32),
        ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldo", C.c_int64),
        ("bsq", C.c_int64), ("bsk", C.c_int64), ("bsv", C.c_int64), ("bso", C.c_int64),
        ("lddq", C.c_int64), ("lddk", C.c_int64), ("lddv", C.c_int64),
        ("bsdq", C.c_int64), ("bsdk", C.c_int64), ("bsdv", C.c_int64),
        ("softmax_scale", C.c_float), ("p_drop", C.c_float), ("seed", C.c_uint32), ("site", C.c_uint32),
        ("o_f32", C.c_void_p),
    ]


HK_LNBWD_MAX_BLOCKS = 256
HK_COLSUM_MAX_SPLIT = 64
HK_COLSUM_STATIC_MAX_C = 9216

# every symbol include/hawkeye_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "hk_gemm_bf16": (C.c_int, [C.POINTER(HkGemmDesc), C.c_void_p]),
    "hk_flash_attn_bf16": (C.c_int, [C.POINTER(HkAttnDesc), C.c_void_p]),
    "hk_layernorm": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(HkRows), C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "hk_layernorm_parts": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.POINTER(HkRows), C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "hk_row_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "hk_residual_add_parts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_int32,
                                        C.c_float, C.c_void_p]),
    "hk_row_stats_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "hk_patchify_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p]),
    "hk_patchify_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]),
    "hk_pos_embed": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hk_rmsnorm": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "hk_rope_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                               C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "hk_gemv_bf16": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hk_decode_rope_append": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                        C.c_float, C.c_void_p]),
    "hk_decode_attn_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "hk_decode_attn_rope_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                           C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    "hk_argmax_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "hk_silu_mul_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "hk_rmsnorm_add_parts": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_int32, C.c_float, C.c_void_p]),
    "hk_embed_splice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32,
                                  C.c_void_p]),
    "hk_qformer_embed_ln": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_float, C.c_void_p]),
    "hk_qformer_key_bias": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hk_cast_f32_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hk_dropout_keep_u32": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "hk_attn_train_fwd": (C.c_int, [C.POINTER(HkAttnTrainDesc), C.c_void_p]),
    "hk_attn_train_bwd": (C.c_int, [C.POINTER(HkAttnTrainDesc), C.c_void_p]),
    "hk_drop_add_ln": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(HkRows), C.POINTER(HkDropout), C.POINTER(HkDropout),
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(HkRows),
                                 C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "hk_layernorm_bwd": (C.c_int, [C.c_void_p, C.POINTER(HkRows), C.c_void_p, C.c_void_p, C.POINTER(HkDropout),
                                   C.POINTER(HkDropout), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "hk_colsum": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hk_colsum_static": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "hk_transpose_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(HkRows), C.c_int64, C.c_int32, C.c_void_p,
                                    C.c_int64, C.c_void_p]),
    "hk_transpose_f32_bf16_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "hk_wgrad_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(HkRows), C.c_void_p, C.c_int64, C.c_void_p,
                                C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "hk_gelu_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hk_gelu_bwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hk_qformer_embed_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hk_qformer_embed_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hk_mfma_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hk_gemm_duo_mask": (C.c_int, [C.c_int]),
    "hk_version": (C.c_int, []),
    "hk_arch": (C.c_char_p, []),
    "hk_last_error": (C.c_size_t, [C.c_char_p, C.c_size_t]),
}

_lib = None


def load() -> C.CDLL:
    """Load libhawkeye_hip.so; raises (never falls back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built "
            "(run `make -C hawkeye_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "hawkeye_amd has no CPU or PyTorch fallback for the encode_img path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def source_digest() -> str:
    """sha1 over the kernel sources and the header the library is built from, comments and white space removed:
    profiles taken with another kernel tree (profiles/traffic.json) are recognised as stale by it, while a note added
    to a comment does not orphan them."""
    import glob
    import hashlib
    import re
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "hawkeye_hip.h"))
    for f in files:
        text = open(f, "r", errors="replace").read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)      # block comments
        text = re.sub(r"//[^\n]*", " ", text)                   # line comments (no '//' occurs inside a string here)
        h.update(os.path.basename(f).encode())
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().hk_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")
