"""CPU checks of the C-ABI boundary: the library loads, exports every symbol
include/hawkeye_hip.h declares, and rejects bad descriptors without touching a GPU."""
import ctypes as C
import os
import re

from hawkeye_amd import lib as L

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "hawkeye_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hk_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = L.load()
    names = _declared()
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), n
        assert n in L.SYMBOLS, f"{n} declared in the header but not bound in hawkeye_amd/lib.py"
    assert sorted(L.SYMBOLS) == names


def test_version_and_arch():
    lib = L.load()
    assert lib.hk_version() >= 100
    assert lib.hk_arch() == b"gfx950"


def test_struct_layout_matches_header():
    # hk_rows: 3 x int64; hk_gemm_desc: 5 ptr, 3 int32 (+pad), rows, int64, rows, rows, 4 int32 + float (+pad),
    # then row_stats, col_c, row_part, split_k (+pad), split_stride
    assert C.sizeof(L.HkRows) == 24
    assert C.sizeof(L.HkGemmDesc) == 5 * 8 + 16 + 24 + 8 + 24 + 24 + 24 + 3 * 8 + 8 + 8   # 4 int32 + float, padded to 8
    assert L.HkGemmDesc.split_k.offset == 5 * 8 + 16 + 24 + 8 + 24 + 24 + 24 + 3 * 8 and L.HkGemmDesc.split_stride.offset % 8 == 0
    assert C.sizeof(L.HkAttnDesc) == 5 * 8 + 16 + 8 * 8 + 8 + 8 + 16   # + head_dim, causal; key_splits (+pad), o_part, lse_part


def test_bad_descriptors_are_rejected_with_text():
    lib = L.load()
    d = L.HkGemmDesc()
    assert lib.hk_gemm_bf16(C.byref(d), None) == 1
    assert "null operand" in L.last_error()
    d.A = d.W = d.C = 4096
    d.M, d.N, d.K = 16, 16, 48
    assert lib.hk_gemm_bf16(C.byref(d), None) == 1
    assert "multiple of 64" in L.last_error()
    d.K, d.N, d.split_k = 128, 64, 2          # split-K needs fp32 output
    d.a_rows.ld = d.ldw = 128
    d.c_rows.ld = 64
    assert lib.hk_gemm_bf16(C.byref(d), None) == 1 and "split_k" in L.last_error()
    a = L.HkAttnDesc()
    assert lib.hk_flash_attn_bf16(C.byref(a), None) == 1
    assert lib.hk_layernorm(None, 0, None, None, None, None, None, None, 4, 64, 1e-6, None) == 1
    assert lib.hk_patchify_bf16(4096, 4096, 1, 1, 30, 32, 16, None) == 1
    assert "bad shape" in L.last_error()


def test_product_library_carries_no_diagnostic_switches():
    """HK_GEMM_TRACE (hipMalloc + hipStreamSynchronize + hipMemcpy inside the launch path) and
    HK_GEMM_DEBUG (wrong results by design) are compiled only into the -DHK_DIAG build (`make diag`).  The shipped
    library promises 'never allocates, frees or synchronises' (include/hawkeye_hip.h): check its strings and its
    undefined dynamic symbols."""
    import subprocess
    from hawkeye_amd import lib as L
    blob = open(L.LIB_PATH, "rb").read()
    for name in (b"HK_GEMM_TRACE", b"HK_GEMM_DEBUG"):
        assert name not in blob, name
    syms = subprocess.run(["nm", "-D", "--undefined-only", L.LIB_PATH], check=True, capture_output=True, text=True).stdout
    used = {line.split()[-1].split("@")[0] for line in syms.splitlines() if line.strip()}
    for banned in ("hipMalloc", "hipFree", "hipStreamSynchronize", "hipDeviceSynchronize", "hipMemcpy", "hipMallocAsync"):
        assert banned not in used, banned
