"""Tensor-level wrappers over the C ABI (include/hawkeye_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream;
every function passes raw pointers to libhawkeye_hip.so and raises
RuntimeError on a non-zero return code.  Nothing in this file computes.
"""
import ctypes as C
from typing import Optional, Tuple

import torch

from . import lib as L

Rows = Tuple[int, int, int]  # (ld, rows_per_batch, batch_stride) in elements; rows_per_batch 0 = plain


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (hawkeye_amd has no CPU path)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")


def gemm_desc(A, W, C_out, M, N, K, *, bias=None, residual=None, a_rows: Rows, c_rows: Rows,
              r_rows: Rows = (0, 0, 0), ldw=None, act=L.HK_ACT_NONE, out_f32=False, res_bf16=False,
              scale_cols=0, scale=1.0, row_stats=None, col_c=None, row_part=None, split_k=0,
              split_stride=0, w_tiled=False) -> L.HkGemmDesc:
    """Raw-pointer descriptor; A/W/C_out/bias/residual (and row_stats/col_c: LayerNorm of A folded into the GEMM,
    see include/hawkeye_hip.h) are integer device addresses."""
    d = L.HkGemmDesc()
    d.A, d.W, d.bias, d.residual, d.C = A, W, bias, residual, C_out
    d.M, d.N, d.K = M, N, K
    d.a_rows = L.HkRows(*a_rows)
    d.ldw = K if ldw is None else ldw
    d.c_rows = L.HkRows(*c_rows)
    d.r_rows = L.HkRows(*r_rows)
    d.act = act
    d.out_dtype = L.HK_DT_F32 if out_f32 else L.HK_DT_BF16
    d.res_dtype = L.HK_DT_BF16 if res_bf16 else L.HK_DT_F32
    d.scale_cols = scale_cols
    d.scale = scale
    d.row_stats, d.col_c, d.row_part = row_stats, col_c, row_part
    d.split_k, d.split_stride = split_k, split_stride
    d.w_tiled = 1 if w_tiled else 0
    return d


def tile_gemm_weight(w: torch.Tensor) -> torch.Tensor:
    """Row-major bf16 [N, K] -> hk_gemm_desc.w_tiled's layout [N/64, K/64, 64, 64].  A permutation done once at load time."""
    _need(w, torch.bfloat16, "w")
    N, K = w.shape
    if N % 64 or K % 64:
        raise RuntimeError("tile_gemm_weight: N and K must be multiples of 64")
    return w.view(N // 64, 64, K // 64, 64).permute(0, 2, 1, 3).contiguous()


def run_gemm(d: L.HkGemmDesc):
    L.check(L.load().hk_gemm_bf16(C.byref(d), _stream()), "hk_gemm_bf16")


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
           residual: Optional[torch.Tensor] = None, act: str = "none", out_dtype=torch.bfloat16,
           scale_cols: int = 0, scale: float = 1.0, out: Optional[torch.Tensor] = None,
           row_stats: Optional[torch.Tensor] = None, col_c: Optional[torch.Tensor] = None,
           row_part: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act((x @ w.T + bias) * colscale) + residual; x [M,K] bf16, w [N,K] bf16 (or tile_gemm_weight's 4-D layout).
    row_stats [M,2] fp32 + col_c [N] fp32: LayerNorm of x folded into the GEMM (w, bias prepared by fold_layernorm)."""
    _need(x, torch.bfloat16, "x")
    _need(w, torch.bfloat16, "w")
    M, K = x.shape
    tiled = w.dim() == 4      # tile_gemm_weight's [N/64, K/64, 64, 64]
    if tiled:
        if tuple(w.shape[2:]) != (64, 64) or not w.is_contiguous() or w.shape[1] * 64 != K or x.stride(1) != 1:
            raise RuntimeError("linear: a tiled w must be contiguous [N/64, K/64, 64, 64] with matching K")
        N = w.shape[0] * 64
    else:
        N = w.shape[0]
        if x.stride(1) != 1 or w.stride(1) != 1 or w.shape[1] != K:
            raise RuntimeError("linear: x and w must be K-contiguous with matching K")
    if bias is not None:
        _need(bias, torch.float32, "bias")
    if residual is not None:
        if residual.dtype not in (torch.float32, torch.bfloat16) or not residual.is_cuda:
            raise RuntimeError("linear: residual must be an fp32 or bf16 GPU tensor")
        if residual.shape != (M, N) or residual.stride(1) != 1:
            raise RuntimeError("linear: residual must be [M,N] row-major")
    if act == "swiglu":   # w rows interleaved (gate_0, up_0, gate_1, up_1, ...): out [M, N / 2] = silu(gate) * up
        if residual is not None or out_dtype != torch.bfloat16 or N % 32 != 0 or scale_cols != 0:
            raise RuntimeError("linear: act='swiglu' needs bf16 output, N % 32 == 0, no residual, no column scale")
    elif act not in ("none", "gelu"):
        raise RuntimeError(f"linear: unknown act {act!r}")
    NO = N // 2 if act == "swiglu" else N
    if out is None:
        out = torch.empty((M, NO), dtype=out_dtype, device=x.device)
    if tuple(out.shape) != (M, NO) or out.stride(1) != 1:
        raise RuntimeError(f"linear: out must be [{M}, {NO}] row-major")
    _need(out, out_dtype, "out")
    d = gemm_desc(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, bias=_ptr(bias), residual=_ptr(residual),
                  a_rows=(x.stride(0), 0, 0), c_rows=(out.stride(0), 0, 0),
                  r_rows=(residual.stride(0), 0, 0) if residual is not None else (0, 0, 0), ldw=K if tiled else w.stride(0),
                  w_tiled=tiled, act={"gelu": L.HK_ACT_GELU_ERF, "swiglu": L.HK_ACT_SWIGLU}.get(act, L.HK_ACT_NONE), out_f32=out_dtype == torch.float32,
                  res_bf16=residual is not None and residual.dtype == torch.bfloat16,
                  scale_cols=scale_cols, scale=scale)
    if row_stats is not None:
        _need(row_stats, torch.float32, "row_stats")
        _need(col_c, torch.float32, "col_c")
        if tuple(row_stats.shape) != (M, 2) or tuple(col_c.shape) != (N,) or not row_stats.is_contiguous():
            raise RuntimeError("linear: row_stats must be [M,2] and col_c [N]")
        # the persistent kernel fetches the statistics of whole 256-row tiles: the allocation must cover them
        have = row_stats.untyped_storage().nbytes() - row_stats.storage_offset() * 4
        if have < (M + 255) // 256 * 256 * 8:
            raise RuntimeError("linear: row_stats storage must extend to a multiple of 256 rows (ops.row_stats pads it)")
        d.row_stats, d.col_c = row_stats.data_ptr(), col_c.data_ptr()
    if row_part is not None:   # out: [M, N/64, 2] (sum, sum of squares) of the result rows, for the next LayerNorm
        _need(row_part, torch.float32, "row_part")
        if tuple(row_part.shape) != (M, N // 64, 2) or not row_part.is_contiguous():
            raise RuntimeError("linear: row_part must be a contiguous [M, N/64, 2] tensor")
        d.row_part = row_part.data_ptr()
    run_gemm(d)
    return out


def row_stats(x: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(mean, rstd) of each row of a contiguous bf16 [rows, D] matrix -> fp32 [rows, 2]."""
    _need(x, torch.bfloat16, "x")
    if x.dim() != 2 or not x.is_contiguous():
        raise RuntimeError("row_stats: x must be a contiguous [rows, D] matrix")
    rows, D = x.shape
    if out is None:   # padded to whole 256-row tiles (what a folded GEMM may read), returned as the [rows, 2] view
        out = torch.zeros(((rows + 255) // 256 * 256, 2), dtype=torch.float32, device=x.device)[:rows]
    L.check(L.load().hk_row_stats(x.data_ptr(), out.data_ptr(), rows, D, eps, _stream()), "hk_row_stats")
    return out


def row_stats_finalize(part: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[rows, groups, 2] partial sums written by linear(..., row_part=) -> fp32 [rows, 2] (mean, rstd)."""
    _need(part, torch.float32, "part")
    rows, groups, _ = part.shape
    if out is None:
        out = torch.zeros(((rows + 255) // 256 * 256, 2), dtype=torch.float32, device=part.device)[:rows]
    L.check(L.load().hk_row_stats_finalize(part.data_ptr(), out.data_ptr(), rows, groups, eps, _stream()),
            "hk_row_stats_finalize")
    return out


def fold_layernorm(weight: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """Operands of a Linear whose input LayerNorm is folded into the GEMM (hk_gemm_desc.row_stats):
    W' = bf16(gamma[k] * weight[n,k]), col_c[n] = sum_k W'[n,k] (of the rounded values, so the mean term cancels
    exactly against what the MFMA accumulates), bias'[n] = bias[n] + sum_k beta[k] * weight[n,k].  fp32/fp64 host math
    on whatever device the tensors are on."""
    w32 = weight.detach().float()
    wf = (w32 * gamma.detach().float()[None, :]).to(torch.bfloat16).contiguous()
    col_c = wf.double().sum(dim=1).float().contiguous()
    b = (w32.double() @ beta.detach().double()).float()
    if bias is not None:
        b = b + bias.detach().float()
    return wf, col_c, b.contiguous()


def attn_desc(Q, K, V, O, B, H, Nq, Nk, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, key_bias=None, head_dim=0,
              causal=False, key_splits=0, o_part=None, lse_part=None) -> L.HkAttnDesc:
    d = L.HkAttnDesc()
    d.key_splits, d.o_part, d.lse_part = key_splits, o_part, lse_part
    d.head_dim, d.causal = head_dim, 1 if causal else 0
    d.Q, d.K, d.V, d.O, d.key_bias = Q, K, V, O, key_bias
    d.B, d.H, d.Nq, d.Nk = B, H, Nq, Nk
    d.ldq, d.ldk, d.ldv, d.ldo = ldq, ldk, ldv, ldo
    d.bsq, d.bsk, d.bsv, d.bso = bsq, bsk, bsv, bso
    return d


def run_attn(d: L.HkAttnDesc):
    L.check(L.load().hk_flash_attn_bf16(C.byref(d), _stream()), "hk_flash_attn_bf16")


def flash_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, key_bias: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None, causal: bool = False, key_splits: int = 0) -> torch.Tensor:
    """q [B,Nq,H,hd], k/v [B,Nk,H,hd] bf16 views (last two dims dense; hd = 64, or 128 for the LLaMA prefill),
    q pre-scaled by scale*log2(e).  key_bias: optional fp32 [B,Nk] additive (natural-log units).
    causal: query i sees keys <= i + (Nk - Nq).  key_splits > 1: the small-call key split.  Returns [B,Nq,H,hd] bf16."""
    hd = q.shape[3] if q.dim() == 4 else -1
    if hd not in (64, 128):
        raise RuntimeError("flash_attention: head size must be 64 or 128")
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _need(t, torch.bfloat16, n)
        if t.dim() != 4 or t.shape[3] != hd or t.stride(3) != 1 or t.stride(2) != hd:
            raise RuntimeError(f"flash_attention: {n} must be [B,N,H,{hd}] with dense (H,{hd}) dims")
    B, Nq, H, _ = q.shape
    Nk = k.shape[1]
    if out is None:
        out = torch.empty((B, Nq, H, hd), dtype=torch.bfloat16, device=q.device)
    if key_bias is not None:
        _need(key_bias, torch.float32, "key_bias")
        if key_bias.shape != (B, Nk) or not key_bias.is_contiguous():
            raise RuntimeError("flash_attention: key_bias must be contiguous [B,Nk]")
    part = lse = None
    if key_splits > 1:
        part = torch.empty((B * key_splits, Nq, H, hd), dtype=torch.bfloat16, device=q.device)
        lse = torch.empty((B * key_splits, H, Nq), dtype=torch.float32, device=q.device)
    d = attn_desc(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, Nq, Nk,
                  q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                  q.stride(0), k.stride(0), v.stride(0), out.stride(0), _ptr(key_bias), head_dim=hd, causal=causal,
                  key_splits=key_splits, o_part=_ptr(part), lse_part=_ptr(lse))
    run_attn(d)
    return out


def layernorm_raw(x_ptr, in_f32: bool, add_ptr, gamma_ptr, beta_ptr, out_b_ptr, out_f_ptr, rows: int, D: int,
                  eps: float, out_rows: Optional[Rows] = None):
    orow = C.byref(L.HkRows(*out_rows)) if out_rows is not None else None
    L.check(L.load().hk_layernorm(x_ptr, L.HK_DT_F32 if in_f32 else L.HK_DT_BF16, add_ptr, gamma_ptr, beta_ptr,
                                  out_b_ptr, out_f_ptr, orow, rows, D, eps, _stream()), "hk_layernorm")


def linear_split_k(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor],
                   split_k: int) -> torch.Tensor:
    """Split-K partial products [split_k, M, N] fp32 of x @ w.T; slice 0 carries bias and residual (fp32 [M, N])."""
    _need(x, torch.bfloat16, "x")
    _need(w, torch.bfloat16, "w")
    M, K = x.shape
    N = w.shape[0]
    parts = torch.empty((split_k, M, N), dtype=torch.float32, device=x.device)
    d = gemm_desc(x.data_ptr(), w.data_ptr(), parts.data_ptr(), M, N, K, bias=_ptr(bias), residual=_ptr(residual),
                  a_rows=(x.stride(0), 0, 0), c_rows=(N, 0, 0), r_rows=(N, 0, 0), ldw=w.stride(0), out_f32=True,
                  split_k=split_k, split_stride=M * N)
    run_gemm(d)
    return parts


def residual_add_parts(x: torch.Tensor, parts: torch.Tensor, eps: float, want_stats: bool = True):
    """x (bf16 [rows, D], in place) += parts.sum(0) (fp32 [P, rows, D]); returns (mean, rstd) [rows, 2] of the stored rows."""
    _need(x, torch.bfloat16, "x")
    _need(parts, torch.float32, "parts")
    P, rows, D = parts.shape
    st = torch.empty((rows, 2), dtype=torch.float32, device=x.device) if want_stats else None
    L.check(L.load().hk_residual_add_parts(x.data_ptr(), parts.data_ptr(), P, rows * D, _ptr(st), rows, D, eps, _stream()),
            "hk_residual_add_parts")
    return st


def layernorm_parts(parts: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float):
    """LayerNorm of sum(parts, 0); parts fp32 [P, rows, D] (split-K slices).  Returns (bf16, fp32)."""
    _need(parts, torch.float32, "parts")
    P, rows, D = parts.shape
    ob = torch.empty((rows, D), dtype=torch.bfloat16, device=parts.device)
    of = torch.empty((rows, D), dtype=torch.float32, device=parts.device)
    L.check(L.load().hk_layernorm_parts(parts.data_ptr(), L.HK_DT_F32, parts.data_ptr() + 4 * rows * D, P - 1, rows * D,
                                        gamma.data_ptr(), beta.data_ptr(), ob.data_ptr(), of.data_ptr(), None, rows, D,
                                        eps, _stream()), "hk_layernorm_parts")
    return ob, of


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, *,
              add: Optional[torch.Tensor] = None, want_bf16=True, want_f32=False):
    """LayerNorm over the last dim of x [rows, D] (fp32 or bf16); returns (bf16 | None, fp32 | None)."""
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_cuda or not x.is_contiguous():
        raise RuntimeError("layernorm: x must be a contiguous fp32/bf16 GPU tensor")
    _need(gamma, torch.float32, "gamma")
    _need(beta, torch.float32, "beta")
    rows, D = x.shape
    ob = torch.empty((rows, D), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    of = torch.empty((rows, D), dtype=torch.float32, device=x.device) if want_f32 else None
    if add is not None:
        _need(add, torch.float32, "add")
        if add.shape != x.shape or not add.is_contiguous():
            raise RuntimeError("layernorm: add must match x")
    layernorm_raw(x.data_ptr(), x.dtype == torch.float32, _ptr(add), gamma.data_ptr(), beta.data_ptr(), _ptr(ob),
                  _ptr(of), rows, D, eps)
    return ob, of


def patchify(clip: torch.Tensor, patch: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """clip [B,T,3,H,W] fp32 contiguous -> [B*T*(H/p)*(W/p), 3*p*p] bf16."""
    _need(clip, torch.float32, "clip")
    if clip.dim() != 5 or clip.shape[2] != 3 or not clip.is_contiguous():
        raise RuntimeError("patchify: clip must be contiguous [B,T,3,H,W]")
    B, T, _, H, W = clip.shape
    rows = B * T * (H // patch) * (W // patch)
    if out is None:
        out = torch.empty((rows, 3 * patch * patch), dtype=torch.bfloat16, device=clip.device)
    L.check(L.load().hk_patchify_bf16(clip.data_ptr(), out.data_ptr(), B, T, H, W, patch, _stream()),
            "hk_patchify_bf16")
    return out


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # tasks/test_recursive_grounding.py:333-334
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def patchify_u8(frames: torch.Tensor, patch: int, mean=CLIP_MEAN, std=CLIP_STD,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """frames [B,T,H,W,3] uint8 contiguous -> normalised [B*T*(H/p)*(W/p), 3*p*p] bf16 (row f3)."""
    _need(frames, torch.uint8, "frames")
    if frames.dim() != 5 or frames.shape[4] != 3 or not frames.is_contiguous():
        raise RuntimeError("patchify_u8: frames must be contiguous [B,T,H,W,3]")
    B, T, H, W, _ = frames.shape
    rows = B * T * (H // patch) * (W // patch)
    if out is None:
        out = torch.empty((rows, 3 * patch * patch), dtype=torch.bfloat16, device=frames.device)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    L.check(L.load().hk_patchify_u8(frames.data_ptr(), out.data_ptr(), B, T, H, W, patch, m3, s3, _stream()),
            "hk_patchify_u8")
    return out


def pos_embed(T: int, Lp: int, D: int, ckpt_frames: int, device) -> torch.Tensor:
    """fp32 [T*Lp, D] sinusoid table generated on the device (row f4); ckpt_frames <= 0 = no interpolation."""
    out = torch.empty((T * Lp, D), dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        L.check(L.load().hk_pos_embed(out.data_ptr(), T, Lp, D, ckpt_frames, _stream()), "hk_pos_embed")
    return out


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LlamaRMSNorm: x [rows, D] fp32 or bf16 -> bf16 [rows, D]; weight fp32 [D]."""
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_cuda or not x.is_contiguous() or x.dim() != 2:
        raise RuntimeError("rmsnorm: x must be a contiguous 2-D fp32/bf16 GPU tensor")
    _need(weight, torch.float32, "weight")
    rows, D = x.shape
    if out is None:
        out = torch.empty((rows, D), dtype=torch.bfloat16, device=x.device)
    L.check(L.load().hk_rmsnorm(x.data_ptr(), L.HK_DT_F32 if x.dtype == torch.float32 else L.HK_DT_BF16,
                                weight.data_ptr(), out.data_ptr(), rows, D, eps, _stream()), "hk_rmsnorm")
    return out


def rmsnorm_add_parts(parts: torch.Tensor, weight: torch.Tensor, eps: float):
    """Split-K slices fp32 [P, rows, D] (slice 0 carries the residual stream) -> (x fp32 [rows, D] = their sum, bf16 RMSNorm(x) * weight)."""
    _need(parts, torch.float32, "parts")
    _need(weight, torch.float32, "weight")
    if parts.dim() != 3 or not parts.is_contiguous():
        raise RuntimeError("rmsnorm_add_parts: parts must be a contiguous [P, rows, D] tensor")
    P, rows, D = parts.shape
    x = torch.empty((rows, D), dtype=torch.float32, device=parts.device)
    out = torch.empty((rows, D), dtype=torch.bfloat16, device=parts.device)
    L.check(L.load().hk_rmsnorm_add_parts(parts.data_ptr(), P, rows * D, x.data_ptr(), weight.data_ptr(), out.data_ptr(), rows, D,
                                          eps, _stream()), "hk_rmsnorm_add_parts")
    return x, out


def rope_(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, H: int, head_dim: int, seq_len: int,
          q_scale: float = 1.0, position_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Rotary embedding in place on the q|k thirds of qkv bf16 [rows, 3*H*hd]; cos/sin fp32 [P, hd/2]."""
    _need(qkv, torch.bfloat16, "qkv")
    _need(cos, torch.float32, "cos")
    _need(sin, torch.float32, "sin")
    if qkv.dim() != 2 or qkv.shape[1] != 3 * H * head_dim or not qkv.is_contiguous():
        raise RuntimeError("rope_: qkv must be contiguous [rows, 3*H*head_dim]")
    if cos.shape != sin.shape or cos.shape[1] != head_dim // 2 or not cos.is_contiguous() or not sin.is_contiguous():
        raise RuntimeError("rope_: cos/sin must be contiguous [P, head_dim/2]")
    if position_ids is not None:
        _need(position_ids, torch.int32, "position_ids")
        if position_ids.numel() != qkv.shape[0] or int(position_ids.max()) >= cos.shape[0] or int(position_ids.min()) < 0:
            raise RuntimeError("rope_: position_ids must hold one in-table position per row")
    L.check(L.load().hk_rope_bf16(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), _ptr(position_ids), qkv.shape[0], H,
                                  head_dim, seq_len, cos.shape[0], q_scale, _stream()), "hk_rope_bf16")
    return qkv


def tile_gemv_weight(w: torch.Tensor) -> torch.Tensor:
    """Row-major bf16 [N, K] -> hk_gemv_desc.w_tiled's block layout [ceil(N/16), K/64, 2, 64, 8] (rows past N zero).
    A layout permutation done once at load time (torch as plumbing)."""
    _need(w, torch.bfloat16, "w")
    N, K = w.shape
    if K % 64:
        raise RuntimeError("tile_gemv_weight: K must be a multiple of 64")
    R = (N + 15) // 16
    if R * 16 != N:
        w = torch.cat([w, torch.zeros((R * 16 - N, K), dtype=w.dtype, device=w.device)])
    # [r, i, b, q, h, e] -> [r, b, h, q, i, e]
    return w.view(R, 16, K // 64, 4, 2, 8).permute(0, 2, 4, 3, 1, 5).contiguous().view(R, K // 64, 2, 64, 8)


def gemv_desc(x, W, out, M, N, K, *, ldx=None, ldo=None, residual=None, out_f32=False, act="none", norm_weight=None,
              norm_eps=0.0, w_tiled=False) -> L.HkGemvDesc:
    """Descriptor of one hk_gemv_bf16 call on raw pointers (the decode plan builds these once)."""
    d = L.HkGemvDesc()
    d.x, d.W, d.out, d.residual = x, W, out, residual
    d.M, d.N, d.K, d.out_f32 = M, N, K, 1 if out_f32 else 0
    d.act = {"none": L.HK_ACT_NONE, "swiglu": L.HK_ACT_SWIGLU}[act]
    d.ldx = K if ldx is None else ldx
    d.ldo = (N // 2 if act == "swiglu" else N) if ldo is None else ldo
    d.norm_weight, d.norm_eps = norm_weight, norm_eps
    d.w_tiled, d.reserved = 1 if w_tiled else 0, 0
    return d


def gemv(x: torch.Tensor, w: torch.Tensor, *, residual: Optional[torch.Tensor] = None, out_f32: bool = False,
         act: str = "none", norm_weight: Optional[torch.Tensor] = None, norm_eps: float = 1e-6) -> torch.Tensor:
    """x bf16 [M <= 16, K] . w bf16 [N, K]^T -> [M, N] (bf16, or fp32 with out_f32; + fp32 residual), or
    [M, N/2] = silu(gate) * up for act="swiglu" on interleaved (gate, up) weight rows.  The decode step's Linear.
    norm_weight fp32 [K]: x is fp32 [M <= 4, K] and the operand is RMSNorm(x) * norm_weight (fused hk_rmsnorm)."""
    if norm_weight is not None:
        _need(norm_weight, torch.float32, "norm_weight")
    _need(x, torch.float32 if norm_weight is not None else torch.bfloat16, "x")
    _need(w, torch.bfloat16, "w")
    if x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1] or not x.is_contiguous() or not w.is_contiguous():
        raise RuntimeError("gemv: x [M, K] and w [N, K] must be contiguous with equal K")
    M, K = x.shape
    N = w.shape[0]
    if residual is not None:
        _need(residual, torch.float32, "residual")
        if tuple(residual.shape) != (M, N) or not residual.is_contiguous():
            raise RuntimeError("gemv: residual must be contiguous fp32 [M, N]")
    out = torch.empty((M, N // 2 if act == "swiglu" else N), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    d = gemv_desc(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, residual=_ptr(residual), out_f32=out_f32, act=act,
                  norm_weight=_ptr(norm_weight), norm_eps=norm_eps if norm_weight is not None else 0.0)
    L.check(L.load().hk_gemv_bf16(C.byref(d), _stream()), "hk_gemv_bf16")
    return out


def decode_rope_append(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos0: torch.Tensor, pos_add: int,
                       k_cache: torch.Tensor, v_cache: torch.Tensor, slot: int, H: int, head_dim: int, q_scale: float) -> torch.Tensor:
    """qkv bf16 [B, 3*H*hd] of the new token -> q (rotary at pos0[b] + pos_add, scaled) bf16 [B, H*hd]; rotary k and v written to
    row `slot` of k_cache / v_cache bf16 [B, rows, H*hd]."""
    _need(qkv, torch.bfloat16, "qkv")
    _need(pos0, torch.int32, "pos0")
    _need(k_cache, torch.bfloat16, "k_cache")
    _need(v_cache, torch.bfloat16, "v_cache")
    B, D = qkv.shape[0], H * head_dim
    if tuple(qkv.shape) != (B, 3 * D) or not qkv.is_contiguous() or k_cache.shape != v_cache.shape or k_cache.dim() != 3 \
            or k_cache.shape[0] != B or k_cache.shape[2] != D or not k_cache.is_contiguous() or not v_cache.is_contiguous():
        raise RuntimeError("decode_rope_append: qkv [B, 3*H*hd], caches contiguous [B, rows, H*hd]")
    if pos0.numel() != B or int(pos0.min()) < 0 or int(pos0.max()) + pos_add >= cos.shape[0]:
        raise RuntimeError("decode_rope_append: positions outside the rotary table")
    q = torch.empty((B, D), dtype=torch.bfloat16, device=qkv.device)
    L.check(L.load().hk_decode_rope_append(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos0.data_ptr(), pos_add, q.data_ptr(),
                                           k_cache.data_ptr(), v_cache.data_ptr(), B, H, head_dim, slot, k_cache.shape[1],
                                           cos.shape[0], q_scale, _stream()), "hk_decode_rope_append")
    return q


def _split_ws(B, H, head_dim, key_splits, device):
    if key_splits <= 1:
        return None
    return torch.empty((B * H * min(key_splits, 16) * (head_dim + 2),), dtype=torch.float32, device=device)


def decode_attn(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, n_keys: int, H: int, head_dim: int,
                key_bias: Optional[torch.Tensor] = None, key_splits: int = 1) -> torch.Tensor:
    """One query row per (sequence, head) over cache rows [0, n_keys): bf16 [B, H*hd]."""
    _need(q, torch.bfloat16, "q")
    _need(k_cache, torch.bfloat16, "k_cache")
    _need(v_cache, torch.bfloat16, "v_cache")
    B = q.shape[0]
    if key_bias is not None:
        _need(key_bias, torch.float32, "key_bias")
        if key_bias.dim() != 2 or key_bias.shape[0] != B or key_bias.stride(1) != 1:
            raise RuntimeError("decode_attn: key_bias must be fp32 [B, >= n_keys]")
    out = torch.empty_like(q)
    ws = _split_ws(B, H, head_dim, key_splits, q.device)
    L.check(L.load().hk_decode_attn_bf16(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), _ptr(key_bias), out.data_ptr(), B, H,
                                         head_dim, n_keys, k_cache.shape[1], 0 if key_bias is None else key_bias.stride(0),
                                         key_splits, _ptr(ws), _stream()), "hk_decode_attn_bf16")
    return out


def decode_attn_rope(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos0: torch.Tensor, pos_add: int,
                     k_cache: torch.Tensor, v_cache: torch.Tensor, slot: int, H: int, head_dim: int, q_scale: float,
                     key_bias: Optional[torch.Tensor] = None, key_splits: int = 1) -> torch.Tensor:
    """decode_rope_append + decode_attn over cache rows [0, slot] in one launch: bf16 [B, H*hd]."""
    _need(qkv, torch.bfloat16, "qkv")
    _need(pos0, torch.int32, "pos0")
    _need(k_cache, torch.bfloat16, "k_cache")
    _need(v_cache, torch.bfloat16, "v_cache")
    B, D = qkv.shape[0], H * head_dim
    if tuple(qkv.shape) != (B, 3 * D) or not qkv.is_contiguous() or k_cache.shape != v_cache.shape or k_cache.dim() != 3 \
            or k_cache.shape[0] != B or k_cache.shape[2] != D or not k_cache.is_contiguous() or not v_cache.is_contiguous():
        raise RuntimeError("decode_attn_rope: qkv [B, 3*H*hd], caches contiguous [B, rows, H*hd]")
    if pos0.numel() != B or int(pos0.min()) < 0 or int(pos0.max()) + pos_add >= cos.shape[0]:
        raise RuntimeError("decode_attn_rope: positions outside the rotary table")
    if key_bias is not None:
        _need(key_bias, torch.float32, "key_bias")
    out = torch.empty((B, D), dtype=torch.bfloat16, device=qkv.device)
    ws = _split_ws(B, H, head_dim, key_splits, qkv.device)
    L.check(L.load().hk_decode_attn_rope_bf16(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos0.data_ptr(), pos_add,
                                              k_cache.data_ptr(), v_cache.data_ptr(), _ptr(key_bias), out.data_ptr(), B, H, head_dim,
                                              slot, k_cache.shape[1], 0 if key_bias is None else key_bias.stride(0), cos.shape[0],
                                              q_scale, key_splits, _ptr(ws), _stream()), "hk_decode_attn_rope_bf16")
    return out


def argmax(logits: torch.Tensor) -> torch.Tensor:
    """fp32 [rows, V] -> int64 [rows], the lowest index of the largest value."""
    _need(logits, torch.float32, "logits")
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError("argmax: logits must be fp32 [rows, V] with unit column stride")
    out = torch.empty((logits.shape[0],), dtype=torch.int64, device=logits.device)
    L.check(L.load().hk_argmax_f32(logits.data_ptr(), out.data_ptr(), logits.shape[0], logits.shape[1], logits.stride(0), _stream()),
            "hk_argmax_f32")
    return out


def silu_mul(gate_up: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gate_up bf16 [rows, 2*I] (gate | up) -> silu(gate) * up, bf16 [rows, I]."""
    _need(gate_up, torch.bfloat16, "gate_up")
    if gate_up.dim() != 2 or gate_up.shape[1] % 2 or not gate_up.is_contiguous():
        raise RuntimeError("silu_mul: gate_up must be contiguous [rows, 2*I]")
    rows, I = gate_up.shape[0], gate_up.shape[1] // 2
    if out is None:
        out = torch.empty((rows, I), dtype=torch.bfloat16, device=gate_up.device)
    L.check(L.load().hk_silu_mul_bf16(gate_up.data_ptr(), out.data_ptr(), rows, I, _stream()), "hk_silu_mul_bf16")
    return out


def embed_splice(src: torch.Tensor, table: torch.Tensor, video: Optional[torch.Tensor],
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src int64 [rows]: >= 0 token id, < 0 video row -src-1; table bf16 [V,D]; video [*,D] bf16/fp32 -> fp32 [rows,D]."""
    _need(src, torch.int64, "src")
    _need(table, torch.bfloat16, "table")
    D = table.shape[1]
    rows = src.numel()
    if int(src.max()) >= table.shape[0]:
        raise RuntimeError("embed_splice: token id outside the embedding table")
    nvid = 0
    if video is not None:
        if video.dtype not in (torch.float32, torch.bfloat16) or not video.is_cuda or not video.is_contiguous():
            raise RuntimeError("embed_splice: video must be a contiguous fp32/bf16 GPU tensor")
        video = video.reshape(-1, D)
        nvid = video.shape[0]
    if int(src.min()) < -nvid:
        raise RuntimeError("embed_splice: video row index outside the video tensor")
    if out is None:
        out = torch.empty((rows, D), dtype=torch.float32, device=table.device)
    vdt = L.HK_DT_BF16 if (video is None or video.dtype == torch.bfloat16) else L.HK_DT_F32
    L.check(L.load().hk_embed_splice(src.contiguous().data_ptr(), table.data_ptr(), _ptr(video), vdt, out.data_ptr(), rows, D,
                                     _stream()), "hk_embed_splice")
    return out


def qformer_embed_ln(query_tokens, word_emb, pos_emb, input_ids, gamma, beta, eps, out_f32, out_bf16):
    """query_tokens [nq,d], word_emb [V,d], pos_emb [P,d] fp32; input_ids [B,Lt] int64 (or None)."""
    nq, d = query_tokens.shape
    B, S, _ = out_f32.shape
    Lt = S - nq
    if Lt > 0:
        _need(input_ids, torch.int64, "input_ids")
        if input_ids.shape != (B, Lt) or not input_ids.is_contiguous():
            raise RuntimeError("qformer_embed_ln: input_ids must be contiguous [B,Lt]")
        if Lt > pos_emb.shape[0]:
            raise RuntimeError("qformer_embed_ln: instruction longer than the position table")
    L.check(L.load().hk_qformer_embed_ln(
        query_tokens.data_ptr(), _ptr(word_emb), _ptr(pos_emb), _ptr(input_ids) if Lt > 0 else None,
        gamma.data_ptr(), beta.data_ptr(), out_f32.data_ptr(), out_bf16.data_ptr(), B, nq, Lt, d,
        word_emb.shape[0] if word_emb is not None else 1, eps, _stream()), "hk_qformer_embed_ln")


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    _need(x, torch.float32, "x")
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    L.check(L.load().hk_cast_f32_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "hk_cast_f32_bf16")
    return out


def mfma_probe_tflops(kind: int = 0, *, zeros: bool = False, grid: int = 2048, iters: int = 4000, reps: int = 5) -> float:
    """Measured dense-bf16 MFMA rate of this chip (register-resident probe, include/hawkeye_hip.h hk_mfma_probe).
    zeros=True feeds all-zero operands (lowest switching power, highest sustained clock)."""
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(7)
    seed = (torch.zeros(2048) if zeros else torch.randn(2048, generator=g)).to(torch.bfloat16).to(dev)
    out = torch.empty(grid * 256, dtype=torch.float32, device=dev)
    lib = L.load()
    st = _stream()
    L.check(lib.hk_mfma_probe(seed.data_ptr(), out.data_ptr(), grid, iters // 10, kind, st), "hk_mfma_probe")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.hk_mfma_probe(seed.data_ptr(), out.data_ptr(), grid, iters, kind, st), "hk_mfma_probe")
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return grid * iters * 524288.0 / (ms * 1e-3) / 1e12
