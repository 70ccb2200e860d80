"""Tensor-level wrappers over the training entry points of the C ABI (row f2, include/hawkeye_hip.h).

Same rules as hawkeye_amd.ops: PyTorch owns memory and the stream, nothing here computes.
"""
import ctypes as C
from typing import Optional, Tuple

import torch

from . import lib as L
from .ops import _need, _ptr, _stream, Rows


def dropout(p: float, seed: int, site: int, p_path: float = 0.0, path_site: int = 0, rows_per_sample: int = 1) -> L.HkDropout:
    d = L.HkDropout()
    d.p, d.p_path, d.seed, d.site, d.path_site, d.rows_per_sample = p, p_path, seed & 0xFFFFFFFF, site, path_site, rows_per_sample
    return d


def _rows(r: Optional[Rows]):
    return None if r is None else L.HkRows(*r)


def _ref(s):
    return None if s is None else C.byref(s)


def attn_train_desc(q, k, v, o, lse, key_bias, B, H, Nq, Nk, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, *, scale,
                    p_drop=0.0, seed=0, site=0, do=None, dq=None, dk=None, dv=None, dsum=None,
                    lddq=0, lddk=0, lddv=0, bsdq=0, bsdk=0, bsdv=0, o_f32=None) -> L.HkAttnTrainDesc:
    """Raw-pointer descriptor (integer device addresses)."""
    d = L.HkAttnTrainDesc()
    d.Q, d.K, d.V, d.O, d.lse, d.key_bias = q, k, v, o, lse, key_bias
    d.dO, d.dQ, d.dK, d.dV, d.dsum = do, dq, dk, dv, dsum
    d.B, d.H, d.Nq, d.Nk = B, H, Nq, Nk
    d.ldq, d.ldk, d.ldv, d.ldo, d.bsq, d.bsk, d.bsv, d.bso = ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso
    d.lddq, d.lddk, d.lddv, d.bsdq, d.bsdk, d.bsdv = lddq, lddk, lddv, bsdq, bsdk, bsdv
    d.softmax_scale, d.p_drop, d.seed, d.site = scale, p_drop, seed & 0xFFFFFFFF, site
    d.o_f32 = o_f32
    return d


def _check_qkv(q, k, v):
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _need(t, torch.bfloat16, n)
        if t.dim() != 4 or t.shape[3] != 64 or t.stride(3) != 1 or t.stride(2) != 64:
            raise RuntimeError(f"attention(train): {n} must be [B,N,H,64] with dense (H,64) dims")


def attn_train_fwd(q, k, v, key_bias=None, *, scale, p_drop=0.0, seed=0, site=0, o_f32=None):
    """q [B,Nq,H,64] (pre-scaled by scale*log2 e), k/v [B,Nk,H,64] bf16 -> (o bf16 [B,Nq,H,64], lse fp32 [B,H,Nq]).
    o_f32: optional dense fp32 [B,Nq,H,64] that receives the unrounded output."""
    _check_qkv(q, k, v)
    B, Nq, H, _ = q.shape
    Nk = k.shape[1]
    o = torch.empty((B, Nq, H, 64), dtype=torch.bfloat16, device=q.device)
    lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
    d = attn_train_desc(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), _ptr(key_bias), B, H, Nq, Nk,
                        q.stride(1), k.stride(1), v.stride(1), o.stride(1), q.stride(0), k.stride(0), v.stride(0),
                        o.stride(0), scale=scale, p_drop=p_drop, seed=seed, site=site, o_f32=_ptr(o_f32))
    L.check(L.load().hk_attn_train_fwd(C.byref(d), _stream()), "hk_attn_train_fwd")
    return o, lse


def attn_train_bwd(q, k, v, o, lse, do, key_bias=None, *, scale, p_drop=0.0, seed=0, site=0, o_f32=None):
    """Gradients (bf16) of attn_train_fwd w.r.t. the unscaled q, k and v."""
    _check_qkv(q, k, v)
    _need(do, torch.bfloat16, "do")
    B, Nq, H, _ = q.shape
    Nk = k.shape[1]
    if do.shape != o.shape or do.stride() != o.stride():
        raise RuntimeError("attention(train): do must share o's layout")
    dq = torch.empty((B, Nq, H, 64), dtype=torch.bfloat16, device=q.device)
    dk = torch.empty((B, Nk, H, 64), dtype=torch.bfloat16, device=q.device)
    dv = torch.empty((B, Nk, H, 64), dtype=torch.bfloat16, device=q.device)
    dsum = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
    d = attn_train_desc(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), _ptr(key_bias), B, H, Nq, Nk,
                        q.stride(1), k.stride(1), v.stride(1), o.stride(1), q.stride(0), k.stride(0), v.stride(0),
                        o.stride(0), scale=scale, p_drop=p_drop, seed=seed, site=site, do=do.data_ptr(),
                        dq=dq.data_ptr(), dk=dk.data_ptr(), dv=dv.data_ptr(), dsum=dsum.data_ptr(),
                        lddq=dq.stride(1), lddk=dk.stride(1), lddv=dv.stride(1), bsdq=dq.stride(0), bsdk=dk.stride(0),
                        bsdv=dv.stride(0), o_f32=_ptr(o_f32))
    L.check(L.load().hk_attn_train_bwd(C.byref(d), _stream()), "hk_attn_train_bwd")
    return dq, dk, dv


def drop_add_ln_raw(t, res, res_rows, drop, post, gamma, beta, z_out, y_f, y_b, out_rows, rows, D, eps):
    """All pointers are integer device addresses (or None); drop/post are HkDropout or None."""
    rr, orr = _rows(res_rows), _rows(out_rows)
    L.check(L.load().hk_drop_add_ln(t, res, _ref(rr), _ref(drop), _ref(post), gamma, beta, z_out, y_f, y_b, _ref(orr),
                                    rows, D, eps, _stream()), "hk_drop_add_ln")


def drop_add_ln(t, res, gamma, beta, eps, drop=None, post=None):
    """t, res fp32 [rows, D] (either may be None) -> (z fp32, y fp32, y bf16)."""
    ref = t if t is not None else res
    rows, D = ref.shape
    z = torch.empty((rows, D), dtype=torch.float32, device=ref.device)
    yf = torch.empty_like(z)
    yb = torch.empty((rows, D), dtype=torch.bfloat16, device=ref.device)
    drop_add_ln_raw(_ptr(t), _ptr(res), None, drop, post, gamma.data_ptr(), beta.data_ptr(), z.data_ptr(), yf.data_ptr(),
                    yb.data_ptr(), None, rows, D, eps)
    return z, yf, yb


def lnbwd_ws_floats(D: int) -> int:
    return L.HK_LNBWD_MAX_BLOCKS * 2 * D


def layernorm_bwd_raw(dy, dy_rows, z, gamma, drop, post, dz, dt, dgb, ws, rows, D, eps):
    """dgb: address of the contiguous [2, D] fp32 block (dgamma | dbeta)."""
    dr = _rows(dy_rows)
    L.check(L.load().hk_layernorm_bwd(dy, _ref(dr), z, gamma, _ref(drop), _ref(post), dz, dt, dgb, dgb + 4 * D, ws, rows, D,
                                      eps, _stream()), "hk_layernorm_bwd")


def layernorm_bwd(dy, z, gamma, eps, drop=None, post=None):
    """dy, z fp32 [rows, D] -> (dz fp32, dt bf16, dgamma, dbeta)."""
    rows, D = z.shape
    dz = torch.empty_like(z)
    dt = torch.empty((rows, D), dtype=torch.bfloat16, device=z.device)
    dgb = torch.empty((2, D), dtype=torch.float32, device=z.device)
    ws = torch.empty(lnbwd_ws_floats(D), dtype=torch.float32, device=z.device)
    layernorm_bwd_raw(dy.data_ptr(), None, z.data_ptr(), gamma.data_ptr(), drop, post, dz.data_ptr(), dt.data_ptr(),
                      dgb.data_ptr(), ws.data_ptr(), rows, D, eps)
    return dz, dt, dgb[0], dgb[1]


def colsum_raw(x, bf16: bool, ld, R, Cc, out, ws):
    L.check(L.load().hk_colsum(x, L.HK_DT_BF16 if bf16 else L.HK_DT_F32, ld, R, Cc, out, ws, _stream()), "hk_colsum")


def colsum(x: torch.Tensor) -> torch.Tensor:
    R, Cc = x.shape
    out = torch.empty(Cc, dtype=torch.float32, device=x.device)
    ws = torch.empty(L.HK_COLSUM_MAX_SPLIT * Cc, dtype=torch.float32, device=x.device)
    colsum_raw(x.data_ptr(), x.dtype == torch.bfloat16, x.stride(0), R, Cc, out.data_ptr(), ws.data_ptr())
    return out


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def transpose_raw(x, f32: bool, x_rows, R, Cc, out, ldo):
    xr = _rows(x_rows)
    L.check(L.load().hk_transpose_bf16(x, L.HK_DT_F32 if f32 else L.HK_DT_BF16, _ref(xr), R, Cc, out, ldo, _stream()),
            "hk_transpose_bf16")


def transpose(x: torch.Tensor) -> torch.Tensor:
    """x [R, C] fp32/bf16 -> bf16 [C, roundup(R, 64)], zero padded."""
    R, Cc = x.shape
    out = torch.empty((Cc, pad64(R)), dtype=torch.bfloat16, device=x.device)
    transpose_raw(x.data_ptr(), x.dtype == torch.float32, (x.stride(0), 0, 0), R, Cc, out.data_ptr(), out.stride(0))
    return out


def wgrad(dy: torch.Tensor, x: torch.Tensor, x_rows: Optional[Rows] = None, rows: Optional[int] = None,
          bias: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """dW [N, K] fp32 = dy^T x and db [N] = column sums of dy, in one launch (hk_wgrad_bf16).  dy bf16 [R, N] (last dim
    dense); x bf16: [R, K], or any buffer addressed through x_rows = (ld, rows_per_batch, batch_stride) with `rows` given."""
    _need(dy, torch.bfloat16, "dy")
    _need(x, torch.bfloat16, "x")
    R, N = (dy.shape[0] if rows is None else rows), dy.shape[1]
    if x.dim() != 2 or x.stride(1) != 1:
        raise RuntimeError("wgrad: x must be a 2-D tensor with a dense last dimension (K)")
    K = x.shape[1]
    xr = L.HkRows(x.stride(0), 0, 0) if x_rows is None else L.HkRows(*x_rows)
    dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    db = torch.empty((N,), dtype=torch.float32, device=dy.device) if bias else None
    L.check(L.load().hk_wgrad_bf16(dy.data_ptr(), dy.stride(0), x.data_ptr(), C.byref(xr), dw.data_ptr(), K,
                                   None if db is None else db.data_ptr(), R, N, K, _stream()), "hk_wgrad_bf16")
    return dw, db


def gelu(u: torch.Tensor) -> torch.Tensor:
    _need(u, torch.bfloat16, "u")
    g = torch.empty_like(u)
    L.check(L.load().hk_gelu_bf16(u.data_ptr(), g.data_ptr(), u.numel(), _stream()), "hk_gelu_bf16")
    return g


def gelu_bwd(dg: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    _need(u, torch.bfloat16, "u")
    _need(dg, torch.bfloat16, "dg")
    du = torch.empty_like(u)
    L.check(L.load().hk_gelu_bwd_bf16(dg.data_ptr(), u.data_ptr(), du.data_ptr(), u.numel(), _stream()), "hk_gelu_bwd_bf16")
    return du


def embed_gather(query, word, pos, ids, B, nq, Lt) -> torch.Tensor:
    d = query.shape[1]
    z = torch.empty((B, nq + Lt, d), dtype=torch.float32, device=query.device)
    L.check(L.load().hk_qformer_embed_gather(query.data_ptr(), _ptr(word), _ptr(pos), _ptr(ids) if Lt else None, z.data_ptr(),
                                             B, nq, Lt, d, word.shape[0] if word is not None else 1, _stream()),
            "hk_qformer_embed_gather")
    return z


def embed_bwd(dz, ids, nq, vocab, max_pos) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    B, S, d = dz.shape
    Lt = S - nq
    dq = torch.empty((nq, d), dtype=torch.float32, device=dz.device)
    dw = torch.zeros((vocab, d), dtype=torch.float32, device=dz.device)
    dp = torch.zeros((max_pos, d), dtype=torch.float32, device=dz.device)
    L.check(L.load().hk_qformer_embed_bwd(dz.data_ptr(), _ptr(ids) if Lt else None, dq.data_ptr(), dw.data_ptr(), dp.data_ptr(),
                                          B, nq, Lt, d, vocab, _stream()), "hk_qformer_embed_bwd")
    return dq, dw, dp
