"""Row f2 kernels on a real MI355X, through the C ABI, against fp32 PyTorch (autograd) evaluated on the same
bf16-rounded inputs with the SAME dropout masks (oracle/dropout_masks.py restates the device hash).

Tolerances: bf16 outputs carry rel 2^-8 rounding; attention gradients additionally round P / dS to bf16 before
the second MFMA, so they are compared at 2e-2 of the tensor's max.  Integer / mask results are exact.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hawkeye_amd import ops, train_ops as T  # noqa: E402
from oracle import dropout_masks as DM  # noqa: E402

DEV = "cuda:0"
LOG2E = 1.4426950408889634


def _rand(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV).to(dtype)


def _rel(a, ref):
    return ((a.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)).item()


def _mask(seed, site, shape, p):
    return torch.from_numpy(DM.keep_mask(seed, site, shape, p)).to(DEV)


def test_device_dropout_mask_equals_numpy_restatement():
    rows, D, p, seed, site = 37, 128, 0.3, 1234, 77
    t = torch.ones((rows, D), device=DEV)
    g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    z, _, _ = T.drop_add_ln(t, None, g, b, 1e-12, drop=T.dropout(p, seed, site))
    keep = _mask(seed, site, (rows, D), p)
    assert torch.equal(z != 0, keep)
    assert abs(keep.float().mean().item() - (1 - p)) < 0.03
    assert torch.equal(z[keep], torch.full_like(z[keep], float(DM.scale(p))))


@pytest.mark.parametrize("B,H,Nq,Nk,p,bias", [(2, 2, 24, 24, 0.0, True), (2, 12, 128, 128, 0.1, True),
                                              (1, 3, 96, 400, 0.1, False), (2, 2, 16, 197, 0.25, False),
                                              (1, 2, 130, 70, 0.1, True), (2, 12, 96, 2353, 0.1, False), (1, 1, 5, 30, 0.0, True)])
def test_attention_train_fwd_bwd(B, H, Nq, Nk, p, bias):
    seed, site = 99, DM.site(3, DM.SELF_PROB)
    scale = 1.0 / 8.0
    q = _rand((B, Nq, H, 64), 1, dtype=torch.bfloat16)
    k = _rand((B, Nk, H, 64), 2, dtype=torch.bfloat16)
    v = _rand((B, Nk, H, 64), 3, dtype=torch.bfloat16)
    do = _rand((B, Nq, H, 64), 4, dtype=torch.bfloat16)
    kb = None
    if bias:
        kb = torch.zeros((B, Nk), device=DEV)
        kb[:, Nk - Nk // 4:] = -10000.0
        kb[0, :] = 0.0
    qs = (q.float() * (scale * LOG2E)).to(torch.bfloat16)    # what the producing GEMM's epilogue hands over
    o32 = torch.empty((B, Nq, H, 64), dtype=torch.float32, device=DEV)
    o, lse = T.attn_train_fwd(qs, k, v, kb, scale=scale, p_drop=p, seed=seed, site=site, o_f32=o32)
    assert torch.equal(o32.to(torch.bfloat16), o)
    # fp32 reference with autograd; the unscaled query it differentiates is qs / (scale * log2e)
    qu = (qs.float() / (scale * LOG2E)).requires_grad_(True)
    kf, vf = k.float().requires_grad_(True), v.float().requires_grad_(True)
    s = torch.einsum("bqhd,bkhd->bhqk", qu, kf) * scale
    if kb is not None:
        s = s + kb[:, None, None, :]
    pr = s.softmax(-1)
    keep = _mask(seed, site, (B, H, Nq, Nk), p)
    prd = pr * keep * float(DM.scale(p))
    ref = torch.einsum("bhqk,bkhd->bqhd", prd, vf)
    assert _rel(o.float(), ref) < 1.5e-2
    ref_lse = torch.logsumexp(s, -1) * LOG2E
    assert (lse - ref_lse).abs().max().item() < 2e-2
    ref.backward(do.float())
    dq, dk, dv = T.attn_train_bwd(qs, k, v, o, lse, do, kb, scale=scale, p_drop=p, seed=seed, site=site, o_f32=o32)
    assert _rel(dv.float(), vf.grad) < 2e-2
    assert _rel(dk.float(), kf.grad) < 2e-2
    assert _rel(dq.float(), qu.grad) < 2e-2
    # deterministic: a second call is bit-identical
    dq2, dk2, dv2 = T.attn_train_bwd(qs, k, v, o, lse, do, kb, scale=scale, p_drop=p, seed=seed, site=site, o_f32=o32)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)
    # without the fp32 output the bf16 O serves (less exact row sums, same tolerance on these peaky random scores)
    dq3, dk3, dv3 = T.attn_train_bwd(qs, k, v, o, lse, do, kb, scale=scale, p_drop=p, seed=seed, site=site)
    assert _rel(dq3.float(), qu.grad) < 3e-2 and _rel(dk3.float(), kf.grad) < 3e-2 and torch.equal(dv3, dv)


def test_attention_train_strided_views_match_inference_kernel_when_p_is_zero():
    # q|k|v thirds of one fused [B*S, 3d] buffer, as the engine lays them out
    B, S, H = 2, 117, 2
    d = H * 64
    qkv = _rand((B * S, 3 * d), 5, dtype=torch.bfloat16)
    v3 = qkv.view(B, S, 3, H, 64)
    q, k, v = v3[:, :, 0], v3[:, :, 1], v3[:, :, 2]
    o, _ = T.attn_train_fwd(q, k, v, None, scale=0.125)
    ref = ops.flash_attention(q, k, v)
    assert _rel(o.float(), ref.float()) < 1.5e-2


@pytest.mark.parametrize("rows,D,p,pp", [(50, 768, 0.0, 0.0), (256, 768, 0.1, 0.2), (33, 128, 0.1, 0.5)])
def test_drop_add_ln_forward_and_backward(rows, D, p, pp):
    seed, site, psite, rps = 5, 40, 41, 16
    t, res, dy = _rand((rows, D), 1), _rand((rows, D), 2), _rand((rows, D), 3)
    gamma, beta = _rand((D,), 4) * 0.1 + 1.0, _rand((D,), 5) * 0.1
    eps = 1e-12
    dr = T.dropout(p, seed, site, pp, psite, rps) if (p or pp) else None
    z, yf, yb = T.drop_add_ln(t, res, gamma, beta, eps, drop=dr)
    keep = _mask(seed, site, (rows, D), p).float() * float(DM.scale(p))
    nsmp = (rows + rps - 1) // rps
    pk = _mask(seed, psite, (nsmp,), pp).float() * float(DM.scale(pp))
    pk = pk.repeat_interleave(rps)[:rows, None]
    tt, rr = t.clone().requires_grad_(True), res.clone().requires_grad_(True)
    gg, bb = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    zr = tt * keep * pk + rr
    yr = torch.nn.functional.layer_norm(zr, (D,), gg, bb, eps)
    assert _rel(z, zr) < 1e-6
    assert _rel(yf, yr) < 1e-5
    assert _rel(yb.float(), yr) < 5e-3
    yr.backward(dy)
    dz, dt, dg, db = T.layernorm_bwd(dy, z, gamma, eps, drop=dr)
    assert _rel(dz, rr.grad) < 1e-4
    assert _rel(dt.float(), tt.grad) < 5e-3
    assert _rel(dg, gg.grad) < 1e-4
    assert _rel(db, bb.grad) < 1e-4


def test_embedding_layernorm_then_dropout_and_row_mapped_gradient():
    # BertEmbeddings: y = dropout(LayerNorm(z)); the backward reads dy through a row map ([:, :nq] of [B, S, D])
    B, S, nq, D, p, seed = 3, 20, 8, 128, 0.1, 11
    z0 = _rand((B * S, D), 1)
    gamma, beta = _rand((D,), 2) * 0.1 + 1.0, _rand((D,), 3) * 0.1
    post = T.dropout(p, seed, DM.EMBED_SITE)
    z, yf, _ = T.drop_add_ln(None, z0, gamma, beta, 1e-12, post=post)
    keep = _mask(seed, DM.EMBED_SITE, (B * S, D), p).float() * float(DM.scale(p))
    zz = z0.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(zz, (D,), gamma, beta, 1e-12) * keep
    assert torch.equal(z, z0) and _rel(yf, yr) < 1e-5
    dy = _rand((B * S, D), 4)
    yr.backward(dy)
    dz, _, _, _ = T.layernorm_bwd(dy, z, gamma, 1e-12, post=post)
    assert _rel(dz, zz.grad) < 1e-4
    # row-mapped dy: only the query rows of a [B, S, D] gradient
    zq = z0.view(B, S, D)[:, :nq].reshape(B * nq, D).contiguous()
    dzq = torch.empty_like(zq)
    dgb = torch.empty((2, D), device=DEV)
    ws = torch.empty(T.lnbwd_ws_floats(D), device=DEV)
    T.layernorm_bwd_raw(dy.data_ptr(), (D, nq, S * D), zq.data_ptr(), gamma.data_ptr(), None, None, dzq.data_ptr(), None,
                        dgb.data_ptr(), ws.data_ptr(), B * nq, D, 1e-12)
    z2 = zq.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(z2, (D,), gamma, beta, 1e-12).backward(dy.view(B, S, D)[:, :nq].reshape(B * nq, D))
    assert _rel(dzq, z2.grad) < 1e-4


@pytest.mark.parametrize("R,C", [(234, 768), (4704, 1536), (1, 64), (70000, 128)])
def test_colsum_and_transpose(R, C):
    x = _rand((R, C), 1, dtype=torch.bfloat16)
    assert _rel(T.colsum(x), x.float().sum(0)) < 1e-5
    xf = _rand((R, C), 2)
    assert _rel(T.colsum(xf), xf.double().sum(0)) < 1e-5
    xt = T.transpose(x)
    assert xt.shape == (C, T.pad64(R))
    assert torch.equal(xt[:, :R], x.t())
    assert not xt[:, R:].any()
    xft = T.transpose(xf)
    assert torch.equal(xft[:, :R], xf.to(torch.bfloat16).t())


def test_wgrad_and_dgrad_through_the_gemm_on_transposed_operands():
    # dW = dY^T X, dX = dY W : the two backward GEMMs of a Linear, contraction over a ragged row count
    R, N, K = 234, 768, 3072
    x = _rand((R, K), 1, dtype=torch.bfloat16)
    dy = _rand((R, N), 2, dtype=torch.bfloat16)
    w = _rand((N, K), 3, 0.03)
    dw = ops.linear(T.transpose(dy), T.transpose(x), None, out_dtype=torch.float32)
    assert dw.shape == (N, K)
    assert _rel(dw, dy.float().t() @ x.float()) < 1e-5
    dx = ops.linear(dy, T.transpose(w)[:, :N], None, out_dtype=torch.float32)
    assert _rel(dx, dy.float() @ w.to(torch.bfloat16).float()) < 1e-5


def test_gelu_forward_backward():
    u = _rand((96, 3072), 1, 2.0, dtype=torch.bfloat16)
    dg = _rand((96, 3072), 2, dtype=torch.bfloat16)
    uf = u.float().requires_grad_(True)
    ref = torch.nn.functional.gelu(uf)
    assert _rel(T.gelu(u).float(), ref) < 5e-3
    ref.backward(dg.float())
    assert _rel(T.gelu_bwd(dg, u).float(), uf.grad) < 5e-3


def test_embedding_gather_and_backward():
    B, nq, Lt, d, V, P = 3, 8, 6, 128, 50, 16
    query, word, pos = _rand((nq, d), 1), _rand((V, d), 2), _rand((P, d), 3)
    ids = torch.tensor([[5, 7, 7, 9, 0, 0], [5, 5, 5, 1, 2, 3], [49, 48, 7, 7, 7, 0]], device=DEV)
    z = T.embed_gather(query, word, pos, ids, B, nq, Lt)
    qq, ww, pp = (t.clone().requires_grad_(True) for t in (query, word, pos))
    ref = torch.cat([qq[None].expand(B, -1, -1), ww[ids] + pp[:Lt][None]], 1)
    assert torch.equal(z, ref)
    dz = _rand((B, nq + Lt, d), 4)
    ref.backward(dz)
    dq, dw, dp = T.embed_bwd(dz, ids, nq, V, P)
    assert _rel(dq, qq.grad) < 1e-6 and _rel(dw, ww.grad) < 1e-6 and _rel(dp, pp.grad) < 1e-6


def test_wgrad_exact_on_integer_data_catches_any_transposition():
    """hk_wgrad_bf16 on small integers (exact in bf16 and in the fp32 accumulator): dW = dY^T X and db = column sums
    must come out EXACT.  dY and X are asymmetric and R is not a multiple of 32, so a swapped output index, a permuted
    contraction order between the two transposed fragment reads, or an unmasked tail row shows as a wrong integer."""
    R, N, K = 75, 256, 384
    g = torch.Generator(device="cpu").manual_seed(5)
    dy = torch.randint(-3, 4, (R, N), generator=g).float()
    x = torch.randint(-4, 5, (R, K), generator=g).float()
    x[:, 0] = torch.arange(R) % 7          # structure along the contraction index
    dy[:, 1] = (torch.arange(R) % 5) - 2
    dw, db = T.wgrad(dy.to(DEV, torch.bfloat16), x.to(DEV, torch.bfloat16))
    assert torch.equal(dw.cpu(), dy.t() @ x)
    assert torch.equal(db.cpu(), dy.sum(0))
    # the rows past R inside the last 32-row step are never read as data: poison the memory behind the operands
    big_y = torch.full((R + 40, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    big_x = torch.full((R + 40, K), float("nan"), dtype=torch.bfloat16, device=DEV)
    big_y[:R], big_x[:R] = dy.to(DEV, torch.bfloat16), x.to(DEV, torch.bfloat16)
    dw2, db2 = T.wgrad(big_y[:R], big_x[:R])
    assert torch.equal(dw2, dw) and torch.equal(db2, db)


@pytest.mark.parametrize("R,N,K", [(256, 768, 768), (192, 3072, 768), (64, 768, 3072), (4704, 1536, 1024), (1, 128, 128), (33, 128, 256)])
def test_wgrad_matches_fp32_autograd_shapes_of_the_stage3_step(R, N, K):
    """The Q-Former's dense layers at B = 2 (R = 256 / 192 / 64 rows) and a slice of the fused cross K/V projection
    (R = B N = 4704): against fp32 matmul on the same bf16-rounded operands; bit-identical on a repeat call."""
    dy, x = _rand((R, N), 21, 0.5, torch.bfloat16), _rand((R, K), 22, 1.0, torch.bfloat16)
    dw, db = T.wgrad(dy, x)
    ref_w, ref_b = dy.float().t() @ x.float(), dy.float().sum(0)
    assert _rel(dw, ref_w) < 2e-5 and _rel(db, ref_b) < 2e-5
    dw2, db2 = T.wgrad(dy, x)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    dw3, none = T.wgrad(dy, x, bias=False)
    assert none is None and torch.equal(dw3, dw)


def test_wgrad_reads_x_through_a_row_map():
    """x = the query rows [:, :96] of a [B, S, d] hidden state (rows_per_batch = 96, batch_stride = S d), as the
    backward addresses them in place (no slice copy)."""
    B, S, nq, d, N = 3, 128, 96, 256, 384
    h = _rand((B, S, d), 31, 1.0, torch.bfloat16)
    dy = _rand((B * nq, N), 32, 0.5, torch.bfloat16)
    dw, db = T.wgrad(dy, h.view(B * S, d), x_rows=(d, nq, S * d), rows=B * nq)
    ref = dy.float().t() @ h[:, :nq].reshape(B * nq, d).float()
    assert _rel(dw, ref) < 2e-5
    # text rows: offset base pointer + the same map
    Lt = S - nq
    dyt = _rand((B * Lt, N), 33, 0.5, torch.bfloat16)
    xt = h.view(-1)[nq * d:]
    dwt, _ = T.wgrad(dyt, xt.view(B * S - nq, d), x_rows=(d, Lt, S * d), rows=B * Lt)
    assert _rel(dwt, dyt.float().t() @ h[:, nq:].reshape(B * Lt, d).float()) < 2e-5


def test_wgrad_rejects_bad_shapes():
    from hawkeye_amd import lib as L
    dy, x = _rand((64, 100), 1, 1.0, torch.bfloat16), _rand((64, 128), 2, 1.0, torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiples of 128"):
        T.wgrad(dy, x)
    assert L.load().hk_wgrad_bf16(None, 0, None, None, None, 0, None, 0, 0, 0, None) == 1


def test_batched_transpose_of_many_weights_in_one_launch():
    """hk_transpose_f32_bf16_batch: the W^T operands of all Linear layers after an optimizer step in one launch (a job
    table on the device).  Ragged sizes, two jobs sharing one output buffer side by side (the fused q|k|v weight)."""
    import ctypes as C
    from hawkeye_amd import lib as L
    shapes = [(130, 70), (64, 64), (768, 200), (1, 5), (257, 129)]
    xs = [_rand(s, 40 + i) for i, s in enumerate(shapes)]
    outs = [torch.full((s[1], s[0] + 3), -7.0, dtype=torch.bfloat16, device=DEV) for s in shapes]
    a, b = _rand((96, 80), 50), _rand((32, 80), 51)           # [a; b] fused along the output dim -> one [80, 128] W^T
    fused = torch.zeros((80, 128), dtype=torch.bfloat16, device=DEV)
    jobs, tile0 = [], 0
    for x, o in zip(xs, outs):
        jobs.append(L.HkTransposeJob(x.data_ptr(), o.data_ptr(), x.stride(0), o.stride(0), x.shape[0], x.shape[1], tile0, 0))
        tile0 += -(-x.shape[0] // 64) * -(-x.shape[1] // 64)
    for x, col in ((a, 0), (b, 96)):
        jobs.append(L.HkTransposeJob(x.data_ptr(), fused.data_ptr() + 2 * col, 80, 128, x.shape[0], 80, tile0, 0))
        tile0 += -(-x.shape[0] // 64) * 2
    raw = (L.HkTransposeJob * len(jobs))(*jobs)
    table = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).to(DEV)
    L.check(L.load().hk_transpose_f32_bf16_batch(table.data_ptr(), len(jobs), tile0, torch.cuda.current_stream().cuda_stream),
            "hk_transpose_f32_bf16_batch")
    for x, o in zip(xs, outs):
        assert torch.equal(o[:, :x.shape[0]], x.t().to(torch.bfloat16))
        assert (o[:, x.shape[0]:] == -7.0).all()                   # nothing written past R
    assert torch.equal(fused, torch.cat([a, b]).t().to(torch.bfloat16))
