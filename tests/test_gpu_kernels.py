"""Per-kernel numerics on a real MI355X, through the C ABI, against plain fp32
PyTorch references evaluated on the same (bf16-rounded) inputs.

Tolerances: the kernels multiply bf16 operands exactly and accumulate in fp32, so a
GEMM/attention result differs from the fp32 reference only by summation order and by
the final bf16 rounding of the output (rel 2^-8).  Stated per test.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from hawkeye_amd import lib as L, ops  # noqa: E402

DEV = "cuda:0"


def _rand(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV).to(dtype)


def _rel(a, ref):
    return ((a.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 1024), (300, 272, 128), (1, 16, 64),
                                   (100352 // 8, 1024, 1024), (65600, 512, 256), (3136 * 2, 3072, 1024),
                                   (66000, 256, 64), (66000, 256, 128), (65537, 512, 192), (16640, 4096, 320)])
def test_gemm_plain_bf16_out(M, N, K):
    x = _rand((M, K), 1, dtype=torch.bfloat16)
    w = _rand((N, K), 2, 0.05, dtype=torch.bfloat16)
    b = _rand((N,), 3)
    out = ops.linear(x, w, b)
    ref = x.float() @ w.float().t() + b
    # bf16 output rounding dominates: rel 2^-8 of each element, measured against the max
    assert _rel(out.float(), ref) < 6e-3
    # asymmetric-operand orientation check: an exact small-integer problem must be exact
    xi = torch.randint(-3, 4, (M, K), device=DEV).to(torch.bfloat16)
    wi = torch.randint(-2, 3, (N, K), device=DEV).to(torch.bfloat16)
    oi = ops.linear(xi, wi, None, out_dtype=torch.float32)
    assert torch.equal(oi, xi.float() @ wi.float().t())


def test_gemm_epilogues_and_row_maps():
    M, N, K = 2 * 96, 768, 768
    B, S, nq = 2, 120, 96
    hb = _rand((B * S, K), 4, dtype=torch.bfloat16)
    w = _rand((N, K), 5, 0.03, dtype=torch.bfloat16)
    bias = _rand((N,), 6)
    res = _rand((B * S, N), 7)
    out = torch.zeros((M, N), dtype=torch.float32, device=DEV)
    d = ops.gemm_desc(hb.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, bias=bias.data_ptr(),
                      residual=res.data_ptr(), a_rows=(K, nq, S * K), c_rows=(N, 0, 0), r_rows=(N, nq, S * N),
                      out_f32=True, scale_cols=256, scale=0.5)
    ops.run_gemm(d)
    a = hb.view(B, S, K)[:, :nq].reshape(M, K).float()
    ref = a @ w.float().t() + bias
    ref[:, :256] *= 0.5
    ref = ref + res.view(B, S, N)[:, :nq].reshape(M, N)
    assert _rel(out, ref) < 2e-5          # fp32 out: summation order only
    # GELU(erf) epilogue, bf16 out
    g = ops.linear(a.to(torch.bfloat16), w, bias, act="gelu")
    gref = torch.nn.functional.gelu(a @ w.float().t() + bias)
    assert _rel(g.float(), gref) < 6e-3
    g32 = ops.linear(a.to(torch.bfloat16), w, bias, act="gelu", out_dtype=torch.float32)
    assert (g32 - gref).abs().max().item() < 2e-5   # exact-erf GELU to fp32 noise
    # broadcast residual (position table): rows_per_batch = N tokens, batch_stride = 0
    ntok = 48
    tab = _rand((ntok, N), 8)
    out2 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    x2 = a.to(torch.bfloat16).contiguous()
    d2 = ops.gemm_desc(x2.data_ptr(), w.data_ptr(), out2.data_ptr(), M, N, K, bias=bias.data_ptr(),
                       residual=tab.data_ptr(), a_rows=(K, 0, 0), c_rows=(N, 0, 0), r_rows=(N, ntok, 0), out_f32=True)
    ops.run_gemm(d2)
    ref2 = x2.float() @ w.float().t() + bias + tab.repeat(M // ntok, 1)
    assert _rel(out2, ref2) < 2e-5


@pytest.mark.parametrize("M,N,K", [(2816, 4096, 512), (2716, 4096, 320), (1400, 8192, 128),
                                   (704, 12288, 4096), (1400, 12288, 2752),   # weights >= 64 MB, few rounds: 192x192 tiles, three stages
                                   (352, 12288, 1024), (700, 3072, 256)])      # skinny: 192x128 tiles, three stages / persistent 192x256
def test_gemm_192_row_tile_shapes(M, N, K):
    """One partial round of 256-row tiles that 192-row tiles cover better (prefill o_proj / down_proj: 176 -> 240
    workgroups) runs the persistent kernel's 192 x 256 instantiation: bf16-out and fp32-out-with-residual epilogues,
    whole and partial last row tiles, an exact small-integer problem (a transposed or shifted tile cannot pass)."""
    x = _rand((M, K), 91, dtype=torch.bfloat16)
    w = _rand((N, K), 92, 0.05, dtype=torch.bfloat16)
    b = _rand((N,), 93)
    res = _rand((M, N), 94)
    ref = x.float() @ w.float().t() + b
    assert _rel(ops.linear(x, w, b).float(), ref) < 6e-3
    assert _rel(ops.linear(x, w, b, residual=res, out_dtype=torch.float32), ref + res) < 2e-5
    xi = torch.randint(-3, 4, (M, K), device=DEV).to(torch.bfloat16)
    wi = torch.randint(-2, 3, (N, K), device=DEV).to(torch.bfloat16)
    assert torch.equal(ops.linear(xi, wi, None, out_dtype=torch.float32), xi.float() @ wi.float().t())
    assert torch.equal(ops.linear(xi, wi, None).float(), (xi.float() @ wi.float().t()).to(torch.bfloat16).float())


@pytest.mark.parametrize("M,N,K,bias", [(1280, 8192, 256, False), (1300, 8192, 128, True), (100, 512, 64, True),
                                        (2816, 22016, 128, False), (257, 1056, 192, False), (66000, 512, 64, True),
                                        (352, 22016, 128, False), (330, 24576, 64, True),     # skinny prompts: 192x192 tiles, three stages
                                        (704, 22016, 64, False)])                             # ... and the persistent 192x256 tile
def test_gemm_swiglu_epilogue_on_interleaved_gate_up_rows(M, N, K, bias):
    """HK_ACT_SWIGLU (modeling_llama.py:143-155): W rows interleaved (gate_0, up_0, ...), C [M, N/2] = silu(gate) * up, on the
    persistent kernel (whole and partial tiles, with and without its tail launch) and on the multi-stage kernel; vs fp32."""
    x = _rand((M, K), 71, dtype=torch.bfloat16)
    I = N // 2
    gate = _rand((I, K), 72, 0.08, dtype=torch.bfloat16)
    up = _rand((I, K), 73, 0.08, dtype=torch.bfloat16)
    b = _rand((N,), 74) if bias else None
    w = torch.stack([gate, up], 1).reshape(N, K).contiguous()
    out = ops.linear(x, w, b, act="swiglu")
    assert out.shape == (M, I)
    g = x.float() @ gate.float().t()
    u = x.float() @ up.float().t()
    if bias:
        g, u = g + b[0::2], u + b[1::2]
    ref = torch.nn.functional.silu(g) * u
    assert _rel(out.float(), ref) < 6e-3          # bf16 output rounding
    # a strided output (the caller's [M, I] view of a wider buffer)
    wide = torch.zeros((M, I + 64), dtype=torch.bfloat16, device=DEV)
    ops.linear(x, w, b, act="swiglu", out=wide[:, :I])
    assert torch.equal(wide[:, :I], out) and not wide[:, I:].any()


@pytest.mark.parametrize("M,N,K", [(66000, 1024, 256), (300, 768, 128), (5000, 1024, 128)])   # the last: 160 tiles of 256x128
def test_gemm_bf16_residual_in_place(M, N, K):
    """The bf16 residual stream: C = bf16(x @ w.T + b + float(C)), rounded once, C updated in place."""
    x = _rand((M, K), 81, dtype=torch.bfloat16)
    w = _rand((N, K), 82, 0.05, dtype=torch.bfloat16)
    b = _rand((N,), 83)
    res = _rand((M, N), 84, 2.0, dtype=torch.bfloat16)
    ref = (x.float() @ w.float().t() + b + res.float()).to(torch.bfloat16)
    out = res.clone()
    ops.linear(x, w, b, residual=out, out=out)
    # single rounding of an fp32 sum that differs from the reference only by summation order
    diff = (out.float() - ref.float()).abs()
    assert (diff <= 2 ** -7 * ref.float().abs() + 1e-6).all()
    assert (out != ref).float().mean().item() < 0.02
    # fp32 out with bf16 residual
    o32 = ops.linear(x, w, b, residual=res, out_dtype=torch.float32)
    assert _rel(o32, x.float() @ w.float().t() + b + res.float()) < 2e-5


@pytest.mark.parametrize("M,N,K", [(256 * 552 + 37, 256, 128), (100352, 1024, 256), (256 * 140, 1024, 64)])
def test_gemm_split_tail_rows_match_a_plain_launch(M, N, K):
    """Tile counts with a small last round (552 = 2*256 + 40; 1568 = 6*256 + 32; 560 = 2*256 + 48) take the
    two-launch path (persistent kernel over whole rounds + multi-stage kernel over the tail rows): every
    output row is written exactly once, with the same epilogue, including row maps and the in-place residual."""
    x = _rand((M, K), 31, dtype=torch.bfloat16)
    w = _rand((N, K), 32, 0.05, dtype=torch.bfloat16)
    b = _rand((N,), 33)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.linear(x, w, b, act="gelu", out=out)
    ref = torch.nn.functional.gelu(x.float() @ w.float().t() + b)
    assert not torch.isnan(out).any()
    assert _rel(out.float(), ref) < 6e-3
    del ref
    # in-place bf16 residual stream + broadcast-free row map (rows_per_batch) across the split point
    res = _rand((M, N), 34, dtype=torch.bfloat16)
    want = (x.float() @ w.float().t() + b + res.float())
    ops.linear(x, w, b, residual=res, out=res)
    assert _rel(res.float(), want) < 6e-3
    rpb = 1000
    nb = M // rpb
    Mm = nb * rpb
    big = torch.zeros((nb, rpb + 8, N), dtype=torch.float32, device=DEV)
    d = ops.gemm_desc(x.data_ptr(), w.data_ptr(), big.data_ptr(), Mm, N, K, bias=b.data_ptr(),
                      a_rows=(K, 0, 0), c_rows=(N, rpb, (rpb + 8) * N), out_f32=True)
    ops.run_gemm(d)
    ref3 = (x[:Mm].float() @ w.float().t() + b).view(nb, rpb, N)
    assert _rel(big[:, :rpb], ref3) < 2e-5
    assert torch.count_nonzero(big[:, rpb:]) == 0


@pytest.mark.parametrize("M,N,K,act", [(512, 768, 1024, "none"), (66000, 1024, 1024, "gelu"), (300, 272, 128, "gelu"),
                                         (100352 // 4 + 7, 3072, 1024, "none"),
                                         # one round of 256x128 tiles (B = 1 calls): qkv at T = 12 (240 tiles), fc1 at T = 8 (224)
                                         (2352, 3072, 1024, "none"), (1568 - 5, 4096, 1024, "gelu"),
                                         # the persistent kernel's 192 x 256 tile (fc1 at B = 1, T = 12: 208 tiles; ragged last row tile)
                                         (2353, 4096, 1024, "gelu"), (2300, 4096, 256, "none")])
def test_gemm_with_folded_layernorm(M, N, K, act):
    """hk_row_stats + row_stats/col_c epilogue == LayerNorm -> Linear (-> GELU) in fp32 (vit.py:125-127), on a residual
    stream with a per-row offset and outlier channels (the mean term must cancel, not merely be small)."""
    x = _rand((M, K), 21)
    x = x * (1.0 + 4.0 * (torch.arange(K, device=DEV) % 97 == 0)) + _rand((M, 1), 22, 3.0)
    x = x.to(torch.bfloat16)
    w = _rand((N, K), 23, 0.05)
    b = _rand((N,), 24)
    g = 1.0 + 0.3 * _rand((K,), 25)
    bt = 0.2 * _rand((K,), 26)
    eps = 1e-6
    st = ops.row_stats(x, eps)
    xf = x.float()
    mean = xf.mean(1)
    rstd = (xf.var(1, unbiased=False) + eps).rsqrt()
    assert torch.allclose(st[:, 0], mean, rtol=1e-5, atol=1e-6) and torch.allclose(st[:, 1], rstd, rtol=1e-5, atol=1e-7)
    wf, cc, bf = ops.fold_layernorm(w, b, g, bt)
    out = ops.linear(x, wf, bf, act=act, row_stats=st, col_c=cc, scale_cols=256 if act == "none" else 0, scale=0.25)
    ref = torch.nn.functional.layer_norm(xf, (K,), g, bt, eps) @ w.t() + b
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    else:
        ref[:, :256] *= 0.25
    # same error budget as LayerNorm (bf16 out) -> bf16 GEMM -> bf16 out
    assert _rel(out.float(), ref) < 8e-3
    # and it agrees with the unfolded pipeline of this library to the same budget
    xn, _ = ops.layernorm(x, g, bt, eps)
    plain = ops.linear(xn, w.to(torch.bfloat16), b, act=act, scale_cols=256 if act == "none" else 0, scale=0.25)
    assert _rel(out.float(), plain.float()) < 8e-3


@pytest.mark.parametrize("M,N,K", [(512, 1024, 256), (100352 // 2 + 3, 1024, 1024), (300, 256, 128), (66000, 1024, 128),
                                   (5000 - 9, 1024, 256),    # one round of 256x128 tiles (160)
                                   (2048, 1024, 192), (1990, 1024, 64), (100352, 1024, 64)])   # 256 tiles of 128x64 (the headline shape's tail rows)
def test_gemm_row_part_feeds_the_next_layernorm(M, N, K):
    """row_part of the in-place residual epilogue (persistent kernel, its split tail and the small kernel) ->
    hk_row_stats_finalize == the statistics hk_row_stats takes from the stored bf16 rows: the sums are taken from
    the values as stored, so the mean the consuming GEMM subtracts is the mean of the matrix it multiplies."""
    x = _rand((M, K), 31, dtype=torch.bfloat16)
    w = _rand((N, K), 32, 0.05, dtype=torch.bfloat16)
    b = _rand((N,), 33)
    res = (_rand((M, N), 34) * 2.0 + _rand((M, 1), 35, 1.5)).to(torch.bfloat16)
    ref = x.float() @ w.float().t() + b + res.float()
    part = torch.full((M, N // 64, 2), float("nan"), device=DEV)
    out = res.clone()
    ops.linear(x, w, b, residual=out, out_dtype=torch.bfloat16, out=out, row_part=part)
    plain = res.clone()
    ops.linear(x, w, b, residual=plain, out_dtype=torch.bfloat16, out=plain)
    assert torch.equal(out, plain)                      # the output itself is untouched by the extra sums
    assert (out.float() - ref).abs().max() <= 2 ** -8 * ref.abs().max()
    g = out.float().view(M, N // 64, 64)                # the stored values, not the fp32 ones
    assert torch.allclose(part[:, :, 0], g.sum(2), rtol=1e-5, atol=1e-4)
    assert torch.allclose(part[:, :, 1], (g * g).sum(2), rtol=1e-5, atol=1e-4)
    st = ops.row_stats_finalize(part, 1e-6)
    st_ref = ops.row_stats(out, 1e-6)
    assert torch.allclose(st[:, 0], st_ref[:, 0], atol=2e-6 * float(ref.abs().max()))
    assert torch.allclose(st[:, 1], st_ref[:, 1], rtol=2e-5)


@pytest.mark.parametrize("M,N,K,S", [(96, 768, 3072, 6), (32, 768, 3072, 8), (200, 768, 3072, 4), (96, 768, 768, 3),
                                     (2352, 1024, 4096, 2), (130, 144, 320, 5),
                                     # slices that make one round of 256x128 tiles: fc2 at T = 12 (240) and T = 8 (224)
                                     (2352, 1024, 4096, 3), (1568 - 3, 1024, 4096, 4)])
def test_gemm_split_k_partials_and_layernorm_of_their_sum(M, N, K, S):
    """hk_gemm_desc.split_k: slice z multiplies its own K range; slice 0 carries bias + residual; the slices sum to the
    unsplit GEMM (fp32 reassociation only), and hk_layernorm_parts normalises the sum without a reduction pass."""
    x = _rand((M, K), 41, dtype=torch.bfloat16)
    w = _rand((N, K), 42, 0.05, dtype=torch.bfloat16)
    b = _rand((N,), 43)
    res = _rand((M, N), 44)
    parts = ops.linear_split_k(x, w, b, res, S)
    assert parts.shape == (S, M, N)
    ref = x.float() @ w.float().t() + b + res
    kper = -(-(K // 64) // S) * 64
    for z in range(S):                       # every slice is exactly its K range
        pz = x[:, z * kper:(z + 1) * kper].float() @ w[:, z * kper:(z + 1) * kper].float().t()
        if z == 0:
            pz = pz + b + res
        assert _rel(parts[z], pz) < 2e-5, z
    whole = ops.linear(x, w, b, residual=res, out_dtype=torch.float32)
    assert _rel(parts.sum(0), whole) < 1e-5 and _rel(whole, ref) < 1e-4
    if N % 4 == 0 and N <= 4096:
        g, bt = _rand((N,), 45, 0.3) + 1.0, _rand((N,), 46, 0.3)
        ob, of = ops.layernorm_parts(parts, g, bt, 1e-12)
        want = torch.nn.functional.layer_norm(parts.sum(0), (N,), g, bt, 1e-12)
        assert _rel(of, want) < 1e-5 and _rel(ob.float(), want) < 1e-2
        ob2, of2 = ops.layernorm_parts(parts, g, bt, 1e-12)
        assert torch.equal(of, of2)              # fixed summation order: bit-reproducible


@pytest.mark.parametrize("rows,D,S", [(2352, 1024, 3), (197, 1024, 2), (64, 256, 4), (5, 1536, 1)])
def test_residual_add_parts_updates_the_stream_and_leaves_row_statistics(rows, D, S):
    """Tail of a split-K fc2 on a small call: x += sum of the slices, rounded once; (mean, rstd) of the rows as stored."""
    x = (_rand((rows, D), 51) * 2 + _rand((rows, 1), 52, 1.5)).to(torch.bfloat16)
    parts = _rand((S, rows, D), 53)
    want = (x.float() + parts.sum(0)).to(torch.bfloat16)
    got = x.clone()
    st = ops.residual_add_parts(got, parts, 1e-6)
    # one rounding of the fp32 sum; the slice order is fixed, torch's sum may associate differently: allow 1 bf16 ulp
    assert (got.float() - want.float()).abs().max() <= 2 ** -7 * want.float().abs().max()
    assert (got != want).float().mean() < 1e-3
    ref = ops.row_stats(got, 1e-6)
    assert torch.equal(st, ref)
    again = x.clone()
    assert ops.residual_add_parts(again, parts, 1e-6, want_stats=False) is None and torch.equal(again, got)


@pytest.mark.parametrize("B,H,Nq,Nk,S", [(1, 16, 2352, 2352, 2), (1, 12, 96, 2352, 8), (2, 3, 130, 1000, 3), (1, 2, 64, 300, 4),
                                        (1, 1, 32, 64, 4)])
def test_attention_key_split_matches_the_unsplit_launch(B, H, Nq, Nk, S):
    """hk_attn_desc.key_splits: slices of the keys on separate workgroups + the combine launch = the same attention
    (partial outputs are bf16: one extra rounding), incl. a ragged last slice and a split that degenerates to one slice."""
    q = _rand((B, Nq, H, 64), 61, 1.3, dtype=torch.bfloat16)
    k = _rand((B, Nk, H, 64), 62, dtype=torch.bfloat16)
    v = _rand((B, Nk, H, 64), 63, dtype=torch.bfloat16)
    k[:, Nk // 3] *= 6.0                                  # a key that dominates: the slices' maxima differ widely
    ref = _attn_ref(q, k, v)
    whole = ops.flash_attention(q, k, v)
    split = ops.flash_attention(q, k, v, key_splits=S)
    assert _rel(whole.float(), ref) < 2e-2 and _rel(split.float(), ref) < 2e-2
    assert _rel(split.float(), whole.float()) < 1.5e-2
    assert torch.equal(split, ops.flash_attention(q, k, v, key_splits=S))     # fixed combine order


def test_gemm_rejects_bad_shapes():
    x = _rand((8, 48), 1, dtype=torch.bfloat16)
    w = _rand((16, 48), 2, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.linear(x, w)
    # SwiGLU epilogue: bf16 [M, N / 2] output only, no residual, N % 32 == 0 (host wrapper and C ABI both refuse)
    x2, w2 = _rand((8, 64), 1, dtype=torch.bfloat16), _rand((48, 64), 2, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="swiglu"):
        ops.linear(x2, w2, act="swiglu")
    out = torch.empty((8, 32), dtype=torch.float32, device=DEV)
    w3 = _rand((64, 64), 2, dtype=torch.bfloat16)
    d = ops.gemm_desc(x2.data_ptr(), w3.data_ptr(), out.data_ptr(), 8, 64, 64, a_rows=(64, 0, 0), c_rows=(32, 0, 0),
                      act=L.HK_ACT_SWIGLU, out_f32=True)
    with pytest.raises(RuntimeError, match="HK_ACT_SWIGLU"):
        ops.run_gemm(d)


def _attn_ref(q, k, v, key_bias=None):
    # q is pre-scaled by scale*log2(e): softmax over exp2 == exp of q/log2e
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) / math.log2(math.e)
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    p = s.softmax(-1)
    return torch.einsum("bhqk,bkhd->bqhd", p, v.float())


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 4, 128, 64), (1, 2, 200, 200), (2, 3, 96, 500), (1, 16, 1568, 1568),
                                       (3, 12, 108, 108), (1, 1, 1, 1), (2, 2, 33, 65)])
def test_flash_attention_matches_fp32_softmax(B, H, Nq, Nk):
    q = _rand((B, Nq, H, 64), 11, 0.6, torch.bfloat16)
    k = _rand((B, Nk, H, 64), 12, 1.0, torch.bfloat16)
    v = _rand((B, Nk, H, 64), 13, 1.0, torch.bfloat16)
    o = ops.flash_attention(q, k, v)
    ref = _attn_ref(q, k, v)
    # P is rounded to bf16 before P.V and O to bf16 on store: abs error <= ~2^-8 * max|v|
    assert (o.float() - ref).abs().max().item() < 2.5e-2
    assert _rel(o.float(), ref) < 1.5e-2


def test_flash_attention_strided_views_and_key_bias():
    B, S, H = 3, 108, 12
    d = H * 64
    qkv = _rand((B, S, 3 * d), 21, 0.5, torch.bfloat16)
    q = qkv[:, :, :d].unflatten(2, (H, 64))
    k = qkv[:, :, d:2 * d].unflatten(2, (H, 64))
    v = qkv[:, :, 2 * d:].unflatten(2, (H, 64))
    mask = torch.ones(B, S, device=DEV)
    mask[1, 100:] = 0
    mask[2, 97:] = 0
    kb = ((1.0 - mask) * -10000.0).contiguous()
    o = ops.flash_attention(q, k, v, key_bias=kb)
    ref = _attn_ref(q, k, v, kb)
    assert (o.float() - ref).abs().max().item() < 2.5e-2


def test_flash_attention_rescale_branch_with_spiked_keys():
    """Force the running max to jump at a late tile (guide rule 26): one key row aligned with the queries."""
    B, H, Nq, Nk = 1, 2, 64, 640
    q = _rand((B, Nq, H, 64), 31, 0.3, torch.bfloat16)
    k = _rand((B, Nk, H, 64), 32, 0.3, torch.bfloat16)
    v = _rand((B, Nk, H, 64), 33, 1.0, torch.bfloat16)
    k[:, 70] = q[:, 5] * 30.0       # tile 1
    k[:, 333] = q[:, 5] * 90.0      # tile 5: max jumps again
    k[:, 639] = q[:, 9] * 120.0     # last key of the last tile
    o = ops.flash_attention(q, k, v)
    ref = _attn_ref(q, k, v)
    assert (o.float() - ref).abs().max().item() < 3e-2


def test_flash_attention_all_scores_far_below_zero():
    """Every score of a row near -300 (log2 units): exp2 of the raw scores underflows, so the no-maximum first pass of the
    default kernel must notice the vanishing row totals and hand over to the running-maximum pass."""
    B, H, Nq, Nk = 1, 2, 160, 200
    u = _rand((1, 1, H, 64), 41, 0.5, torch.bfloat16)
    q = (u + _rand((B, Nq, H, 64), 42, 0.02, torch.bfloat16)).to(torch.bfloat16)
    k = (-20.0 * u + _rand((B, Nk, H, 64), 43, 0.5, torch.bfloat16)).to(torch.bfloat16)
    v = _rand((B, Nk, H, 64), 44, 1.0, torch.bfloat16)
    o = ops.flash_attention(q, k, v)
    ref = _attn_ref(q, k, v)
    assert torch.isfinite(o.float()).all()
    assert (o.float() - ref).abs().max().item() < 3e-2


@pytest.mark.parametrize("env", [{"HK_ATTN_PIPE": "2"}, {"HK_ATTN_PIPE": "8"}, {"HK_ATTN_MODE": "2"}, {"HK_ATTN_LOADER": "2"}])
def test_flash_attention_measured_variants_stay_correct(env):
    """The attention variants kept for A/B (software-pipelined kernel with 4 / 8 waves, no-maximum first pass, resident
    -m block) are selected by environment variables the library reads once per process: run this file's attention
    tests in a child process per variant."""
    import os
    import subprocess
    import sys
    if os.environ.get("HK_ATTN_VARIANT_CHILD"):
        pytest.skip("already inside a variant run")
    child_env = dict(os.environ, HK_ATTN_VARIANT_CHILD="1", **env)
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-x", "-q", "-k", "flash_attention and not variants",
                        "--timeout", "120", "--timeout-method", "thread", "-p", "no:cacheprovider"],
                       env=child_env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("D,eps,in_bf16", [(1024, 1e-6, False), (768, 1e-12, False), (128, 1e-12, True), (256, 1e-6, False)])
def test_layernorm(D, eps, in_bf16):
    rows = 1003
    x = _rand((rows, D), 41, 3.0) + 0.7
    if in_bf16:
        x = x.to(torch.bfloat16)
    add = _rand((rows, D), 42)
    g, b = _rand((D,), 43) * 0.1 + 1.0, _rand((D,), 44) * 0.1
    ob, of = ops.layernorm(x, g, b, eps, add=add, want_f32=True)
    ref = torch.nn.functional.layer_norm(x.float() + add, (D,), g, b, eps)
    assert (of - ref).abs().max().item() < 2e-5
    assert _rel(ob.float(), ref) < 5e-3
    ob2, _ = ops.layernorm(x, g, b, eps)
    ref2 = torch.nn.functional.layer_norm(x.float(), (D,), g, b, eps)
    assert _rel(ob2.float(), ref2) < 5e-3


@pytest.mark.parametrize("D", [1024, 256, 768])
def test_layernorm_bf16_fast_path(D):
    rows = 4099
    x = (_rand((rows, D), 46, 3.0) + 0.5).to(torch.bfloat16)
    g, b = _rand((D,), 47) * 0.1 + 1.0, _rand((D,), 48) * 0.1
    ob, _ = ops.layernorm(x, g, b, 1e-6)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), g, b, 1e-6)
    assert _rel(ob.float(), ref) < 5e-3
    assert (ob.float() - ref).abs().max().item() < 0.05


def test_layernorm_out_row_map():
    B, S, nq, D = 3, 20, 8, 128
    x = _rand((B * nq, D), 45)
    g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    of = torch.zeros((B, S, D), device=DEV)
    obf = torch.zeros((B, S, D), device=DEV, dtype=torch.bfloat16)
    ops.layernorm_raw(x.data_ptr(), True, None, g.data_ptr(), b.data_ptr(), obf.data_ptr(), of.data_ptr(), B * nq, D,
                      1e-12, out_rows=(D, nq, S * D))
    ref = torch.nn.functional.layer_norm(x, (D,), g, b, 1e-12).view(B, nq, D)
    assert (of[:, :nq] - ref).abs().max().item() < 2e-5
    assert of[:, nq:].abs().max().item() == 0.0


def test_patchify_is_an_exact_gather():
    B, T, H, W, P = 2, 3, 64, 48, 16
    clip = _rand((B, T, 3, H, W), 51)
    out = ops.patchify(clip, P)
    ref = clip.view(B, T, 3, H // P, P, W // P, P).permute(0, 1, 3, 5, 2, 4, 6).reshape(-1, 3 * P * P)
    assert torch.equal(out, ref.to(torch.bfloat16))


def test_patchify_u8_is_the_reference_transform_bit_for_bit(golden_dir):
    """Row f3: uint8 frames -> ((x/255) - mean)/std -> bf16 patches must equal the fp32 patch gather of the clip the
    reference's transform classes produce (tests/golden/frame_transform.npz), and of a larger random clip
    normalised with the same separately rounded fp32 steps."""
    import os
    import numpy as np
    g = np.load(os.path.join(golden_dir, "frame_transform.npz"))
    frames = torch.from_numpy(g["frames"]).unsqueeze(0).to(DEV)            # [1,T,H,W,3]
    video = torch.from_numpy(g["video"]).to(DEV)                           # [1,T,3,H,W] by the reference classes
    got = ops.patchify_u8(frames, 16, g["mean"].tolist(), g["std"].tolist())
    assert torch.equal(got, ops.patchify(video.contiguous(), 16))
    gen = torch.Generator(device="cpu").manual_seed(5)
    fr = torch.randint(0, 256, (2, 3, 224, 224, 3), generator=gen, dtype=torch.uint8).to(DEV)
    x = fr.permute(0, 1, 4, 2, 3).contiguous().float().div(255)
    for c in range(3):
        x[:, :, c].sub_(ops.CLIP_MEAN[c]).div_(ops.CLIP_STD[c])
    assert torch.equal(ops.patchify_u8(fr, 16), ops.patchify(x, 16))


@pytest.mark.parametrize("T", [1, 4, 8, 12, 16])
def test_pos_embed_on_device_matches_reference_table(golden_dir, T):
    """Row f4: the device-generated table vs rows of the reference's get_sinusoid_encoding_table
    (tests/golden/pos_embed.npz) and vs the host restatement.  The angle is float64 on both sides; libm and the
    device math library may differ in the last float64 bit, which after the cast shows as an occasional 1-ulp
    fp32 difference: tolerance 1.2e-7 absolute (values are in [-1, 1]), and < 0.1 % of elements may differ at all."""
    import os
    import numpy as np
    from hawkeye_amd import posembed
    g = np.load(os.path.join(golden_dir, "pos_embed.npz"))
    tab = ops.pos_embed(T, 196, 1024, -1 if T == 1 else 4, DEV)
    rows = torch.from_numpy(g[f"T{T}.rows"]).to(DEV)
    ref_rows = torch.from_numpy(g[f"T{T}.vals"]).to(DEV)
    assert (tab[rows] - ref_rows).abs().max().item() <= 1.2e-7
    host = (posembed.image_pos_embed(196, 1024) if T == 1 else posembed.video_pos_embed(T, 196, 1024, 4)).reshape(-1, 1024)
    diff = (tab.cpu() - host)
    assert diff.abs().max().item() <= 1.2e-7
    assert (diff != 0).float().mean().item() < 1e-3
    assert abs(tab.double().norm().item() - float(g[f"T{T}.l2"])) < 1e-3


def test_qformer_embed_ln():
    B, nq, Lt, d, V = 3, 16, 7, 128, 1000
    qt, we, pe = _rand((nq, d), 61, 0.02), _rand((V, d), 62, 0.02), _rand((64, d), 63, 0.02)
    ids = torch.randint(0, V, (B, Lt), device=DEV)
    g, b = _rand((d,), 64) * 0.1 + 1.0, _rand((d,), 65) * 0.1
    of = torch.empty((B, nq + Lt, d), device=DEV)
    ob = torch.empty((B, nq + Lt, d), device=DEV, dtype=torch.bfloat16)
    ops.qformer_embed_ln(qt, we, pe, ids, g, b, 1e-12, of, ob)
    emb = torch.cat([qt.expand(B, -1, -1), we[ids] + pe[:Lt]], dim=1)
    ref = torch.nn.functional.layer_norm(emb, (d,), g, b, 1e-12)
    assert (of - ref).abs().max().item() < 2e-5
    assert _rel(ob.float(), ref) < 5e-3


def test_cast_is_round_to_nearest_even():
    x = _rand((1000003,), 71, 10.0)
    assert torch.equal(ops.cast_bf16(x), x.to(torch.bfloat16))


@pytest.mark.parametrize("kind", [0, 3, 4, 1])
def test_gemm_two_workgroups_per_cu_kernel_gives_the_256x256_kernels_bits(kind):
    """hk_gemm_duo_kernel (256x128 tiles, two persistent workgroups per CU, K-step 32, 3-stage ring; round 3, measured
    slower and off by default -- profiles/r03_duo_pmc.txt) accumulates every output element over the same K order with
    the same epilogue code as the 256x256 kernel: bit-identical results for each epilogue kind, including a row count
    that leaves a partial 256-row tile to the tail launch, and a multi-tile-per-workgroup walk (M = 70 000: 274 x 8 tiles)."""
    import os
    import sys
    from hawkeye_amd import lib as L
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import duo_check as DC
    lib = L.load()
    assert lib.hk_gemm_duo_mask(-1) == 0                     # the default path is the 256x256 kernel
    try:
        for (M, N, K) in ((70000, 1024, 1024 if kind != 0 else 768), (25088, 3072 if kind in (0, 3) else 1024, 256)):
            args = DC.make(M, N, K, kind)
            outs = []
            for mask in (0, 31):
                lib.hk_gemm_duo_mask(mask)
                assert lib.hk_gemm_duo_mask(-1) == mask
                out = args[-1].clone()
                rp = args[6].clone() if args[6] is not None else None
                DC.run(kind, *args[:6], rp, out)
                torch.cuda.synchronize()
                outs.append((out, rp))
            assert torch.equal(outs[0][0], outs[1][0]), (kind, M, N, K)
            if outs[0][1] is not None:
                whole = M // 256 * 256     # the rows of a last partial tile take the multi-stage kernel's summation order
                assert torch.equal(outs[0][1][:whole], outs[1][1][:whole])
                assert torch.allclose(outs[0][1][whole:], outs[1][1][whole:], rtol=1e-5, atol=1e-4)
            lib.hk_gemm_duo_mask(0)
    finally:
        lib.hk_gemm_duo_mask(0)


# hk_gemm_desc.w_tiled: the same GEMM with W in 64 x 64 blocks -- every kernel family the LLM's shapes reach (persistent 256- and
# 192-row tiles, multi-stage 192x192 / 192x128 / 256x128 / 128x128 three- and four-stage, split launches with a tail)
@pytest.mark.parametrize("M,N,K,act,f32", [(2816, 12288, 4096, "none", False), (2816, 4096, 4096, "none", True), (2816, 22016, 4096, "swiglu", False),
                                           (352, 12288, 4096, "none", False), (352, 22016, 4096, "swiglu", False), (352, 4096, 11008, "none", True),
                                           (704, 4096, 4096, "none", True), (1408, 4096, 11008, "none", True), (96, 768, 256, "none", False),
                                           (5000, 1024, 512, "none", False)])
def test_gemm_tiled_weight_layout_gives_the_row_major_bits(M, N, K, act, f32):
    x, w = _rand((M, K), 31, 1.0, torch.bfloat16), _rand((N, K), 32, 0.03, torch.bfloat16)
    wt = ops.tile_gemm_weight(w)
    kw = dict(act=act, out_dtype=torch.float32 if f32 else torch.bfloat16)
    res = _rand((M, N), 33) if f32 else None
    a = ops.linear(x, w, residual=res, **kw)
    b = ops.linear(x, wt, residual=res, **kw)
    assert torch.equal(a, b)
