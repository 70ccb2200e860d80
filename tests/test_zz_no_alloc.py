"""hk_colsum_static -- the entry point that allocates nothing (include/hawkeye_hip.h): its partial sums live in a
static device array of the library.  Checked through the C ABI against hk_colsum (caller workspace), bit for bit,
on two calls in a row, the second with a smaller input that leaves stale partials of the first call in the array."""
import ctypes as C

import pytest
import torch

from hawkeye_amd import lib as L

pytestmark = pytest.mark.gpu


def _both(x: torch.Tensor):
    lib = L.load()
    R, Cc = x.shape
    dt = L.HK_DT_BF16 if x.dtype == torch.bfloat16 else L.HK_DT_F32
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.full((L.HK_COLSUM_MAX_SPLIT * Cc,), float("nan"), dtype=torch.float32, device=x.device)
    ref = torch.empty(Cc, dtype=torch.float32, device=x.device)
    out = torch.empty(Cc, dtype=torch.float32, device=x.device)
    L.check(lib.hk_colsum(x.data_ptr(), dt, x.stride(0), R, Cc, ref.data_ptr(), ws.data_ptr(), stream), "hk_colsum")
    torch.cuda.synchronize()
    # torch's counter sees only torch's caching allocator; the driver-level free memory also sees a hipMalloc / hipFree
    # made by the library itself.  The first call on a device does the one-time symbol lookup, so the driver-level
    # comparison is made from the second call on.
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    L.check(lib.hk_colsum_static(x.data_ptr(), dt, x.stride(0), R, Cc, out.data_ptr(), stream), "hk_colsum_static")
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    if _both.calls > 0:
        assert torch.cuda.mem_get_info()[0] == free_before, "device memory changed across an hk_colsum_static call"
    _both.calls += 1
    return out, ref


_both.calls = 0


def test_static_scratch_colsum_matches_workspace_colsum_twice():
    g = torch.Generator(device="cpu").manual_seed(11)
    # call 1: the largest width the static array takes (the fused cross-K/V bias gradient), 64-way split
    x1 = torch.randn(20000, L.HK_COLSUM_STATIC_MAX_C, generator=g).to("cuda", torch.bfloat16)
    o1, r1 = _both(x1)
    assert torch.equal(o1, r1)
    want = x1.double().sum(0)
    assert float((o1.double() - want).abs().max() / want.abs().max()) < 1e-5
    # call 2: smaller in both dimensions, fp32, a strided view, fewer splits -- the array still holds call 1's partials
    big = torch.randn(3000, 1024, generator=g).cuda()
    x2 = big[:2999, :777]
    o2, r2 = _both(x2)
    assert torch.equal(o2, r2)
    want = x2.double().sum(0)
    assert float((o2.double() - want).abs().max() / want.abs().max()) < 1e-5
    # call 3: too short for a split at all (single launch, the array is not touched)
    x3 = torch.randn(100, 768, generator=g).to("cuda", torch.bfloat16)
    o3, r3 = _both(x3)
    assert torch.equal(o3, r3)


def test_static_scratch_colsum_rejects_oversize():
    lib = L.load()
    x = torch.zeros(512, L.HK_COLSUM_STATIC_MAX_C + 64, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(x.shape[1], dtype=torch.float32, device="cuda")
    rc = lib.hk_colsum_static(x.data_ptr(), L.HK_DT_BF16, x.stride(0), x.shape[0], x.shape[1], out.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    assert rc == 1 and "exceeds the static scratch" in L.last_error()
    assert lib.hk_colsum_static(None, 0, 0, 0, 0, None, None) == 1
