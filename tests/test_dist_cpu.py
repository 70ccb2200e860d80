"""world_size-2 gloo run of the clip-sharding helpers (hawkeye_amd/dist.py): block partition,
output all_gather (equal and ragged shards), max-over-ranks timing.  The HIP path itself needs a
GPU; here each rank's "encode_img" is a stand-in that tags rows with the global clip index."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from hawkeye_amd import dist as hd
    r, lr, w = hd.init("gloo")
    assert (r, w) == (rank, world)
    lo, hi = hd.shard_range(total, rank, world)
    counts = [hd.shard_range(total, i, world)[1] - hd.shard_range(total, i, world)[0] for i in range(world)]
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 3, 4).contiguous()
    full = hd.gather_outputs(local, counts)
    ok = full.shape == (total, 3, 4) and torch.equal(full[:, 0, 0], torch.arange(total, dtype=torch.float32))
    if total % world == 0:
        full2 = hd.gather_outputs(local)
        ok = ok and torch.equal(full2, full)
    if total % world == 0:
        # the overlapped form bench.py uses: submit() hands back the previous call's batch, collect() the last one
        g = hd.OutputGatherer()
        first = g.submit(local)
        second = g.submit(local + 100.0)
        last = g.collect()
        ok = ok and first is None and torch.equal(second, full) and torch.equal(last, full + 100.0) and g.collect() is None
    t = hd.max_over_ranks(1.0 + rank, "cpu")
    seen = hd.ranks_seen("cpu")          # what bench.py reports as `ranks_seen`
    ok = ok and [(e["rank"], e["local_rank"]) for e in seen] == [(i, i) for i in range(world)]
    hd.barrier()
    q.put((rank, bool(ok), t))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_two_rank_gloo_sharding_and_gather(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(abs(t - 2.0) < 1e-9 for _, _, t in res)


def test_shard_range_covers_everything():
    from hawkeye_amd.dist import shard_range
    for total in (0, 1, 7, 32, 3720):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from hawkeye_amd import dist as hd
    hd.init("gloo")
    n = 1000
    base = torch.arange(n, dtype=torch.float32)
    grad = base * (rank + 1)                      # rank r holds (r+1) * i  ->  average = 1.5 * i
    red = hd.BucketReducer(grad, average=True, bucket_bytes=4 * 128)      # 128-element buckets: 8 buckets (last short)
    ok = len(red.bounds) == 8 and red.bounds[-1] == (896, 1000)
    red.ready(100)                                # nothing whole yet
    ok = ok and red.log == []
    red.ready(300)                                # buckets 0,1 -> one range [0, 256)
    ok = ok and red.log == [(0, 256)]
    mid = grad.clone()
    ok = ok and torch.equal(mid[:256], base[:256] * 1.5) and torch.equal(mid[256:], base[256:] * (rank + 1))
    red.ready(900)                                # buckets 2..6
    red.finish()                                  # the short tail
    ok = ok and red.log == [(0, 256), (256, 896), (896, 1000)] and torch.equal(grad, base * 1.5)
    # blocking variant, sum instead of average
    g2 = base * (rank + 1)
    hd.all_reduce_flat(g2, average=False, bucket_bytes=4 * 300)
    ok = ok and torch.equal(g2, base * 3.0)
    hd.barrier()
    q.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_bucketed_gradient_all_reduce():
    """Row f2's gradient exchange (hawkeye_amd/dist.py BucketReducer / all_reduce_flat) on a flat arena, world size 2."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _param_grad_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from hawkeye_amd import dist as hd
    from hawkeye_amd.train import ParamArena
    hd.init("gloo")
    torch.manual_seed(0)                                  # identical replicas
    names = ["a", "b", "c", "d", "e"]
    params = {n: torch.nn.Parameter(torch.randn(4, 8)) for n in names}
    params["e"].requires_grad = False                     # a frozen tensor inside the arena
    ar = ParamArena(params, names, "cpu")
    ok = ar.attached()
    g = {n: torch.full((4, 8), float(i + 1) * (rank + 1)) for i, n in enumerate(names)}
    # what autograd can leave behind after the node returned arena views:
    ar.grad_view("a").copy_(g["a"]); params["a"].grad = ar.grad_view("a")        # adopted the view
    params["b"].grad = g["b"].clone()                                            # cloned / accumulated elsewhere
    if rank == 0:
        params["c"].grad = g["c"].clone()                                        # used on rank 0 only (e.g. Lt = 0 on rank 1)
    ar.grad_view("d").fill_(123.0)                                               # unused everywhere: stale arena content
    used = hd.reduce_param_grads(ar, average=True, bucket_bytes=4 * 48)
    ok = ok and used == [True, True, True, False, False]
    ok = ok and all(ar.grad_aliases(n, params[n]) for n in "abc") and params["d"].grad is None and params["e"].grad is None
    ok = ok and torch.equal(params["a"].grad, torch.full((4, 8), 1.5)) and torch.equal(params["b"].grad, torch.full((4, 8), 3.0))
    ok = ok and torch.equal(params["c"].grad, torch.full((4, 8), 1.5))           # (3 + 0) / 2 on BOTH ranks
    # the optimizer reads p.grad: one SGD step must leave both replicas identical
    opt = torch.optim.SGD([p for p in params.values() if p.requires_grad], lr=0.1)
    opt.step()
    flat = ar.flat.clone()
    other = [torch.empty_like(flat) for _ in range(world)]
    torch.distributed.all_gather(other, flat)
    ok = ok and torch.equal(other[0], other[1]) and ar.attached()
    hd.barrier()
    q.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_param_grads_end_identical_on_every_replica():
    """The autograd route of row f2: whatever `p.grad` is after backward (arena view, separate tensor, None on one
    rank), reduce_param_grads leaves every replica with the same averaged `p.grad` -- the tensors the optimizer
    reads -- and keeps globally unused parameters at None (tasks/shared_utils.py:54-59 find_unused_parameters)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_param_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_bucket_bounds_and_single_process_reducer():
    from hawkeye_amd import dist as hd
    assert hd.bucket_bounds(10, 4) == [(0, 4), (4, 8), (8, 10)]
    g = torch.ones(10)
    red = hd.BucketReducer(g, bucket_bytes=16)
    red.ready(9)
    red.finish()
    assert red.log == [(0, 8), (8, 10)] and torch.equal(g, torch.ones(10))      # no process group: values untouched


def _stand_in_encode(frames, instruction):
    """encode_frames stand-in: a [b, 4, 8] tensor that is a fixed function of each clip's bytes (so a clip's result does
    not depend on which rank or micro-batch carried it)."""
    b = frames.shape[0]
    f = frames.reshape(b, -1).to(torch.float64)
    feats = torch.stack([f.mean(1), f.std(1), f[:, ::97].sum(1), f[:, 5::1013].sum(1)], dim=1)
    return (feats[:, :, None] * torch.arange(1, 9, dtype=torch.float64)).float(), False


def _answer(example, turn, row, bound):
    from hawkeye_amd.grounding import CHOICES
    return CHOICES[(int(row.abs().sum().item()) + 3 * int(example) + turn) % 4]


def _batcher_worker(rank, world, port, n_examples, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from hawkeye_amd import dist as hd
    from hawkeye_amd.grounding import RecursiveGroundingBatcher, SyntheticFrameSource
    hd.init("gloo")
    examples = list(range(n_examples))
    b = RecursiveGroundingBatcher(_stand_in_encode, SyntheticFrameSource(size=32, seed=5), _answer, "sys", max_turns=4,
                                  num_segments=3, micro_batch=4, size=32, device="cpu", keep_outputs=True)
    mine = b.run(examples, rank, world)
    lo, hi = hd.shard_range(n_examples, rank, world)
    ok = sorted(mine) == examples[lo:hi]                       # --start-idx/--end-idx block of this rank
    # the only exchange of the path: all_gather of the first turn's outputs, uneven counts, in example order
    counts = [hd.shard_range(n_examples, r, world)[1] - hd.shard_range(n_examples, r, world)[0] for r in range(world)]
    local = torch.stack([mine[e]["outputs"][0] for e in examples[lo:hi]]) if hi > lo else torch.zeros(0, 4, 8)
    full = hd.gather_outputs(local, counts)
    hd.barrier()
    q.put((rank, bool(ok), {e: (r["bound"], r["answers"]) for e, r in mine.items()}, full))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_recursive_grounding_batcher_equals_single_rank():
    """BASELINE config 3 shards by EXAMPLE (tasks/test_recursive_grounding.py:481-482): both ranks of a gloo
    group drive RecursiveGroundingBatcher.run(examples, rank, world); the union of their results must equal the
    single-rank run, and gather_outputs with uneven counts must reassemble the outputs in example order."""
    from hawkeye_amd.grounding import RecursiveGroundingBatcher, SyntheticFrameSource
    n = 11                                                     # 6 + 5: uneven shards, micro-batches of 4 + tail
    single = RecursiveGroundingBatcher(_stand_in_encode, SyntheticFrameSource(size=32, seed=5), _answer, "sys", max_turns=4,
                                       num_segments=3, micro_batch=4, size=32, device="cpu", keep_outputs=True).run(list(range(n)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batcher_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    union = {}
    for _, _, part, _ in res:
        assert not (set(part) & set(union))
        union.update(part)
    assert union == {e: (r["bound"], r["answers"]) for e, r in single.items()}
    assert any(len(a) > 1 for _, a in union.values()) and any(a[-1] == "throughout" for _, a in union.values())
    want = torch.stack([single[e]["outputs"][0] for e in range(n)])
    for _, _, _, full in res:
        assert torch.equal(full, want)


def test_output_gatherer_without_a_process_group_passes_the_local_batch_through():
    from hawkeye_amd.dist import OutputGatherer
    g = OutputGatherer()
    a, b = torch.ones(2, 3), torch.zeros(2, 3)
    assert g.submit(a) is None
    assert g.submit(b) is a
    assert g.collect() is b and g.collect() is None
