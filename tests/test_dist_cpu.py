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
    t = hd.max_over_ranks(1.0 + rank, "cpu")
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
