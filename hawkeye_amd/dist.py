"""Clip-sharded multi-GPU helpers.

encode_img has no cross-clip arithmetic, so the path shards by clip with a full
weight replica per GPU and NO data-path collective.  The only exchange is the
optional all_gather of the [b_local, 96, 4096] outputs for a consumer that wants the
whole batch (RCCL over xGMI on GPUs; gloo in the CPU tests).  One process per GPU,
rendezvous from the torchrun environment.
"""
import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: str = "nccl") -> Tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Block partition of `total` clips (or recursive-grounding examples, the reference's
    --start-idx/--end-idx, tasks/test_recursive_grounding.py:481-482): rank r gets [lo, hi)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_outputs(local: torch.Tensor, counts: List[int] = None) -> torch.Tensor:
    """all_gather of per-rank [b_r, nq, C] outputs along the clip dim (equal b_r unless counts given)."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()
    local = local.contiguous()
    if counts is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out
    bmax = max(counts)
    pad = torch.zeros((bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


class OutputGatherer:
    """The same all_gather, overlapped with the NEXT call's kernels: submit(local) enqueues the collective on the
    backend's own stream (RCCL: it starts when the compute stream reaches this point and runs beside what is
    launched after it) and hands back the PREVIOUS call's gathered batch; collect() returns the last one.  A consumer
    that works one call behind (a throughput loop, the grounding batcher's next micro-batch) never waits on xGMI.
    The local output is copied first: the engine's output buffer is pooled and the next call writes it again."""

    def __init__(self):
        self.pending = None

    def submit(self, local: torch.Tensor):
        prev = self.collect()
        if not dist.is_initialized():
            self.pending = (None, local)
            return prev
        src = local.clone(memory_format=torch.contiguous_format)
        out = torch.empty((dist.get_world_size() * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        work = dist.all_gather_into_tensor(out, src, async_op=True)
        self.pending = (work, out)
        return prev

    def collect(self):
        """The gathered batch of the last submit() (the current stream waits for it), or None if there is none."""
        if self.pending is None:
            return None
        work, out = self.pending
        self.pending = None
        if work is not None:
            work.wait()
        return out


def max_over_ranks(seconds: float, device) -> float:
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()


def ranks_seen(device) -> List[dict]:
    """What every rank of the job reports about itself, gathered over the process group's own backend (RCCL on GPUs):
    [{"rank", "local_rank", "device_count", "device_index"}] in rank order.  bench.py puts it into its JSON line so a
    multi-GPU run shows that the collective really spanned N ranks on N distinct devices."""
    rank, local_rank, world = env_world()
    device = torch.device(device)
    on_gpu = device.type == "cuda"
    mine = torch.tensor([rank, local_rank, torch.cuda.device_count() if on_gpu else 0,
                         device.index if on_gpu and device.index is not None else (torch.cuda.current_device() if on_gpu else -1)],
                        dtype=torch.int64, device=device)
    if not dist.is_initialized():
        rows = mine.view(1, 4)
    else:
        rows = torch.empty(dist.get_world_size() * 4, dtype=torch.int64, device=device)   # flat: what gloo accepts too
        dist.all_gather_into_tensor(rows, mine)
        rows = rows.view(-1, 4)
    return [dict(zip(("rank", "local_rank", "device_count", "device_index"), map(int, r))) for r in rows.cpu().tolist()]


# ---------------------------------------------------------------------------------------------------------------
# Row f2: data-parallel gradient exchange of the stage-3 step (the reference relies on
# DistributedDataParallel's bucketed all-reduce, tasks/shared_utils.py:54-59).  The trainable parameters' gradients
# live in ONE flat fp32 arena laid out in backward-completion order (hawkeye_amd/train.py), so the exchange is an
# all-reduce over contiguous ranges of that arena: no per-tensor flattening, no copy-in / copy-out.
# ---------------------------------------------------------------------------------------------------------------

def _all_reduce_avg(t: torch.Tensor, average: bool):
    if average and dist.get_backend() == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if average:
            t.div_(dist.get_world_size())


def bucket_bounds(total: int, bucket_elems: int) -> List[Tuple[int, int]]:
    """[lo, hi) element ranges of the arena's buckets (the last one may be short)."""
    bucket_elems = max(int(bucket_elems), 1)
    return [(lo, min(lo + bucket_elems, total)) for lo in range(0, total, bucket_elems)]


def all_reduce_flat(grad: torch.Tensor, average: bool = True, bucket_bytes: int = 64 << 20):
    """Blocking variant: all-reduce the whole arena bucket by bucket on the current stream."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return grad
    for lo, hi in bucket_bounds(grad.numel(), bucket_bytes // grad.element_size()):
        _all_reduce_avg(grad[lo:hi], average)
    return grad


def used_anywhere(used: List[bool], device) -> List[bool]:
    """Element-wise OR over the ranks of a per-parameter 'has a gradient' list.  A parameter no rank used keeps
    grad None everywhere (what DDP's find_unused_parameters bookkeeping does, tasks/shared_utils.py:54-59); one used
    on any rank must get the reduced gradient on every rank, or the replicas drift apart."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(used)
    flags = torch.tensor([1 if u else 0 for u in used], dtype=torch.int32, device=device)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    return [bool(f) for f in flags.tolist()]


def reduce_param_grads(arena, average: bool = True, bucket_bytes: int = 64 << 20):
    """Blocking data-parallel exchange for the autograd route: `arena` is a hawkeye_amd.train.ParamArena.  Gradients
    that live outside the arena (autograd cloned or accumulated them) are copied in and `p.grad` is rebound to the
    arena view, parameters without a gradient on this rank but with one elsewhere contribute zeros and receive the
    result, the flat arena is all-reduced, and every rank ends with identical `p.grad`s."""
    used = used_anywhere([p.requires_grad and p.grad is not None for p in arena.params], arena.grad.device)
    used = [u and p.requires_grad for u, p in zip(used, arena.params)]
    arena.bind_grads({n for n, u in zip(arena.names, used) if u})
    with torch.no_grad():
        for n, p, u in zip(arena.names, arena.params, used):
            if not u and p.requires_grad:
                arena.grad_view(n, p).zero_()        # unused everywhere: nothing to add, p.grad stays None
    all_reduce_flat(arena.grad, average=average, bucket_bytes=bucket_bytes)
    return used


class BucketReducer:
    """Overlaps the gradient all-reduce with the backward.  QFormerTrainEngine.backward calls ready(n) when arena
    elements [0, n) are final; every whole bucket below n that has not been sent yet is all-reduced on a side
    stream (RCCL over xGMI runs beside the remaining backward kernels); finish() sends the rest and makes the
    compute stream wait.  On CPU tensors (gloo, tests) the same logic runs synchronously.

    Bucket size: xGMI is point-to-point, a ring all-reduce is bound by one ~153 GB/s link, so buckets are large
    (default 64 MB: ~1 ms on the wire at 8 GPUs) -- a dozen collectives per step instead of DDP's 25 MB default."""

    def __init__(self, grad: torch.Tensor, average: bool = True, bucket_bytes: int = 64 << 20, force: bool = False):
        """force: run the collectives even in a one-rank group (tests: the overlapped path on one GPU)."""
        self.grad, self.average = grad, average
        self.bounds = bucket_bounds(grad.numel(), bucket_bytes // grad.element_size())
        self.active = dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.cuda = grad.is_cuda
        self.stream = torch.cuda.Stream(device=grad.device) if (self.cuda and self.active) else None
        self.sent = 0
        self.log: List[Tuple[int, int]] = []     # ranges in the order they were reduced (inspection / tests)

    def reset(self):
        self.sent = 0
        self.log = []

    def _send(self, upto_bucket: int):
        if upto_bucket <= self.sent:
            return
        lo, hi = self.bounds[self.sent][0], self.bounds[upto_bucket - 1][1]
        self.log.append((lo, hi))
        if self.active:
            if self.stream is not None:
                self.stream.wait_stream(torch.cuda.current_stream(self.grad.device))
                with torch.cuda.stream(self.stream):
                    for b in range(self.sent, upto_bucket):
                        l, h = self.bounds[b]
                        _all_reduce_avg(self.grad[l:h], self.average)
            else:
                for b in range(self.sent, upto_bucket):
                    l, h = self.bounds[b]
                    _all_reduce_avg(self.grad[l:h], self.average)
        self.sent = upto_bucket

    def ready(self, n_final: int):
        k = self.sent
        while k < len(self.bounds) and self.bounds[k][1] <= n_final:
            k += 1
        self._send(k)

    def finish(self):
        self._send(len(self.bounds))
        if self.stream is not None:
            torch.cuda.current_stream(self.grad.device).wait_stream(self.stream)
        self.sent = 0
