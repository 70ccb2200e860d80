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


def max_over_ranks(seconds: float, device) -> float:
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()
