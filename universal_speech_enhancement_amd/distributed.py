"""Multi-GPU layout of the sampling path: one process per GPU, utterances sharded over ranks, no data-path
collective.  The only communication is a one-time broadcast of the packed weight blob from rank 0 (RCCL over xGMI
when the process group's backend is "nccl"; ``gloo`` in the CPU tests).

The reference's own multi-device predict is Lightning DDP with ``batch_size // world_size`` per rank
(``src/data/loadwav_datamodule.py:53-60``): independent replicas on disjoint file shards.  The Langevin corrector's
step size is a mean over the *local* batch (``sampling/correctors.py:55-57``), so sharding reproduces the
reference's per-rank semantics exactly.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; initialises the default group if needed.  Under a launcher
    (RANK and WORLD_SIZE set) the group is created at world size 1 as well, so that the one-GPU run of a launched job goes
    through the same RCCL calls - communicator creation with ``device_id``, the weight broadcast, the MAX all-reduce - as
    the N-GPU run; a plain ``python bench.py`` (no launcher environment) stays without a process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of ``n_items`` for ``rank`` (first n%world ranks get one extra)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_list(items: Sequence, rank: int, world: int) -> List:
    lo, hi = shard_bounds(len(items), rank, world)
    return list(items[lo:hi])


def broadcast_blob(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast of a contiguous byte tensor (the packed weights) from ``src`` to every rank."""
    if dist.is_initialized():                               # (world size 1 included: the call path of the N-GPU run)
        dist.broadcast(blob, src=src)
    return blob


def broadcast_weights(engine, state_dict=None, src: int = 0):
    """Rank ``src`` packs + uploads ``state_dict``; the others allocate the blob; one broadcast fills them."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == src:
        if state_dict is None:
            raise ValueError("the source rank needs the state dict")
        engine.load_state_dict(state_dict)
    else:
        engine.alloc_weight_blob()
    broadcast_blob(engine.weight_blob(), src=src)
    if dist.is_initialized():
        torch.cuda.synchronize()


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
