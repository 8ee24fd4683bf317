"""Multi-GPU layout of the sampling path: one process per GPU, utterances sharded over ranks, no data-path
collective.  The only communication is a one-time broadcast of the packed weight blob from rank 0 (RCCL over xGMI
when the process group's backend is "nccl"; ``gloo`` in the CPU tests).  Training (``SGMSEModule.training_step``) is data-parallel:
each rank takes its share of the batch and ``allreduce_gradients`` averages the gradients in a few large buckets after ``backward()``.

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
        if blob.is_cuda and dist.get_backend() == "gloo":    # gloo moves host memory: stage the blob (tests: two ranks sharing one GPU)
            host = blob.cpu()
            dist.broadcast(host, src=src)
            if dist.get_rank() != src:
                blob.copy_(host)
        else:
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


def allreduce_gradients(params, bucket_bytes: int = 256 << 20, average: bool = True) -> int:
    """Data-parallel training step, exchange half: average ``p.grad`` of ``params`` over the ranks (what Lightning's DDP strategy does
    for the reference's ``training_step``, SGMSE_module.py:46-54 with trainer strategy ddp).  Gradients are packed into contiguous
    buckets of ``bucket_bytes`` (default 256 MB: NCSN++ large's 65 M fp32 gradients travel as ONE ring all-reduce - xGMI is
    point-to-point, a ring is bound per link, so few large messages beat DDP's 25 MB default), reduced in place and copied back.
    Parameters without a gradient on this rank contribute zeros; those without one on ANY rank are left alone (one small MAX all-reduce
    of the has-gradient mask finds them; every rank must call with the same parameter list).  Returns the number of collectives issued
    (0 without a process group)."""
    params = [p for p in params if p.requires_grad]
    if not dist.is_initialized() or not params:
        return 0
    world = dist.get_world_size()
    # a parameter without a gradient on EVERY rank (frozen in effect: e.g. detached in the taped forward) keeps grad = None, as in
    # single-process training - materialising zeros would let Adam's weight decay move it
    has = torch.tensor([0 if p.grad is None else 1 for p in params], dtype=torch.int32, device=params[0].device)
    dist.all_reduce(has, op=dist.ReduceOp.MAX)
    params = [p for p, k in zip(params, has.tolist()) if k]
    if not params:
        return 1
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nb = p.numel() * p.element_size()
        if cur and (cur_bytes + nb > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
            buckets.append(cur); cur, cur_bytes = [], 0
        cur.append(p); cur_bytes += nb
    if cur:
        buckets.append(cur)
    for b in buckets:
        flat = torch.zeros(sum(p.numel() for p in b), dtype=b[0].dtype, device=b[0].device)
        o = 0
        for p in b:
            if p.grad is not None:
                flat[o:o + p.numel()].copy_(p.grad.reshape(-1))
            o += p.numel()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat.div_(world)
        o = 0
        for p in b:
            g = flat[o:o + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            o += p.numel()
    return len(buckets) + 1
