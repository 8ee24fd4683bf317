"""world_size-2 gloo test of the multi-GPU layout: utterance sharding is a disjoint cover and the one collective
of the path (weight-blob broadcast) delivers rank 0's bytes to every rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from universal_speech_enhancement_amd import distributed as D


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 4, 8):
            spans = [D.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_bounds(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env(backend="gloo")
    blob = (torch.arange(4096, dtype=torch.int64) % 251).to(torch.uint8) if r == 0 else torch.zeros(4096, dtype=torch.uint8)
    D.broadcast_blob(blob, src=0)
    items = [f"utt{i}" for i in range(9)]
    mine = D.shard_list(items, r, w)
    mx = D.max_over_ranks(float(r + 1))
    q.put((r, int(blob.to(torch.int64).sum()), mine, mx))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_sharding():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    want = int((torch.arange(4096) % 251).sum())
    assert res[0][1] == want and res[1][1] == want
    assert res[0][2] + res[1][2] == [f"utt{i}" for i in range(9)]
    assert res[0][3] == 2.0 and res[1][3] == 2.0


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D.init_from_env(backend="gloo")
    torch.manual_seed(7)                                             # same parameters on both ranks
    params = [torch.nn.Parameter(torch.randn(n)) for n in (5, 300, 2, 1000, 64, 7)]
    params[2].requires_grad_(False)                                  # frozen: skipped
    for i, p in enumerate(params):
        if i == 5:                                                   # trainable on paper, but no rank has a gradient (detached in the
            continue                                                 # taped forward): must stay None, not become zeros
        if p.requires_grad and not (rank == 1 and i == 4):           # rank 1 has no gradient for this one: contributes zeros
            p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    n = D.allreduce_gradients(params, bucket_bytes=2048)             # the has-gradient mask + 2 KB buckets: [5, 300] | [1000] | [64]
    q.put((rank, n, [None if p.grad is None else float(p.grad[0]) for p in params],
           all(p.grad is None or bool((p.grad == p.grad[0]).all()) for p in params)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_in_buckets():
    """Data-parallel training exchange: gradients averaged over the ranks in contiguous buckets; frozen parameters skipped; a parameter
    without a gradient on one rank counts as zero there; one without a gradient on every rank keeps grad = None."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for r, n, firsts, uniform in res:
        assert n == 4 and uniform
        assert firsts == [1.5, 3.0, None, 6.0, 2.5, None]             # mean of (1, 2) x (i + 1); index 4: (5 + 0) / 2; index 5: untouched
    assert D.allreduce_gradients([torch.nn.Parameter(torch.zeros(3))]) == 0     # no process group: nothing to do
