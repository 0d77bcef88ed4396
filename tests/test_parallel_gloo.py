"""CPU, world_size 2 over gloo: the data-parallel exchange of the hot path (one all-reduce of the flat gradient
buffer, 1/world fused into the optimiser, rank-local BN buffers, even pair-preserving shards)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from sound_event_detection_dcase2017_task4_amd import parallel
    r, w, lr = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    # parameters: rank 0's values win
    flat = torch.full((1000,), float(rank + 1))
    parallel.broadcast_flat(flat)
    assert torch.all(flat == 1.0)
    # gradients: sum over ranks, in one call and in 3 buckets, sync and async
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    parallel.allreduce_flat_grad(g)
    assert torch.allclose(g, torch.arange(1000, dtype=torch.float32) * 3)
    g2 = torch.ones(1001) * (rank + 1)
    hs = parallel.allreduce_flat_grad(g2, buckets=3, async_op=True)
    for h in hs:
        h.wait()
    assert torch.all(g2 == 3.0)
    # buffers broadcast once at start
    bn = torch.nn.BatchNorm2d(4)
    bn.running_mean.fill_(float(rank))
    parallel.broadcast_buffers(bn)
    assert torch.all(bn.running_mean == 0.0)
    # data-parallel SGD-equivalence on a toy problem: mean of per-rank gradients == gradient of the global batch
    torch.manual_seed(0)
    wgt = torch.randn(5, requires_grad=True)
    X = torch.randn(8, 5); Y = torch.randn(8)
    lo, hi = parallel.shard_range(8, rank, world)
    assert (hi - lo) == 4 and lo % 2 == 0                          # mixup pairs (2i, 2i+1) stay on one rank
    loss = ((X[lo:hi] @ wgt - Y[lo:hi]) ** 2).mean()
    loss.backward()
    gl = wgt.grad.clone()
    parallel.allreduce_flat_grad(gl)
    gl /= world
    full = torch.autograd.grad(((X @ wgt.detach().requires_grad_(True) - Y) ** 2).mean(), [])[0] if False else None
    w2 = wgt.detach().clone().requires_grad_(True)
    ((X @ w2 - Y) ** 2).mean().backward()
    assert torch.allclose(gl, w2.grad, atol=1e-6)
    dist.barrier()
    if rank == 0:
        open(out, "w").write("ok")
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    out = str(tmp_path / "ok.txt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_single_process_is_a_noop():
    sys.path.insert(0, REPO)
    from sound_event_detection_dcase2017_task4_amd import parallel
    g = torch.ones(10)
    assert parallel.allreduce_flat_grad(g) == [] and torch.all(g == 1)
    assert parallel.world_size() == 1 and parallel.shard_range(10, 0, 1) == (0, 10)
