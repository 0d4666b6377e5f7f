"""CPU suite: the N>1 data-parallel logic with gloo, world_size 2 (no GPU needed).

Covers what bench.py / train.py do around the kernels when frames shard across ranks:
gradient-arena all-reduce + 1/world scaling, rank-0 broadcast of the replicas, frame
sharding, and the max-over-ranks timing reduction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import PKG


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    train = importlib.import_module(PKG + ".train")
    assert train.world() == world
    # 1) gradient arena: sum over ranks, then the optimiser scales by 1/world  ==  mean of per-rank grads
    g = torch.arange(10, dtype=torch.float32) * (rank + 1)
    n = train.allreduce_grads(g)
    assert n == world
    expect = torch.arange(10, dtype=torch.float32) * sum(r + 1 for r in range(world))
    assert torch.equal(g, expect)
    # 2) replicas start identical: rank 0's arena wins
    flat = torch.full((7,), float(rank + 5))
    dist.broadcast(flat, 0)
    assert torch.equal(flat, torch.full((7,), 5.0))
    # 3) frames shard by rank with no overlap: global batch = world * per-rank batch
    B = 2
    mine = [rank * B + i for i in range(B)]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    assert sorted(sum(gathered, [])) == list(range(world * B))
    # 3b) the evaluation fences are host barriers on the side group init_distributed() makes (never an RCCL collective)
    os.environ["WORLD_SIZE"] = str(world)
    assert train.init_distributed() == world and train._HOST_PG is not None
    import time
    if rank == 1:
        time.sleep(0.3)                     # rank 0 "evaluates" meanwhile: the others just wait on the host
    train._eval_barrier(timeout_s=60)
    # 4) timing: the bench reports MAX over ranks
    t = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert abs(t.item() - 0.010 * world) < 1e-12
    # 5) data-parallel equivalence on a toy quadratic: 2 ranks x 1 sample == 1 rank x 2 samples (sum loss, eval-BN analogue)
    w = torch.tensor([1.0, -2.0, 0.5])
    xs = torch.tensor([[1.0, 2.0, 3.0], [0.5, -1.0, 2.0]])
    local = 2 * (w * xs[rank]).sum() * xs[rank]                 # d/dw of (w.x)^2 for this rank's sample
    buf = local.clone()
    train.allreduce_grads(buf)
    full = sum(2 * (w * xs[r]).sum() * xs[r] for r in range(world))
    assert torch.allclose(buf, full)
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"


def test_single_process_world_is_one():
    import importlib
    train = importlib.import_module(PKG + ".train")
    assert train.world() == 1
    g = torch.ones(4)
    assert train.allreduce_grads(g) == 1 and torch.equal(g, torch.ones(4))
