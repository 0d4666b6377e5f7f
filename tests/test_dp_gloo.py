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


def _bucket_worker(rank, world, port, out):
    """One of `world` ranks: the backend's bucket bookkeeping (HipBackend.bucket_ready / end_backward over the cfg2 layer table)
    drives Train._bucket_ready exactly as a backward does -- lidar_hi, lidar+fusion, image_hi, remainder -- on a CPU gradient
    arena; the collectives are gloo's asynchronous all-reduces."""
    import importlib
    import sys
    import types
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    engine = importlib.import_module(PKG + ".engine")
    backend_hip = importlib.import_module(PKG + ".backend_hip")
    train = importlib.import_module(PKG + ".train")
    plan = engine.Plan(bench.kitti_config(2), with_image=True)
    n = plan.table.n_params
    K = backend_hip.HipBackend.__new__(backend_hip.HipBackend)          # bookkeeping only: no device, no library call
    K.params = torch.zeros(n)
    K.nconv = len(plan.layers)
    finalised = []
    K._finalize = lambda lo, hi: finalised.append((lo, hi))
    K._flush_wgrads = lambda: None
    T = train.Train.__new__(train.Train)
    g = torch.arange(n, dtype=torch.float32).remainder_(97.0) * float(rank + 1)
    T.model = types.SimpleNamespace(flat_grads=g)
    T.grad_bucket_dtype, T.allreduce_premul = "f32", None
    T._pending, T._reduced, T._widen = [], 0, []
    calls = []

    def hook(ranges):
        calls.append(list(ranges))
        T._bucket_ready(ranges)
    K.bucket_hook = hook
    for which in ("lidar_hi", "lidar+fusion", "image_hi"):
        K.bucket_ready(plan.layers, which)
    K.end_backward(plan.layers)
    for w in T._pending:
        w.wait()
    # four hand-overs whose arena ranges tile [0, n) exactly once
    assert len(calls) == 4, calls
    flat = sorted(r for c in calls for r in c)
    assert flat[0][0] == 0 and flat[-1][1] == n and all(flat[i][1] == flat[i + 1][0] for i in range(len(flat) - 1)), flat
    assert T._reduced == n
    # every layer finalised exactly once
    cover = sorted(finalised)
    assert cover[0][0] == 0 and cover[-1][1] == len(plan.layers) and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
    expect = torch.arange(n, dtype=torch.float32).remainder_(97.0) * float(sum(r + 1 for r in range(world)))
    assert torch.equal(g, expect)
    sizes = [sum(b - a for a, b in c) * 4 / 1e6 for c in calls]
    if rank == 0:
        out.put(sizes)
    dist.destroy_process_group()


def test_four_gradient_buckets_world_size_8_gloo():
    """cfg3's exchange on 8 ranks (reference train.py:24: DDP's bucketed gradient all-reduce): the four buckets the backward hands
    over cover the cfg2 gradient arena exactly once and every rank ends with the sum over ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    sizes = q.get(timeout=5)
    assert len(sizes) == 4 and all(s > 5.0 for s in sizes) and 90.0 < sum(sizes) < 100.0, sizes      # MB: 96 MB in all, none of them tiny


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """`python bench.py --gpus 2` under a launcher that exported WORLD_SIZE=4 must not print an n_gpus line at all."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr and not p.stdout.strip()


def test_bench_rank_environments():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    envs = bench.rank_envs(8, 29511, base={"PATH": "/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert [e["RANK"] for e in envs] == [str(r) for r in range(8)] and [e["LOCAL_RANK"] for e in envs] == [str(r) for r in range(8)]
    assert all(e["WORLD_SIZE"] == "8" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511" and
               e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
