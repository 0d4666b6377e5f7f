"""GPU suite: data-parallel equivalence of Train.one_step with world_size 2 ON ONE GPU (N ranks x B frames
must equal one rank x N*B frames up to re-association).  Two rank processes share GPU 0 and all-reduce the flat gradient
arena through gloo; a third process runs the same two frames as one batch.  The children are started by conftest.py at
session start, BEFORE this pytest process touches the GPU (a process that has initialised HIP must not exec another
program on this pool), and run while the other GPU tests do.  Reference: /root/reference/train.py:24,51-56 (DDP averages the
gradients of identical replicas)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _wait(request, prefix):
    kids = getattr(request.config, "_dcf_dp_children", None)
    if not kids:
        pytest.skip("data-parallel children were not started (no GPU at session start)")
    outdir, procs = kids
    for name, p in procs:
        if not name.startswith(prefix):
            continue
        try:
            rc = p.wait(timeout=900)
        except Exception:
            p.kill()
            raise AssertionError("child %s did not finish" % name)
        log = open(os.path.join(outdir, name + ".log")).read()
        assert rc == 0, "child %s failed:\n%s" % (name, log[-3000:])
    return outdir


def test_bench_two_ranks_on_one_gpu(request):
    """`python bench.py --gpus 2 --steps 3 --warmup 1` typed WITHOUT a launcher and with a clean environment (the driver's scaling
    command): bench.py starts its two rank processes itself, both here on GPU 0 with gloo in place of RCCL -- cfg2 steps including
    the bucketed gradient all-reduce, barrier-fenced timing, MAX over ranks, ONE JSON line (rank 0's) whose n_gpus is the flag's.
    A functional check of the code path the 8-GPU scaling run takes (reference train.py:24,51-56)."""
    import json
    outdir = _wait(request, "bench_launch")
    lines = [ln for ln in open(os.path.join(outdir, "bench_launch.out")).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                                      # only rank 0 prints the line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["parallelism"] == "dp2" and d["roofline"] is not None


def test_two_ranks_equal_one_rank_with_twice_the_batch(request):
    outdir = _wait(request, "w")
    one = torch.load(os.path.join(outdir, "w1_r0.pt"))
    r0 = torch.load(os.path.join(outdir, "w2_r0.pt"))
    r1 = torch.load(os.path.join(outdir, "w2_r1.pt"))
    lr = one["lr"]
    # step 1 starts from identical replicas: the averaged gradient of the two ranks IS the gradient of the two-frame batch,
    # up to fp32 re-association (and the order of the float atomics in the fusion backward)
    a, b, c = one["grads"][0], r0["grads"][0], r1["grads"][0]
    assert torch.equal(b, c), "the ranks hold different reduced gradients"
    assert float((a - b).abs().max()) < 1e-5 * float(a.abs().max()), "step 0: averaged gradient differs by %g" % float((a - b).abs().max())
    for step in range(2):
        a, b, c = one["params"][step], r0["params"][step], r1["params"][step]
        assert torch.equal(b, c), "replicas diverged at step %d" % step          # same all-reduced gradient, same Adam step
        d = (a - b).abs()
        # Adam divides by sqrt(v): a parameter whose gradient is at rounding-noise level can move by up to +-lr per step in
        # either run.  So: all but a sliver of the parameters agree to 1e-5, and nothing is further apart than Adam can put it.
        frac = float((d > 1e-5 * float(a.abs().max())).float().mean())
        assert frac < 1e-3, "step %d: %.2e of the parameters differ" % (step, frac)
        assert float(d.max()) <= 2.5 * lr * (step + 1), "step %d: parameters %g apart" % (step, float(d.max()))
    # the steps moved the parameters at all (learning rate 1e-3, Adam: ~1e-3 per step)
    assert float((one["params"][1] - one["params"][0]).abs().max()) > 1e-4


def test_rccl_bucketed_allreduce_world_size_one(request):
    """The product's overlapped, bucketed gradient all-reduce on RCCL itself (backend "nccl", a one-rank process group on the
    one GPU of this box; tests/dp_child.py run_rccl1): the LiDAR + fusion bucket is finalised and handed to
    dist.all_reduce(async_op=True) from the autograd thread while the camera stream's backward still runs, the camera bucket
    follows, Adam waits for both.  Gradients over three steps equal those of the plain single-rank step (to the 1e-4 of the
    largest gradient that the float atomics of the fusion backward leave between any two runs; parameters as far as Adam keeps noise-level gradients
    together).  With a PreMulSum reduction the collective DOUBLES the buffer: the arena then holds exactly twice the plain
    gradient -- which it can only when RCCL's stream ran behind the finalisation launch -- and the negative control (hook
    called before the finalisation: the launch overwrites the doubled, stale arena) is detected.
    Reference: /root/reference/train.py:24,51-56."""
    outdir = _wait(request, "rccl1")
    r = torch.load(os.path.join(outdir, "rccl1.pt"))
    assert r["backend"] == "nccl"
    plain, ov = r["plain"], r["overlap"]
    assert "error" not in plain and "error" not in ov, (plain.get("error"), ov.get("error"))
    # the hook really ran per bucket: (LiDAR, fusion) then (camera), together covering the arena
    # four buckets per step, in the order the backward completes them: LiDAR stages 4-5 + FPN + heads; the rest of the LiDAR
    # stream and the fusion layers (two arena ranges); camera layer4 + FPN; the rest of the camera stream
    assert [len(c) for c in ov["hook_calls"]] == [1, 2, 1, 1] * 3
    assert sum(b - a for c in ov["hook_calls"][:4] for a, b in c) == ov["numel"]
    assert not plain["hook_calls"]
    lr = 1e-3

    def close(a, b, what):
        # fp32 model: two runs differ by the summation order of the float atomics (fusion dW1d / db1 / slice-crossing dP rows,
        # point-sample scatter), up to ~1e-5 of the largest gradient on sums that cancel (measured over repeated runs: 0.4-2e-5);
        # an all-reduce that raced the finalisation launch is O(1) off (the negative control below)
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), "%s: %g apart (max %g)" % (what, float((a - b).abs().max()), float(b.abs().max()))

    def params_close(a, b, step, what):
        d = (a - b).abs()
        # (Adam divides by sqrt(v): parameters whose gradients are at noise level drift apart by up to +-lr per step in any two runs)
        assert float((d > 1e-5 * float(a.abs().max())).float().mean()) < 1e-3 * (step + 1) and float(d.max()) <= 2.5 * lr * (step + 1), what

    close(ov["grads"][0], plain["grads"][0], "bucketed gradients, step 0")
    for step in range(3):
        params_close(ov["params"][step], plain["params"][step], step, "bucketed path, parameters after step %d" % step)
    pm, wrong = r["premul"], r["wrong"]
    assert "error" not in pm, pm.get("error")
    close(pm["grads"][0], 2.0 * plain["grads"][0], "PreMulSum reduction, step 0")
    for step in range(3):
        params_close(pm["params"][step], plain["params"][step], step, "PreMulSum path, parameters after step %d" % step)
    # bf16 buckets: every gradient rounded to bf16 once (one rank: the sum is the rounded value itself)
    b16 = r["bf16"]
    assert "error" not in b16, b16.get("error")
    g, g16 = plain["grads"][0], b16["grads"][0]
    assert float((g16 - g).abs().max()) <= 2.0 ** -8 * float(g.abs().max()) and float(((g16 - g).abs() > 2.0 ** -8 * g.abs() + 1e-4 * float(g.abs().max())).float().mean()) == 0.0
    assert torch.equal(g16, g16.bfloat16().float()), "the arena does not hold bf16 values"
    for step in range(3):
        assert float((b16["params"][step] - plain["params"][step]).abs().max()) <= 2.5 * lr * (step + 1)
    assert "error" not in wrong, wrong.get("error")
    (a0, b0), _ = wrong["hook_calls"][0]            # LiDAR-stream range of the arena
    got, want = wrong["grads"][0][a0:b0], 2.0 * plain["grads"][0][a0:b0]
    assert float((got - want).abs().max()) > 0.25 * float(want.abs().max()), \
        "negative control: an all-reduce started before the finalisation launch went unnoticed"
    # (what it holds instead is a race between the collective and the launch: the undoubled gradient, or twice the stale arena)


def test_two_devices_rccl_equal_one_rank_with_twice_the_batch(request):
    """Where the box has TWO GPUs: rank r on device r, gradient buckets exchanged by RCCL over xGMI while the backward runs
    (the product's launch mode, /root/reference/train.py:24,51-56 replaced by one process per GPU); 2 x (B = 1) must equal
    1 x (B = 2).  Skipped on a one-GPU box (conftest.py starts the children only when two devices are visible)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the two-device RCCL exchange needs two")
    outdir = _wait(request, "n2_r")
    _wait(request, "w1_r0")
    one = torch.load(os.path.join(outdir, "w1_r0.pt"))
    r0 = torch.load(os.path.join(outdir, "n2_r0.pt"))
    r1 = torch.load(os.path.join(outdir, "n2_r1.pt"))
    a, b, c = one["grads"][0], r0["grads"][0], r1["grads"][0]
    assert torch.equal(b, c), "the ranks hold different reduced gradients"
    assert float((a - b).abs().max()) < 1e-4 * float(a.abs().max())
    lr = one["lr"]
    for step in range(2):
        assert torch.equal(r0["params"][step], r1["params"][step]), "replicas diverged at step %d" % step
        assert float((one["params"][step] - r0["params"][step]).abs().max()) <= 2.5 * lr * (step + 1)
