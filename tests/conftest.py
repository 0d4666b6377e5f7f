import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Start the child processes of tests/test_gpu_dp.py (two data-parallel ranks sharing GPU 0 + the one-rank run they
    must equal) BEFORE this process initialises the GPU: torch.cuda.device_count() does not, anything later does, and a
    process that has initialised HIP must not exec another program on the GPU pool."""
    import socket
    import subprocess
    import tempfile
    import torch
    config = session.config
    expr = getattr(config.option, "markexpr", "") or ""
    if "not gpu" in expr or os.environ.get("DCF_NO_DP_CHILDREN") or torch.cuda.device_count() < 1:
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    outdir = tempfile.mkdtemp(prefix="dcf_dp_")
    child = os.path.join(ROOT, "tests", "dp_child.py")
    procs = []
    for name, world, rank in (("w1_r0", 1, 0), ("w2_r0", 2, 0), ("w2_r1", 2, 1)):
        log = open(os.path.join(outdir, name + ".log"), "w")
        procs.append((name, subprocess.Popen([sys.executable, child, str(world), str(rank), port, outdir], stdout=log, stderr=subprocess.STDOUT,
                                             cwd=ROOT)))
    # ... and bench.py itself with two ranks on this one GPU (gloo instead of RCCL): the N > 1 code path of the benchmark
    s2 = socket.socket()
    s2.bind(("127.0.0.1", 0))
    port2 = str(s2.getsockname()[1])
    s2.close()
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port2, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", DCF_DIST_BACKEND="gloo")
        out = open(os.path.join(outdir, "bench_r%d.out" % rank), "w")
        log = open(os.path.join(outdir, "bench_r%d.log" % rank), "w")
        procs.append(("bench_r%d" % rank, subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                                                            "--no-cpu-baseline"], stdout=out, stderr=log, cwd=ROOT, env=env)))
    config._dcf_dp_children = (outdir, procs)


@pytest.fixture(scope="session")
def dcf():
    """The product package (its directory name has a hyphen, so it is imported by name)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped automatically where no GPU is visible (the CPU container).
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
