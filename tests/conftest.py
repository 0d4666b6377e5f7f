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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    return port


def _start_dp_children(config):
    """Child processes of tests/test_gpu_dp.py: two data-parallel ranks sharing GPU 0 + the one-rank run they must equal, a
    world-size-1 RCCL rank (the bucketed, overlapped all-reduce on the real backend), and bench.py with two ranks.  They are
    started BEFORE this process initialises the GPU: torch.cuda.device_count() does not, anything later does, and a process
    that has initialised HIP must not exec another program on the GPU pool."""
    import subprocess
    import tempfile
    outdir = tempfile.mkdtemp(prefix="dcf_dp_")
    child = os.path.join(ROOT, "tests", "dp_child.py")
    procs = []
    port = _free_port()
    for name, world, rank in (("w1_r0", 1, 0), ("w2_r0", 2, 0), ("w2_r1", 2, 1)):
        log = open(os.path.join(outdir, name + ".log"), "w")
        procs.append((name, subprocess.Popen([sys.executable, child, str(world), str(rank), port, outdir], stdout=log, stderr=subprocess.STDOUT,
                                             cwd=ROOT)))
    log = open(os.path.join(outdir, "rccl1.log"), "w")
    procs.append(("rccl1", subprocess.Popen([sys.executable, child, "rccl1", "0", _free_port(), outdir], stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)))
    import torch
    if torch.cuda.device_count() >= 2:
        # two REAL devices: the same two ranks as above on GPU 0 and GPU 1, exchanging their gradient buckets through RCCL
        port3 = _free_port()
        for rank in range(2):
            log = open(os.path.join(outdir, "n2_r%d.log" % rank), "w")
            procs.append(("n2_r%d" % rank, subprocess.Popen([sys.executable, child, "rccl2", str(rank), port3, outdir], stdout=log,
                                                            stderr=subprocess.STDOUT, cwd=ROOT)))
    # ... and bench.py itself exactly as the driver types it -- `python bench.py --gpus 2 ...`, NO launcher and none of its variables
    # in the environment: bench.py has to start its two ranks itself (both on this one GPU, gloo instead of RCCL)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DCF_DIST_BACKEND"] = "gloo"
    out = open(os.path.join(outdir, "bench_launch.out"), "w")
    log = open(os.path.join(outdir, "bench_launch.log"), "w")
    procs.append(("bench_launch", subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                                                    "--no-cpu-baseline"], stdout=out, stderr=log, cwd=ROOT, env=env)))
    # ... and the repeatability stress runs of tests/test_gpu_stress.py (they start their own sibling process before touching the GPU)
    if not os.environ.get("DCF_NO_STRESS_CHILDREN"):
        for mode in ("stream", "sibling"):
            log = open(os.path.join(outdir, "stress_%s.log" % mode), "w")
            procs.append(("stress_" + mode, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "stress_child.py"), mode, outdir,
                                                              os.environ.get("DCF_STRESS_RUNS", "50")], stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)))
    config._dcf_dp_children = (outdir, procs)


def pytest_sessionfinish(session, exitstatus):
    """Children that are still running (an aborted or deselected session) are ended here: none outlives the session."""
    kids = getattr(session.config, "_dcf_dp_children", None)
    if not kids:
        return
    for name, p in kids[1]:
        if p.poll() is None:
            p.kill()
            try:
                p.wait(timeout=30)
            except Exception:
                pass


@pytest.fixture(scope="session")
def dcf():
    """The product package (its directory name has a hyphen, so it is imported by name)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.hookimpl(trylast=True)          # after -m / -k have deselected
def pytest_collection_modifyitems(config, items):
    import torch
    # The data-parallel children are started only when their tests were collected AND will run (a GPU is visible, the
    # selection keeps them) -- and before anything below initialises the GPU in this process.  Under pytest-xdist every
    # worker would start its own set on the one GPU: the tests skip there instead.
    expr = getattr(config.option, "markexpr", "") or ""
    wanted = [it for it in items if it.fspath.basename in ("test_gpu_dp.py", "test_gpu_stress.py")]
    if (wanted and "not gpu" not in expr and not os.environ.get("DCF_NO_DP_CHILDREN") and not os.environ.get("PYTEST_XDIST_WORKER")
            and torch.cuda.device_count() >= 1 and not hasattr(config, "_dcf_dp_children")):
        _start_dp_children(config)
    # GPU tests are skipped automatically where no GPU is visible (the CPU container).
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
