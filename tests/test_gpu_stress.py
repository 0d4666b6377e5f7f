"""GPU suite: the cfg2 forward + backward is BITWISE repeatable under load (Weak #2: one undiagnosed one-off
failure of a deterministic fp32 forward on a slow box is what a timing-dependent LDS slot hand-over looks like).  The runs happen in
child processes (tests/stress_child.py) that conftest.py starts at session start, before this process initialises the GPU -- a
process that has must not start another GPU program on this pool -- one with a bandwidth hog on a second stream, one with a
second PROCESS (bench.py --steps 400, per-layer launches) training on the same GPU meanwhile.  The static side of the same
question is tools/audit_barrier_lds.py (no raw s_barrier of the ring kernels is reached with LDS reads in flight), run by
tests/test_cabi.py on the built objects.  Reference: /root/reference/model.py:194-204."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def _result(request, mode):
    kids = getattr(request.config, "_dcf_dp_children", None)
    if not kids:
        pytest.skip("stress children were not started (no GPU at session start)")
    outdir, procs = kids
    for name, p in procs:
        if name != "stress_" + mode:
            continue
        try:
            rc = p.wait(timeout=2400)
        except Exception:
            p.kill()
            raise AssertionError("child %s did not finish" % name)
        log = open(os.path.join(outdir, name + ".log")).read()
        path = os.path.join(outdir, "stress_%s.json" % mode)
        assert os.path.exists(path), "child %s wrote no result (rc %d):\n%s" % (name, rc, log[-3000:])
        res = json.load(open(path))
        print("stress %s: %s" % (mode, json.dumps(res)))            # (visible with -rA / in pytest.log also when green)
        return res, rc, log
    pytest.skip("no stress_%s child in this session" % mode)


@pytest.mark.parametrize("mode", ["stream", "sibling"])
def test_cfg2_forward_backward_is_bitwise_repeatable_under_load(request, mode):
    res, rc, log = _result(request, mode)
    for dt, r in res["dtypes"].items():
        assert not r["bad"], "%s, %s: %d of %d runs differ from run 0: %s" % (mode, dt, len(r["bad"]), res["runs"], r["bad"][:3])
        assert r["camera_fusion_worst_rel"] <= r["camera_fusion_bound"]
        assert 0 < r["lidar_arena_elements"] < r["arena_elements"]
    if mode == "sibling":
        assert max(r["runs_with_sibling_alive"] for r in res["dtypes"].values()) > 0, "the sibling process was never running beside the step"
    assert rc == 0, log[-2000:]
