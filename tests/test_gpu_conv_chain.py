"""GPU parity of the CHAIN launches (csrc/conv_chain.hip, conv_rs_kernel.h): a residual stage's 3x3 / stride-1 layers
(/root/reference/model.py:32-41, :48-60) in one persistent launch must give, bit for bit, what the same layers give as separate
dcf_conv2d_fwd / dcf_conv2d_dgrad launches of the row-sharing kernel -- the kernel body is the same; what is new is the
hand-off between workgroups inside the launch (arrival counters, write-through stores, sc1 loads), so every word of every
layer's output is compared, over repeated launches into recycled buffers, with an L1-warming read of the buffers in between
and with another kernel competing for the CUs.  One separate launch is also checked against torch's fp32 convolution."""
import pytest
import torch
import torch.nn.functional as F

from _util import TORCH_DT, from_dev, pkg, q, rel_err, rnd, to_dev

pytestmark = pytest.mark.gpu

# (B, H, W, C, layers): the cfg2 stages (LiDAR stages 3-5 at batch 2; camera layers 1-4), odd sizes, a one-layer chain
SHAPES = [
    (2, 44, 50, 256, 21),
    (2, 88, 100, 192, 21),
    (2, 176, 200, 128, 13),
    (2, 94, 311, 64, 4),
    (2, 47, 156, 128, 3),
    (2, 24, 78, 256, 3),
    (2, 12, 39, 512, 3),
    (1, 9, 13, 64, 5),
    (3, 17, 23, 128, 6),
    (1, 31, 7, 192, 2),
    (2, 44, 50, 256, 1),
]


@pytest.fixture(autouse=True)
def _wide_chains():
    """The 512-channel shapes are outside the automatic policy (conv_chain.hip: more than four channel tiles); the hand-off is
    tested on them all the same."""
    H = pkg("_hip")
    H.set_option("CHAIN_WIDE", 1)
    yield
    H.set_option("CHAIN_WIDE", None)


def _weights(C, n, dtype, seed):
    return [to_dev(q(rnd((C, C, 3, 3), seed + i, -0.05, 0.05), dtype), dtype).contiguous() for i in range(n)]    # [Cout][kh][kw][Cin]


def _forward_spec(n, ext_res):
    """Layers of a residual stage starting at block 0's conv2: (res, relu) per layer; res = 'ext' | int | None."""
    spec = []
    for l in range(n):
        if l % 2 == 0:                               # conv2 of a block: += shortcut
            spec.append(("ext" if l == 0 else l - 2, True))
        else:                                        # conv1 of the next block
            spec.append((None, True))
    return spec


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("shape", SHAPES)
def test_chain_forward_equals_separate_launches(shape, dtype):
    ops, H = pkg("ops"), pkg("_hip")
    B, Hh, W, C, n = shape
    if dtype == 2 and n > 6:
        n = 6                                        # (fp16: the short chains only; same kernel template)
    assert ops.conv3x3_chain_supported(dtype, B, Hh, W, C, n)
    ws = ops.conv3x3_chain_workspace(dtype, B, Hh, W, C, n, "cuda")
    wts = _weights(C, n, dtype, 100)
    shifts = [rnd((C,), 300 + i, -0.1, 0.1).cuda() for i in range(n)]
    spec = _forward_spec(n, True)
    for rep in range(3):
        x = to_dev(q(rnd((B, C, Hh, W), 7 + rep), dtype), dtype)
        ext = to_dev(q(rnd((B, C, Hh, W), 17 + rep), dtype), dtype)
        # separate launches (the row-sharing kernel: CONV_LC off so that both sides run the same function)
        H.set_option("CONV_LC", 0)
        try:
            want, cur = [], x
            for l, (r, relu) in enumerate(spec):
                rr = ext if r == "ext" else (None if r is None else want[r])
                cur = ops.conv2d_fwd(dtype, cur, wts[l], shifts[l], rr, 3, 3, 1, 1, relu, C)
                want.append(cur)
        finally:
            H.set_option("CONV_LC", None)
        layers = [(wts[l], shifts[l], ext if r == "ext" else r, None, relu) for l, (r, relu) in enumerate(spec)]
        got = ops.conv3x3_chain(dtype, x, layers, 0, ws)
        assert ops.conv3x3_chain_status(ws) == 0, "a workgroup gave up waiting"
        for l in range(n):
            assert torch.equal(got[l].view(torch.int16), want[l].view(torch.int16)), "layer %d of %d differs (rep %d)" % (l, n, rep)
        assert int(ws.abs().sum().item()) == 0, "the launch must leave its counters zero"
        if rep == 0:
            ref = torch.relu(F.conv2d(from_dev(x), from_dev(wts[0]), None, 1, 1) + shifts[0].cpu().view(1, -1, 1, 1) + from_dev(ext))
            assert rel_err(from_dev(got[0]), ref) < (1.2e-2 if dtype == 1 else 2e-3)
        # recycle: drop the outputs so that the next repetition's tensors land on the same addresses with other contents
        del got, want, cur


@pytest.mark.parametrize("shape", SHAPES[:7] + SHAPES[8:9])
def test_chain_dgrad_equals_separate_launches(shape):
    """The input-gradient list of a stage, last block first: dgrad(conv2) masked by y1, dgrad(conv1) + g2 masked by x
    (engine.Block.backward)."""
    ops, H = pkg("ops"), pkg("_hip")
    dtype = 1
    B, Hh, W, C, n = shape
    ws = ops.conv3x3_chain_workspace(dtype, B, Hh, W, C, n, "cuda")
    wts = [w.permute(3, 1, 2, 0).contiguous() for w in _weights(C, n, dtype, 500)]          # [Cin][kh][kw][Cout]
    masks = [to_dev(q(rnd((B, C, Hh, W), 700 + i), dtype), dtype) for i in range(n)]
    for rep in range(2):
        g = to_dev(q(rnd((B, C, Hh, W), 27 + rep), dtype), dtype)
        H.set_option("CONV_LC", 0)
        try:
            want, cur = [], g
            for l in range(n):
                r = None if l % 2 == 0 else (g if l == 1 else want[l - 2])
                cur = ops.conv2d_dgrad(dtype, cur, wts[l], r, (B, Hh, W, C), 3, 3, 1, 1, masks[l] if l != n - 1 else None)
                want.append(cur)
        finally:
            H.set_option("CONV_LC", None)
        layers = [(wts[l], None, None if l % 2 == 0 else (g if l == 1 else l - 2), masks[l] if l != n - 1 else None, False) for l in range(n)]
        got = ops.conv3x3_chain(dtype, g, layers, 1, ws)
        assert ops.conv3x3_chain_status(ws) == 0
        for l in range(n):
            assert torch.equal(got[l].view(torch.int16), want[l].view(torch.int16)), "layer %d of %d differs (rep %d)" % (l, n, rep)
        del got, want, cur


def test_chain_hand_off_under_load_with_warm_l1():
    """Guideline 16's test conditions: the consumer's caches hold OLD copies of the handed-off lines (the buffers are read by
    another kernel right before the launch and re-used with new contents), and a second stream keeps the CUs unevenly busy
    while the chain runs.  Every word is compared, many times."""
    ops, H = pkg("ops"), pkg("_hip")
    dtype, (B, Hh, W, C, n) = 1, (2, 44, 50, 256, 21)
    ws = ops.conv3x3_chain_workspace(dtype, B, Hh, W, C, n, "cuda")
    wts = _weights(C, n, dtype, 900)
    spec = _forward_spec(n, True)
    side = torch.cuda.Stream()
    a = torch.randn((2048, 2048), device="cuda")
    bad = 0
    for rep in range(12):
        x = to_dev(q(rnd((B, C, Hh, W), 40 + rep), dtype), dtype)
        ext = to_dev(q(rnd((B, C, Hh, W), 60 + rep), dtype), dtype)
        H.set_option("CONV_LC", 0)
        try:
            want, cur = [], x
            for l, (r, relu) in enumerate(spec):
                cur = ops.conv2d_fwd(dtype, cur, wts[l], None, ext if r == "ext" else (None if r is None else want[r]), 3, 3, 1, 1, relu, C)
                want.append(cur)
        finally:
            H.set_option("CONV_LC", None)
        want = [w.clone() for w in want]
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 3):
                a = (a @ a).clamp_(-1, 1)                     # uneven load beside the chain
        layers = [(wts[l], None, ext if r == "ext" else r, None, relu) for l, (r, relu) in enumerate(spec)]
        got = ops.conv3x3_chain(dtype, x, layers, 0, ws)
        assert ops.conv3x3_chain_status(ws) == 0
        bad += sum(int((g.view(torch.int16) != w.view(torch.int16)).sum().item()) for g, w in zip(got, want))
        # warm the caches with THIS repetition's outputs; the next repetition re-uses the addresses with other contents
        s = sum(float(g.float().sum().item()) for g in got)
        assert s == s
        del got
    torch.cuda.synchronize()
    assert bad == 0, "%d stale / wrong words" % bad


def test_chain_rejects_what_it_cannot_run():
    ops, H = pkg("ops"), pkg("_hip")
    assert not ops.conv3x3_chain_supported(0, 2, 44, 50, 256, 4)            # fp32
    assert not ops.conv3x3_chain_supported(1, 2, 44, 50, 96, 4)             # channels not a multiple of 64
    assert not ops.conv3x3_chain_supported(1, 8, 176, 200, 128, 4)          # more than one round of workgroups
    assert not ops.conv3x3_chain_supported(1, 2, 44, 50, 256, H.CHAIN_MAX_LAYERS + 1)
    H.set_option("CHAIN_WIDE", None)
    assert not ops.conv3x3_chain_supported(1, 2, 12, 39, 512, 3)           # eight channel tiles per position tile: not automatic
    H.set_option("CHAIN_WIDE", 1)
    assert ops.conv3x3_chain_supported(1, 2, 12, 39, 512, 3)
    x = torch.zeros((2, 44, 50, 256), device="cuda", dtype=torch.bfloat16)
    w = torch.zeros((256, 3, 3, 256), device="cuda", dtype=torch.bfloat16)
    ws = ops.conv3x3_chain_workspace(1, 2, 44, 50, 256, 2, "cuda")
    with pytest.raises(H.DcfError):
        ops.conv3x3_chain(1, x, [(w, None, 1, None, True), (w, None, None, None, True)], 0, ws)       # layer 0's residual is layer 1's output


def test_give_up_is_reported_one_step_later():
    """HipBackend.check_chains: a chain workgroup that gives up leaves a record in its workspace; the backend copies the records out
    asynchronously once per step and raises when the previous step's copy shows one."""
    import copy
    import yaml, os
    from _util import ROOT, PKG
    cfg = yaml.safe_load(open(os.path.join(ROOT, PKG, "config", "config_carla.yaml")))
    cfg.update(dict(voxel_length=64, voxel_width=64, dtype="bf16", conv_chain=True))
    cfg["lidar_module"] = dict(out_feature1=32, out_feature2=64, out_feature3=128, out_feature4=192, out_feature5=256,
                               num_res_block1=1, num_res_block2=2, num_res_block3=2, num_res_block4=2, num_res_block5=2)
    net = pkg("model").ObjectDetection_DCF(cfg).cuda()
    pkg("detfill").fill_state_dict(net)
    x = torch.rand((1, 32, 64, 64), device="cuda")
    img = torch.zeros((1, 3, 8, 8), dtype=torch.uint8, device="cuda")
    with torch.no_grad():
        net(x, img)
        K = net._backend
        assert K._chain_ws, "the stages of this model should run as chain launches"
        net(x, img); torch.cuda.synchronize(); net(x, img)          # clean steps: nothing raised
        next(iter(K._chain_ws.values()))[1] = 0x40000000 | (3 << 16) | 5      # what a workgroup that gave up would leave
        torch.cuda.synchronize()
        K._chain_pending = None
        net(x, img)                                                 # copies the record out
        torch.cuda.synchronize()
        with pytest.raises(pkg("_hip").DcfError) as e:
            net(x, img)
        assert "layer 3" in str(e.value) and "tile 5" in str(e.value)


def test_chain_launches_replay_inside_captured_graphs():
    """The opt-in chain launches inside the captured step (config hip_graphs + conv_chain): a chain launch resets its own arrival
    counters (the workgroup that finishes last), so a graph replay finds them zero like a first launch -- same prediction and
    gradients, bit for bit, as the eager chained step and as the eager per-layer step (LiDAR stream only: nothing order-dependent)."""
    import yaml, os
    from _util import ROOT, PKG
    base = yaml.safe_load(open(os.path.join(ROOT, PKG, "config", "config_carla.yaml")))
    base.update(dict(voxel_length=64, voxel_width=64, dtype="bf16"))
    base["lidar_module"] = dict(out_feature1=32, out_feature2=64, out_feature3=128, out_feature4=192, out_feature5=256,
                                num_res_block1=1, num_res_block2=2, num_res_block3=3, num_res_block4=2, num_res_block5=2)
    det = pkg("detfill")
    x = torch.from_numpy(det.uniform((1, 32, 64, 64), 21, 0.0, 1.0)).cuda()
    img = torch.zeros((1, 3, 8, 8), dtype=torch.uint8, device="cuda")
    R = torch.from_numpy(det.uniform((1, 32, 16, 16), 22, -1.0, 1.0)).cuda()
    outs = {}
    for tag, chain, graphs in (("plain", False, False), ("chain", True, False), ("chain+graphs", True, True)):
        cfg = dict(base, conv_chain=chain, hip_graphs=graphs)
        net = pkg("model").ObjectDetection_DCF(cfg).cuda()
        det.fill_state_dict(net)
        for rep in range(4):                       # (graphs: one capture step, then replays)
            pred = net(x * (1.0 + 0.1 * rep), img)
            (pred * R).sum().backward()
        torch.cuda.synchronize()
        if chain:
            assert net._backend._chain_ws and all(int(ws[1].item()) == 0 for ws in net._backend._chain_ws.values())
        outs[tag] = (pred.detach().clone(), net.flat_grads.clone())
    for tag in ("chain", "chain+graphs"):
        assert torch.equal(outs[tag][0], outs["plain"][0]), tag
        assert torch.equal(outs[tag][1], outs["plain"][1]), tag
