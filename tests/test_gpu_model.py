"""GPU parity of the whole LiDAR stream (model surface -> engine -> HIP kernels) against the
golden vectors generated from the imported reference, forward and backward, plus the
train step (loss + Adam) trajectory.  Tolerance from BASELINE.json north_star: 1e-3 relative
(fp32 path); the bf16 path is checked against the same vectors at bf16 precision."""
import copy

import numpy as np
import pytest
import torch

from _util import golden_cfg, load_golden, pkg

pytestmark = pytest.mark.gpu


def tiny_input():
    det = pkg("detfill")
    u = det.uniform((2, 32, 64, 32), 4242, 0.0, 1.0)
    m = det.uniform((2, 32, 64, 32), 4242 + 17, 0.0, 1.0) < 0.12
    return torch.from_numpy((u * m).astype(np.float32))


def build(cfg, dtype="f32", **over):
    cfg = copy.deepcopy(cfg)
    cfg["dtype"] = dtype
    cfg.update(over)
    net = pkg("model").ObjectDetection_DCF(cfg)
    pkg("detfill").fill_state_dict(net)
    return net.cuda(), cfg


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-3), ("bf16", 6e-2), ("f16", 8e-3)])
def test_tiny_forward_matches_reference(dtype, tol):
    z = load_golden("model_tiny.npz")
    net, cfg = build(golden_cfg(z), dtype)
    net.eval()
    with torch.no_grad():
        pred = net(tiny_input().cuda(), torch.zeros(2, 3, 8, 8, dtype=torch.uint8, device="cuda")).cpu().numpy()
    ref = z["pred_eval"]
    assert pred.shape == ref.shape == (2, 32, 16, 8)
    for name, sl in (("cls", slice(0, 4)), ("reg", slice(4, 18)), ("bbox", slice(18, 32))):
        err = np.abs(pred[:, sl] - ref[:, sl]).max() / np.abs(ref[:, sl]).max()
        assert err < tol, "%s %s: rel err %g" % (dtype, name, err)


def test_state_dict_surface_matches_reference():
    from oracle import model_ref
    z = load_golden("model_tiny.npz")
    cfg = golden_cfg(z)
    net = pkg("model").ObjectDetection_DCF(cfg)
    want = model_ref.lidar_state_shapes(cfg)
    got = net.state_dict()
    assert list(got.keys()) == list(want.keys())
    assert all(tuple(got[k].shape) == tuple(want[k]) for k in want)
    # 'module.'-prefixed (DDP) checkpoints load too (train.py:79)
    net.load_state_dict({"module." + k: v.clone() for k, v in got.items()})


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-3), ("bf16", 2.5e-1), ("f16", 1e-1)])
def test_tiny_backward_matches_reference(dtype, tol):
    z = load_golden("model_tiny.npz")
    net, cfg = build(golden_cfg(z), dtype)
    det = pkg("detfill")
    R = torch.from_numpy(det.uniform((1, 32, 16, 8), 777, -1.0, 1.0)).cuda()
    pred = net(tiny_input()[:1].cuda(), torch.zeros(1, 3, 8, 8, dtype=torch.uint8, device="cuda"))
    (pred * R).sum().backward()
    named = dict(net.named_parameters())
    worst = 0.0
    for k in [str(s) for s in z["grad_keys"]]:
        ref = z["g_eval_" + k]
        got = named[k].grad.detach().cpu().numpy()
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        worst = max(worst, err)
        assert err < tol, "%s grad of %s: rel err %g" % (dtype, k, err)
    # every parameter: sum |g| checksum in the reference's named_parameters order
    gabs = np.array([named[k].grad.abs().sum().item() for k in named])
    ref_abs = z["gabs_eval"]
    assert gabs.shape == ref_abs.shape
    rel = np.abs(gabs - ref_abs) / (ref_abs + 1e-6)
    assert rel.max() < (5e-3 if dtype == "f32" else 2e-1), "checksum rel err %g at %d" % (rel.max(), int(rel.argmax()))


def test_full_carla_frame_cfg1():
    """BASELINE configs[0]: synthetic CARLA frame (10k pts) -> HIP voxeliser -> HIP model, B=1, fp32,
    against 64 output pixels and channel sums of the reference's CPU forward."""
    g = load_golden("geometry_carla.npz")
    z = load_golden("model_carla_full.npz")
    cfg = golden_cfg(g)
    net, cfg = build(cfg, "f32")
    geo = pkg("data_import_carla").FrameGeometry(cfg, g["crt"])
    voxel, pc, uv, cnt, _ = geo(torch.from_numpy(g["n10k_pts"]))
    with torch.no_grad():
        pred = net(voxel.unsqueeze(0), torch.zeros(1, 3, 240, 320, dtype=torch.uint8, device="cuda")).cpu().numpy()[0]
    got = pred[:, z["sample_h"], z["sample_w"]]
    err = np.abs(got - z["sample_pred"]).max() / np.abs(z["sample_pred"]).max()
    assert err < 1e-3, err
    assert np.allclose(pred.astype(np.float64).sum((1, 2)), z["chan_sum"], rtol=2e-3, atol=0.5)


def test_train_step_trajectory_matches_reference():
    """3 steps of Train.one_step semantics (eval-BN, Adam lr 1e-4, np.random.seed(100+step)), B=1, fp32."""
    z = load_golden("model_tiny.npz")
    traj = load_golden("adam_traj.npz")
    lz = load_golden("loss.npz")
    net, cfg = build(golden_cfg(z), "f32")
    train = pkg("train")
    L = pkg("loss").LossTotal(cfg)
    opt = train.FlatAdam(net, cfg["learning_rate"], (cfg["beta1"], 0.999))
    x = tiny_input()[:1].cuda()
    img = torch.zeros(1, 3, 8, 8, dtype=torch.uint8, device="cuda")
    boxes, nb = torch.from_numpy(lz["bboxes"])[:1].cuda(), torch.from_numpy(lz["nbox"])[:1]
    losses = []
    for step in range(3):
        pred = net(x, img)
        cls, reg, _ = torch.split(pred, [4, 14, 14], dim=1)
        np.random.seed(100 + step)
        val = L(boxes, nb, cls, reg)
        val.backward()
        opt.step()
        losses.append(val.item())
    ref = traj["losses"]
    assert np.abs(np.array(losses) - ref).max() < 2e-3 * abs(ref[0]), (losses, ref.tolist())
    w = dict(net.named_parameters())["lidar_backbone.conv3.weight"].detach().cpu().numpy()[:4, :4]
    assert np.abs(w - traj["conv3_after"]).max() < 2e-5


def test_cpu_tensors_fail_loudly():
    z = load_golden("model_tiny.npz")
    cfg = golden_cfg(z)
    net = pkg("model").ObjectDetection_DCF(cfg)
    with pytest.raises(Exception) as e:
        net(tiny_input(), torch.zeros(2, 3, 8, 8, dtype=torch.uint8))
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_train_mode_batchnorm_forward_backward():
    """bn_mode=train (batch statistics, what a literal .train() reference module does): forward (B=2) and
    backward (B=1) against the train-mode golden vectors of the imported reference, fp32 path."""
    z = load_golden("model_tiny.npz")
    net, cfg = build(golden_cfg(z), "f32", bn_mode="train")
    img = torch.zeros(2, 3, 8, 8, dtype=torch.uint8, device="cuda")
    with torch.no_grad():
        pred = net(tiny_input().cuda(), img).cpu().numpy()
    ref = z["pred_train"]
    for name, sl in (("cls", slice(0, 4)), ("reg", slice(4, 18)), ("bbox", slice(18, 32))):
        err = np.abs(pred[:, sl] - ref[:, sl]).max() / np.abs(ref[:, sl]).max()
        assert err < 1e-3, "train-BN %s: rel err %g" % (name, err)
    # running statistics moved (momentum 0.1) and the batch counter advanced
    sd = net.state_dict()
    k = "lidar_backbone.backbone.layer1.sequential.resblock_0.bn1."
    assert int(sd[k + "num_batches_tracked"]) == 1
    fresh, _ = build(golden_cfg(z), "f32", bn_mode="train")
    assert not torch.equal(sd[k + "running_mean"], fresh.state_dict()[k + "running_mean"])
    # backward, B=1
    net, cfg = build(golden_cfg(z), "f32", bn_mode="train")
    R = torch.from_numpy(pkg("detfill").uniform((1, 32, 16, 8), 777, -1.0, 1.0)).cuda()
    out = net(tiny_input()[:1].cuda(), img[:1])
    (out * R).sum().backward()
    named = dict(net.named_parameters())
    for k in [str(s) for s in z["grad_keys"]]:
        ref = z["g_train_" + k]
        got = named[k].grad.detach().cpu().numpy()
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err < 5e-3, "train-BN grad of %s: rel err %g" % (k, err)
    gabs = np.array([named[k].grad.abs().sum().item() for k in named])
    rel = np.abs(gabs - z["gabs_train"]) / (z["gabs_train"] + 1e-6)
    assert rel.max() < 1e-2, "checksum rel err %g at %d" % (rel.max(), int(rel.argmax()))


def test_module_mode_follows_training_flag():
    z = load_golden("model_tiny.npz")
    net, cfg = build(golden_cfg(z), "f32", bn_mode="module")
    img = torch.zeros(2, 3, 8, 8, dtype=torch.uint8, device="cuda")
    x = tiny_input().cuda()
    with torch.no_grad():
        net.eval()
        a = net(x, img).cpu().numpy()
        net.train()
        b = net(x, img).cpu().numpy()
    assert np.abs(a - z["pred_eval"]).max() / np.abs(z["pred_eval"]).max() < 1e-3
    assert np.abs(b - z["pred_train"]).max() / np.abs(z["pred_train"]).max() < 1e-3


def test_hip_graph_replay_matches_eager():
    """config hip_graphs: forward/backward captured once and replayed -- same prediction and gradients as eager."""
    z = load_golden("model_tiny.npz")
    x = tiny_input()[:1].cuda()
    img = torch.zeros(1, 3, 8, 8, dtype=torch.uint8, device="cuda")
    R = torch.from_numpy(pkg("detfill").uniform((1, 32, 16, 8), 777, -1.0, 1.0)).cuda()
    outs = []
    for graphs in (False, True):
        net, cfg = build(golden_cfg(z), "f32", hip_graphs=graphs)
        for rep in range(3):                       # replays after the capture step
            pred = net(x * (1.0 + 0.1 * rep), img)
            (pred * R).sum().backward()
        outs.append((pred.detach().clone(), net.flat_grads.clone()))
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-6)
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-6)


def test_checkpoint_resume_is_exact(tmp_path):
    """save after 2 steps, resume in a fresh Train, take step 3: identical parameters to an uninterrupted run."""
    z = load_golden("model_tiny.npz")
    lz = load_golden("loss.npz")
    cfg = golden_cfg(z)
    cfg["dtype"] = "f32"
    T = pkg("train")
    x = tiny_input()[:1].cuda()
    img = torch.zeros(1, 3, 8, 8, dtype=torch.uint8, device="cuda")
    boxes, nb = torch.from_numpy(lz["bboxes"])[:1], torch.from_numpy(lz["nbox"])[:1]

    def run(trainer, steps, first):
        for s in range(first, first + steps):
            np.random.seed(100 + s)
            trainer.one_step(x, img, boxes, nb)

    a = T.Train(cfg)
    pkg("detfill").fill_state_dict(a.model)
    run(a, 2, 0)
    a.save_checkpoint(str(tmp_path / "ck.pt"), epoch=7)
    run(a, 1, 2)
    b = T.Train(cfg)
    assert b.load_checkpoint(str(tmp_path / "ck.pt")) == 7
    run(b, 1, 2)
    assert torch.equal(a.model.flat_params, b.model.flat_params)
    assert b.optimizer.step_count == 3


@pytest.mark.parametrize("reduction", ["last", "sum", "mean"])
def test_fused_loss_kernel_matches_torch_path(reduction):
    """dcf_loss_fwd_bwd (one launch, CUDA tensors) against the same LossTotal on CPU tensors (torch ops; that path is
    pinned to the reference by tests/golden/loss.npz in the CPU suite): loss value and both gradients."""
    z = load_golden("loss.npz")
    cfg = golden_cfg(load_golden("model_tiny.npz"))
    cfg["loss_reduction"] = reduction
    L = pkg("loss").LossTotal(cfg)
    boxes, nb = torch.from_numpy(z["bboxes"]), torch.from_numpy(z["nbox"])
    for seed in (0, 1):
        c0 = torch.from_numpy(z["cls"]).clone().requires_grad_(True)
        r0 = torch.from_numpy(z["reg"]).clone().requires_grad_(True)
        np.random.seed(seed)
        ref = L(boxes, nb, c0, r0)
        ref.backward()
        # separate tensors
        c1 = torch.from_numpy(z["cls"]).cuda().requires_grad_(True)
        r1 = torch.from_numpy(z["reg"]).cuda().requires_grad_(True)
        np.random.seed(seed)
        got = L(boxes, nb, c1, r1)
        got.backward()
        assert abs(got.item() - ref.item()) < 2e-6 * max(1.0, abs(ref.item()))
        if reduction == "last":                       # the reference's own numbers (gen_golden.py)
            assert abs(got.item() - float(z["loss_seed%d" % seed])) < 2e-6
            assert np.abs(c1.grad.cpu().numpy() - z["gcls_seed%d" % seed]).max() < 2e-7
            assert np.abs(r1.grad.cpu().numpy() - z["greg_seed%d" % seed]).max() < 2e-7
        assert torch.allclose(c1.grad.cpu(), c0.grad, rtol=1e-5, atol=1e-7)
        assert torch.allclose(r1.grad.cpu(), r0.grad, rtol=1e-5, atol=1e-7)
        # views of one [B,32,h,w] head tensor (what the model hands over): the gradient goes straight to the base
        B, _, h, w = z["cls"].shape
        base = torch.zeros(B, 32, h, w)
        base[:, 0:4] = torch.from_numpy(z["cls"])
        base[:, 4:18] = torch.from_numpy(z["reg"])
        base = base.cuda().requires_grad_(True)
        np.random.seed(seed)
        got2 = L(boxes, nb, base[:, 0:4], base[:, 4:18])
        got2.backward()
        assert abs(got2.item() - ref.item()) < 2e-6 * max(1.0, abs(ref.item()))
        assert torch.allclose(base.grad[:, 0:4].cpu(), c0.grad, rtol=1e-5, atol=1e-7)
        assert torch.allclose(base.grad[:, 4:18].cpu(), r0.grad, rtol=1e-5, atol=1e-7)
        assert float(base.grad[:, 18:].abs().max()) == 0.0


def test_eval_harness_one_step():
    """test.py surface: Test(net, cfg).get_eval_value_onestep -> loss, score-thresholded boxes, SAT suppression and the
    precision / recall counters (host post-processing, pinned separately by tests/golden/eval.npz)."""
    z = load_golden("model_tiny.npz")
    lz = load_golden("loss.npz")
    net, cfg = build(golden_cfg(z), "f32")
    cfg["score_threshold"] = 0.5
    T = pkg("test").Test(net, cfg)
    x = tiny_input().cuda()
    img = torch.zeros(x.shape[0], 3, 8, 8, dtype=torch.uint8, device="cuda")
    boxes, nb = torch.from_numpy(lz["bboxes"]), torch.from_numpy(lz["nbox"])
    np.random.seed(3)
    loss, sel = T.get_eval_value_onestep(x, img, boxes, nb)
    assert np.isfinite(loss) and len(sel) == x.shape[0]
    assert T.get_num_T() == int((boxes[..., -1] == 1).sum())
    assert T.get_num_P() == sum(len(k) for k in T.refined_bbox) <= sum(b.shape[0] for b in sel)
    assert all(0 <= T.get_num_TP_set()[t] <= T.get_num_P() for t in T.IOU_threshold)
