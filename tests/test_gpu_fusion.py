"""GPU parity of the continuous-fusion path (image stream + KNN + gather + MLP), which the reference
leaves as a TODO (model.py:199-203): checked against this repo's literal CPU statement of SURVEY.md
App. D (oracle/model_ref.py) -- "parity unpinned" by the reference, pinned by the oracle."""
import copy

import numpy as np
import pytest
import torch

from _util import golden_cfg, load_golden, pkg
from oracle import geometry_ref, model_ref

pytestmark = pytest.mark.gpu


def small_crt():
    K = np.array([[60.0, 0.0, 64.0], [0.0, 60.0, 48.0], [0.0, 0.0, 1.0]])
    return pkg("calib").crt_from(K, pkg("calib").R_LIDAR_TO_CAM)


def setup(dtype="f32", K=3, zero_last=False, arch="resnet18"):
    cfg = golden_cfg(load_golden("model_tiny.npz"))
    cfg.update(dict(image_height=96, image_width=128, max_num_pc=2048, projection_mode="correct", dtype=dtype))
    cfg["fusion"] = dict(enabled=True, K=K, r_max=None, image_channels=64, image_stream=arch, zero_init_last=zero_last)
    det = pkg("detfill")
    lim6 = (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"], cfg["lidar_z_min"], cfg["lidar_z_max"])
    B = 2
    pts = [det.synthetic_points(1500, lim6, 50 + b) for b in range(B)]
    pts[1][:, 0] = pts[1][:, 0] * 0.5 + 1.0          # a different density for the second frame
    img = torch.stack([torch.from_numpy(det.synthetic_image(96, 128, 9 + b)) for b in range(B)], 0)
    return cfg, pts, img, small_crt()


def oracle_inputs(cfg, pts, crt):
    grids, pcs, uvs, ns = [], [], [], []
    for p in pts:
        g, pc, uv, n, _ = geometry_ref.voxelization_projection(p, cfg, crt, proj_mode="correct")
        grids.append(g); pcs.append(pc); uvs.append(uv); ns.append(n)
    return (torch.from_numpy(np.stack(grids)), torch.from_numpy(np.stack(pcs)), torch.from_numpy(np.stack(uvs)), ns)


def full_shapes(cfg):
    s = {}
    s.update(model_ref.lidar_state_shapes(cfg))
    s.update(model_ref.image_state_shapes(64, arch=cfg["fusion"].get("image_stream", "resnet18")))
    s.update(model_ref.fusion_state_shapes(cfg, 64))
    return s


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_image_stream_feature_map(arch):
    """ResNet-18 / ResNet-50 trunk + FPN (stem7x7, max-pool, Basic / Bottleneck blocks, resize-add, smooth) against the
    CPU statement."""
    cfg, pts, img, crt = setup("f32", arch=arch)
    net = pkg("model").ObjectDetection_DCF(cfg)
    pkg("detfill").fill_state_dict(net)
    net = net.cuda()
    K = net._ensure_backend(torch.device("cuda", 0))
    K.prepare()
    fmap = net._plan._image_forward(K, img.cuda(), False).float().cpu().permute(0, 3, 1, 2)
    sd = model_ref.make_state_dict(full_shapes(cfg))
    ref = model_ref.image_stream(sd, img, "eval")
    assert fmap.shape == ref.shape
    err = float((fmap - ref).abs().max() / ref.abs().max())
    assert err < 1e-3, err


@pytest.mark.parametrize("dtype,tol,K", [("f32", 1e-3, 3), ("bf16", 6e-2, 3), ("f16", 8e-3, 5), ("f32", 1e-3, 5), ("f32", 1e-3, 1)])
def test_fused_forward(dtype, tol, K):
    cfg, pts, img, crt = setup(dtype, K=K)
    net = pkg("model").ObjectDetection_DCF(cfg)
    pkg("detfill").fill_state_dict(net)
    net = net.cuda().eval()
    geo = pkg("data_import_carla").FrameGeometry(cfg, crt)
    vox, pcs, uvs, cnts = [], [], [], []
    for p in pts:
        v, pc, uv, cnt, _ = geo(torch.from_numpy(p))
        vox.append(v); pcs.append(pc); uvs.append(uv); cnts.append(cnt)
    with torch.no_grad():
        pred = net(torch.stack(vox), img.cuda(), points=torch.stack(pcs), uv=torch.stack(uvs), n_valid=torch.cat(cnts)).cpu()
    x, pc, uv, ns = oracle_inputs(cfg, pts, crt)
    assert [int(c.item()) for c in cnts] == ns and min(ns) > 50
    assert torch.equal(torch.stack(vox).cpu(), x)
    sd = model_ref.make_state_dict(full_shapes(cfg))
    g = geometry_ref.grid_constants(cfg)
    with torch.no_grad():
        ref = model_ref.forward(sd, cfg, x, img, pc, uv, ns, "eval", fusion={"K": cfg["fusion"]["K"], "aff": g["aff"], "rmax": None})
        nofuse = model_ref.forward(sd, cfg, x, bn_mode="eval")
    assert float((ref - nofuse).abs().max()) > 1e-2          # the fusion branch really contributes
    for name, sl in (("cls", slice(0, 4)), ("reg", slice(4, 18)), ("bbox", slice(18, 32))):
        err = float((pred[:, sl] - ref[:, sl]).abs().max() / ref[:, sl].abs().max())
        assert err < tol, "%s %s rel err %g" % (dtype, name, err)


def test_zero_init_last_reproduces_lidar_only_reference():
    """SURVEY.md App. D bridge: with fc2 = 0 the fused model equals the reference's LiDAR-only forward (golden)."""
    z = load_golden("model_tiny.npz")
    cfg, pts, img, crt = setup("f32")
    net = pkg("model").ObjectDetection_DCF(cfg)
    pkg("detfill").fill_state_dict(net)
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "fusion" in k and ".fc2." in k:
                p.zero_()
    net = net.cuda().eval()
    det = pkg("detfill")
    u = det.uniform((2, 32, 64, 32), 4242, 0.0, 1.0)
    m = det.uniform((2, 32, 64, 32), 4242 + 17, 0.0, 1.0) < 0.12
    x = torch.from_numpy((u * m).astype(np.float32)).cuda()
    geo = pkg("data_import_carla").FrameGeometry(cfg, crt)
    pcs, uvs, cnts = [], [], []
    for p in pts:
        _, pc, uv, cnt, _ = geo(torch.from_numpy(p))
        pcs.append(pc); uvs.append(uv); cnts.append(cnt)
    with torch.no_grad():
        pred = net(x, img.cuda(), points=torch.stack(pcs), uv=torch.stack(uvs), n_valid=torch.cat(cnts)).cpu().numpy()
    ref = z["pred_eval"]
    assert np.abs(pred - ref).max() / np.abs(ref).max() < 1e-3


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_fused_backward_all_parameters(arch):
    """d<pred,R>/d(every parameter) of the fused model (fp32 path) against autograd through the CPU statement."""
    cfg, pts, img, crt = setup("f32", arch=arch)
    net = pkg("model").ObjectDetection_DCF(cfg)
    det = pkg("detfill")
    det.fill_state_dict(net)
    net = net.cuda()
    geo = pkg("data_import_carla").FrameGeometry(cfg, crt)
    vox, pcs, uvs, cnts = [], [], [], []
    for p in pts:
        v, pc, uv, cnt, _ = geo(torch.from_numpy(p))
        vox.append(v); pcs.append(pc); uvs.append(uv); cnts.append(cnt)
    R = torch.from_numpy(det.uniform((2, 32, 16, 8), 778, -1.0, 1.0))
    pred = net(torch.stack(vox), img.cuda(), points=torch.stack(pcs), uv=torch.stack(uvs), n_valid=torch.cat(cnts))
    (pred * R.cuda()).sum().backward()
    x, pc, uv, ns = oracle_inputs(cfg, pts, crt)
    sd = model_ref.make_state_dict(full_shapes(cfg))
    params = {k: (v.clone().requires_grad_(True) if (v.dtype.is_floating_point and "running" not in k) else v) for k, v in sd.items()}
    g = geometry_ref.grid_constants(cfg)
    out = model_ref.forward(params, cfg, x, img, pc, uv, ns, "eval", fusion={"K": 3, "aff": g["aff"], "rmax": None})
    (out * R).sum().backward()
    named = dict(net.named_parameters())
    bad = []
    for k, p in named.items():
        ref = params[k].grad
        got = p.grad.detach().cpu()
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max()) / (scale + 1e-9)
        if err > 5e-3 and float((got - ref).abs().max()) > 1e-5:
            bad.append((k, err, scale))
    assert not bad, "%d parameters off, worst %s" % (len(bad), sorted(bad, key=lambda t: -t[1])[:5])


def test_side_stream_geometry_matches_inline():
    """train.Train.geometry_async (voxelise / project / KNN on a side stream with events) gives the same
    prediction as the inline path."""
    cfg, pts, img, crt = setup("f32")
    T = pkg("train")
    trainer = T.Train(cfg)
    pkg("detfill").fill_state_dict(trainer.model)
    geo = pkg("data_import_carla").FrameGeometry(cfg, crt)
    dev_pts = [torch.from_numpy(p).cuda() for p in pts]
    with torch.no_grad():
        x_lidar, geom = trainer.geometry_async(geo, dev_pts)
        a = trainer.model(x_lidar, img.cuda(), geom=geom)
        vox, pcs, uvs, cnts = [], [], [], []
        for p in dev_pts:
            v, pc, uv, cnt, _ = geo(p)
            vox.append(v); pcs.append(pc); uvs.append(uv); cnts.append(cnt)
        b = trainer.model(torch.stack(vox), img.cuda(), points=torch.stack(pcs), uv=torch.stack(uvs), n_valid=torch.cat(cnts))
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # with gradients: the side-stream path sizes the per-point tensors by the valid count (a third of max_num_pc here)
    # and inverts the KNN maps on the side stream; parameters' gradients must agree with the inline path
    R = torch.from_numpy(pkg("detfill").uniform(tuple(a.shape), 780, -1.0, 1.0)).cuda()
    grads = []
    for mode in ("side", "inline"):
        if mode == "side":
            x_lidar, geom = trainer.geometry_async(geo, dev_pts)
            out = trainer.model(x_lidar, img.cuda(), geom=geom)
            assert geom["n_rows"] < geom["xyz"].shape[1] and geom["n_rows"] % 256 == 0
        else:
            out = trainer.model(torch.stack(vox), img.cuda(), points=torch.stack(pcs), uv=torch.stack(uvs), n_valid=torch.cat(cnts))
        (out * R).sum().backward()
        grads.append(trainer.model._gradflat.clone())
    scale = float(grads[1].abs().max())
    assert float((grads[0] - grads[1]).abs().max()) < 2e-5 * scale


@pytest.mark.parametrize("Cb,K,case", [(64, 3, "random"), (128, 5, "random"), (192, 1, "random"), (256, 3, "random"),
                                       (64, 3, "one_point"), (128, 3, "no_points")])
def test_fusion_backward_by_point_matches_pixel_run_kernel(Cb, K, case):
    """dcf_fusion_invert + dcf_fusion_gather_bwd_inv (pairs sorted by point) against dcf_fusion_gather_bwd (pixel runs)
    on the same KNN map: dP, dW1d, db1.  Edge cases: one point owning every pixel (runs far longer than a wave's
    slice), no valid point at all (every index -1)."""
    ops, H = pkg("ops"), pkg("_hip")
    g = torch.Generator().manual_seed(7)
    h, w, stride, n_max = 24, 40, 4, 300
    aff = (10.0, 0.0, 10.0, 400.0)
    n = {"random": 200, "one_point": 1, "no_points": 0}[case]
    xyz = torch.zeros(n_max, 3)
    xyz[:n, 0] = torch.rand(n, generator=g) * (h * stride / aff[0])
    xyz[:n, 1] = torch.rand(n, generator=g) * (w * stride / aff[2]) - aff[3] / aff[2]
    xyz[:n, 2] = torch.rand(n, generator=g) * 2 - 1
    xyz = xyz.cuda()
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    idx = ops.knn_bev(xyz, cnt, K, h, w, stride, aff)
    assert int((idx >= 0).sum()) == (min(K, n) * h * w)
    P = (torch.rand(n_max, Cb, generator=g) - 0.5).cuda()
    ghs = (torch.rand(h, w, Cb, generator=g) - 0.5).cuda()
    w1d = ((torch.rand(Cb, 3, generator=g) - 0.5) * 0.2).cuda().reshape(-1)
    b1 = ((torch.rand(Cb, generator=g) - 0.5) * 0.2).cuda()
    for dtype, tol in ((H.F32, 2e-5), (H.BF16, 2e-5)):
        Pd = P.to(H.torch_dtype(dtype))
        gd = ghs.to(H.torch_dtype(dtype))
        ref = [torch.zeros(n_max, Cb, device="cuda"), torch.zeros(Cb * 3, device="cuda"), torch.zeros(Cb, device="cuda")]
        ops.fusion_gather_bwd(dtype, Pd, xyz, idx, stride, aff, w1d, b1, gd, *ref)
        got = [torch.zeros_like(t) for t in ref]
        inv = ops.fusion_invert([idx], n_max)
        start = inv[0].cpu()
        assert int(start[0]) == 0 and int(start[n_max]) == min(K, n) * h * w
        ops.fusion_gather_bwd_inv(dtype, Pd, xyz, inv, n_max, 0, (K, h, w), stride, aff, w1d, b1, gd, *got)
        got_ws = [torch.zeros_like(t) for t in ref]          # ... and with the workspace (last-arriver reduction of dW1d / db1)
        ops.fusion_gather_bwd_inv(dtype, Pd, xyz, inv, n_max, 0, (K, h, w), stride, aff, w1d, b1, gd, *got_ws, ws=ops.fusion_bwd_workspace("cuda", Cb))
        for a, b in zip(got_ws, ref):
            scale = max(float(b.abs().max()), 1e-6)
            assert float((a - b).abs().max()) <= tol * scale * max(1.0, (h * w) ** 0.5), (case, Cb, "workspace")
        for a, b in zip(got, ref):
            scale = max(float(b.abs().max()), 1e-6)
            assert float((a - b).abs().max()) <= tol * scale * max(1.0, (h * w) ** 0.5), (case, Cb, float((a - b).abs().max()), scale)
        # one-writer-per-point flavour (dcf_fusion_gather_bwd_pts): dP comes out whole in the compute dtype -- rows of points
        # nobody chose are written as zeros (the buffer starts as NaN here), on fewer rows than n_max too
        for rows in (n_max, 256):
            gp = torch.full((rows, Cb), float("nan"), device="cuda").to(H.torch_dtype(dtype))
            gw, gb = torch.zeros(Cb * 3, device="cuda"), torch.zeros(Cb, device="cuda")
            ops.fusion_gather_bwd_pts(dtype, Pd[:rows].contiguous(), xyz, inv, n_max, 0, (K, h, w), stride, aff, w1d, b1, gd, gp, gw, gb)
            want = ref[0][:rows].to(H.torch_dtype(dtype)).float() if dtype != H.F32 else ref[0][:rows]
            assert torch.isfinite(gp.float()).all()
            rtol = tol if dtype == H.F32 else 2.0 ** -7
            assert float((gp.float() - want).abs().max()) <= rtol * max(float(want.abs().max()), 1e-6) * max(1.0, (h * w) ** 0.5 if dtype == H.F32 else 1.0)
            for a, b in ((gw, ref[1]), (gb, ref[2])):
                scale = max(float(b.abs().max()), 1e-6)
                assert float((a - b).abs().max()) <= tol * scale * max(1.0, (h * w) ** 0.5)


def test_cfg4_shape_model_step_fp16():
    """BASELINE configs[3] in miniature: ResNet-50 camera stream, K=5, fp16 MFMA, batch 4 -- one fused forward against
    the CPU statement at fp16 precision and one backward + Adam step that changes every parameter."""
    cfg, pts, img, crt = setup("f16", K=5, arch="resnet50")
    det = pkg("detfill")
    pts = pts + [det.synthetic_points(1200, (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"],
                                             cfg["lidar_z_min"], cfg["lidar_z_max"]), 60 + b) for b in range(2)]
    img = torch.cat([img, torch.flip(img, dims=[3])], 0)
    net = pkg("model").ObjectDetection_DCF(cfg)
    det.fill_state_dict(net)
    net = net.cuda()
    geo = pkg("data_import_carla").FrameGeometry(cfg, crt)
    vox, pcs, uvs, cnts = [], [], [], []
    for p in pts:
        v, pc, uv, cnt, _ = geo(torch.from_numpy(p))
        vox.append(v); pcs.append(pc); uvs.append(uv); cnts.append(cnt)
    pred = net(torch.stack(vox), img.cuda(), points=torch.stack(pcs), uv=torch.stack(uvs), n_valid=torch.cat(cnts))
    x, pc, uv, ns = oracle_inputs(cfg, pts, crt)
    sd = model_ref.make_state_dict(full_shapes(cfg))
    g = geometry_ref.grid_constants(cfg)
    ref = model_ref.forward(sd, cfg, x, img, pc, uv, ns, "eval", fusion={"K": 5, "aff": g["aff"], "rmax": None})
    err = float((pred.detach().cpu() - ref).abs().max() / ref.abs().max())
    assert pred.shape[0] == 4 and err < 1e-2, err
    before = net.flat_params.clone()
    opt = pkg("train").FlatAdam(net, 1e-4, (0.9, 0.999))
    R = torch.from_numpy(det.uniform(tuple(pred.shape), 779, -1.0, 1.0)).cuda()
    # fp16 activations need loss scaling (the deep camera stream's gradients underflow otherwise); Adam is scale invariant
    ((pred * R).sum() * 4096.0).backward()
    opt.step()
    assert torch.isfinite(net.flat_params).all()
    # (with the deterministic fill ~55 % of the camera trunk's channels are dead ReLUs -- in the fp32 CPU statement too)
    assert float((net.flat_params != before).float().mean()) > 0.4


def test_cfg5_fp8_forward_path_model_step():
    """BASELINE configs[4] in miniature: dtype "fp8" = e4m3 operands in the forward convolutions with Cin >= fp8_min_cin
    (camera trunk, LiDAR stages, FPN, fusion fc2), bf16 storage and backward.  The fused prediction stays within e4m3
    noise of the bf16 model and of the CPU statement, the delayed activation scales fill in after the first step, and a
    train step through the path updates the parameters."""
    cfg, pts, img, crt = setup("fp8", K=3)
    cfg["fp8_min_cin"], cfg["fp8_min_blocks"] = 64, 1          # every eligible layer, whatever its size
    cfg["lidar_module"].update(out_feature1=32, out_feature2=64, out_feature3=128, out_feature4=192, out_feature5=256,
                               num_res_block1=1, num_res_block2=1, num_res_block3=2, num_res_block4=1, num_res_block5=1)
    det = pkg("detfill")
    geo = pkg("data_import_carla").FrameGeometry(cfg, crt)
    vox, pcs, uvs, cnts = [], [], [], []
    for p in pts:
        v, pc, uv, cnt, _ = geo(torch.from_numpy(p))
        vox.append(v); pcs.append(pc); uvs.append(uv); cnts.append(cnt)
    args = (torch.stack(vox), img.cuda())
    kw = dict(points=torch.stack(pcs), uv=torch.stack(uvs), n_valid=torch.cat(cnts))
    nets = {}
    for dt in ("fp8", "bf16"):
        c = copy.deepcopy(cfg)
        c["dtype"] = dt
        nets[dt] = pkg("model").ObjectDetection_DCF(c)
        det.fill_state_dict(nets[dt])
        nets[dt] = nets[dt].cuda()
    with torch.no_grad():
        p16 = nets["bf16"](*args, **kw).cpu()
        p8_first = nets["fp8"](*args, **kw).cpu()          # step 0: activation scales are 1
        p8 = nets["fp8"](*args, **kw).cpu()                # step 1: scales from step 0's maxima
    K = nets["fp8"]._backend
    f8_layers = [L for L in K.plan.layers if L.w8_off >= 0]
    assert K.has_fp8 and len(f8_layers) >= 20 and any(L.name.startswith("lidar") for L in f8_layers)
    used = [float(K._amax(L)[1].item()) for L in f8_layers]
    # (a conv that shares its input's image with another consumer -- conv1 next to a down-sampling shortcut -- uses
    # that consumer's slot and leaves its own at 0)
    assert sum(1 for u in used if u > 0) >= 0.7 * len(used), used
    x, pc, uv, ns = oracle_inputs(cfg, pts, crt)
    sd = model_ref.make_state_dict(full_shapes(cfg))
    g = geometry_ref.grid_constants(cfg)
    ref = model_ref.forward(sd, cfg, x, img, pc, uv, ns, "eval", fusion={"K": 3, "aff": g["aff"], "rmax": None})
    e16 = float((p16 - ref).abs().max() / ref.abs().max())
    for p in (p8_first, p8):
        assert torch.isfinite(p).all()
        e8 = float((p - ref).abs().max() / ref.abs().max())
        d = float((p - p16).norm() / p16.norm())
        assert e8 < max(0.12, 4 * e16) and d < 0.06, (e8, e16, d)
    assert not torch.equal(p8, p16)
    # train step through the fp8 forward
    net = nets["fp8"]
    before = net.flat_params.clone()
    opt = pkg("train").FlatAdam(net, 1e-4, (0.9, 0.999))
    pred = net(*args, **kw)
    R = torch.from_numpy(det.uniform(tuple(pred.shape), 779, -1.0, 1.0)).cuda()
    (pred * R).sum().backward()
    g8 = net._gradflat.clone()
    opt.step()
    assert torch.isfinite(net.flat_params).all() and float((net.flat_params != before).float().mean()) > 0.4
    # straight-through backward: gradients close to the bf16 model's
    pred16 = nets["bf16"](*args, **kw)
    (pred16 * R).sum().backward()
    g16 = nets["bf16"]._gradflat
    cos = float((g8 * g16).sum() / (g8.norm() * g16.norm()))
    assert cos > 0.9, cos


def test_graph_replay_with_side_stream_geometry_matches_eager():
    """config hip_graphs with the fusion path: camera-stream graph, then (after the side-stream geometry) the
    LiDAR-stream graph, then the backward graph -- same predictions and gradients as the eager launches, over several
    steps with different frames (replays, and a second graph set when the valid-point count changes bucket)."""
    cfg, pts, img, crt = setup("f32")
    T = pkg("train")
    det = pkg("detfill")
    geo = pkg("data_import_carla").FrameGeometry(cfg, crt)
    lim6 = (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"], cfg["lidar_z_min"], cfg["lidar_z_max"])
    frames = [[torch.from_numpy(p).cuda() for p in pts],
              [torch.from_numpy(det.synthetic_points(1500, lim6, 70 + b)).cuda() for b in range(2)],
              [torch.from_numpy(det.synthetic_points(400, lim6, 90 + b)).cuda() for b in range(2)]]     # fewer valid points
    order = [0, 1, 0, 2, 1, 2]
    res = {}
    for graphs in (False, True):
        c = copy.deepcopy(cfg)
        c["hip_graphs"] = graphs
        trainer = T.Train(c)
        det.fill_state_dict(trainer.model)
        out = []
        for step, k in enumerate(order):
            x_lidar, geom = trainer.geometry_async(geo, frames[k])
            pred = trainer.model(x_lidar, img.cuda(), geom=geom)
            R = torch.from_numpy(det.uniform(tuple(pred.shape), 700 + step, -1.0, 1.0)).cuda()
            (pred * R).sum().backward()
            out.append((pred.detach().clone(), trainer.model._gradflat.clone()))
        res[graphs] = out
        if graphs:
            assert trainer.model._graphs is not None and 1 <= len(trainer.model._graphs.sets) <= 3
    for step, ((p0, g0), (p1, g1)) in enumerate(zip(res[False], res[True])):
        assert torch.allclose(p0, p1, rtol=1e-5, atol=1e-6), "step %d: predictions differ by %g" % (step, float((p0 - p1).abs().max()))
        assert float((g0 - g1).abs().max()) < 2e-5 * float(g0.abs().max()), "step %d" % step


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_voxel_image_fast_path_matches_fp32_grid_path(dtype):
    """16-bit compute types: train.geometry_async hands the model the voxeliser's [B,L,W,Cz] image instead of the fp32
    [B,Cz,L,W] grid; the prediction must be the same bit for bit."""
    cfg, pts, img, crt = setup(dtype)
    trainer = pkg("train").Train(cfg)
    pkg("detfill").fill_state_dict(trainer.model)
    geo = pkg("data_import_carla").FrameGeometry(cfg, crt)
    dev_pts = [torch.from_numpy(p).cuda() for p in pts]
    with torch.no_grad():
        x_lidar, geom = trainer.geometry_async(geo, dev_pts)
        assert x_lidar.dtype != torch.float32 and x_lidar.shape[-1] == cfg["voxel_channel"]
        a = trainer.model(x_lidar, img.cuda(), geom=geom)
        grids = torch.stack([geo.voxelize(p) for p in dev_pts])
        x2, geom2 = trainer.geometry_async(geo, dev_pts)
        b = trainer.model(grids, img.cuda(), geom=geom2)
    assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [0, 1])
def test_batched_fusion_kernels_equal_per_frame_calls(dtype):
    """dcf_point_sample_fwd/bwd_batch, dcf_fusion_gather_fwd_batch, dcf_fusion_gather_bwd_inv_batch (grid.y = frame: one launch per
    batch instead of one per frame) against the single-frame entry points on three frames of different density."""
    ops, H = pkg("ops"), pkg("_hip")
    tdt = H.torch_dtype(dtype)
    g = torch.Generator().manual_seed(11)
    B, n_max, K, h, w, stride, Cb, Cf, Hf, Wf = 3, 700, 3, 24, 40, 4, 128, 64, 24, 32
    aff = (10.0, 0.0, 10.0, 400.0)
    ns = [500, 37, 0]
    xyz = torch.zeros(B, n_max, 3)
    uv = torch.zeros(B, n_max, 2)
    for b, n in enumerate(ns):
        xyz[b, :n, 0] = torch.rand(n, generator=g) * (h * stride / aff[0])
        xyz[b, :n, 1] = torch.rand(n, generator=g) * (w * stride / aff[2]) - aff[3] / aff[2]
        xyz[b, :n, 2] = torch.rand(n, generator=g) * 2 - 1
        uv[b, :n, 0] = torch.rand(n, generator=g) * (Wf * 4 - 1)
        uv[b, :n, 1] = torch.rand(n, generator=g) * (Hf * 4 - 1)
    xyz, uv = xyz.cuda(), uv.cuda()
    cnt = torch.tensor(ns, dtype=torch.int32, device="cuda")
    fmap = (torch.rand(B, Hf, Wf, Cf, generator=g) - 0.5).cuda().to(tdt)
    # point sampling forward / backward
    # (the output needs no clearing since round 4: rows past a frame's point count are written as zeros)
    fp_b = ops.point_sample_fwd_batch(dtype, fmap, uv, cnt, n_max, torch.full((B, n_max, Cf), float("nan"), device="cuda", dtype=tdt))
    for b in range(B):
        assert float(fp_b[b, ns[b]:].float().abs().max()) == 0.0 if ns[b] < n_max else True
    assert bool(torch.isfinite(fp_b.float()).all())
    gfp = (torch.rand(B, n_max, Cf, generator=g) - 0.5).cuda().to(tdt)
    gF_b = ops.point_sample_bwd_batch(dtype, gfp, uv, cnt, n_max, torch.zeros(B, Hf, Wf, Cf, device="cuda"))
    for b in range(B):
        fp1 = ops.point_sample_fwd(dtype, fmap[b], uv[b], cnt[b:b + 1], n_max)
        assert torch.equal(fp1, fp_b[b])
        gF1 = ops.point_sample_bwd(dtype, gfp[b], uv[b], cnt[b:b + 1], n_max, torch.zeros(Hf, Wf, Cf, device="cuda"))
        assert float((gF1 - gF_b[b]).abs().max()) <= 1e-5 * max(float(gF1.abs().max()), 1e-6)
    # gather forward / backward by point
    idx = torch.stack([ops.knn_bev(xyz[b], cnt[b:b + 1], K, h, w, stride, aff) for b in range(B)], 0)
    P = (torch.rand(B, n_max, Cb, generator=g) - 0.5).cuda().to(tdt)
    w1d = ((torch.rand(Cb, 3, generator=g) - 0.5) * 0.2).cuda().reshape(-1)
    b1 = ((torch.rand(Cb, generator=g) - 0.5) * 0.2).cuda()
    hs_b, c_b = ops.fusion_gather_fwd_batch(dtype, P, xyz, idx, stride, aff, w1d, b1, torch.empty(B, h, w, Cb, device="cuda", dtype=tdt),
                                            torch.empty(B, h * w, device="cuda"))
    ghs = (torch.rand(B, h, w, Cb, generator=g) - 0.5).cuda().to(tdt)
    inv = ops.fusion_invert([idx[b] for b in range(B)], n_max)
    ws = ops.fusion_bwd_workspace("cuda")
    gP_b, gw_b, gb_b = torch.zeros(B, n_max, Cb, device="cuda"), torch.zeros(Cb * 3, device="cuda"), torch.zeros(Cb, device="cuda")
    ops.fusion_gather_bwd_inv_batch(dtype, P, xyz, inv, n_max, 0, (K, h, w), stride, aff, w1d, b1, ghs, gP_b, gw_b, gb_b, ws)
    gw_1, gb_1 = torch.zeros(Cb * 3, device="cuda"), torch.zeros(Cb, device="cuda")
    for b in range(B):
        hs1, c1 = ops.fusion_gather_fwd(dtype, P[b], xyz[b], idx[b], stride, aff, w1d, b1)
        assert torch.equal(hs1, hs_b[b]) and torch.equal(c1, c_b[b])
        gP1 = torch.zeros(n_max, Cb, device="cuda")
        ops.fusion_gather_bwd_inv(dtype, P[b], xyz[b], inv, n_max, b, (K, h, w), stride, aff, w1d, b1, ghs[b], gP1, gw_1, gb_1, ws)
        assert float((gP1 - gP_b[b]).abs().max()) <= 2e-5 * max(float(gP1.abs().max()), 1e-6)
    for a_, b_ in ((gw_b, gw_1), (gb_b, gb_1)):
        assert float((a_ - b_).abs().max()) <= 1e-4 * float(b_.abs().max())
    assert float(ws.abs().max()) == 0.0


@pytest.mark.parametrize("Cb,K,hw,ns", [(64, 3, (24, 40), (200, 1)), (128, 5, (24, 40), (300, 0)), (192, 1, (24, 40), (7, 150)),
                                        (256, 3, (24, 40), (3, 299)), (64, 3, (176, 200), (3000, 40)), (128, 3, (88, 100), (2500, 2))])
def test_fusion_backward_direct_rows_match_the_fp32_accumulator_form(Cb, K, hw, ns):
    """dcf_fusion_gather_bwd_direct_batch (dP stored in the compute dtype by the single writer of a row; rows whose pairs cross a
    slice boundary summed in the workspace and stored by the run's last slice) against dcf_fusion_gather_bwd_inv_batch (fp32
    accumulator + atomics), two frames per launch: a dense frame and one with a handful of points / one point / none (runs far
    longer than a slice).  The workspace must come back zero, and a second launch on it must give the same rows."""
    ops, H = pkg("ops"), pkg("_hip")
    g = torch.Generator().manual_seed(23)
    h, w = hw
    stride, n_max, B = 4, 3000 if max(ns) > 300 else 300, 2
    aff = (10.0, 0.0, 10.0, 400.0)
    xyz = torch.zeros(B, n_max, 3)
    for b, n in enumerate(ns):
        xyz[b, :n, 0] = torch.rand(n, generator=g) * (h * stride / aff[0])
        xyz[b, :n, 1] = torch.rand(n, generator=g) * (w * stride / aff[2]) - aff[3] / aff[2]
        xyz[b, :n, 2] = torch.rand(n, generator=g) * 2 - 1
    xyz = xyz.cuda()
    cnt = torch.tensor(list(ns), dtype=torch.int32, device="cuda")
    idx = torch.stack([ops.knn_bev(xyz[b], cnt[b:b + 1], K, h, w, stride, aff, 1.0e4) for b in range(B)], 0)
    inv = ops.fusion_invert([idx[b] for b in range(B)], n_max)
    w1d = ((torch.rand(Cb, 3, generator=g) - 0.5) * 0.2).cuda().reshape(-1)
    b1 = ((torch.rand(Cb, generator=g) - 0.5) * 0.2).cuda()
    for dtype in (H.BF16, H.F32, H.F16):
        tdt = H.torch_dtype(dtype)
        P = (torch.rand(B, n_max, Cb, generator=g) - 0.5).cuda().to(tdt)
        ghs = (torch.rand(B, h, w, Cb, generator=g) - 0.5).cuda().to(tdt)
        ws = ops.fusion_bwd_workspace("cuda")
        ref = [torch.zeros(B, n_max, Cb, device="cuda"), torch.zeros(Cb * 3, device="cuda"), torch.zeros(Cb, device="cuda")]
        ops.fusion_gather_bwd_inv_batch(dtype, P, xyz, inv, n_max, 0, (K, h, w), stride, aff, w1d, b1, ghs, *ref, ws)
        dws = ops.fusion_bwd_direct_workspace("cuda", K * h * w, Cb, B)
        runs = []
        for _ in range(2):
            got = [torch.zeros(B, n_max, Cb, device="cuda", dtype=tdt), torch.zeros(Cb * 3, device="cuda"), torch.zeros(Cb, device="cuda")]
            ops.fusion_gather_bwd_direct_batch(dtype, P, xyz, inv, n_max, 0, (K, h, w), stride, aff, w1d, b1, ghs, *got, ws, dws)
            runs.append(got)
            assert float(dws.abs().max()) == 0.0 and float(ws.abs().max()) == 0.0
        got = runs[0]
        # rows: one rounding to the storage type on top of the fp32 sums (whose order differs between the two kernels)
        ulp = {H.F32: 2.0 ** -22, H.BF16: 2.0 ** -8, H.F16: 2.0 ** -11}[dtype]
        scale = max(float(ref[0].abs().max()), 1e-6)
        err = (got[0].float() - ref[0]).abs()
        assert bool((err <= ulp * ref[0].abs() + 2e-5 * scale * max(1.0, (h * w) ** 0.5 / 8)).all()), (Cb, ns, dtype, float(err.max()), scale)
        assert float((runs[1][0].float() - got[0].float()).abs().max()) <= 2 * ulp * scale + 1e-5 * scale * (h * w) ** 0.5 / 8
        for a, b in ((got[1], ref[1]), (got[2], ref[2])):
            assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-6)
    with pytest.raises(H.DcfError):
        ops.fusion_gather_bwd_direct_batch(dtype, P, xyz, inv, n_max, 0, (K, h, w), stride, aff, w1d, b1, ghs, *got, ws, dws[:8])
