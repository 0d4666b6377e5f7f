#!/usr/bin/env python3
"""Whose noise is it?  cfg2-size fp32 backward: HIP fp32 and CPU fp32 (oracle/model_ref.py) against the same statement in fp64,
with every ReLU argument of the fp64 pass recorded: what the decisions within fp32 noise of zero can move in the bias gradients.
Usage (GPU box): python tools/bwd_noise.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _util import pkg
from oracle import geometry_ref, model_ref
import test_gpu_benchsize as tb

det, calib, D = pkg("detfill"), pkg("calib"), pkg("data_import_carla")
cfg = tb._cfg2_config("f32")
crt = calib.kitti_like_crt()
pts = det.synthetic_points(100000, (0.0, 70.4, -40.0, 40.0, -2.4, 0.8), 23)
img = torch.from_numpy(det.synthetic_image(375, 1242, 23)).unsqueeze(0)
net = pkg("model").ObjectDetection_DCF(cfg)
det.fill_state_dict(net)
net = net.cuda()
geo = D.FrameGeometry(cfg, crt)
vox, pc, uv, cnt, _ = geo(torch.from_numpy(pts))
R = torch.from_numpy(det.uniform((1, 32, 176, 200), 97, -1.0, 1.0)); R[:, 18:] = 0
pred = net(vox.unsqueeze(0), img.cuda(), points=pc.unsqueeze(0), uv=uv.unsqueeze(0), n_valid=cnt)
(pred * R.cuda()).sum().backward()
torch.cuda.synchronize()
grid, pc_ref, uv_ref, n_ref, _ = geometry_ref.voxelization_projection(pts, cfg, crt, proj_mode="correct")
shapes = {}
shapes.update(model_ref.lidar_state_shapes(cfg)); shapes.update(model_ref.image_state_shapes(64)); shapes.update(model_ref.fusion_state_shapes(cfg, 64))
sd = model_ref.make_state_dict(shapes)
gc = geometry_ref.grid_constants(cfg)
maps = [[torch.from_numpy(geometry_ref.knn_bev(pc_ref[:n_ref], 3, 704 // s, 800 // s, s, gc["aff"], None))] for s in (2, 4, 8, 16)]
res = {}
for dt in (torch.float32, torch.float64):
    params = {k: (v.detach().to(dt).clone().requires_grad_(True) if (v.dtype.is_floating_point and "running" not in k) else (v.detach().to(dt) if v.dtype.is_floating_point else v)) for k, v in sd.items()}
    import torch.nn.functional as F
    x = torch.from_numpy(grid).unsqueeze(0).to(dt)
    # model_ref casts the image to float32 internally: patch through a double image stream by monkeypatching
    orig = model_ref.image_stream
    if dt == torch.float64:
        def img64(sd_, img_u8, bn_mode="eval", pfx="image_backbone", fpn="image_fpn"):
            class U(object):
                def to(self, _): return img_u8.to(torch.float64)
            return orig(sd_, U(), bn_mode, pfx, fpn)
        model_ref.image_stream = img64
        model_ref.anchors_orig = model_ref.anchors
        model_ref.anchors = lambda c: model_ref.anchors_orig(c).double()
    relus = []
    if dt == torch.float64:                      # record every ReLU's argument and (after backward) the gradient at its output
        real_relu = F.relu
        def spy(t, *a, **k):
            o = real_relu(t, *a, **k)
            o.retain_grad()
            relus.append((t.detach(), o))
            return o
        F.relu = spy
    out = model_ref.forward(params, cfg, x, img, torch.from_numpy(pc_ref).unsqueeze(0).to(dt), torch.from_numpy(uv_ref).unsqueeze(0).to(dt), [n_ref], "eval",
                            fusion={"K": 3, "aff": gc["aff"], "rmax": None}, knn_maps=maps)
    (out * R.to(dt)).sum().backward()
    if dt == torch.float64:
        F.relu = real_relu
        amb = []
        for pre, o in relus:                      # [1,C,H,W]
            if pre.dim() != 4:
                amb.append((0, None))
                continue
            m = pre.abs() <= 4e-6 * float(pre.abs().max())
            gsl = (o.grad.abs() * m).sum((0, 2, 3)) if o.grad is not None else torch.zeros(pre.shape[1], dtype=dt)
            amb.append((int(m.sum()), gsl))
    res[dt] = ({k: v.grad for k, v in params.items() if torch.is_tensor(v) and v.grad is not None}, out.detach())
    model_ref.image_stream = orig
g64, o64 = res[torch.float64]
g32, o32 = res[torch.float32]
print("forward: hip vs f64 %.3g, cpu32 vs f64 %.3g" % (float((pred.detach().cpu().double() - o64).abs().max() / o64.abs().max()), float((o32.double() - o64).abs().max() / o64.abs().max())))
rows = []
for k, p in net.named_parameters():
    w = g64[k]; s = float(w.abs().max()) + 1e-30
    l2 = float(w.norm()) + 1e-30
    rows.append((float((p.grad.cpu().double() - w).abs().max()) / s, float((g32[k].double() - w).abs().max()) / s, s, k,
                 float((p.grad.cpu().double() - w).norm()) / l2, float((g32[k].double() - w).norm()) / l2))
rows.sort(reverse=True)
print("worst by HIP error (hip_err, cpu32_err, scale, key):")
for r in rows[:12]: print("  %.3g %.3g %.3g %s  L2: hip %.3g cpu32 %.3g" % r)
rows.sort(key=lambda r: -r[1])
print("worst by CPU-fp32 error:")
for r in rows[:8]: print("  %.3g %.3g %.3g %s  L2: hip %.3g cpu32 %.3g" % r)
print("tensors with max-rel > 1e-3: hip %d, cpu32 %d of %d; worst L2-rel: hip %.3g, cpu32 %.3g" % (
    sum(r[0] > 1e-3 for r in rows), sum(r[1] > 1e-3 for r in rows), len(rows), max(r[4] for r in rows), max(r[5] for r in rows)))

# ReLU decisions within fp32 noise of zero (fp64 statement): what they can move in the BatchNorm bias gradients
names = []
for li, nb in ((1, 2), (2, 2), (3, 2), (4, 2)):
    for bi in range(nb):
        names += [("image_backbone.layer%d.%d.bn1.bias" % (li, bi)), ("image_backbone.layer%d.%d.bn2.bias" % (li, bi))]
lm = cfg["lidar_module"]
for si in range(1, 6):
    for bi in range(lm["num_res_block%d" % si]):
        p = "lidar_backbone.backbone.layer%d.sequential.resblock_%d" % (si, bi)
        names += [p + ".bn1.bias", p + ".bn2.bias"]
    if si >= 2:
        names.append(None)                     # the fusion site's MLP ReLU
amb_l = amb[1:]                                # amb[0] = the stem's ReLU
assert len(amb_l) == len(names), (len(amb_l), len(names))
byk = dict((r[3], r) for r in rows)
print("bias gradients: observed HIP error vs the most the ambiguous ReLU decisions can move them")
for (cnt_, gsl), k in zip(amb_l, names):
    if k is None:
        continue
    r = byk[k]
    if r[0] > 5e-4:
        d = (dict(net.named_parameters())[k].grad.cpu().double() - g64[k]).abs()
        print("  %-70s hip err %.3g (abs %.3g at ch %d) ambiguous elements %d, their |g| in that channel %.3g" % (k, r[0], float(d.max()), int(d.argmax()), cnt_, float(gsl[int(d.argmax())])))
