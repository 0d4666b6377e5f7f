import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _util import pkg
from test_gpu_fusion import setup
cfg, pts, img, crt = setup("f32")
T = pkg("train"); det = pkg("detfill")
geo = pkg("data_import_carla").FrameGeometry(cfg, crt)
lim6 = (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"], cfg["lidar_z_min"], cfg["lidar_z_max"])
frames = [[torch.from_numpy(p).cuda() for p in pts],
          [torch.from_numpy(det.synthetic_points(1500, lim6, 70 + b)).cuda() for b in range(2)],
          [torch.from_numpy(det.synthetic_points(400, lim6, 90 + b)).cuda() for b in range(2)]]
order = [0, 1, 0, 2, 1, 2]
res = {}
for mode in ("eager_dyn", "eager_static", "graphs_static", "graphs_dyn"):
    c = copy.deepcopy(cfg)
    c["hip_graphs"] = mode.startswith("graphs")
    c["static_geometry"] = mode.endswith("static")
    trainer = T.Train(c)
    det.fill_state_dict(trainer.model)
    out = []
    for step, k in enumerate(order):
        x_lidar, geom = trainer.geometry_async(geo, frames[k])
        pred = trainer.model(x_lidar, img.cuda(), geom=geom)
        R = torch.from_numpy(det.uniform(tuple(pred.shape), 700 + step, -1.0, 1.0)).cuda()
        (pred * R).sum().backward()
        torch.cuda.synchronize()
        out.append((pred.detach().clone(), trainer.model._gradflat.clone()))
    res[mode] = out
for mode in res:
    print(mode, [("%.2e" % float((a[0] - b[0]).abs().max()), "%.2e" % float((a[1] - b[1]).abs().max() / b[1].abs().max())) for a, b in zip(res[mode], res["eager_dyn"])])
