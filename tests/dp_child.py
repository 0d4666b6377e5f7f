"""Child process of tests/test_gpu_dp.py (not a test module): one data-parallel rank of a tiny fused model on GPU 0.

    python tests/dp_child.py <world> <rank> <port> <outdir>

world 1: one process takes both frames (B = 2).  world 2: rank r takes frame r (B = 1); the two ranks share GPU 0 and
exchange gradients through gloo (DCF_DIST_BACKEND=gloo) -- functionally what RCCL does on an 8-GPU node, on one GPU.
Writes the parameter arena after each step to <outdir>/w<world>_r<rank>.pt.
"""
import copy
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    world, rank, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),
                           DCF_DIST_BACKEND="gloo"))
    import numpy as np
    import torch
    from _util import golden_cfg, load_golden, pkg
    T, D, det, calib = pkg("train"), pkg("data_import_carla"), pkg("detfill"), pkg("calib")
    assert T.init_distributed() == world
    if world == 1:
        torch.cuda.set_device(0)
    cfg = golden_cfg(load_golden("model_tiny.npz"))
    cfg.update(dict(image_height=96, image_width=128, max_num_pc=2048, projection_mode="correct", dtype="f32",
                    loss_reduction="mean", bn_mode="eval", learning_rate=1e-3))
    cfg["fusion"] = dict(enabled=True, K=3, r_max=None, image_channels=64, image_stream="resnet18", zero_init_last=False)
    lim6 = (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"], cfg["lidar_z_min"], cfg["lidar_z_max"])
    Kc = np.array([[60.0, 0.0, 64.0], [0.0, 60.0, 48.0], [0.0, 0.0, 1.0]])
    crt = calib.crt_from(Kc, calib.R_LIDAR_TO_CAM)
    geo = D.FrameGeometry(cfg, crt)
    frames = []
    for f in range(2):
        pts = torch.from_numpy(det.synthetic_points(1500, lim6, 60 + f)).cuda()
        img = torch.from_numpy(det.synthetic_image(96, 128, 60 + f)).cuda()
        boxes, nb = D.synthetic_boxes(cfg, 60 + f, n=3)
        frames.append((pts, img, boxes, nb))
    torch.manual_seed(1000 + rank)                      # deliberately different: the rank-0 broadcast must make replicas equal
    trainer = T.Train(cfg)
    if rank == 0:
        det.fill_state_dict(trainer.model)
    else:
        with torch.no_grad():
            trainer.model.flat_params.add_(0.25)        # a replica that starts out wrong ...
    trainer.sync_replicas()                             # ... until rank 0's arenas are broadcast (Train does this at construction too)
    mine = list(range(2)) if world == 1 else [rank]
    H, W = cfg["voxel_length"] // 4, cfg["voxel_width"] // 4
    outs, grads = [], []
    for step in range(2):
        np.random.seed(100 + step)
        # the loss draws its negative samples from numpy's global generator, frame after frame: a rank that owns frame r
        # first consumes the draws of the frames before it, so that every frame sees the same samples in both runs
        for f in range(mine[0]):
            trainer.loss_total.assign(frames[f][2][:frames[f][3]], H, W)
        x_lidar, geom = trainer.geometry_async(geo, [frames[f][0] for f in mine])
        img = torch.stack([frames[f][1] for f in mine], 0)
        boxes = torch.stack([frames[f][2] for f in mine], 0)
        nb = torch.tensor([frames[f][3] for f in mine])
        trainer.one_step(x_lidar, img, boxes, nb, geom=geom)
        torch.cuda.synchronize()
        outs.append(trainer.model.flat_params.detach().cpu().clone())
        grads.append((trainer.model.flat_grads.detach() / world).cpu().clone())     # what Adam consumed (gscale = 1 / world)
    torch.save({"params": outs, "grads": grads, "loss": float(trainer.loss_value.item()), "lr": cfg["learning_rate"]},
               os.path.join(outdir, "w%d_r%d.pt" % (world, rank)))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
