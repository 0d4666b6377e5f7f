"""Child process of tests/test_gpu_dp.py (not a test module): one data-parallel rank of a tiny fused model on GPU 0.

    python tests/dp_child.py <world> <rank> <port> <outdir>

world 1: one process takes both frames (B = 2).  world 2: rank r takes frame r (B = 1); the two ranks share GPU 0 and
exchange gradients through gloo (DCF_DIST_BACKEND=gloo) -- functionally what RCCL does on an 8-GPU node, on one GPU.
Writes the parameter arena after each step to <outdir>/w<world>_r<rank>.pt.

world "rccl2": the same two ranks on devices 0 and 1 with RCCL ("nccl") -- started by conftest.py only where two GPUs are visible;
writes <outdir>/n2_r<rank>.pt.

world "rccl1": ONE rank in a world-size-1 RCCL ("nccl") process group -- the product's bucketed, overlapped all-reduce
(Train._bucket_ready: dist.all_reduce(arena slice, async_op=True) from the autograd thread, behind dcf_wgrad_finalize_rows)
on the real backend with the one GPU a test box has.  Writes <outdir>/rccl1.pt (see run_rccl1).
"""
import copy
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _tiny_setup(lr=1e-3):
    import numpy as np
    import torch
    from _util import golden_cfg, load_golden, pkg
    D, det, calib = pkg("data_import_carla"), pkg("detfill"), pkg("calib")
    cfg = golden_cfg(load_golden("model_tiny.npz"))
    cfg.update(dict(image_height=96, image_width=128, max_num_pc=2048, projection_mode="correct", dtype="f32",
                    loss_reduction="mean", bn_mode="eval", learning_rate=lr))
    cfg["fusion"] = dict(enabled=True, K=3, r_max=None, image_channels=64, image_stream="resnet18", zero_init_last=False)
    lim6 = (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"], cfg["lidar_z_min"], cfg["lidar_z_max"])
    Kc = np.array([[60.0, 0.0, 64.0], [0.0, 60.0, 48.0], [0.0, 0.0, 1.0]])
    geo = D.FrameGeometry(cfg, calib.crt_from(Kc, calib.R_LIDAR_TO_CAM))
    frames = []
    for f in range(2):
        pts = torch.from_numpy(det.synthetic_points(1500, lim6, 60 + f)).cuda()
        img = torch.from_numpy(det.synthetic_image(96, 128, 60 + f)).cuda()
        boxes, nb = D.synthetic_boxes(cfg, 60 + f, n=3)
        frames.append((pts, img, boxes, nb))
    return cfg, geo, frames


def run_rccl1(port, outdir):
    """Four trainers on the same frames and weights, three Train.one_step each:
      plain    -- overlap off: one finalisation launch, no collective (world size 1)
      overlap  -- overlap_allreduce "force": LiDAR + fusion bucket finalised and all-reduced (RCCL, async) under the camera
                  stream's backward, camera bucket at the end                              -> must equal `plain` BITWISE
      premul   -- the same with allreduce_premul 2 (ncclRedOp PreMulSum inside RCCL, undone by Adam's gradient scale): the
                  collective then CHANGES the buffer, so its stream order against the finalisation launch is observable
                                                                                           -> must equal `plain` BITWISE
      bf16     -- overlap with grad_bucket_dtype "bf16": the buckets are rounded to bf16, all-reduced as bf16 and widened back
                                                                                           -> `plain` to bf16 rounding
      wrong    -- premul with the hook called (and the collective drained) BEFORE the finalisation launch (negative control: the
                  all-reduce doubles the stale arena, the finalisation then overwrites it with the undoubled gradient) -> must DIFFER."""
    os.environ.update(dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", DCF_FORCE_DIST="1",
                           DCF_DIST_BACKEND="nccl"))
    import copy
    import numpy as np
    import torch
    import torch.distributed as dist
    from _util import pkg
    T, det = pkg("train"), pkg("detfill")
    HB = pkg("backend_hip").HipBackend
    T.init_distributed()
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    cfg, geo, frames = _tiny_setup()
    out = {"backend": dist.get_backend()}

    def run(mode):
        c = copy.deepcopy(cfg)
        c["overlap_allreduce"] = False if mode == "plain" else "force"
        if mode in ("premul", "wrong"):
            c["allreduce_premul"] = 2.0
        if mode == "bf16":
            c["grad_bucket_dtype"] = "bf16"
        tr = T.Train(c)
        det.fill_state_dict(tr.model)
        calls = []
        orig = tr._bucket_ready
        tr._bucket_ready = lambda ranges: (calls.append(list(ranges)), orig(ranges))[1]
        if mode == "wrong":
            def bad_bucket_ready(self, layers, which):
                if self.bucket_hook is None or which != "lidar+fusion":
                    return
                i0, f0 = self._layer_split(layers)
                self._flush_wgrads()
                done = self.__dict__.setdefault("_done", [False] * len(layers))
                todo = [(a, b) for a, b in ((0, i0), (f0, len(layers)))]
                self.bucket_hook([self._param_ranges(layers, a, b) for a, b in todo])                              # too early
                torch.cuda.synchronize()      # (the collective has doubled the STALE arena before the launch below overwrites it:
                                              # without this the two race and the control passes or fails by chance)
                for a, b in todo:
                    self._finalize(a, b)
                    for i in range(a, b):
                        done[i] = True
            backend = tr.model._ensure_backend(tr.model.flat_params.device)
            backend.bucket_ready = bad_bucket_ready.__get__(backend, HB)
        params, grads = [], []
        for step in range(3):
            np.random.seed(100 + step)
            x_lidar, geom = tr.geometry_async(geo, [f[0] for f in frames])
            tr.one_step(x_lidar, torch.stack([f[1] for f in frames], 0), torch.stack([f[2] for f in frames], 0),
                        torch.tensor([f[3] for f in frames]), geom=geom)
            torch.cuda.synchronize()
            params.append(tr.model.flat_params.detach().cpu().clone())
            grads.append(tr.model.flat_grads.detach().cpu().clone())
        return {"params": params, "grads": grads, "hook_calls": calls, "numel": tr.model.flat_grads.numel()}

    for mode in ("plain", "overlap", "premul", "wrong", "bf16"):
        try:
            out[mode] = run(mode)
        except Exception as e:                       # e.g. PreMulSum not available in this RCCL build: reported, not hidden
            import traceback
            out[mode] = {"error": "%s: %s" % (type(e).__name__, e), "trace": traceback.format_exc()}
    torch.save(out, os.path.join(outdir, "rccl1.pt"))
    dist.destroy_process_group()


def main():
    if sys.argv[1] == "rccl1":
        return run_rccl1(sys.argv[3], sys.argv[4])
    two_gpus = sys.argv[1] == "rccl2"          # two ranks on devices 0 and 1, gradients exchanged by RCCL (boxes with >= 2 GPUs)
    world, rank, port, outdir = (2 if two_gpus else int(sys.argv[1])), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), LOCAL_RANK=str(rank) if two_gpus else "0",
                           WORLD_SIZE=str(world), DCF_DIST_BACKEND="nccl" if two_gpus else "gloo"))
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
               os.path.join(outdir, ("n%d_r%d.pt" if two_gpus else "w%d_r%d.pt") % (world, rank)))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
