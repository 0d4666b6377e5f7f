# experiment: wall time per step with geometry recomputed every step (the benchmark) vs reused (NOT a valid benchmark mode)
import sys, os, time, torch
sys.path.insert(0, "/root/repo")
import bench
train = bench.pkg("train")
cfg = bench.kitti_config(2)
torch.cuda.set_device(0)
trainer = train.Train(cfg)
bench.pkg("detfill").fill_state_dict(trainer.model)
pool = bench.FramePool(cfg, 4, 100000, 0)
def run(reuse, steps=12):
    cache = {}
    def step(s):
        ids = pool.batch(s, 2)
        key = tuple(ids)
        if reuse and key in cache:
            x_lidar, geom = cache[key]
            geom = {k: v for k, v in geom.items() if not k.startswith("_")}
        else:
            x_lidar, geom = trainer.geometry_async(pool.geometry, [pool.pts[i] for i in ids])
            if reuse:
                torch.cuda.synchronize()
                cache[key] = (x_lidar, {k: v for k, v in geom.items() if k not in ("event", "voxel_event", "inv_event")})
        x_image = torch.stack([pool.img[i] for i in ids], 0)
        boxes = torch.stack([pool.boxes[i] for i in ids], 0)
        nb = torch.tensor([pool.nb[i] for i in ids])
        trainer.one_step(x_lidar, x_image, boxes, nb, geom=geom)
    for s in range(4): step(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(steps): step(4 + s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for r in range(3):
    print("recompute %.3f ms   reuse %.3f ms" % (run(False), run(True)))
