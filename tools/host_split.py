#!/usr/bin/env python3
"""Host enqueue time of the cfg2 step by phase (the GPU runs behind, unsynchronised): geometry, forward, loss, backward + Adam,
and the number of C-ABI calls per phase.  Usage (GPU box): python tools/host_split.py
With the launch-free build (make -C <pkg>/csrc nolaunch; DCF_HIP_LIB=<pkg>/libdcf_hip_nolaunch.so) the GPU never holds the host
back (no queue back-pressure, the valid-count event is instantly done): the figures are then the host's own work per step, without
the ~3 us of every hipLaunchKernel."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
H = bench.pkg("_hip")
train = bench.pkg("train")
cfg = bench.kitti_config(2)
torch.cuda.set_device(0)
trainer = train.Train(cfg)
bench.pkg("detfill").fill_state_dict(trainer.model)
pool = bench.FramePool(cfg, 4, 100000, 0)
for s in range(5):
    bench.train_step(trainer, pool, pool.batch(s, 2))
torch.cuda.synchronize()
calls = collections.Counter()
orig = H.call
phase = ["?"]
def counted(name, *a):
    calls[phase[0]] += 1
    return orig(name, *a)
for m in ("_hip", "ops", "backend_hip", "train", "loss", "model", "engine", "data_import_carla"):
    mod = bench.pkg(m)
    if hasattr(mod, "H") and mod.H is H:
        pass
H.call = counted
T = collections.defaultdict(float)
N = 20
for s in range(N):
    ids = pool.batch(s, 2)
    t0 = time.perf_counter(); phase[0] = "geometry"
    x_lidar, geom = trainer.geometry_async(pool.geometry, [pool.pts[i] for i in ids])
    x_image = pool.image_batch(ids)
    boxes = torch.stack([pool.boxes[i] for i in ids], 0); nb = torch.tensor([pool.nb[i] for i in ids])
    t1 = time.perf_counter(); phase[0] = "forward"
    pred_cls, pred_reg, _ = trainer._predict(x_lidar, x_image, {"geom": geom})
    t2 = time.perf_counter(); phase[0] = "loss"
    trainer.loss_value = trainer.loss_total(boxes, nb, pred_cls, pred_reg)
    t3 = time.perf_counter(); phase[0] = "backward"
    trainer.loss_value.backward()
    t4 = time.perf_counter(); phase[0] = "adam"
    trainer.optimizer.step(1.0)
    st = geom.get("_set")
    if st is not None:
        st["free_event"] = torch.cuda.Event(); st["free_event"].record()
    t5 = time.perf_counter()
    for k, v in (("geometry", t1 - t0), ("forward", t2 - t1), ("loss", t3 - t2), ("backward", t4 - t3), ("adam", t5 - t4)):
        T[k] += v
torch.cuda.synchronize()
tot = sum(T.values())
for k in ("geometry", "forward", "loss", "backward", "adam"):
    print("%-9s %.3f ms/step  %5.1f C-ABI calls/step" % (k, T[k] / N * 1e3, calls[k] / N))
print("total     %.3f ms/step host" % (tot / N * 1e3))
