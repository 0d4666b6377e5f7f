#!/usr/bin/env python3
"""Does the host hold the GPU back at the loss (between forward and backward)?  A busy-wait is inserted in front of the loss call;
a step whose host thread is ahead of the GPU there absorbs it, one that is not gets longer by the same amount."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
train = bench.pkg("train")
cfg = bench.kitti_config(2)
torch.cuda.set_device(0)
trainer = train.Train(cfg)
bench.pkg("detfill").fill_state_dict(trainer.model)
pool = bench.FramePool(cfg, 4, 100000, 0)
delay = [0.0]
orig = trainer.loss_total.forward
def slow(*a, **k):
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < delay[0]:
        pass
    return orig(*a, **k)
trainer.loss_total.forward = slow
for s in range(5):
    bench.train_step(trainer, pool, pool.batch(s, 2))
torch.cuda.synchronize()
for d in (0.0, 0.1e-3, 0.2e-3, 0.4e-3, 0.8e-3, 0.0):
    delay[0] = d
    t0 = time.perf_counter()
    for s in range(40):
        bench.train_step(trainer, pool, pool.batch(s, 2))
    torch.cuda.synchronize()
    print("host delay before the loss %.1f ms -> %.3f ms/step" % (d * 1e3, (time.perf_counter() - t0) / 40 * 1e3))
