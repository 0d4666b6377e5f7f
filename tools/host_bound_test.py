#!/usr/bin/env python3
"""Is the step host-bound?  Add a host-side delay per step and watch the step time: a GPU-bound step absorbs it."""
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
for s in range(5):
    bench.train_step(trainer, pool, pool.batch(s, 2))
torch.cuda.synchronize()
for delay in (0.0, 0.5e-3, 1.0e-3, 2.0e-3, 0.0):
    t0 = time.perf_counter()
    for s in range(30):
        bench.train_step(trainer, pool, pool.batch(s, 2))
        if delay:
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < delay:
                pass
    torch.cuda.synchronize()
    print("host delay %.1f ms/step -> %.3f ms/step" % (delay * 1e3, (time.perf_counter() - t0) / 30 * 1e3))
# CPU time actually burnt per step (main + autograd thread), against the wall time of the step
c0, t0 = time.process_time(), time.perf_counter()
for s in range(30):
    bench.train_step(trainer, pool, pool.batch(s, 2))
c1 = time.process_time()
torch.cuda.synchronize()
print("CPU time %.2f ms/step (all threads), wall %.2f ms/step" % ((c1 - c0) / 30 * 1e3, (time.perf_counter() - t0) / 30 * 1e3))
