#!/usr/bin/env python3
"""cProfile of the host side of bench.py's train step (GPU box)."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
train = bench.pkg("train")
cfg = bench.kitti_config(2)
torch.cuda.set_device(0)
trainer = train.Train(cfg)
bench.pkg("detfill").fill_state_dict(trainer.model)
pool = bench.FramePool(cfg, 4, 100000, 0)
for s in range(3):
    bench.train_step(trainer, pool, pool.batch(s, 2))
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(5):
    bench.train_step(trainer, pool, pool.batch(s, 2))
t_host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 5
print("host enqueue time per step %.2f ms ; with final sync %.2f ms" % (t_host * 1e3, t_all * 1e3))
pr = cProfile.Profile()
pr.enable()
for s in range(5):
    bench.train_step(trainer, pool, pool.batch(s, 2))
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(32)
print(st.getvalue()[:9000])
