#!/usr/bin/env python3
"""Which Python lines of the cfg2 step issue copy / fill kernels (torch ops, not the C ABI): torch.profiler with stacks over
three steps, grouped by the innermost frame inside this repository.  Usage (GPU box): python tools/copy_sites.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
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
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for s in range(N):
        bench.train_step(trainer, pool, pool.batch(5 + s, 2))
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sites = collections.Counter()
dev = collections.Counter()
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue
    where = "?"
    for fr in ev.stack:
        if root in fr or "bench.py" in fr:
            where = fr.replace(root + "/", "")
            break
    sites[(ev.name, where)] += 1
    dev[(ev.name, where)] += ev.device_time_total
for (name, where), n in sorted(sites.items(), key=lambda kv: -dev[kv[0]]):
    print("%6.1f us/step %5.1f /step  %-22s %s" % (dev[(name, where)] / N, n / N, name, where[:150]))
