#!/usr/bin/env python3
"""Which Python call sites launch torch's own copy / fill kernels during one cfg2 train step (GPU box)."""
import collections, os, sys, traceback
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
train = bench.pkg("train")
cfg = bench.kitti_config(2, "bf16", 100000, 3, "resnet18", (1242, 375))
cfg["bn_mode"] = "eval"
torch.cuda.set_device(0)
trainer = train.Train(cfg)
bench.pkg("detfill").fill_state_dict(trainer.model)
pool = bench.FramePool(cfg, n_frames=4, n_points=100000, seed0=0)
for s in range(4):
    bench.train_step(trainer, pool, pool.batch(s, 2))
torch.cuda.synchronize()
counts = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "site-packages/torch" not in fr.filename and "_glue_trace" not in fr.filename:
            return "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
    return "?"
def wrap(obj, name):
    orig = getattr(obj, name)
    def f(*a, **k):
        r = orig(*a, **k)
        t = a[0] if a and torch.is_tensor(a[0]) else r
        if torch.is_tensor(t) and t.is_cuda or (torch.is_tensor(r) and r.is_cuda):
            counts[(name, site())] += 1
        return r
    setattr(obj, name, f)
for n in ("copy_", "zero_", "fill_", "clone", "contiguous", "to", "cuda", "float", "bfloat16"):
    wrap(torch.Tensor, n)
for n in ("zeros", "zeros_like", "cat", "stack", "ones", "full", "tensor"):
    wrap(torch, n)
bench.train_step(trainer, pool, pool.batch(5, 2))
torch.cuda.synchronize()
for (n, s), c in sorted(counts.items(), key=lambda kv: -kv[1]):
    print("%3d  %-12s %s" % (c, n, s))
