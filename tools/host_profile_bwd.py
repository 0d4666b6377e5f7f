#!/usr/bin/env python3
"""cProfile of the host side of the hand-written backward (engine.Plan.backward called directly in the main thread, the way
autograd's worker thread calls it) and of the forward, cfg2.  Usage (GPU box): python tools/host_profile_bwd.py"""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
m = trainer.model
K = m._backend
def fwd(s):
    ids = pool.batch(s, 2)
    x_lidar, geom = trainer.geometry_async(pool.geometry, [pool.pts[i] for i in ids])
    K.prepare()
    return m._plan.forward(K, x_lidar, pool.image_batch(ids), geom, save=True), geom
for which in ("forward", "backward"):
    pr = cProfile.Profile()
    t = 0.0
    for s in range(6):
        if which == "forward":
            t0 = time.perf_counter(); pr.enable(); pred, geom = fwd(s); pr.disable(); t += time.perf_counter() - t0
            g = torch.zeros_like(pred)
            m._plan.backward(K, g)
        else:
            pred, geom = fwd(s)
            g = torch.zeros_like(pred)
            t0 = time.perf_counter(); pr.enable(); m._plan.backward(K, g); pr.disable(); t += time.perf_counter() - t0
        st = geom.get("_set")
        if st is not None:
            st["free_event"] = torch.cuda.Event(); st["free_event"].record()
    torch.cuda.synchronize()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(22)
    print("==== %s: %.3f ms/step host (profiled)" % (which, t / 6 * 1e3))
    print("\n".join(out.getvalue().splitlines()[6:34]))
