#!/usr/bin/env python3
"""Where the host time of a FrameLoader-fed step goes (cProfile around the --from-host loop of bench.py)."""
import cProfile, pstats, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
train = bench.pkg("train")
cfg = bench.kitti_config(2)
cfg["bn_mode"] = "eval"; cfg["hip_graphs"] = False
torch.cuda.set_device(0)
trainer = train.Train(cfg)
bench.pkg("detfill").fill_state_dict(trainer.model)
pool = bench.FramePool(cfg, 4, 100000, 0)
loader = iter(bench.pkg("frame_loader").FrameLoader(bench.HostFrames(pool, 40, 2), 2))
for _ in range(3):
    trainer.one_step_raw(pool.geometry, next(loader))
torch.cuda.synchronize()
tn = ts = 0.0
for _ in range(10):
    t0 = time.perf_counter(); b = next(loader); t1 = time.perf_counter(); trainer.one_step_raw(pool.geometry, b); t2 = time.perf_counter()
    tn += t1 - t0; ts += t2 - t1
torch.cuda.synchronize()
print("next(loader) %.2f ms, one_step_raw enqueue %.2f ms" % (tn * 100, ts * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    trainer.one_step_raw(pool.geometry, next(loader))
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
# GPU time of the steps themselves
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); t0 = time.perf_counter(); a.record()
for _ in range(10):
    trainer.one_step_raw(pool.geometry, next(loader))
b.record(); torch.cuda.synchronize()
print("10 host-fed steps: wall %.1f ms, GPU (main stream) %.1f ms" % ((time.perf_counter() - t0) * 1e3, a.elapsed_time(b)))
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(10):
    bench.train_step(trainer, pool, pool.batch(s, 2))
torch.cuda.synchronize()
print("10 resident steps: wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
Hm = bench.pkg("_hip")
for mode in ("host", "resident"):
    Hm.call("dcf_prof_reset"); Hm.call("dcf_prof_enable", 1)
    for s in range(4):
        if mode == "host":
            trainer.one_step_raw(pool.geometry, next(loader))
        else:
            bench.train_step(trainer, pool, pool.batch(s, 2))
    torch.cuda.synchronize(); Hm.call("dcf_prof_enable", 0)
    prof = Hm.prof_read()
    tab = sorted(((n, v[0] / 4, v[1] / 4) for n, v in prof.items()), key=lambda t: -t[1])
    print(mode, "sum of kernel ms/step: %.2f" % sum(t[1] for t in tab))
    for n, ms, c in tab[:8]:
        print("   %-40s %.3f ms  x%.0f" % (n, ms, c))
