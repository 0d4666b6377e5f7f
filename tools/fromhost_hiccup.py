#!/usr/bin/env python3
"""Where a slow step of the from-host loop spends its HOST time (bench.py --input host shows single steps of 30-90 ms on some boxes):
the bench's own loop with wall-clock stamps around its phases; prints every step slower than 3x the median."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
train = bench.pkg("train")
B = int(os.environ.get("HB", "2"))
cfg = bench.kitti_config(B, "bf16", 100000, 3, "resnet18", (1242, 375))
cfg["bn_mode"] = "eval"; cfg["hip_graphs"] = "auto"
torch.cuda.set_device(0)
trainer = train.Train(cfg)
bench.pkg("detfill").fill_state_dict(trainer.model)
pool = bench.FramePool(cfg, n_frames=max(2 * B, 4), n_points=100000, seed0=0)
for s in range(10):
    bench.train_step(trainer, pool, pool.batch(s, B))
torch.cuda.synchronize()
import gc
gc.collect(); gc.freeze()
FL = bench.pkg("frame_loader")
# stamps inside the staging thread: what it does between batches
_log = []
_orig_stage, _orig_sets = FL.FrameLoader._stage, FL.FrameLoader._ensure_sets
def _stage(self, host, slot, consumer_stream=None):
    t0 = time.perf_counter()
    st = self._sets[slot]
    if st.done is not None:
        st.done.synchronize()
    t1 = time.perf_counter()
    out = _orig_stage(self, host, slot, consumer_stream)
    _log.append(("stage slot %d" % slot, t0, t1 - t0, time.perf_counter() - t1))
    return out
def _sets(self, host, n):
    t0 = time.perf_counter()
    _orig_sets(self, host, n)
    _log.append(("ensure_sets", t0, 0.0, time.perf_counter() - t0))
FL.FrameLoader._stage, FL.FrameLoader._ensure_sets = _stage, _sets
for rep in range(int(os.environ.get("REPS", "4"))):
    N = 50
    loader = iter(FL.FrameLoader(bench.HostFrames(pool, N + 4, B), B))
    for _ in range(3):
        trainer.one_step_raw(pool.geometry, next(loader))
    torch.cuda.synchronize()
    rows = []
    t_prev = time.perf_counter()
    for s in range(N):
        t0 = time.perf_counter()
        batch = next(loader)
        t1 = time.perf_counter()
        batch.wait()
        x_lidar, geom = trainer.geometry_async(pool.geometry, batch["points"], crts=batch.get("crt"), wait_event=batch.event)
        t2 = time.perf_counter()
        trainer.one_step(x_lidar, batch["image"], batch["bboxes"], batch["num_bboxes"], geom=geom)
        t3 = t4 = t5 = t6 = time.perf_counter()
        rows.append((t6 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5))
    torch.cuda.synchronize()
    tot = time.perf_counter() - t_prev
    loader.close()
    if rep == 0:
        tb = _log[0][1]
        for name, t0, a, b in _log[:12]:
            print("   worker +%.1f ms: %s: wait previous copies %.1f ms, pack + enqueue %.1f ms" % ((t0 - tb) * 1e3, name, a * 1e3, b * 1e3))
    med = sorted(r[0] for r in rows)[N // 2]
    print("rep %d: %.2f ms per step wall; median host step %.2f ms" % (rep, tot / N * 1e3, med * 1e3))
    for i, r in enumerate(rows):
        if r[0] > 3 * med:
            print("   step %2d host %.1f ms: next(loader) %.1f | geometry_async %.1f | one_step %.1f | %.1f %.1f %.1f" % ((i,) + tuple(v * 1e3 for v in r)))
