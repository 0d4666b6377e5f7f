#!/usr/bin/env python3
"""Per-site timing of the geometry kernels at cfg2 (GPU box)."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ops = bench.pkg("ops"); H = bench.pkg("_hip")
cfg = bench.kitti_config(2)
pool = bench.FramePool(cfg, 2, 100000, 0)
geo = pool.geometry
g = geo.grid


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


v, pc, uv, cnt, _ = geo(pool.pts[0])
print("n_valid", int(cnt.item()))
print("voxelize+project us (GPU timeline)", timeit(lambda: geo(pool.pts[0])))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): geo(pool.pts[0])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue us", (t1 - t0) / 10 * 1e6, "with sync", (t2 - t0) / 10 * 1e6)
H.call("dcf_prof_reset"); H.call("dcf_prof_enable", 1)
for _ in range(5): geo(pool.pts[0])
torch.cuda.synchronize(); H.call("dcf_prof_enable", 0)
print({k: (round(v[0] / 5 * 1e3, 1), v[1] // 5) for k, v in H.prof_read().items()})
for si in range(1, 5):
    s = 2 ** si
    h, w = 704 // s, 800 // s
    ws = torch.empty((H.lib().dcf_knn_workspace_bytes(pc.shape[0], h, w),), dtype=torch.uint8, device="cuda")
    H.call("dcf_prof_reset"); H.call("dcf_prof_enable", 1)
    for _ in range(5):
        ops.knn_bev(pc, cnt, 3, h, w, s, g.aff, None, ws)
    torch.cuda.synchronize(); H.call("dcf_prof_enable", 0)
    pr = H.prof_read()
    print("site stride %2d (%dx%d):" % (s, h, w), {k: round(v[0] / v[1] * 1e3, 1) for k, v in pr.items()})
