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

# the engine's call: both frames of a cfg2 batch, the four sites in one dcf_knn_bev_sites call; waves per tile of k_knn_search 1 / 2 / 4 / automatic
B = 2
fr = [geo(pool.pts[b]) for b in range(B)]
n_max = max(f[1].shape[0] for f in fr)
pts = torch.zeros((B, n_max, 3), device="cuda")
for b, f in enumerate(fr):
    pts[b, :f[1].shape[0]] = f[1]
cnts = torch.cat([f[3].reshape(1) for f in fr]).to(torch.int32)
sites = []
for si in range(1, 5):
    s = 2 ** si
    h, w = 704 // s, 800 // s
    sites.append((h, w, s, 0 if h * w <= 20000 else -1, torch.empty((B, ops.knn_ws_stride(n_max, h, w)), dtype=torch.uint8, device="cuda"),
                  torch.empty((B, 3, h, w), dtype=torch.int32, device="cuda")))
for tw in ("unmerged", "1", "4", None):
    H.set_option("KNN_MERGED_SEARCH", "0" if tw == "unmerged" else None)
    H.set_option("KNN_TILE_WAVES", None if tw == "unmerged" else tw)
    ops.knn_bev_sites(pts, cnts, 3, sites, g.aff, None)
    H.call("dcf_prof_reset"); H.call("dcf_prof_enable", 1)
    for _ in range(5):
        ops.knn_bev_sites(pts, cnts, 3, sites, g.aff, None)
    torch.cuda.synchronize(); H.call("dcf_prof_enable", 0)
    pr = H.prof_read()
    print("sites call, B=2, %s: total %.1f us;" % ({"unmerged": "one search launch per site", None: "one search launch, waves per tile automatic"}.get(tw, "one search launch, waves per tile %s" % tw), sum(v[0] for v in pr.values()) / 5 * 1e3),
          {k: (round(v[0] / v[1] * 1e3, 1), v[1] // 5) for k, v in pr.items()})
    print("   whole call on the GPU timeline: %.1f us" % timeit(lambda: ops.knn_bev_sites(pts, cnts, 3, sites, g.aff, None)))
H.set_option("KNN_TILE_WAVES", None)
H.set_option("KNN_MERGED_SEARCH", None)
