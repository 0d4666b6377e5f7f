#!/usr/bin/env python3
"""Threshold sweep of k_knn_search_fine (option KNN_FINE_MIN: points in a pixel's 5 x 5 window of own cells above which the fine
site's cells are searched) on the cfg2 cloud, coarse sites, batch 2.  Usage (GPU box): python tools/knn_fine_sweep.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conv_bench import timeit
import test_gpu_geometry as tg
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops = importlib.import_module(PKG + ".ops"); H = importlib.import_module(PKG + "._hip")
g, a = tg._cfg2_cloud(seed=9); _, b = tg._cfg2_cloud(seed=10)
n_max = 100000
d = torch.zeros(2, n_max, 3); d[0, :a.shape[0]] = torch.from_numpy(a); d[1, :b.shape[0]] = torch.from_numpy(b); d = d.cuda()
cnt = torch.tensor([a.shape[0], b.shape[0]], dtype=torch.int32, device="cuda")
ws = torch.empty((2, ops.knn_ws_stride(n_max, 352, 400)), dtype=torch.uint8, device="cuda")
ops.knn_bev_batch(d, cnt, 3, 352, 400, 2, g.aff, None, ws=ws)
for stride in (4, 8, 16):
    h, w = 704 // stride, 800 // stride
    wso = torch.empty((2, ops.knn_ws_stride(n_max, h, w)), dtype=torch.uint8, device="cuda")
    out = torch.empty((2, 3, h, w), dtype=torch.int32, device="cuda")
    t0 = timeit(lambda: ops.knn_bev_batch(d, cnt, 3, h, w, stride, g.aff, None, ws=wso, out=out))
    row = ["own %.1f us" % (t0 * 1e6)]
    for thr in (32, 64, 128, 192, 384, 1000000):
        H.set_option("KNN_FINE_MIN", thr)
        t = timeit(lambda: ops.knn_bev_batch_shared(d, cnt, 3, h, w, stride, (352, 400, 2), ws, g.aff, None, ws=wso, out=out))
        row.append("%d: %.1f" % (thr, t * 1e6))
    H.set_option("KNN_FINE_MIN", None)
    print("stride %2d (sort + search, us): %s" % (stride, " | ".join(row)), flush=True)
