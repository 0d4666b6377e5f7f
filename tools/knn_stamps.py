#!/usr/bin/env python3
"""Which tiles a k_knn_search launch waits for (needs a -DKNN_STAMP build of geometry.hip:
    KFILE=geometry bash tools/rw_variants.sh knnstamp="-DKNN_STAMP"; DCF_HIP_LIB=<pkg>/libdcf_hip_vknnstamp.so python tools/knn_stamps.py [stride]).
Per 8x8-pixel tile of frame 0: cycles to the end of the window phase and to the end of the kernel, candidate points of the window
and of the ring phase, rings walked, blocks scanned, lanes still searching after the window."""
import ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ops = bench.pkg("ops"); H = bench.pkg("_hip")
cfg = bench.kitti_config(2)
pool = bench.FramePool(cfg, 2, 100000, 0)
geo = pool.geometry
g = geo.grid
v, pc, uv, cnt, _ = geo(pool.pts[0])
L = ctypes.CDLL(H.LIB_PATH)
for s in [int(a) for a in sys.argv[1:]] or [2, 4]:
    h, w = 704 // s, 800 // s
    for _ in range(3):
        ops.knn_bev(pc, cnt, 3, h, w, s, g.aff)
    torch.cuda.synchronize()
    dims = (ctypes.c_int * 2)()
    buf = np.zeros(4096 * 16, dtype=np.int64)
    L.dcf_knn_stamps_read(buf.ctypes.data_as(ctypes.c_void_p), dims)
    h8, w8 = (h + 7) // 8, (w + 7) // 8
    st = buf.reshape(4096, 16)[:h8 * w8]
    print("stride %d: %d tiles; cycles to end: median %d, p90 %d, max %d; window phase: median %d, max %d" % (
        s, h8 * w8, np.median(st[:, 1]), np.percentile(st[:, 1], 90), st[:, 1].max(), np.median(st[:, 0]), st[:, 0].max()))
    print("  tile(TI,TJ)  cyc_window  cyc_total  pts_window  pts_rings  rings  blocks  lanes_left")
    for t in np.argsort(-st[:, 1])[:12]:
        print("  (%2d,%2d) %10d %10d %10d %10d %6d %6d %8d" % (t // w8, t % w8, st[t, 0], st[t, 1], st[t, 2], st[t, 3], st[t, 4], st[t, 5], st[t, 6]))
    busy = st[st[:, 2] > 0]
    print("  window phase of tiles with points (medians, cycles since kernel entry): count read %d, cell ranges read %d, gather addresses %d, "
          "first points read %d, end of phase %d; median points %d" % tuple(np.median(busy[:, k]) for k in (8, 9, 10, 11, 0, 2)))
    tot = st[:, 1].astype(np.float64)
    print("  sum of tile cycles %.0f (= %.1f us if spread over 1024 SIMDs at 2.4 GHz)" % (tot.sum(), tot.sum() / 1024 / 2400))
