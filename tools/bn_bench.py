#!/usr/bin/env python3
"""Train-mode BatchNorm kernels at the cfg2 activation shapes: time per launch (library profiler) and the tensor bytes per second
they stream.  Usage (GPU box): python tools/bn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
H, ops = bench.pkg("_hip"), bench.pkg("ops")
SHAPES = [(2, 704, 800, 32), (2, 352, 400, 64), (2, 176, 200, 128), (2, 88, 100, 192), (2, 44, 50, 256), (2, 94, 311, 64), (2, 24, 78, 256), (2, 12, 39, 512)]
for shp in SHAPES:
    C = shp[-1]
    x = torch.randn(shp, device="cuda").to(torch.bfloat16)
    g = torch.randn(shp, device="cuda").to(torch.bfloat16)
    res = torch.randn(shp, device="cuda").to(torch.bfloat16)
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    ws = ops.bn_workspace(C, "cuda")
    for _ in range(3):
        y, mean, invstd = ops.bn_train_fwd(1, x, gamma, beta, res, rm, rv, True, ws)
        ops.bn_train_bwd(1, g, x, mean, invstd, gamma, dg, db, ws)
    torch.cuda.synchronize()
    H.call("dcf_prof_reset"); H.call("dcf_prof_enable", 1)
    for _ in range(10):
        y, mean, invstd = ops.bn_train_fwd(1, x, gamma, beta, res, rm, rv, True, ws)
        ops.bn_train_bwd(1, g, x, mean, invstd, gamma, dg, db, ws)
    torch.cuda.synchronize(); H.call("dcf_prof_enable", 0)
    pr = H.prof_read()
    mb = x.numel() * 2 / 1e6
    passes = {"bn_stats_partial": 1, "bn_bwd_partial": 2, "bn_apply_fwd": 3, "bn_apply_bwd": 3}
    print("%-22s %6.1f MB |" % (str(shp), mb), "  ".join("%s %.1f us%s" % (k.replace("bn_", ""), v[0] / v[1] * 1e3, (" (%.2f TB/s)" % (passes[k] * mb / (v[0] / v[1] * 1e3) / 1e6 * 1e6 / 1e6)) if k in passes else "") for k, v in sorted(pr.items())))
