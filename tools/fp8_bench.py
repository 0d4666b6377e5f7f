#!/usr/bin/env python3
"""Forward conv: bf16 kernel vs fp8 kernel (+ the activation cast) on the BEV / camera layer shapes.  usage: fp8_bench.py"""
import os, sys, importlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops = importlib.import_module(PKG + ".ops")

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

SHAPES = [(2, 352, 400, 64, 64, 3, 1), (2, 352, 400, 64, 128, 3, 2), (2, 176, 200, 128, 128, 3, 1), (2, 88, 100, 192, 192, 3, 1),
          (2, 44, 50, 256, 256, 3, 1), (2, 176, 200, 192, 192, 3, 1), (2, 94, 311, 64, 64, 3, 1), (2, 47, 156, 128, 128, 3, 1),
          (1, 270, 480, 64, 64, 3, 1), (2, 176, 200, 128, 192, 1, 1)]
for (B, H, W, Ci, Co, k, s) in SHAPES:
    x = torch.rand((B, H, W, Ci), device="cuda").to(torch.bfloat16)
    w = (torch.rand((Co, k, k, Ci), device="cuda") - 0.5).to(torch.bfloat16)
    w8 = torch.randint(0, 120, (Co, k, k, Ci), dtype=torch.uint8, device="cuda")
    ws = torch.ones(Co, device="cuda")
    am = torch.ones(1, device="cuda"); cur = torch.zeros(64, device="cuda")
    x8 = ops.cast_fp8(1, x, am, cur)
    t16 = timeit(lambda: ops.conv2d_fwd(1, x, w, None, None, k, k, s, k // 2, True, Co))
    t8 = timeit(lambda: ops.conv2d_fwd_fp8(1, x8, w8, ws, am, None, None, k, k, s, k // 2, True, Co))
    tc = timeit(lambda: ops.cast_fp8(1, x, am, cur))
    t8f = timeit(lambda: ops.conv2d_fwd_fp8(1, x8, w8, ws, am, None, None, k, k, s, k // 2, True, Co, want_y8=True, y8amax=am, y8cur=cur))
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    fl = 2.0 * B * Ho * Wo * Co * Ci * k * k
    print("B%d %dx%d %d->%d k%d s%d: bf16 %.1f us (%.0f TF)  fp8 %.1f us (%.0f TF)  +fp8 output %.1f us  cast %.1f us" % (B, H, W, Ci, Co, k, s, t16, fl / t16 / 1e6, t8, fl / t8 / 1e6, t8f, tc))
