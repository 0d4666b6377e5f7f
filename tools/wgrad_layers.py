#!/usr/bin/env python3
"""Weight-gradient time of single cfg2 layers, one launch each (HIP events over 20 launches): which layers the grouped launches'
time belongs to.  Usage (GPU box): python tools/wgrad_layers.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
H, ops = bench.pkg("_hip"), bench.pkg("ops")
# (name, B, H, W, Cin, Cout, k, stride)
LAYERS = [("l2s conv1 3x3/s2", 2, 704, 800, 32, 64, 3, 2), ("l3s conv1 3x3/s2", 2, 352, 400, 64, 128, 3, 2), ("l4s conv1 3x3/s2", 2, 176, 200, 128, 192, 3, 2),
          ("l5s conv1 3x3/s2", 2, 88, 100, 192, 256, 3, 2), ("l2s down 1x1/s2", 2, 704, 800, 32, 64, 1, 2), ("l3s down 1x1/s2", 2, 352, 400, 64, 128, 1, 2),
          ("latconv2 1x1", 2, 176, 200, 128, 192, 1, 1), ("fusion fc2 site0 1x1", 2, 352, 400, 64, 64, 1, 1), ("fusion fc1 site0 (points)", 2, 35000, 1, 128, 64, 1, 1),
          ("l1 3x3 32ch", 2, 704, 800, 32, 32, 3, 1), ("l2 3x3 64ch", 2, 352, 400, 64, 64, 3, 1), ("l3 3x3 128ch", 2, 176, 200, 128, 128, 3, 1),
          ("l4 3x3 192ch", 2, 88, 100, 192, 192, 3, 1), ("l5 3x3 256ch", 2, 44, 50, 256, 256, 3, 1)]
for name, B, Hh, W, Cin, Cout, k, s in LAYERS:
    pad = k // 2
    Ho, Wo = (Hh + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(B, Hh, W, Cin, device="cuda").to(torch.bfloat16)
    gy = torch.randn(B, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
    ns = ops.conv2d_wgrad_splits(B, Ho, Wo, Cin, Cout, k, k, s)
    slabs = torch.empty((ns, Cout, k, k, Cin), device="cuda")
    gsum = torch.empty((4 * ns, Cout), device="cuda")
    for _ in range(3):
        ops.conv2d_wgrad(1, x, gy, slabs, ns, k, k, s, pad, gsum)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_wgrad(1, x, gy, slabs, ns, k, k, s, pad, gsum)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    byt = (x.numel() + gy.numel()) * 2
    fl = 2.0 * B * Ho * Wo * Cin * Cout * k * k
    print("%-28s nsplit %4d  %7.1f us  %6.2f TB/s of x + gy  %7.1f TFLOP/s  slabs %.1f MB" % (name, ns, us, byt / us / 1e6, fl / us / 1e6, slabs.numel() * 4 / 1e6))
