#!/usr/bin/env python3
"""Stand-alone timing of the 32 -> 32 channel 3x3 weight gradient at the cfg2 size (2 x 704 x 800): the strip-walking kernel
(conv_wgv.hip) against the position-walking one it replaces (option WGRAD3V=0), HIP events over 20 launches each.
Usage (GPU box): python tools/wgv_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
H, ops = bench.pkg("_hip"), bench.pkg("ops")
B, Hh, W, C = 2, 704, 800, 32
x = torch.randn(B, Hh, W, C, device="cuda").to(torch.bfloat16)
gy = torch.randn(B, Hh, W, C, device="cuda").to(torch.bfloat16)
for opt in (None, "0") + tuple(sys.argv[1:]):
    if opt is not None and opt != "0":
        H.set_option("WGRAD3V_UNITS", opt)
    else:
        H.set_option("WGRAD3V", opt)
    ns = ops.conv2d_wgrad_splits(B, Hh, W, C, C, 3, 3, 1)
    slabs = torch.empty((ns, C, 3, 3, C), device="cuda")
    gsum = torch.empty((4 * ns, C), device="cuda")
    for _ in range(3):
        ops.conv2d_wgrad(1, x, gy, slabs, ns, 3, 3, 1, 1, gsum)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_wgrad(1, x, gy, slabs, ns, 3, 3, 1, 1, gsum)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("option %-6s nsplit %4d: %7.1f us  %6.2f TB/s of x + gy" % (opt, ns, us, 2 * x.numel() * 2 / us / 1e6))
    if opt == "0":
        H.set_option("WGRAD3V", None)
