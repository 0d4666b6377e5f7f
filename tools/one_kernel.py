#!/usr/bin/env python3
"""Run one conv kernel shape a few times (for rocprofv3 --pmc passes).  usage: one_kernel.py fwd|dgrad|wgrad NAME"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import conv_bench as cb
ops = cb.ops
kind, name = sys.argv[1], sys.argv[2]
B = 2
row = [r for r in cb.LIDAR + cb.IMAGE if r[0] == name][0]
_, Hh, W, Ci, Co, k, s, _ = row
pad = k // 2
Ho, Wo = (Hh + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
td = torch.bfloat16
x = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).to(td)
w = ((torch.rand((Co, k, k, Ci), device="cuda") - 0.5) * 0.1).to(td)
wt = w.permute(3, 1, 2, 0).contiguous()
gy = (torch.rand((B, Ho, Wo, Co), device="cuda") - 0.5).to(td)
ns = ops.conv2d_wgrad_splits(B, Ho, Wo, Ci, Co, k, k, s)
slabs = torch.empty((ns, Co, k, k, Ci), device="cuda")
for _ in range(5):
    if kind == "fwd":
        ops.conv2d_fwd(1, x, w, None, None, k, k, s, pad, False, Co)
    elif kind == "dgrad":
        ops.conv2d_dgrad(1, gy, wt, None, (B, Hh, W, Ci), k, k, s, pad)
    else:
        ops.conv2d_wgrad(1, x, gy, slabs, ns, k, k, s, pad)
torch.cuda.synchronize()
