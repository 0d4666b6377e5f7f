#!/usr/bin/env python3
"""us per layer of the cfg2 residual stages as separate dcf_conv2d_fwd launches and as one chain launch (forward epilogue:
residual + ReLU), batch 1 and 2: python tools/chain_time.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops = importlib.import_module(PKG + ".ops")


def run(shape, n, chain):
    B, Hh, W, C = shape
    x = (torch.rand((B, Hh, W, C), device="cuda") - 0.5).bfloat16()
    ext = (torch.rand((B, Hh, W, C), device="cuda") - 0.5).bfloat16()
    wts = [((torch.rand((C, 3, 3, C), device="cuda") - 0.5) * 0.05).bfloat16() for _ in range(n)]
    ws = ops.conv3x3_chain_workspace(1, B, Hh, W, C, n, "cuda")
    layers = [(wts[l], None, (ext if l == 0 else l - 2) if l % 2 == 0 else None, None, True) for l in range(n)]

    def sep():
        cur, outs = x, []
        for l, (w, sh, r, m, relu) in enumerate(layers):
            cur = ops.conv2d_fwd(1, cur, w, None, outs[r] if type(r) is int else r, 3, 3, 1, 1, relu, C)
            outs.append(cur)
    f = (lambda: ops.conv3x3_chain(1, x, layers, 0, ws)) if chain else sep
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    assert ops.conv3x3_chain_status(ws) == 0
    return e0.elapsed_time(e1) * 1e3 / 20 / n


for shape, n in (((2, 44, 50, 256), 11), ((2, 88, 100, 192), 11), ((2, 176, 200, 128), 7), ((2, 94, 311, 64), 4), ((2, 47, 156, 128), 3),
                 ((2, 24, 78, 256), 3), ((2, 12, 39, 512), 3), ((1, 44, 50, 256), 11), ((1, 88, 100, 192), 11), ((1, 176, 200, 128), 7)):
    if not ops.conv3x3_chain_supported(1, shape[0], shape[1], shape[2], shape[3], n):
        importlib.import_module(PKG + "._hip").set_option("CHAIN_WIDE", 1)          # (outside the automatic policy: timed all the same)
    print(shape, n, "us/layer: separate %.2f  chain %.2f" % (run(shape, n, False), run(shape, n, True)))
