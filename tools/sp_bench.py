#!/usr/bin/env python3
"""A/B of the spatial-tile streaming kernel (conv_sp.hip) against the kernels it replaces on the HBM-bound 32 / 64-channel 3x3
layers of cfg2 (forward with the residual epilogue, input gradient with residual + mask), per workgroups-per-CU setting.
Usage (GPU box): python tools/sp_bench.py [batch]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import timeit
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops = importlib.import_module(PKG + ".ops")
H = importlib.import_module(PKG + "._hip")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for name, Hh, W, C in (("l1", 704, 800, 32), ("l2", 352, 400, 64), ("i1", 94, 311, 64)):
        x = (torch.rand((B, Hh, W, C), device="cuda") - 0.5).bfloat16()
        w = ((torch.rand((C, 3, 3, C), device="cuda") - 0.5) * 0.1).bfloat16()
        res = (torch.rand((B, Hh, W, C), device="cuda") - 0.5).bfloat16()
        mask = (torch.rand((B, Hh, W, C), device="cuda") - 0.5).bfloat16()
        shift = torch.rand(C, device="cuda")
        byt = x.numel() * 2
        rows = []
        for label, opts in (("old", {"CONV_SP": 0}), ("sp/1", {"CONV_SP_MIN_PIX": 0, "CONV_SP_WGPC": 1}), ("sp/2", {"CONV_SP_MIN_PIX": 0, "CONV_SP_WGPC": 2}),
                            ("sp/auto", {"CONV_SP_MIN_PIX": 0}), ("sp/ring3", {"CONV_SP_MIN_PIX": 0, "CONV_SP_NBUF32": 3})):
            for k in ("CONV_SP", "CONV_SP_MIN_PIX", "CONV_SP_WGPC", "CONV_SP_NBUF32"):
                H.set_option(k, None)
            for k, v in opts.items():
                H.set_option(k, v)
            t0 = timeit(lambda: ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, False, C))
            t1 = timeit(lambda: ops.conv2d_fwd(1, x, w, shift, res, 3, 3, 1, 1, True, C))
            t2 = timeit(lambda: ops.conv2d_dgrad(1, x, w, res, (B, Hh, W, C), 3, 3, 1, 1, mask))
            rows.append("%s: plain %.1f us (%.2f TB/s) | +shift+res+relu %.1f (%.2f) | dgrad+res+mask %.1f (%.2f)" % (
                label, t0 * 1e6, 2 * byt / t0 / 1e12, t1 * 1e6, 3 * byt / t1 / 1e12, t2 * 1e6, 4 * byt / t2 / 1e12))
        print("%s %dx%dx%d B=%d (%.0f MB per tensor)" % (name, Hh, W, C, B, byt / 1e6))
        for r in rows:
            print("   " + r, flush=True)


if __name__ == "__main__":
    main()
