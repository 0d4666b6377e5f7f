#!/usr/bin/env python3
"""A/B of the row-sharing kernel's rotated / prefetching tap loop (option RS_PF, conv_rs_kernel.h) per cfg2 layer shape: forward
and input gradient (with a residual + mask epilogue), us per launch with the option off / on, and whether the outputs are
bit-identical.  Usage (GPU box): python tools/rs_pf_ab.py [--batch 2] [names...]"""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import LIDAR, IMAGE, timeit
ops = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd.ops")
H = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd._hip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--opt", default="RS_PF")
    ap.add_argument("--values", default="0,1", help="the two option values to compare")
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    B = args.batch
    tot = [0.0, 0.0]
    for name, Hh, W, Ci, Co, k, s, cnt in LIDAR + IMAGE:
        if k != 3 or s != 1 or Ci % 64 or Co % 64 or (args.names and name not in args.names):
            continue
        x = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).bfloat16()
        w = ((torch.rand((Co, 3, 3, Ci), device="cuda") - 0.5) * 0.1).bfloat16()
        wt = w.permute(3, 1, 2, 0).contiguous()
        gy = (torch.rand((B, Hh, W, Co), device="cuda") - 0.5).bfloat16()
        res = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).bfloat16()
        mask = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.3).bfloat16()
        fl = 2.0 * B * Hh * W * Co * Ci * 9
        out = {}
        va, vb = [int(t) for t in args.values.split(",")]
        for key, v in ((0, va), (1, vb)):
            H.set_option(args.opt, v)
            yf = ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, True, Co)
            yd = ops.conv2d_dgrad(1, gy, wt, res, (B, Hh, W, Ci), 3, 3, 1, 1, mask=mask)
            tf = timeit(lambda: ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, True, Co), iters=30)
            td = timeit(lambda: ops.conv2d_dgrad(1, gy, wt, res, (B, Hh, W, Ci), 3, 3, 1, 1, mask=mask), iters=30)
            out[key] = (yf.clone(), yd.clone(), tf, td)
        H.set_option(args.opt, None)
        same = torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
        tot[0] += (out[0][2] + out[0][3]) * cnt; tot[1] += (out[1][2] + out[1][3]) * cnt
        print("%-6s %4dx%-4d %3d->%3d  fwd %6.1f -> %6.1f us (%4.0f -> %4.0f TF)   dgrad %6.1f -> %6.1f us   identical: %s  (x%d)" % (
            name, Hh, W, Ci, Co, out[0][2] * 1e6, out[1][2] * 1e6, fl / out[0][2] / 1e12, fl / out[1][2] / 1e12, out[0][3] * 1e6, out[1][3] * 1e6, same, cnt), flush=True)
    print("weighted fwd + dgrad (ms per step): %s=0 %.3f   %s=1 %.3f" % (args.opt, tot[0] * 1e3, args.opt, tot[1] * 1e3))


if __name__ == "__main__":
    main()
