#!/usr/bin/env python3
"""Time of the register-weight kernel's forward launch on cfg2 layer shapes, for whatever library DCF_HIP_LIB names (the
compile-time variants of tools/rw_variants.sh).  Usage: DCF_HIP_LIB=... python tools/rw_time.py [--batch 2] [--tag x] names..."""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import LIDAR, IMAGE, timeit
sys.path.insert(0, os.path.join(ROOT, "tools", "variants"))
import rw_api          # the register-weight kernel lives in a variant library (tools/rw_variants.sh), not in libdcf_hip.so
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops = importlib.import_module(PKG + ".ops")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--tag", default="")
    ap.add_argument("--dgrad", action="store_true")
    ap.add_argument("--std", action="store_true", help="time dcf_conv2d_fwd (standard weight layout: conv_lc / conv_rs) instead of the _wf entry")
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    B = args.batch
    out = []
    for name, Hh, W, Ci, Co, k, s, cnt in LIDAR + IMAGE:
        if name not in args.names:
            continue
        x = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).bfloat16()
        w = ((torch.rand((Co, 3, 3, Ci), device="cuda") - 0.5) * 0.1).bfloat16()
        wf = rw_api.conv3x3_weight_frag(1, w)
        if args.dgrad:
            res = (torch.rand((B, Hh, W, Co), device="cuda") - 0.5).bfloat16()
            mask = (torch.rand((B, Hh, W, Co), device="cuda") - 0.3).bfloat16()
            t = timeit(lambda: rw_api.conv3x3_dgrad_wf(1, x, wf, res, (B, Hh, W, Co), mask=mask), iters=30)
        elif args.std:
            t = timeit(lambda: ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, False, Co), iters=30)
        else:
            t = timeit(lambda: rw_api.conv3x3_fwd_wf(1, x, wf, None, None, False, Co), iters=30)
        out.append("%s %.1f" % (name, t * 1e6))
    print("%-10s b%d %s" % (args.tag, B, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
