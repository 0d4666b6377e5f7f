#!/usr/bin/env python3
"""Sweep of the row-sharing convolution kernel's tile shapes (DCF_RS_KIND / DCF_RS_NPT) per cfg2 layer shape, forward and
dgrad, against the generic implicit GEMM (DCF_CONV_RS=0 is read once per process, so the generic numbers come from
tools/conv_bench.py of an earlier run).  Usage (GPU box): python tools/rs_sweep.py [names...]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import LIDAR, IMAGE, timeit
ops = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd.ops")
H = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd._hip")
KINDS = {0: (128, 10), 1: (64, 12), 2: (64, 4)}


def main():
    want = set(sys.argv[1:])
    B = 2
    for name, Hh, W, Ci, Co, k, s, cnt in LIDAR + IMAGE:
        if k != 3 or s != 1 or Ci % 64 or Co % 64 or (want and name not in want):
            continue
        x = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).bfloat16()
        w = ((torch.rand((Co, 3, 3, Ci), device="cuda") - 0.5) * 0.1).bfloat16()
        fl = 2.0 * B * Hh * W * Co * Ci * 9
        H.set_option("RS_KIND", None); H.set_option("RS_NPT", None)
        t0 = timeit(lambda: ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, False, Co))
        res = []
        for kind, (bn, nmax) in KINDS.items():
            if Co % bn:
                continue
            for npt in range(1, nmax + 1):
                H.set_option("RS_KIND", kind); H.set_option("RS_NPT", npt)
                t = timeit(lambda: ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, False, Co), iters=10)
                res.append((t, kind, npt))
        res.sort()
        print("%-6s auto %6.1f us %5.0f TF | best: %s" % (name, t0 * 1e6, fl / t0 / 1e12, "  ".join(
            "k%d/n%d %.1f" % (kd, n, t * 1e6) for t, kd, n in res[:6])), flush=True)


if __name__ == "__main__":
    main()
