#!/usr/bin/env python3
"""Weight-gradient kernel timing per cfg2 layer shape (single launches) over the number of pixel ranges; DCF_WGRAD3S=0 gives
the row-sharing per-wave kernel, DCF_WGS_DBG the ablations of the shared-staging one.  Usage (GPU box):
python tools/wgs_bench.py [names...]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import LIDAR, IMAGE, timeit
ops = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd.ops")


def main():
    want = set(sys.argv[1:])
    B = 2
    for name, Hh, W, Ci, Co, k, s, cnt in LIDAR + IMAGE:
        if k != 3 or s != 1 or (want and name not in want) or (not want and Ci % 64):
            continue
        x = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).bfloat16()
        gy = (torch.rand((B, Hh, W, Co), device="cuda") - 0.5).bfloat16()
        fl = 2.0 * B * Hh * W * Co * Ci * 9
        ns0 = ops.conv2d_wgrad_splits(B, Hh, W, Ci, Co, 3, 3, 1)
        out = []
        for ns in sorted({ns0, 1, 2, 4, 8, 16, 28, 32, 56, 84}):
            if B * Hh * (W + 2) // ns < 256:
                continue
            slabs = torch.empty((ns, Co, 3, 3, Ci), device="cuda")
            t = timeit(lambda: ops.conv2d_wgrad(1, x, gy, slabs, ns, 3, 3, 1, 1), iters=10)
            out.append("%s%d: %.1f us %.0f TF" % ("*" if ns == ns0 else "", ns, t * 1e6, fl / t / 1e12))
        print("%-6s %s" % (name, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
